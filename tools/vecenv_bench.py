"""Host-path throughput of the SB3 VecEnv adapter (numpy in / numpy out), needs a GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so100_mujoco_rl_amd.vec_env import So100VecEnv
from so100_mujoco_rl_amd.lib import F_REFERENCE
for n in (256, 4096):
    env = So100VecEnv("Env01-v1", n, flags=F_REFERENCE, seed=0, stagger_episodes=True)
    env.reset()
    a = np.random.uniform(-1, 1, (n, 6)).astype(np.float32)
    for _ in range(20): env.step(a)
    t0 = time.perf_counter(); K = 300
    for _ in range(K): env.step(a)
    dt = time.perf_counter() - t0
    print(f"So100VecEnv(Env01-v1, {n}) numpy step: {dt/K*1e6:.0f} us/step  {n*K/dt/1e6:.2f} M env-steps/s")
