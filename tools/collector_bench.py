"""Rollout-collector throughput vs batch size (persistent kernel up to 16384 envs, policy kernel + step kernel above).
    python tools/collector_bench.py"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.vec_env import So100VecEnv
from so100_mujoco_rl_amd.collector import RolloutCollector
from so100_mujoco_rl_amd.lib import F_REFERENCE, F_CUBE_PINNED

for flags, name in ((F_CUBE_PINNED, "contact disabled"), (F_REFERENCE, "reference physics")):
    for n in (4096, 16384, 65536, 262144, 1048576):
        T = 64 if n <= 16384 else 16
        env = So100VecEnv("Env01-v1", n, flags=flags, seed=0, stagger_episodes=True)
        col = RolloutCollector(env, RolloutCollector.random_policy_state(env.sim.obs_dim, env.device), T=T)
        for _ in range(3): col.collect()
        gc.collect(); torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 8
        for _ in range(reps): col.collect()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:18s} N={n:8d}  {'persistent' if col.persistent else 'stepwise  '}  {dt/reps/T*1e6:8.1f} us/step  {n*reps*T/dt/1e6:8.1f} M env-steps/s", flush=True)
        env.close(); del col, env; gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()    # (freeing pinned buffers later would stall a timed loop)
