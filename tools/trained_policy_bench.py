"""How much does a TRAINED policy change the step cost of the reference physics (fewer joints at their limits)?  Trains
Env01 for a few seconds with the built-in PPO, then times the persistent collector with the random initial policy and with
the trained one.    [SO100_LIB=...] python tools/trained_policy_bench.py [weights.pt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.vec_env import So100VecEnv
from so100_mujoco_rl_amd.collector import RolloutCollector
from so100_mujoco_rl_amd.lib import F_REFERENCE
from so100_mujoco_rl_amd.ppo import PPO

wfile = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trained_env01_r02.pt"
env = So100VecEnv("Env01-v1", 4096, flags=F_REFERENCE, seed=0, stagger_episodes=True)
learner = PPO(env.sim.obs_dim, env.device, seed=0)
col = RolloutCollector(env, learner.net.state_dict(), T=64)


def timeit(tag):
    for _ in range(4): col.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(16): col.collect()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cs = env.sim.get_field("contact_stat", dtype=torch.int32)
    print(f"{tag:16s} {dt/16/64*1e6:7.2f} us/step  ({4096*16*64/dt/1e6:6.1f} M env-steps/s)   envs with a pad contact in the last step: {float(((cs & 255) > 0).float().mean()):.3f}", flush=True)


timeit("random policy")
if os.path.exists(wfile):
    learner.net.load_state_dict(torch.load(wfile, map_location=env.device, weights_only=True))
else:
    for it in range(300):
        b = col.collect(); learner.update(b); col.load_policy(learner.net.state_dict())
    os.makedirs(os.path.dirname(wfile), exist_ok=True); torch.save(learner.net.state_dict(), wfile)
col.load_policy(learner.net.state_dict())
timeit("trained policy")
