"""The replayed (hipGraph) PPO update against the eager one on the same seed, with the collector's truncation bootstrap: the check behind
ppo.py's note on stream ordering (variants: clone=True gives the kernels a private copy of the weights, sync=True brackets update()
with device synchronisations).   python tools/ppo_graph_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.vec_env import So100VecEnv
from so100_mujoco_rl_amd.collector import RolloutCollector
from so100_mujoco_rl_amd.ppo import PPO
from so100_mujoco_rl_amd.lib import F_NOPADS
def run(name, boot, clone=False, sync=False, dummy=False, graph=True):
    env = So100VecEnv("Env01-v1", 4096, flags=F_NOPADS, seed=0, stagger_episodes=True)
    learner = PPO(15, env.device, seed=0, use_graph=graph)
    sd = lambda: {k: v.clone() for k, v in learner.net.state_dict().items()} if clone else learner.net.state_dict()
    col = RolloutCollector(env, sd(), T=64, bootstrap_truncated=boot)
    out = []
    for it in range(1, 141):
        b = col.collect()
        if dummy:
            x = torch.tanh(torch.randn(262144, 15, device="cuda") @ torch.randn(15, 64, device="cuda")); y = torch.tanh(x @ torch.randn(64, 64, device="cuda")); del x, y
        if sync: torch.cuda.synchronize()
        st = learner.update(b)
        if sync: torch.cuda.synchronize()
        col.load_policy(sd())
        if it % 20 == 0: out.append("%d: %.3f" % (it, st["mean_reward"]))
    print(name, " | ".join(out), flush=True); env.close()
run("graph boot (update() ends with a device sync)", True)
run("eager boot", True, graph=False)
