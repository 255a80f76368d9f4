#!/usr/bin/env python3
"""Minimal PPO on top of the on-device rollout collector (needs a GPU).  Functional evidence that the batched simulator
is a learnable stand-in for the reference's `main.py -a PPO train -e Env01-v1` loop (main.py:177-238) -- stable-baselines3
is not installed in this image, so the learner here is ~80 lines of plain PyTorch with SB3's PPO defaults
(MlpPolicy 2x64 tanh towers, gamma 0.99, gae_lambda 0.95, clip 0.2, lr 3e-4, 10 epochs are reduced to 4 for speed).

    python examples/train_ppo.py [--env Env01-v1] [--envs 4096] [--iters 150]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from so100_mujoco_rl_amd.vec_env import So100VecEnv               # noqa: E402
from so100_mujoco_rl_amd.collector import RolloutCollector        # noqa: E402
from so100_mujoco_rl_amd.lib import F_REFERENCE                    # noqa: E402


from so100_mujoco_rl_amd.ppo import PPO                             # noqa: E402  (the learner lives in the package)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env01-v1"); ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=150); ap.add_argument("--T", type=int, default=64)
    args = ap.parse_args()
    env = So100VecEnv(args.env, args.envs, flags=F_REFERENCE, seed=0, stagger_episodes=True)
    dev = env.device
    learner = PPO(env.sim.obs_dim, dev, seed=0)
    col = RolloutCollector(env, learner.net.state_dict(), T=args.T)
    t0 = time.time(); steps = 0
    for it in range(args.iters):
        b = col.collect()
        stats = learner.update(b)
        col.load_policy(learner.net.state_dict())
        steps += args.T * args.envs
        if it % 10 == 0 or it == args.iters - 1:
            torch.cuda.synchronize()
            print(f"iter {it:4d}  env-steps {steps/1e6:7.1f} M  mean reward/step {stats['mean_reward']:+.4f}  "
                  f"value loss {stats['value_loss']:.4f}  log_std {learner.net.log_std.mean().item():+.3f}  wall {time.time() - t0:6.1f} s", flush=True)


if __name__ == "__main__":
    main()
