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
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from so100_mujoco_rl_amd.vec_env import So100VecEnv               # noqa: E402
from so100_mujoco_rl_amd.collector import RolloutCollector        # noqa: E402
from so100_mujoco_rl_amd.lib import F_REFERENCE                    # noqa: E402


class ActorCritic(nn.Module):
    """state_dict keys match SB3's ActorCriticPolicy, so RolloutCollector.load_policy() takes it as is."""

    def __init__(self, obs_dim, act_dim=6):
        super().__init__()
        mk = lambda: nn.Sequential(nn.Linear(obs_dim, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh())
        self.mlp_extractor = nn.ModuleDict({"policy_net": mk(), "value_net": mk()})
        self.action_net = nn.Linear(64, act_dim); self.value_net = nn.Linear(64, 1)
        self.log_std = nn.Parameter(torch.zeros(act_dim))
        for m, g in ((self.mlp_extractor, 2 ** 0.5), (self.action_net, 0.01), (self.value_net, 1.0)):
            for l in m.modules():
                if isinstance(l, nn.Linear):
                    nn.init.orthogonal_(l.weight, g); nn.init.zeros_(l.bias)

    def evaluate(self, obs, act):
        mean = self.action_net(self.mlp_extractor["policy_net"](obs))
        value = self.value_net(self.mlp_extractor["value_net"](obs)).squeeze(-1)
        std = self.log_std.exp()
        logp = (-0.5 * ((act - mean) / std) ** 2 - self.log_std - 0.9189385332046727).sum(-1)
        ent = (0.5 + 0.9189385332046727 + self.log_std).sum()
        return value, logp, ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Env01-v1"); ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=150); ap.add_argument("--T", type=int, default=64)
    args = ap.parse_args()
    env = So100VecEnv(args.env, args.envs, flags=F_REFERENCE, seed=0, stagger_episodes=True)
    dev = env.device
    torch.manual_seed(0)
    net = ActorCritic(env.sim.obs_dim).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, eps=1e-5)
    col = RolloutCollector(env, net.state_dict(), T=args.T)
    gamma, lam, clip, epochs, mb = 0.99, 0.95, 0.2, 4, 32768
    t0 = time.time(); steps = 0
    for it in range(args.iters):
        b = col.collect()
        with torch.no_grad():
            last_v = net.value_net(net.mlp_extractor["value_net"](b["last_obs"])).squeeze(-1)
            T = b["rewards"].shape[0]
            adv = torch.zeros_like(b["rewards"]); g = torch.zeros_like(last_v)
            for t in reversed(range(T)):                       # GAE; `dones[t]` ends the episode after step t
                nv = last_v if t == T - 1 else b["values"][t + 1]
                nonterm = 1.0 - b["dones"][t]
                delta = b["rewards"][t] + gamma * nv * nonterm - b["values"][t]
                g = delta + gamma * lam * nonterm * g
                adv[t] = g
            ret = adv + b["values"]
            obs = b["obs"].reshape(-1, env.sim.obs_dim); act = b["actions"].reshape(-1, 6)
            oldlp = b["log_probs"].reshape(-1); adv = adv.reshape(-1); ret = ret.reshape(-1)
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        n = obs.shape[0]
        for _ in range(epochs):
            perm = torch.randperm(n, device=dev)
            for i in range(0, n, mb):
                idx = perm[i:i + mb]
                v, lp, ent = net.evaluate(obs[idx], act[idx])
                ratio = (lp - oldlp[idx]).exp()
                pg = -torch.min(ratio * adv[idx], ratio.clamp(1 - clip, 1 + clip) * adv[idx]).mean()
                loss = pg + 0.5 * (ret[idx] - v).pow(2).mean() - 0.0 * ent
                opt.zero_grad(set_to_none=True); loss.backward()
                nn.utils.clip_grad_norm_(net.parameters(), 0.5); opt.step()
        col.load_policy(net.state_dict())
        steps += T * args.envs
        if it % 10 == 0 or it == args.iters - 1:
            torch.cuda.synchronize()
            print(f"iter {it:4d}  env-steps {steps/1e6:7.1f} M  mean reward/step {b['rewards'].mean().item():+.4f}  "
                  f"value loss {(ret - b['values'].reshape(-1)).pow(2).mean().item():.4f}  log_std {net.log_std.mean().item():+.3f}  "
                  f"wall {time.time() - t0:6.1f} s", flush=True)


if __name__ == "__main__":
    main()
