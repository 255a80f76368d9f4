"""Plain-PyTorch PPO on top of the on-device rollout collector -- the learner used by `main.py train` when
stable-baselines3 is not importable (it is not in this image).  Network and loss follow SB3's PPO
(ref: main.py:56-64 -> stable_baselines3.PPO("MlpPolicy")): 2x64 tanh towers, gamma 0.99, gae_lambda 0.95, clip 0.2,
lr 3e-4, vf_coef 0.5, max_grad_norm 0.5.  NOT SB3's defaults: 4 epochs instead of 10, minibatches of 32 768 instead of
64, rollouts of 64 steps x N envs instead of 2048 x 1 (the batch is 262 144 samples per update at 4096 envs).
TimeLimit truncations are bootstrapped by the collector exactly as SB3 does (rewards += gamma V(terminal_obs)), so
`dones` below ends the GAE recursion with the right target for both terminations and truncations.  (Round 1's
hipGraph-replayed update is gone: see PPO.__init__.)
The network's state_dict keys equal SB3's ActorCriticPolicy keys, so checkpoints and RolloutCollector.load_policy()
interoperate with an SB3 policy."""
import os

import torch
import torch.nn as nn


class ActorCritic(nn.Module):
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

    def value(self, obs):
        return self.value_net(self.mlp_extractor["value_net"](obs)).squeeze(-1)

    def mean_action(self, obs):
        return self.action_net(self.mlp_extractor["policy_net"](obs))

    def evaluate(self, obs, act):
        mean = self.mean_action(obs)
        std = self.log_std.exp()
        logp = (-0.5 * ((act - mean) / std) ** 2 - self.log_std - 0.9189385332046727).sum(-1)
        return self.value(obs), logp


class PPO:
    def __init__(self, obs_dim, device, lr=3e-4, gamma=0.99, gae_lambda=0.95, clip=0.2, epochs=4, minibatch=32768,
                 vf_coef=0.5, max_grad_norm=0.5, seed=0, use_graph=False):
        torch.manual_seed(seed)
        self.net = ActorCritic(obs_dim).to(device)
        on_gpu = torch.device(device).type == "cuda"
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, eps=1e-5)
        self.gamma, self.lam, self.clip, self.epochs, self.mb = gamma, gae_lambda, clip, epochs, minibatch
        self.vf_coef, self.max_grad_norm, self.device = vf_coef, max_grad_norm, device
        # use_graph is accepted and ignored.  Round 1 replayed the update from two captured hipGraphs (+14 % end to end when the
        # rollout ran at 54 M env-steps/s).  With the collector's truncation bootstrap in the loop it produced policy collapses
        # after ~60 updates: replays launched on the legacy default stream were not ordered against the next rollout's raw kernel
        # launch, which read half-updated parameters; fencing fixed the collapses, but from its third update on the replayed
        # learner still differed from the eager one on identical inputs (5e-3 in the weights, cause not found), so it was removed
        # rather than shipped as an option nobody can vouch for.  The eager update is ~60 ms per 262 144 samples.
        del use_graph

    # ---- the two pieces of an update, written over the buffers self._s -------------------------------------------------
    def _gae(self):
        S = self._s; net = self.net
        with torch.no_grad():
            last_v = net.value(S["last_obs"])
            T = S["rewards"].shape[0]
            g = torch.zeros_like(last_v)
            for t in reversed(range(T)):                       # GAE; dones[t] ends the episode after step t
                nv = last_v if t == T - 1 else S["values"][t + 1]
                nonterm = 1.0 - S["dones"][t]
                delta = S["rewards"][t] + self.gamma * nv * nonterm - S["values"][t]
                g = delta + self.gamma * self.lam * nonterm * g
                S["adv"][t] = g
            S["ret"].copy_((S["adv"] + S["values"]).reshape(-1))
            a = S["adv"].reshape(-1)
            S["adv_n"].copy_((a - a.mean()) / (a.std() + 1e-8))

    def _step(self):
        S = self._s; net = self.net; idx = S["idx"]
        obs = S["obs"].reshape(-1, S["obs"].shape[-1]); act = S["actions"].reshape(-1, S["actions"].shape[-1])
        v, lp = net.evaluate(obs.index_select(0, idx), act.index_select(0, idx))
        adv = S["adv_n"].index_select(0, idx)
        ratio = (lp - S["log_probs"].reshape(-1).index_select(0, idx)).exp()
        pg = -torch.min(ratio * adv, ratio.clamp(1 - self.clip, 1 + self.clip) * adv).mean()
        vl = (S["ret"].index_select(0, idx) - v).pow(2).mean()
        loss = pg + self.vf_coef * vl
        loss.backward()
        nn.utils.clip_grad_norm_(net.parameters(), self.max_grad_norm); self.opt.step()
        S["vl"].copy_(vl.detach())

    def _alloc(self, b):
        dev = self.device
        self._s = {k: torch.empty(b[k].shape, dtype=torch.float32, device=dev) for k in ("obs", "actions", "rewards", "dones", "values", "log_probs", "last_obs")}
        T, N = b["rewards"].shape
        self._s.update(adv=torch.zeros(T, N, device=dev), ret=torch.zeros(T * N, device=dev), adv_n=torch.zeros(T * N, device=dev),
                       idx=torch.zeros(min(self.mb, T * N), dtype=torch.long, device=dev), vl=torch.zeros((), device=dev))
        self._shape = tuple(b["obs"].shape)

    def update(self, b):
        """b: RolloutCollector.collect() output ([T, N, ...] device tensors + last_obs)."""
        if getattr(self, "_s", None) is None or self._shape != tuple(b["obs"].shape):
            self._alloc(b)
        S = self._s
        for k in ("obs", "actions", "rewards", "dones", "values", "log_probs", "last_obs"):
            S[k].copy_(b[k])
        n = S["ret"].numel(); mb = S["idx"].numel()
        self._gae()
        for _ in range(self.epochs):
            perm = torch.randperm(n, device=self.device)
            for i in range(0, n, mb):
                S["idx"] = perm[i:i + mb]
                self.opt.zero_grad(set_to_none=True); self._step()
        S["idx"] = torch.zeros(mb, dtype=torch.long, device=self.device)
        # mean_reward = the ENV's mean reward per step (the collector takes it before its TimeLimit bootstrap adds gamma * V to the
        # truncated steps); S["rewards"] holds the bootstrapped rewards the advantages are computed from
        raw = b.get("raw_reward_mean")
        return {"value_loss": S["vl"].item(), "mean_reward": (raw if raw is not None else S["rewards"].mean()).item(),
                "mean_bootstrapped_reward": S["rewards"].mean().item()}
