"""Plain-PyTorch PPO on top of the on-device rollout collector -- the learner used by `main.py train` when
stable-baselines3 is not importable (it is not in this image).  Network and loss follow SB3's PPO
(ref: main.py:56-64 -> stable_baselines3.PPO("MlpPolicy")): 2x64 tanh towers, gamma 0.99, gae_lambda 0.95, clip 0.2,
lr 3e-4, vf_coef 0.5, max_grad_norm 0.5.  NOT SB3's defaults: 4 epochs instead of 10, minibatches of 32 768 instead of
64, rollouts of 64 steps x N envs instead of 2048 x 1 (the batch is 262 144 samples per update at 4096 envs).
TimeLimit truncations are bootstrapped by the collector exactly as SB3 does (rewards += gamma V(terminal_obs)), so
`dones` below ends the GAE recursion with the right target for both terminations and truncations.  On a GPU the
update CAN be replayed from two captured hipGraphs (GAE pass, minibatch step; use_graph=True): opt-in, see __init__.
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
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, eps=1e-5, capturable=on_gpu)
        self.gamma, self.lam, self.clip, self.epochs, self.mb = gamma, gae_lambda, clip, epochs, minibatch
        self.vf_coef, self.max_grad_norm, self.device = vf_coef, max_grad_norm, device
        # One PPO update is ~40 minibatch steps of ~100 tiny kernels each plus a 64-step GAE recursion: launch-bound in eager
        # mode (~60 ms for a 262 144-sample batch).  With use_graph the GAE pass and the minibatch step are captured once as
        # hipGraphs over static buffers and replayed (~5 ms); the arithmetic is the same.
        # OPT-IN since round 2.  What went wrong with it as the default: the simulator's kernels read the learner's LIVE parameters
        # (RolloutCollector.load_policy aliases them) and are launched on the raw current stream; the next rollout did not wait for
        # the end of the last replayed minibatch step (graph work is not joined to the legacy null stream the way eager work is),
        # so it sampled with half-updated weights -> log-probs inconsistent with the actions -> policy collapse after ~60 updates
        # (Env01 / Env05, several seeds; never in eager mode; gone with a device synchronisation after the replays, or with cloned
        # weights: tools/ppo_graph_check.py).  update() now ends the replayed path with torch.cuda.synchronize().
        self.use_graph = use_graph and on_gpu and os.environ.get("SO100_PPO_GRAPH", "1") != "0"
        self._g = None

    # ---- the two pieces of an update, written over the static buffers self._s (also what gets captured) ---------------
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

    def _capture(self):
        """torch's whole-network capture recipe: a few real steps on a side stream, then one captured step."""
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._gae()
            for _ in range(3):
                self.opt.zero_grad(set_to_none=True); self._step()
        torch.cuda.current_stream().wait_stream(s)
        g_gae = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_gae):
            self._gae()
        g_step = torch.cuda.CUDAGraph(); self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g_step):
            self._step()
        self._g = (g_gae, g_step)

    def update(self, b):
        """b: RolloutCollector.collect() output ([T, N, ...] device tensors + last_obs)."""
        if getattr(self, "_s", None) is None or self._shape != tuple(b["obs"].shape):
            self._alloc(b); self._g = None
        S = self._s
        for k in ("obs", "actions", "rewards", "dones", "values", "log_probs", "last_obs"):
            S[k].copy_(b[k])
        n = S["ret"].numel(); mb = S["idx"].numel()
        graph = self.use_graph and n % mb == 0
        if graph and self._g is None:
            S["idx"].copy_(torch.randperm(n, device=self.device)[:mb])
            self._capture()
        if graph: self._g[0].replay()
        else: self._gae()
        for _ in range(self.epochs):
            perm = torch.randperm(n, device=self.device)
            for i in range(0, n, mb):
                if graph:
                    S["idx"].copy_(perm[i:i + mb]); self._g[1].replay()
                else:
                    S["idx"] = perm[i:i + mb]
                    self.opt.zero_grad(set_to_none=True); self._step()
        if graph: torch.cuda.synchronize()                 # the replays must have finished before anyone reads the parameters (see __init__)
        else: S["idx"] = torch.zeros(mb, dtype=torch.long, device=self.device)
        return {"value_loss": S["vl"].item(), "mean_reward": S["rewards"].mean().item()}
