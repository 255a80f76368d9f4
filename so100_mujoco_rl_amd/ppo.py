"""Plain-PyTorch PPO on top of the on-device rollout collector -- the learner used by `main.py train` when
stable-baselines3 is not importable (it is not in this image).  Hyper-parameters are SB3's PPO defaults
(ref: main.py:56-64 -> stable_baselines3.PPO("MlpPolicy")): 2x64 tanh towers, gamma 0.99, gae_lambda 0.95, clip 0.2,
lr 3e-4, vf_coef 0.5, max_grad_norm 0.5; epochs 4 instead of 10 (the batch is 262 144 samples per update).
The network's state_dict keys equal SB3's ActorCriticPolicy keys, so checkpoints and RolloutCollector.load_policy()
interoperate with an SB3 policy."""
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
                 vf_coef=0.5, max_grad_norm=0.5, seed=0):
        torch.manual_seed(seed)
        self.net = ActorCritic(obs_dim).to(device)
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, eps=1e-5)
        self.gamma, self.lam, self.clip, self.epochs, self.mb = gamma, gae_lambda, clip, epochs, minibatch
        self.vf_coef, self.max_grad_norm, self.device = vf_coef, max_grad_norm, device

    def update(self, b):
        """b: RolloutCollector.collect() output ([T, N, ...] device tensors + last_obs)."""
        net = self.net
        with torch.no_grad():
            last_v = net.value(b["last_obs"])
            T = b["rewards"].shape[0]
            adv = torch.zeros_like(b["rewards"]); g = torch.zeros_like(last_v)
            for t in reversed(range(T)):                       # GAE; dones[t] ends the episode after step t
                nv = last_v if t == T - 1 else b["values"][t + 1]
                nonterm = 1.0 - b["dones"][t]
                delta = b["rewards"][t] + self.gamma * nv * nonterm - b["values"][t]
                g = delta + self.gamma * self.lam * nonterm * g
                adv[t] = g
            ret = (adv + b["values"]).reshape(-1)
            obs = b["obs"].reshape(-1, b["obs"].shape[-1]); act = b["actions"].reshape(-1, b["actions"].shape[-1])
            oldlp = b["log_probs"].reshape(-1); adv = adv.reshape(-1)
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        n = obs.shape[0]; vloss = 0.0
        for _ in range(self.epochs):
            perm = torch.randperm(n, device=self.device)
            for i in range(0, n, self.mb):
                idx = perm[i:i + self.mb]
                v, lp = net.evaluate(obs[idx], act[idx])
                ratio = (lp - oldlp[idx]).exp()
                pg = -torch.min(ratio * adv[idx], ratio.clamp(1 - self.clip, 1 + self.clip) * adv[idx]).mean()
                vl = (ret[idx] - v).pow(2).mean()
                loss = pg + self.vf_coef * vl
                self.opt.zero_grad(set_to_none=True); loss.backward()
                nn.utils.clip_grad_norm_(net.parameters(), self.max_grad_norm); self.opt.step()
                vloss = vl.item()
        return {"value_loss": vloss, "mean_reward": b["rewards"].mean().item()}
