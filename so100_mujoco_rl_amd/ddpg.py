"""Plain-PyTorch DDPG over the batched VecEnv's tensor API -- the learner behind `main.py -a DDPG train` when stable-baselines3 is
not importable (it is not in this image).  Restates the reference's DDPG branch (ref: main.py:38-55 ->
stable_baselines3.DDPG("MlpPolicy", policy_kwargs=dict(net_arch=dict(pi=[300, 200], qf=[200, 150])),
action_noise=NormalActionNoise(sigma=0.1))) with SB3 2.6.0's DDPG defaults: ReLU towers, tanh-squashed actor, ONE critic, no target
policy smoothing, lr 1e-3 (Adam), gamma 0.99, tau 0.005 (Polyak), batch 256, learning_starts 100 transitions, replay buffer with
`handle_timeout_termination` semantics (a TimeLimit truncation stores the terminal observation as next_obs and done = 0, so the target
bootstraps through it; a genuine termination stores done = 1).

Differences, forced by the env being N envs wide (documented, not SB3's numbers): one vector step adds N transitions, so
`gradient_steps` updates run per vector step on minibatches of `batch_size` (SB3: train_freq = 1 step of ONE env, 1 update of 256);
the replay buffer lives on the device.  The reference builds its action noise with `np.zeros(2)` / `0.1 * np.ones(2)` (main.py:42-45, a
left-over of the balancing robot the comment mentions): two entries do not broadcast against this arm's six actions, so upstream's DDPG
branch raises on its first step; the noise here has one entry per action, sigma 0.1.
state_dict keys equal SB3's TD3Policy (`actor.mu.*`, `critic.qf0.*`, `actor_target.*`, `critic_target.*`)."""
import torch
import torch.nn as nn


def _mlp(i, hidden, o, squash):
    layers, d = [], i
    for h in hidden:
        layers += [nn.Linear(d, h), nn.ReLU()]; d = h
    layers.append(nn.Linear(d, o))
    if squash:
        layers.append(nn.Tanh())
    return nn.Sequential(*layers)


class _Actor(nn.Module):
    def __init__(self, obs_dim, act_dim, arch):
        super().__init__(); self.mu = _mlp(obs_dim, arch, act_dim, True)

    def forward(self, obs):
        return self.mu(obs)


class _Critic(nn.Module):
    def __init__(self, obs_dim, act_dim, arch):
        super().__init__(); self.qf0 = _mlp(obs_dim + act_dim, arch, 1, False)

    def forward(self, obs, act):
        return self.qf0(torch.cat([obs, act], -1)).squeeze(-1)


class DDPGPolicy(nn.Module):
    def __init__(self, obs_dim, act_dim=6, pi=(300, 200), qf=(200, 150)):
        super().__init__()
        self.actor, self.critic = _Actor(obs_dim, act_dim, pi), _Critic(obs_dim, act_dim, qf)
        self.actor_target, self.critic_target = _Actor(obs_dim, act_dim, pi), _Critic(obs_dim, act_dim, qf)
        self.actor_target.load_state_dict(self.actor.state_dict()); self.critic_target.load_state_dict(self.critic.state_dict())
        for p in list(self.actor_target.parameters()) + list(self.critic_target.parameters()):
            p.requires_grad_(False)

    def mean_action(self, obs):                      # the deterministic policy (what `test` / `record` replay)
        return self.actor(obs)


class ReplayBuffer:
    """Ring buffer of transitions on the device; add() takes a whole vector step."""
    def __init__(self, capacity, obs_dim, act_dim, device):
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        self.obs, self.next_obs, self.act, self.rew, self.done = z(capacity, obs_dim), z(capacity, obs_dim), z(capacity, act_dim), z(capacity), z(capacity)
        self.capacity, self.pos, self.full, self.device = capacity, 0, False, device

    def __len__(self):
        return self.capacity if self.full else self.pos

    def add(self, obs, next_obs, act, rew, done):
        n = obs.shape[0]
        if n > self.capacity:
            raise ValueError(f"replay buffer of {self.capacity} transitions cannot take a vector step of {n}")
        idx = (torch.arange(n, device=self.device) + self.pos) % self.capacity
        self.obs[idx], self.next_obs[idx], self.act[idx], self.rew[idx], self.done[idx] = obs, next_obs, act, rew, done
        self.full = self.full or self.pos + n >= self.capacity
        self.pos = (self.pos + n) % self.capacity

    def sample(self, batch, gen=None):
        idx = torch.randint(0, len(self), (batch,), device=self.device, generator=gen)
        return self.obs[idx], self.next_obs[idx], self.act[idx], self.rew[idx], self.done[idx]


class DDPG:
    def __init__(self, obs_dim, device, act_dim=6, lr=1e-3, gamma=0.99, tau=0.005, batch_size=256, buffer_size=1_000_000,
                 learning_starts=100, gradient_steps=1, noise_sigma=0.1, seed=0):
        torch.manual_seed(seed)
        self.net = DDPGPolicy(obs_dim, act_dim).to(device)
        self.actor_opt = torch.optim.Adam(self.net.actor.parameters(), lr=lr)
        self.critic_opt = torch.optim.Adam(self.net.critic.parameters(), lr=lr)
        self.buf = ReplayBuffer(buffer_size, obs_dim, act_dim, device)
        self.gamma, self.tau, self.batch_size, self.learning_starts, self.gradient_steps = gamma, tau, batch_size, learning_starts, gradient_steps
        self.sigma, self.device = noise_sigma, device
        self._gen = torch.Generator(device=device); self._gen.manual_seed(seed + 1)
        self.n_updates = 0

    @torch.no_grad()
    def act(self, obs, deterministic=False):
        """SB3's OffPolicyAlgorithm._sample_action: uniform random actions before learning_starts, then mu(obs) + N(0, sigma), clipped to [-1, 1]."""
        if deterministic:
            return self.net.actor(obs)
        if len(self.buf) < self.learning_starts:
            return torch.rand(obs.shape[0], self.net.actor.mu[-2].out_features, device=self.device, generator=self._gen) * 2 - 1
        a = self.net.actor(obs)
        return (a + self.sigma * torch.randn(a.shape, device=self.device, generator=self._gen)).clamp(-1, 1)

    def store(self, obs, act, rew, next_obs, done, trunc, terminal_obs):
        """done / trunc: the env's uint8 columns AFTER auto-reset (next_obs of a finished env is its reset observation; `terminal_obs` holds
        the observation the episode ended on).  handle_timeout_termination: truncated => bootstrap through terminal_obs (done stored as 0)."""
        ended = done.bool()
        nxt = torch.where(ended.unsqueeze(-1), terminal_obs, next_obs)
        self.buf.add(obs, nxt, act, rew, (ended & ~trunc.bool()).float())

    def train_step(self):
        """One SB3 TD3.train() gradient step with policy_delay = 1, target_policy_noise = 0 (that is what SB3's DDPG is)."""
        o, o2, a, r, d = self.buf.sample(self.batch_size, self._gen)
        with torch.no_grad():
            target = r + (1.0 - d) * self.gamma * self.net.critic_target(o2, self.net.actor_target(o2))
        critic_loss = nn.functional.mse_loss(self.net.critic(o, a), target)
        self.critic_opt.zero_grad(set_to_none=True); critic_loss.backward(); self.critic_opt.step()
        actor_loss = -self.net.critic(o, self.net.actor(o)).mean()
        self.actor_opt.zero_grad(set_to_none=True); actor_loss.backward(); self.actor_opt.step()
        with torch.no_grad():                         # Polyak update of both targets
            for src, dst in ((self.net.actor, self.net.actor_target), (self.net.critic, self.net.critic_target)):
                for p, pt in zip(src.parameters(), dst.parameters()):
                    pt.mul_(1.0 - self.tau).add_(p, alpha=self.tau)
        self.n_updates += 1
        return {"critic_loss": critic_loss.item(), "actor_loss": actor_loss.item()}

    def learn_steps(self, env, n_vec_steps, obs=None):
        """Drive `env` (tensor API: reset_tensor / step_tensor / sim.terminal_obs) for n vector steps; returns (last obs, statistics)."""
        if obs is None:
            obs = env.reset_tensor().clone()
        rew_sum, stats = 0.0, {}
        for _ in range(n_vec_steps):
            a = self.act(obs).contiguous()
            nobs, rew, done, trunc = env.step_tensor(a)
            self.store(obs, a, rew, nobs, done, trunc, env.sim.terminal_obs)
            rew_sum += rew.mean().item()
            obs = nobs.clone()
            if len(self.buf) >= max(self.learning_starts, self.batch_size):
                for _ in range(self.gradient_steps):
                    stats = self.train_step()
        stats["mean_reward"] = rew_sum / max(1, n_vec_steps)
        return obs, stats
