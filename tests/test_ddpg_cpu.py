"""CPU tests of the built-in DDPG learner (so100_mujoco_rl_amd/ddpg.py; ref: main.py:38-55 -> stable_baselines3.DDPG with
net_arch pi=[300, 200], qf=[200, 150] and NormalActionNoise(sigma 0.1)): network shapes and SB3 state_dict keys, the replay buffer's
handle_timeout_termination semantics, one hand-computed gradient step, and learning on a toy env with the VecEnv's tensor surface."""
import numpy as np
import torch

from so100_mujoco_rl_amd.ddpg import DDPG, DDPGPolicy, ReplayBuffer


class _ToyVecEnv:
    """obs = position in [-1, 1]^6 (+ padding to obs_dim); reward = -|pos + 0.5 * action - target|; TimeLimit of `limit` steps with auto-reset."""
    def __init__(self, n, obs_dim=8, limit=20, seed=0):
        self.num_envs, self.obs_dim, self.limit, self.device = n, obs_dim, limit, torch.device("cpu")
        self.g = torch.Generator().manual_seed(seed)
        self.sim = type("Sim", (), {})()
        self.sim.terminal_obs = torch.zeros(n, obs_dim)
        self.target = torch.full((6,), 0.3)

    def _obs(self):
        o = torch.zeros(self.num_envs, self.obs_dim); o[:, :6] = self.pos
        return o

    def reset_tensor(self):
        self.pos = torch.rand(self.num_envs, 6, generator=self.g) * 2 - 1
        self.t = torch.randint(0, self.limit, (self.num_envs,), generator=self.g)
        return self._obs()

    def step_tensor(self, a):
        self.pos = (self.pos + 0.5 * a).clamp(-1, 1)
        rew = -(self.pos - self.target).abs().sum(-1)
        self.t += 1
        trunc = self.t >= self.limit
        if trunc.any():
            self.sim.terminal_obs[trunc] = self._obs()[trunc]
            self.pos[trunc] = torch.rand(int(trunc.sum()), 6, generator=self.g) * 2 - 1
            self.t[trunc] = 0
        return self._obs(), rew, trunc.to(torch.uint8), trunc.to(torch.uint8)


def test_network_shapes_and_sb3_keys():
    net = DDPGPolicy(15)
    keys = set(net.state_dict())
    for k in ("actor.mu.0.weight", "actor.mu.2.weight", "actor.mu.4.bias", "critic.qf0.0.weight", "critic.qf0.4.weight",
              "actor_target.mu.0.weight", "critic_target.qf0.4.bias"):
        assert k in keys
    sd = net.state_dict()
    assert tuple(sd["actor.mu.0.weight"].shape) == (300, 15) and tuple(sd["actor.mu.2.weight"].shape) == (200, 300)      # pi=[300, 200]
    assert tuple(sd["actor.mu.4.weight"].shape) == (6, 200)
    assert tuple(sd["critic.qf0.0.weight"].shape) == (200, 21) and tuple(sd["critic.qf0.2.weight"].shape) == (150, 200)  # qf=[200, 150]
    a = net.mean_action(torch.randn(5, 15) * 100)
    assert a.shape == (5, 6) and a.abs().max() <= 1.0                                                                   # tanh-squashed
    assert all(torch.equal(sd[f"actor.{k}"], sd[f"actor_target.{k}"]) for k in ("mu.0.weight", "mu.4.bias"))


def test_replay_stores_terminal_observation_and_timeout_flag():
    d = DDPG(4, "cpu", act_dim=2, buffer_size=8, learning_starts=0, batch_size=2)
    obs = torch.arange(12.).reshape(3, 4); nxt = obs + 100; tobs = obs + 1000
    act = torch.zeros(3, 2); rew = torch.tensor([1., 2., 3.])
    done = torch.tensor([0, 1, 1], dtype=torch.uint8); trunc = torch.tensor([0, 1, 0], dtype=torch.uint8)
    d.store(obs, act, rew, nxt, done, trunc, tobs)
    b = d.buf
    assert len(b) == 3
    assert torch.equal(b.next_obs[0], nxt[0])                           # running episode: the next observation
    assert torch.equal(b.next_obs[1], tobs[1]) and b.done[1] == 0       # truncated: terminal obs, bootstraps through it
    assert torch.equal(b.next_obs[2], tobs[2]) and b.done[2] == 1       # terminated: terminal obs, no bootstrap
    # ring: 3 + 3 + 3 transitions into 8 slots wrap around
    d.store(obs, act, rew, nxt, done, trunc, tobs); d.store(obs + 0.5, act, rew, nxt, done, trunc, tobs)
    assert len(b) == 8 and b.pos == 1 and torch.equal(b.obs[0], obs[2] + 0.5)


def test_one_gradient_step_matches_hand_computation():
    torch.manual_seed(0)
    d = DDPG(5, "cpu", act_dim=3, buffer_size=64, learning_starts=0, batch_size=64, tau=0.25)
    o = torch.randn(64, 5); o2 = torch.randn(64, 5); a = torch.rand(64, 3) * 2 - 1; r = torch.randn(64); done = (torch.rand(64) < 0.3).float()
    d.buf.add(o, o2, a, r, done)
    d.buf.sample = lambda n, g=None: (o, o2, a, r, done)              # the whole buffer, in order
    import copy
    ref = copy.deepcopy(d.net)
    with torch.no_grad():
        target = r + (1 - done) * 0.99 * ref.critic_target(o2, ref.actor_target(o2))
    exp_critic = ((ref.critic(o, a) - target) ** 2).mean().item()
    actor0 = {k: v.clone() for k, v in d.net.actor.state_dict().items()}
    st = d.train_step()
    np.testing.assert_allclose(st["critic_loss"], exp_critic, rtol=1e-6)
    # Polyak: target = (1 - tau) * old target + tau * NEW online weights
    for k, v in d.net.actor_target.state_dict().items():
        np.testing.assert_allclose(v.numpy(), (0.75 * actor0[k] + 0.25 * d.net.actor.state_dict()[k]).numpy(), atol=1e-6)
    assert any(not torch.equal(actor0[k], v) for k, v in d.net.actor.state_dict().items())


def test_learns_the_toy_task():
    env = _ToyVecEnv(32)
    d = DDPG(env.obs_dim, "cpu", buffer_size=20000, learning_starts=256, batch_size=128, gradient_steps=2, seed=1)
    obs, first = d.learn_steps(env, 10)
    for _ in range(6):
        obs, stats = d.learn_steps(env, 50, obs)
    assert d.n_updates > 500
    # deterministic policy: one step from anywhere lands near the target
    env2 = _ToyVecEnv(64, seed=5); o = env2.reset_tensor()
    _, r_pol, _, _ = env2.step_tensor(d.act(o, deterministic=True))
    env3 = _ToyVecEnv(64, seed=5); o = env3.reset_tensor()
    _, r_zero, _, _ = env3.step_tensor(torch.zeros(64, 6))
    assert r_pol.mean() > r_zero.mean() + 1.0, (r_pol.mean(), r_zero.mean())
    assert stats["mean_reward"] > first["mean_reward"] + 0.5
