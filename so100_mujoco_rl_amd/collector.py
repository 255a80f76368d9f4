"""On-device rollout collector: what SB3's `OnPolicyAlgorithm.collect_rollouts` does (policy forward, clip, env step,
buffer writes; ref: main.py:234-238 -> model.learn) with two kernel launches per vectorised step and no host round trip.

    env = So100VecEnv("Env01-v1", 4096)
    col = RolloutCollector(env, sb3_policy.state_dict())          # or RolloutCollector.random_policy(env)
    batch = col.collect(64)                                       # dict of [T, N, ...] device tensors
    col.load_policy(sb3_policy.state_dict())                      # after each learner update

`batch["last_values"]` / GAE are left to the learner (SB3's RolloutBuffer.compute_returns_and_advantage needs them).
"""
import torch

from .lib import POLICY_TENSORS, SB3_STATE_DICT_KEYS
from .rollout import RolloutChunk, gather_rollout


class RolloutCollector:
    def __init__(self, vec_env, state_dict, T=64, persistent=None):
        self.env = vec_env; self.sim = vec_env.sim
        self.T = T
        self.chunk = RolloutChunk(T, self.sim.n, self.sim.obs_dim, self.sim.device)
        self.act = torch.zeros(self.sim.n, 6, device=self.sim.device)
        self.counter = 0
        # one launch per chunk (so100_rollout: lowest latency, but one physics wave per CU => best up to 256 CUs x 64
        # envs) vs two launches per step (so100_policy_forward + so100_step: every lane computes physics => best
        # throughput for large batches).  Measured on MI355X (profiles/r01_k_*): persistent 4096 envs 121 M, 16384 envs 478 M
        # env-steps/s (34 us per step up to 256 workgroups = one per CU); stepwise 65536 envs 0.73 G, 1 M envs 1.33 G.
        self.persistent = (self.sim.n <= 16384) if persistent is None else persistent
        self.load_policy(state_dict)
        self._started = False

    @staticmethod
    def random_policy_state(obs_dim, device, seed=0):
        """SB3 ActorCriticPolicy default initialisation (orthogonal, gains sqrt2 / 0.01 / 1, log_std 0)."""
        g = torch.Generator(device="cpu"); g.manual_seed(seed)

        def lin(o, i, gain):
            w = torch.empty(o, i); torch.nn.init.orthogonal_(w, gain=gain, generator=g)
            return w.to(device), torch.zeros(o, device=device)
        s2 = 2 ** 0.5
        sd = {}
        for tower, key in (("policy_net", "pi"), ("value_net", "vf")):
            w0, b0 = lin(64, obs_dim, s2); w1, b1 = lin(64, 64, s2)
            sd[f"mlp_extractor.{tower}.0.weight"] = w0; sd[f"mlp_extractor.{tower}.0.bias"] = b0
            sd[f"mlp_extractor.{tower}.2.weight"] = w1; sd[f"mlp_extractor.{tower}.2.bias"] = b1
        sd["action_net.weight"], sd["action_net.bias"] = lin(6, 64, 0.01)
        sd["value_net.weight"], sd["value_net.bias"] = lin(1, 64, 1.0)
        sd["log_std"] = torch.zeros(6, device=device)
        return sd

    def load_policy(self, state_dict):
        t = {k: state_dict[SB3_STATE_DICT_KEYS[k]].detach().to(self.sim.device, torch.float32).contiguous() for k in POLICY_TENSORS}
        self.sim.set_policy(t)

    def collect(self, T=None, gather_dst=None):
        """Run T vectorised steps; returns the unpacked chunk (views into a reused buffer).  With torch.distributed
        initialised and gather_dst set, the packed chunk is gathered to that rank (RCCL) and unpacked there."""
        T = self.T if T is None else T
        assert T <= self.T
        if not self._started:
            self.env.reset_tensor(); self._started = True
        if self.persistent:
            self.sim.rollout(self.chunk.buf[:T], self.counter)
            self.counter += T
        else:
            for t in range(T):
                row = self.chunk.buf[t]
                self.sim.policy_forward(self.sim.obs, self.act, self.counter, rollout_row=row)
                self.sim.step(self.act, rollout_row=row)
                self.counter += 1
        buf = self.chunk.buf[:T]
        last_obs = self.sim.obs
        if gather_dst is not None:
            buf = gather_rollout(buf.contiguous(), dst=gather_dst)
            last_obs = gather_rollout(last_obs.unsqueeze(0).contiguous(), dst=gather_dst)      # [1, N_total, obs_dim] on the learner rank
            if buf is None:
                return None
            last_obs = last_obs[0]
        out = self.chunk.unpack(buf)
        out["last_obs"] = last_obs
        return out
