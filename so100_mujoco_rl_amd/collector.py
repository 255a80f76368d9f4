"""On-device rollout collector: what SB3's `OnPolicyAlgorithm.collect_rollouts` does (policy forward, clip, env step,
buffer writes; ref: main.py:234-238 -> model.learn) with two kernel launches per vectorised step and no host round trip.

    env = So100VecEnv("Env01-v1", 4096)
    col = RolloutCollector(env, sb3_policy.state_dict())          # or RolloutCollector.random_policy(env)
    batch = col.collect(64)                                       # dict of [T, N, ...] device tensors
    col.load_policy(sb3_policy.state_dict())                      # after each learner update

TimeLimit truncations are bootstrapped like SB3's collect_rollouts does (`rewards += gamma * V(terminal_observation)`,
done code 2 in the rollout row; see rollout.bootstrap_truncated); `batch["last_values"]` / GAE are left to the learner
(SB3's RolloutBuffer.compute_returns_and_advantage needs them).
"""
import torch

from .lib import POLICY_TENSORS, SB3_STATE_DICT_KEYS
from .rollout import RolloutChunk, bootstrap_truncated, gather_rollout, mean_over_ranks


class RolloutCollector:
    def __init__(self, vec_env, state_dict, T=64, persistent=None, gamma=0.99, bootstrap_truncated=True):
        self.env = vec_env; self.sim = vec_env.sim
        self.T = T
        self.chunk = RolloutChunk(T, self.sim.n, self.sim.obs_dim, self.sim.device)
        # SB3 adds gamma * V(terminal_observation) to the reward of a step that ended by TimeLimit truncation alone
        # (OnPolicyAlgorithm.collect_rollouts); every episode end of Env01/02/06 is such a truncation.  The kernels mark them
        # (done column == 2) and deliver the terminal observations; the value net is applied here, before any gather.
        self.gamma = gamma; self.bootstrap = bootstrap_truncated
        self.tobs = torch.zeros(T, self.sim.n, self.sim.obs_dim, device=self.sim.device) if bootstrap_truncated else None
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
        """Copies the weights into buffers this collector owns (the kernels read THESE: a learner that keeps updating its
        parameters in place -- or asynchronously, from replayed hipGraphs -- cannot change the policy under a running rollout)."""
        with torch.no_grad():
            if getattr(self, "_wbuf", None) is None:
                self._wbuf = {k: state_dict[SB3_STATE_DICT_KEYS[k]].detach().to(self.sim.device, torch.float32).clone().contiguous() for k in POLICY_TENSORS}
                self.sim.set_policy(self._wbuf)
            else:
                for k in POLICY_TENSORS:
                    self._wbuf[k].copy_(state_dict[SB3_STATE_DICT_KEYS[k]])

    @torch.no_grad()
    def _value(self, obs):
        """V(obs) with the weights the kernels are using (SB3 value tower: 2 x 64 tanh + linear)."""
        t = self.sim._policy_tensors
        h = torch.tanh(torch.addmm(t["vf_b0"], obs, t["vf_w0"].t()))
        h = torch.tanh(torch.addmm(t["vf_b1"], h, t["vf_w1"].t()))
        return torch.addmm(t["v_b"], h, t["v_w"].t()).squeeze(-1)

    def state_dict(self):
        """What a resumed run needs beside the sim checkpoint: the policy-noise Philox counter (so that noise is not
        reused from 0) and whether the first reset has happened."""
        return {"counter": int(self.counter), "started": bool(self._started)}

    def load_state_dict(self, sd):
        self.counter = int(sd["counter"]); self._started = bool(sd["started"])

    def collect(self, T=None, gather_dst=None):
        """Run T vectorised steps; returns the unpacked chunk (views into a reused buffer).  With torch.distributed
        initialised and gather_dst set, the packed chunk is gathered to that rank (RCCL) and unpacked there."""
        T = self.T if T is None else T
        assert T <= self.T
        if not self._started:
            self.env.reset_tensor(); self._started = True
        if self.persistent:
            self.sim.rollout(self.chunk.buf[:T], self.counter, terminal_obs_chunk=None if self.tobs is None else self.tobs[:T])
            self.counter += T
        else:
            for t in range(T):
                row = self.chunk.buf[t]
                self.sim.policy_forward(self.sim.obs, self.act, self.counter, rollout_row=row)
                self.sim.step(self.act, rollout_row=row, terminal_obs=None if self.tobs is None else self.tobs[t])
                self.counter += 1
        buf = self.chunk.buf[:T]
        o = self.sim.obs_dim
        # statistics are taken from the ENV's rewards, before the bootstrap adds the critic's gamma * V(terminal_observation) to the
        # truncated steps: model selection / early stopping / the "reward/step" log line must not depend on the value estimates
        raw_mean = buf[..., o + 6].mean()
        if self.bootstrap:
            bootstrap_truncated(buf[..., o + 6], buf[..., o + 7], self.tobs[:T], self._value, self.gamma)
        last_obs = self.sim.obs
        if gather_dst is not None:
            raw_mean = mean_over_ranks(raw_mean)
            buf = gather_rollout(buf.contiguous(), dst=gather_dst)
            last_obs = gather_rollout(last_obs.unsqueeze(0).contiguous(), dst=gather_dst)      # [1, N_total, obs_dim] on the learner rank
            if buf is None:
                return None
            last_obs = last_obs[0]
        out = self.chunk.unpack(buf)
        out["last_obs"] = last_obs
        out["raw_reward_mean"] = raw_mean                    # 0-d tensor: mean env reward per step of this chunk (all ranks), bootstrap excluded
        return out
