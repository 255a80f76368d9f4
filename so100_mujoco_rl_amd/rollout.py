"""Env sharding across the GPUs of one node and the ONE collective of the path: gathering rollouts to the learner.

Envs are independent (each reference env owns its own MjData, ref: envs/env_base_01.py:42-51), so the batch shards
embarrassingly: rank r owns global envs [r*N, (r+1)*N) (`env_id_offset = r*N` keys the device RNG, so a sharded run
reproduces the unsharded one env by env).  No collective runs while stepping.  Once per rollout chunk the packed
[T, N_local, k] block (obs, action, reward, done, value, log-prob) goes to the learner rank with a single RCCL gather
over xGMI (`torch.distributed` backend "nccl" on ROCm); on CPU the same code runs over gloo (tests).
"""
import torch
import torch.distributed as dist


def shard_range(num_envs_total, rank, world_size):
    """Contiguous env range of `rank`; sizes differ by at most one."""
    base, rem = divmod(num_envs_total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_width(obs_dim, act_dim=6):
    return obs_dim + act_dim + 4


class RolloutChunk:
    """[T, N, k] on-device buffer: obs | action | reward | done | value | logp.  The done column is 0 (running),
    1 (terminated) or 2 (TimeLimit-truncated only), as the kernels write it (include/so100_sim.h)."""

    def __init__(self, T, n, obs_dim, device, act_dim=6):
        self.T, self.n, self.obs_dim, self.act_dim = T, n, obs_dim, act_dim
        self.buf = torch.zeros(T, n, pack_width(obs_dim, act_dim), device=device)

    def write(self, t, obs, act, rew, done, value, logp):
        row = self.buf[t]
        o, a = self.obs_dim, self.act_dim
        row[:, :o] = obs; row[:, o:o + a] = act
        row[:, o + a] = rew; row[:, o + a + 1] = done; row[:, o + a + 2] = value; row[:, o + a + 3] = logp

    def unpack(self, buf=None):
        b = self.buf if buf is None else buf
        o, a = self.obs_dim, self.act_dim
        code = b[..., o + a + 1]
        return {"obs": b[..., :o], "actions": b[..., o:o + a], "rewards": b[..., o + a], "dones": (code != 0).to(b.dtype),
                "truncated": code == 2, "values": b[..., o + a + 2], "log_probs": b[..., o + a + 3]}


def bootstrap_truncated(rewards, done_code, terminal_obs, value_fn, gamma):
    """SB3's TimeLimit handling in OnPolicyAlgorithm.collect_rollouts: where an episode was only truncated
    (`infos["TimeLimit.truncated"]`, done code 2) the learner must not treat the step as terminal, so
    `rewards += gamma * V(terminal_observation)` there, in place.  Dense and sync-free: V is evaluated on the whole
    [T, N, obs_dim] terminal-observation chunk (entries that are not episode ends hold stale values and are masked)."""
    mask = done_code == 2
    v = value_fn(terminal_obs.reshape(-1, terminal_obs.shape[-1])).reshape(rewards.shape)
    rewards.add_(torch.where(mask, gamma * v, torch.zeros_like(v)))
    return mask


def gather_rollout(chunk_buf, dst=0, group=None):
    """Gather every rank's [T, N_local, k] chunk on `dst`; returns [T, N_total, k] there (rank order = env order),
    None elsewhere.  All ranks must hold the same N_local (weak scaling: fixed envs per GPU)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return chunk_buf
    world = dist.get_world_size(group); rank = dist.get_rank(group)
    out = [torch.empty_like(chunk_buf) for _ in range(world)] if rank == dst else None
    dist.gather(chunk_buf, out, dst=dst, group=group)
    return torch.cat(out, dim=1) if rank == dst else None


def mean_over_ranks(x, group=None):
    """mean of a per-rank scalar statistic (equal shard sizes); every rank calls it"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        x = x.clone(); dist.all_reduce(x, group=group); x = x / dist.get_world_size(group)
    return x


def broadcast_policy(tensors, src=0, group=None):
    """Send the learner's updated policy weights (SB3 MlpPolicy: ~11k parameters) back to every rank."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
