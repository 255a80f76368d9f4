"""The N > 1 path on CPU: world_size 2 over gloo.  Covers (a) the one collective of the path -- the rollout gather to the
learner rank and the policy broadcast back -- and (b) the sharding contract: rank r steps global envs [rN, (r+1)N) with
`env_id_offset = rN`, so a sharded run reproduces the unsharded one env by env.  (b) is exercised here with the CPU
oracle standing in for the simulator (same Philox keying as the device; the GPU version of the same property is
tests/test_gpu_parity.py::test_full_size_properties)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_local, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import so100_oracle as O
        from so100_mujoco_rl_amd.rollout import RolloutChunk, gather_rollout, broadcast_policy, shard_range, mean_over_ranks
        total = n_local * world
        lo, hi = shard_range(total, rank, world)
        assert hi - lo == n_local
        flags = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_CUBE_PINNED
        envs = O.OracleBatch(1, n_local, flags, 3, seed=77, env_id0=lo)
        obs = envs.reset().copy()
        rs = np.random.RandomState(5)
        acts_all = rs.uniform(-1, 1, (T, total, 6)).astype(np.float32)
        chunk = RolloutChunk(T, n_local, 15, "cpu")
        for t in range(T):
            a = acts_all[t, lo:hi]
            o2, r, term, trunc = envs.step(a)
            chunk.write(t, torch.from_numpy(obs), torch.from_numpy(a), torch.from_numpy(r.copy()),
                        torch.from_numpy((term | trunc).astype(np.float32)), torch.full((n_local,), float(rank)), torch.zeros(n_local))
            obs = o2.copy()
        full = gather_rollout(chunk.buf, dst=0)
        # bench.py's pattern: two chunk buffers (send AND receive side), async gather of buffer i while buffer 1-i is produced
        bufs = [torch.zeros(T, n_local, 25), torch.zeros(T, n_local, 25)]
        recv = [[torch.zeros(T, n_local, 25) for _ in range(world)] for _ in range(2)] if rank == 0 else None
        pending = [None, None]; seen = []
        ci = 0
        for c in range(5):
            Tc = T if c < 4 else 2                           # trailing partial chunk
            ci ^= 1
            if pending[ci] is not None:
                pending[ci].wait(); pending[ci] = None
                if rank == 0: seen.append([float(r[0, 0, 0]) for r in recv[ci]])
            bufs[ci][:Tc] = 100.0 * c + rank
            pending[ci] = dist.gather(bufs[ci][:Tc], [r[:Tc] for r in recv[ci]] if rank == 0 else None, dst=0, async_op=True)
        for i, wk in enumerate(pending):
            if wk is not None:
                wk.wait()
        if rank == 0:
            assert seen == [[100.0 * c + k for k in range(world)] for c in range(3)], seen       # chunk c was complete when its buffer was reused
            assert [float(r[0, 0, 0]) for r in recv[1]] == [400.0 + k for k in range(world)]    # last (partial) chunk, rank order
            assert float(recv[1][1][2, 0, 0]) == 201.0                                         # rows beyond Tc keep that buffer's previous chunk
            assert [float(r[0, 0, 0]) for r in recv[0]] == [300.0 + k for k in range(world)]
        w = torch.full((4,), float(rank + 1)); broadcast_policy([w], src=0)
        assert torch.all(w == 1.0)
        # the per-chunk statistic of the collector (mean env reward BEFORE the TimeLimit bootstrap) is averaged over the ranks, on every rank
        mine = torch.tensor(float(10*rank + 1)); m = mean_over_ranks(mine)
        assert float(m) == sum(10*k + 1 for k in range(world))/world and float(mine) == 10*rank + 1
        if rank == 0:
            q.put(("ok", full.numpy(), acts_all))
        else:
            assert full is None
            q.put(("ok", None, None))
        dist.barrier()                                       # nobody tears its sockets down while a peer is still receiving
    except Exception as e:                                   # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_sharded_rollout_gather_world2():
    from oracle import so100_oracle as O
    from so100_mujoco_rl_amd.rollout import RolloutChunk
    world, n_local, T = 2, 6, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_local, T, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    if not all(r[0] == "ok" for r in res):
        pytest.fail("worker failed:\n" + "\n".join(str(r[1]) for r in res if r[0] != "ok"))
    full, acts_all = [(r[1], r[2]) for r in res if r[1] is not None][0]
    total = world * n_local
    assert full.shape == (T, total, 25)
    # unsharded single-process run of the same global envs
    flags = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_CUBE_PINNED
    envs = O.OracleBatch(1, total, flags, 3, seed=77, env_id0=0)
    obs = envs.reset().copy()
    u = RolloutChunk(T, total, 15, "cpu").unpack(torch.from_numpy(full))
    for t in range(T):
        np.testing.assert_array_equal(u["obs"][t].numpy(), obs)                       # rank order == env order
        np.testing.assert_array_equal(u["actions"][t].numpy(), acts_all[t])
        o2, r, term, trunc = envs.step(acts_all[t])
        np.testing.assert_array_equal(u["rewards"][t].numpy(), r)
        obs = o2.copy()
    np.testing.assert_array_equal(u["values"][0].numpy(), np.repeat(np.arange(world, dtype=np.float32), n_local))
