#!/usr/bin/env python3
"""bench.py -- env-steps/s of the Env01 PPO rollout at 4096 envs/GPU (BASELINE.json metric).

One "step" = one vectorised env step of the rollout loop on every rank: policy forward (SB3 MlpPolicy shape:
separate 2x64 tanh towers for pi and V, state-independent log-std Gaussian), action sampling + clipping, the fused
HIP env step (reward -> ctrl -> 16 physics substeps -> obs -> TimeLimit -> auto-reset), and the write of
obs/action/reward/done/value/log-prob into the on-device rollout buffer; every ROLLOUT_T steps the rollout chunk is
gathered to the learner rank over RCCL (N > 1 only).  Default collector (`--policy persistent`): ONE launch per rollout chunk of 64 steps (so100_rollout: persistent
workgroups, policy phase on all waves, physics phase on wave 0, env state in registers, weights in LDS).
`--policy fused`: two launches per step (so100_policy_forward + so100_step), no PyTorch op in the loop.
`--policy torch`: the same rollout with the policy as plain PyTorch ops (what an unmodified SB3 policy costs).
--steps / --warmup count vectorised env steps in every mode (a trailing partial chunk is one shorter launch).  Workload = BASELINE.json configs[1]: Env01, 4096 envs per
GPU, contact disabled / no constraint solver (cube pinned), synthetic randomized-reset batches, random-init policy.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (so100_step_fused) with the algorithmic
452 B/env-step of SURVEY.md section 8(d); `cpu_baseline` times the CPU oracle (a port, not the reference) on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = 452.0        # SURVEY.md section 8(d), Env01, fp32 SoA, 16 substeps fused
FLOP_PER_ENV_STEP_SURVEY = 6.0e4  # SURVEY.md section 8(d) estimate, constraint-free
# measured: PMC SQ_INSTS_VALU = 22.95 k VALU instructions per env-step lane (profiles/r01_c), of which ~45 % are FMAs
# (ISA count: 1017 fma/fmac of 2100 float ops per substep) => ~1.45 FLOP per instruction => 3.3e4 FLOP per env-step.
# The VALU fraction below uses the MEASURED figure (the survey estimate would overstate utilisation 1.8x).
FLOP_PER_ENV_STEP = 3.3e4
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
VALU_PEAK_TFLOPS = 157.3
ROLLOUT_T = 64


class MlpPolicy:
    """SB3 ActorCriticPolicy("MlpPolicy") forward for a Box action space, random init, on the device."""

    def __init__(self, obs_dim, act_dim, device, seed):
        g = torch.Generator(device="cpu"); g.manual_seed(seed)
        def lin(i, o, gain):
            w = torch.empty(o, i); torch.nn.init.orthogonal_(w, gain=gain, generator=g)
            return w.t().contiguous().to(device), torch.zeros(o, device=device)
        s2 = 2 ** 0.5
        self.pi = [lin(obs_dim, 64, s2), lin(64, 64, s2)]; self.vf = [lin(obs_dim, 64, s2), lin(64, 64, s2)]
        self.mu = lin(64, act_dim, 0.01); self.v = lin(64, 1, 1.0)
        self.log_std = torch.zeros(act_dim, device=device)
        self.act_dim = act_dim

    @torch.no_grad()
    def forward(self, obs, noise):
        h = obs
        for w, b in self.pi:
            h = torch.tanh(torch.addmm(b, h, w))
        mean = torch.addmm(self.mu[1], h, self.mu[0])
        g = obs
        for w, b in self.vf:
            g = torch.tanh(torch.addmm(b, g, w))
        value = torch.addmm(self.v[1], g, self.v[0]).squeeze(1)
        std = self.log_std.exp()
        act = mean + std * noise
        logp = (-0.5 * noise.pow(2) - self.log_std - 0.9189385332046727).sum(1)
        return act, value, logp


def cpu_baseline(kind, flags, iters, seconds=12.0):
    """The oracle (a CPU port of the same step) on the host cores: one env slice per thread."""
    import numpy as np
    from oracle import so100_oracle as O
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    n = 64 * threads
    b = O.OracleBatch(kind, n, flags, iters, seed=1234)
    b.reset()
    rs = np.random.RandomState(0)
    acts = rs.uniform(-1, 1, (n, 6)).astype(np.float32)
    b.step(acts, threads=threads)                          # warm-up
    t0 = time.perf_counter(); steps = 0
    while time.perf_counter() - t0 < seconds:
        b.step(acts, threads=threads); steps += 1
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} Env01 envs x {steps} vec-steps ({dt:.1f} s), fp64 C oracle, {threads} threads, same flags/solver iterations"}


def large_batch_roofline(kind, flags, dev, n=1 << 20, reps=20):
    """Supplementary: the same fused env-step kernel (so100_step_fused, every lane computes physics) with the chip
    filled -- 1,048,576 envs (0.37 GB of state) -- priced with the same 452 B / 3.3e4 FLOP per env-step.  At the BASELINE
    batch of 4096 envs (64 physics waves on 1024 SIMDs) no kernel can approach a roofline; this is what the kernel
    sustains when it can."""
    from so100_mujoco_rl_amd.lib import So100Sim
    sim = So100Sim(kind, n, device=dev, flags=flags, seed=99)
    sim.reset()
    a = (torch.rand(n, 6, device=dev) * 2 - 1).contiguous()
    for _ in range(3):
        sim.step(a)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev); e0.record()
    for _ in range(reps):
        sim.step(a)
    e1.record(); torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    sim.close()
    gbs = BYTES_PER_ENV_STEP * n / (ms * 1e-3) / 1e9
    tf = FLOP_PER_ENV_STEP * n / (ms * 1e-3) / 1e12
    return {"kernel": "so100_step_fused", "envs": n, "kernel_ms": ms, "env_steps_per_s": n / (ms * 1e-3),
            "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "valu_tflops": tf, "valu_frac": tf / VALU_PEAK_TFLOPS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--workload", default="env01_free", choices=["env01_free", "env01_arm", "env01_reference", "env02_reference", "env05_reference"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the supplementary 1M-env kernel measurement")
    ap.add_argument("--policy", default="persistent", choices=["persistent", "fused", "torch"])
    args = ap.parse_args()
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when the
    # first communicator is created), so file descriptor 1 points at stderr for the whole run and the result line is
    # written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1); os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # SO100_FORCE_DIST=1 runs the distributed code path (RCCL init, async gather, barrier, all-reduce) even with one
    # rank: a way to exercise it on a one-GPU box
    use_dist = world > 1 or os.environ.get("SO100_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED, F_REFERENCE, F_FRICTIONLOSS, F_LIMITS
    kind, flags = {"env01_free": (1, F_CUBE_PINNED), "env01_arm": (1, F_FRICTIONLOSS | F_LIMITS | F_CUBE_PINNED), "env01_reference": (1, F_REFERENCE),
                   "env02_reference": (2, F_REFERENCE), "env05_reference": (5, F_REFERENCE)}[args.workload]
    n = args.envs
    sim = So100Sim(kind, n, device=dev, flags=flags, solver_iters=2, contact_iters=6, seed=1234 + rank, env_id_offset=rank * n)
    obs = sim.reset()
    # stagger the episodes so TimeLimit resets are spread over the rollout
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    sim.set_field("elapsed_steps", torch.randint(0, sim.cfg.max_episode_steps, (n,), device=dev, generator=g, dtype=torch.int32))
    pol = MlpPolicy(sim.obs_dim, 6, dev, seed=0)
    T = ROLLOUT_T
    k = sim.obs_dim + 6 + 4                                  # obs, action, reward, done, value, logp
    # two chunk buffers: while chunk i is gathered to the learner over RCCL (async, on the collective's own stream),
    # the next rollout chunk is already being produced into buffer 1-i
    chunks = [torch.zeros(T, n, k, device=dev) for _ in range(2 if use_dist else 1)]
    # (the learner rank double-buffers the receive side too: a consumer of chunk i never races the gather of chunk i+1)
    gathered = [[torch.zeros(T, n, k, device=dev) for _ in range(world)] for _ in range(2)] if (use_dist and rank == 0) else None
    pending = [None, None]
    noise = torch.empty(n, 6, device=dev)

    act = torch.zeros(n, 6, device=dev)
    if args.policy in ("fused", "persistent"):
        sim.set_policy({"pi_w0": pol.pi[0][0].t().contiguous(), "pi_b0": pol.pi[0][1], "pi_w1": pol.pi[1][0].t().contiguous(), "pi_b1": pol.pi[1][1],
                        "mu_w": pol.mu[0].t().contiguous(), "mu_b": pol.mu[1], "log_std": pol.log_std,
                        "vf_w0": pol.vf[0][0].t().contiguous(), "vf_b0": pol.vf[0][1], "vf_w1": pol.vf[1][0].t().contiguous(), "vf_b1": pol.vf[1][1],
                        "v_w": pol.v[0].t().contiguous(), "v_b": pol.v[1]})
    counter = [0]

    def run(nsteps):
        """exactly nsteps vectorised env steps, in rollout chunks of at most T steps"""
        ci = 0
        for c0 in range(0, nsteps, T):
            Tc = min(T, nsteps - c0)
            ci ^= (len(chunks) - 1)
            if pending[ci] is not None:                      # buffer about to be overwritten: its gather must be done
                pending[ci].wait(); pending[ci] = None
            chunk = chunks[ci][:Tc]
            if args.policy == "persistent":
                sim.rollout(chunk, counter[0]); counter[0] += Tc               # ONE launch for Tc steps
            else:
                for t in range(Tc):
                    row = chunk[t]
                    if args.policy == "fused":
                        sim.policy_forward(sim.obs, act, counter[0], rollout_row=row)      # obs | action | value | logp -> row
                        sim.step(act, rollout_row=row)                                     # reward | done -> row
                        counter[0] += 1
                    else:
                        noise.normal_(generator=g)
                        a, value, logp = pol.forward(sim.obs, noise)
                        row[:, :sim.obs_dim] = sim.obs
                        row[:, sim.obs_dim:sim.obs_dim + 6] = a
                        a = a.clamp_(-1.0, 1.0)
                        ob, rew, done, trunc = sim.step(a)
                        row[:, -4] = rew; row[:, -3] = done; row[:, -2] = value; row[:, -1] = logp
            if use_dist:
                # RCCL: rollout chunk -> learner rank.  The collective is ordered after the producing kernel on this
                # stream and runs on its own stream; nothing waits for it until this buffer is reused (two chunks later).
                pending[ci] = dist.gather(chunk, [gb[:Tc] for gb in gathered[ci]] if gathered is not None else None, dst=0, async_op=True)

    def sync():
        for i, wk in enumerate(pending):
            if wk is not None:
                wk.wait(); pending[i] = None
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())

    # dominant kernel alone: HIP events on the launch stream (torch's current stream, where the C ABI enqueues) around
    # back-to-back launches.  persistent mode: so100_rollout_fused, one launch = T x N env-steps; otherwise so100_step_fused.
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    if args.policy == "persistent":
        kernel_name, units, reps = "so100_rollout_fused", n * T, 12
        for _ in range(2):
            sim.rollout(chunks[0], counter[0]); counter[0] += T
        torch.cuda.synchronize(dev); e0.record()
        for _ in range(reps):
            sim.rollout(chunks[0], counter[0]); counter[0] += T
        e1.record(); torch.cuda.synchronize(dev)
    else:
        kernel_name, units, reps = "so100_step_fused", n, 200
        a2 = (torch.rand(n, 6, device=dev) * 2 - 1).contiguous()
        for _ in range(20):
            sim.step(a2)
        torch.cuda.synchronize(dev); e0.record()
        for _ in range(reps):
            sim.step(a2)
        e1.record(); torch.cuda.synchronize(dev)
    kern_ms = e0.elapsed_time(e1) / reps

    # HBM traffic of the dominant kernel: rocprofv3 PMC counters cannot be collected from inside this process; the
    # committed PMC pass (profiles/r01_c_pmc_summary.json, same workload / batch size) is quoted when it matches.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_j_pmc_summary.json" if args.policy == "persistent" else "r01_c_pmc_summary.json")))
        if args.workload == "env01_free" and n == 4096:
            key = "rollout_fused" if args.policy == "persistent" else "step_fused"
            traffic = [v for k, v in pmc["kernels"].items() if key in k][0]["hbm_traffic_bytes_per_launch"]
    except Exception:
        traffic = None

    if rank == 0:
        value = world * n * args.steps / dt
        achieved = BYTES_PER_ENV_STEP * units / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env steps/sec at 4096 envs/GPU, Env01 PPO rollout", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: Env{kind:02d} x {n} envs/GPU, frame_skip 16, "
                                   + ("contact disabled, no constraint solver (BASELINE.json configs[1])" if args.workload == "env01_free" else "friction-loss + limits + cube/floor contact")
                                   + ", SB3-MlpPolicy-shaped rollout (" + args.policy + " policy), randomized resets, staggered episodes",
                       "envs_per_gpu": n, "rollout_chunk": T, "parallelism": f"env-sharded x{world}, RCCL gather per chunk" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (FETCH_SIZE+WRITE_SIZE, PMC pass in profiles/)", "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * units, "kernel": kernel_name, "kernel_ms": kern_ms, "env_steps_per_launch": units,
                         "bytes_per_env_step": BYTES_PER_ENV_STEP, "kernel_env_steps_per_s": units / (kern_ms * 1e-3),
                         "valu": {"flop_per_env_step": FLOP_PER_ENV_STEP, "flop_per_env_step_survey_estimate": FLOP_PER_ENV_STEP_SURVEY,
                                  "achieved_tflops": FLOP_PER_ENV_STEP * units / (kern_ms * 1e-3) / 1e12, "peak_tflops": VALU_PEAK_TFLOPS,
                                  "frac": FLOP_PER_ENV_STEP * units / (kern_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, flags, 2)
        if world == 1 and args.workload == "env01_free" and not args.no_large_batch:
            out["roofline"]["large_batch"] = large_batch_roofline(kind, flags, dev)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
