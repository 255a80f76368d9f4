#!/usr/bin/env python3
"""bench.py -- env-steps/s of the Env01 PPO rollout at 4096 envs/GPU (BASELINE.json metric).

One "step" = one vectorised env step of the rollout loop on every rank: policy forward (SB3 MlpPolicy shape:
separate 2x64 tanh towers for pi and V, state-independent log-std Gaussian), action sampling + clipping, the fused
HIP env step (reward -> ctrl -> 16 physics substeps -> obs -> TimeLimit -> auto-reset), and the write of
obs/action/reward/done/value/log-prob into the on-device rollout buffer; every ROLLOUT_T steps the rollout chunk is
gathered to the learner rank over RCCL (N > 1 only).  Default collector (`--policy persistent`): ONE launch per rollout
chunk of 64 steps (so100_rollout_fused: persistent workgroups, policy phase on all waves, physics split over the waves,
env state in registers).  `--policy fused`: two launches per step (so100_policy_forward + so100_step), no PyTorch op in
the loop.  `--policy torch`: the same rollout with the policy as plain PyTorch ops (what an unmodified SB3 policy costs).
--steps / --warmup count vectorised env steps in every mode (a trailing partial chunk is one shorter launch).
Default workload = BASELINE.json configs[1]: Env01, 4096 envs per GPU, contact disabled / no constraint solver (cube
pinned), synthetic randomized-reset batches, random-init policy.  `--workload env01_contact` = configs[4]'s per-GPU shape.

`python bench.py --gpus N` launches the N ranks itself (torch.distributed.run, one process per GPU, RCCL) when it is not
already running under a launcher (WORLD_SIZE unset) and relays rank 0's line; under a launcher it is a rank.

Prints ONE JSON line (rank 0).  The timed region of exactly --steps steps is repeated (`repeats`) and `value` is the
MEDIAN repeat (min / max alongside): one 0.7 ms launch is not a measurement.  `roofline` prices the dominant kernel
(so100_rollout_fused in the default mode) with the algorithmic 452 B/env-step of SURVEY.md section 8(d), its duration
measured live with HIP events around every launch of the timed repeats; `cpu_baseline` times the CPU oracle (a port,
not the reference) on the host cores; `sb3_vecenv_path` is the numpy-in / numpy-out `So100VecEnv.step` round trip that an
unmodified SB3 learner drives (ref: main.py:57-63, 234-238), same batch, same physics flags, same run.
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
# SURVEY.md section 8(d) per env kind: Env02 500 B, Env05 536 B, contact configs ~600 B (key: (kind, pad contacts on))
BYTES_PER_ENV_STEP_BY_KIND = {(1, False): 452.0, (2, False): 500.0, (5, False): 536.0, (1, True): 600.0, (2, True): 648.0, (5, True): 684.0}
CONTACT_BITS = 16 | 32 | 64 | 128 # SO100_F_PADS_FLOOR | SO100_F_PADS_CUBE | SO100_F_LINKS_FLOOR | SO100_F_LINKS_CUBE (include/so100_sim.h)
FLOP_PER_ENV_STEP_SURVEY = 6.0e4  # SURVEY.md section 8(d) estimate, constraint-free
# measured: PMC SQ_INSTS_VALU = 22.95 k VALU instructions per env-step lane (profiles/r01_c), of which ~45 % are FMAs
# (ISA count: 1017 fma/fmac of 2100 float ops per substep) => ~1.45 FLOP per instruction => 3.3e4 FLOP per env-step.
# The VALU fraction below uses the MEASURED figure (the survey estimate would overstate utilisation 1.8x).
FLOP_PER_ENV_STEP = 3.3e4
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
VALU_PEAK_TFLOPS = 157.3
ROLLOUT_T = 64
# name -> (env kind, physics flag names, description); flags are resolved against so100_mujoco_rl_amd.lib at run time
WORKLOADS = {
    "env01_free": (1, ("F_CUBE_PINNED",), "contact disabled, no constraint solver (BASELINE.json configs[1])"),
    "env01_arm": (1, ("F_FRICTIONLOSS", "F_LIMITS", "F_CUBE_PINNED"), "friction-loss + joint-limit rows, cube pinned"),
    "env01_nopads": (1, ("F_NOPADS",), "round-1 'reference': friction-loss + limits + cube/floor contact, no arm contact at all"),
    "env01_reference": (1, ("F_REFERENCE",), "reference physics: friction-loss + limits + cube/floor + finger-pad/floor contact"),
    "env02_reference": (2, ("F_REFERENCE",), "reference physics (BASELINE.json configs[2] at this batch size)"),
    "env05_reference": (5, ("F_REFERENCE",), "reference physics (BASELINE.json configs[3] per-GPU shape)"),
    "env01_reference_links": (1, ("F_REFERENCE_LINKS",), "reference physics + capsule proxies of the arm links' collision meshes vs the floor (a documented stand-in; run-time-flags kernels)"),
    "env01_reference_proxies": (1, ("F_REFERENCE_PROXIES",), "reference physics + every capsule proxy pair: links vs the floor, Rotation_Pitch / Upper_Arm vs the cube (SURVEY.md Q7; stand-ins; run-time-flags kernels)"),
    "env01_contact": (1, ("F_CONTACT5",), "reference physics + finger-pad/cube box-box contact, coupled arm+cube solve (BASELINE.json configs[4] per-GPU shape)"),
}


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


def cpu_baseline(kind, flags, iters, seconds=10.0):
    """The oracle (a CPU port of the same step) on ALL the host's cores: one env slice per thread, one thread per CPU the process
    may run on (north star: "timed on the node's own host cores").  The 64-thread figure of earlier rounds is reported beside it."""
    import numpy as np
    from oracle import so100_oracle as O
    host = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))               # what this process can actually use (a taskset may be narrower) ...
    except AttributeError:
        cores = host
    quota = None
    try:                                                   # ... and what the container's CPU quota allows (cgroup v2 cpu.max / v1 cfs quota)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q)/float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q/per
        except Exception:
            pass
    if quota is not None and quota >= 1.0:
        cores = max(1, min(cores, int(quota + 0.5)))

    def measure(threads, secs):
        n = 64 * threads
        b = O.OracleBatch(kind, n, flags, iters, seed=1234)
        b.reset()
        acts = np.random.RandomState(0).uniform(-1, 1, (n, 6)).astype(np.float32)
        b.step(acts, threads=threads, native_threads=True)  # warm-up (pthreads inside the oracle library: one slice of 64 envs per thread)
        t0 = time.perf_counter(); steps = 0
        while time.perf_counter() - t0 < secs:
            b.step(acts, threads=threads, native_threads=True); steps += 1
        dt = time.perf_counter() - t0
        return n * steps / dt, n, steps, dt
    v, n, steps, dt = measure(cores, seconds)
    out = {"value": v, "unit": "env-steps/s", "cores": cores, "host_cpus": host, "cpu_quota": quota, "kind": "port",
           "sample": f"{n} Env{kind:02d} envs x {steps} vec-steps ({dt:.1f} s), fp64 C oracle, {cores} threads (one per usable CPU) on a {host}-CPU host, same flags, "
                     + ("primal Newton to convergence" if iters < 0 else f"{iters} PGS sweeps")}
    if cores > 64:
        out["value_64_threads"] = measure(64, 4.0)[0]
    return out


def sb3_vecenv_path(kind, flags, n, dev, steps=200):
    """The SB3-facing path (ref: main.py:57-63 hands the env to SB3, whose collect_rollouts calls VecEnv.step with numpy
    actions): So100VecEnv.step = ONE launch of the fused step kernel, which reads the actions from and writes its results to pinned host
    memory itself, one stream sync, + the Python info bookkeeping.  Same batch size and physics flags as the headline, policy not included (SB3 runs its own)."""
    import numpy as np
    from so100_mujoco_rl_amd.vec_env import So100VecEnv
    env = So100VecEnv(kind, n, device=dev, flags=flags, seed=77, stagger_episodes=True)
    env.reset()
    a = np.random.RandomState(1).uniform(-1, 1, (n, 6)).astype(np.float32)
    for _ in range(20):
        env.step(a)
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(a)
    dt = time.perf_counter() - t0
    env.close()
    return {"us_per_step": dt / steps * 1e6, "env_steps_per_s": n * steps / dt, "envs": n, "steps": steps,
            "what": "So100VecEnv.step(numpy actions) -> numpy obs/rew/done/infos: one kernel launch that reads / writes pinned host memory, one sync, no policy"}


def source_sha16():
    """Fingerprint of the kernel sources (csrc/ only: editing this script does not change what the kernels do): the committed
    PMC pass is only quoted while it describes THIS code."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "so100_mujoco_rl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".h", ".inc")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(n_gpus, argv, worker=None, port=None):
    """Start n_gpus fresh rank processes (torch.distributed.run, one per GPU) BEFORE this process touches the GPU and
    relay rank 0's JSON line.  Returns the children's exit code."""
    import socket
    import subprocess
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), worker or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    # rank 0's result line; the ranks share one stdout pipe, so another rank's output can land on the same line
    result = None
    dec = json.JSONDecoder()
    for l in proc.stdout.splitlines():
        i = l.find("{")
        if i >= 0:
            try:
                obj, _ = dec.raw_decode(l[i:])
                if isinstance(obj, dict):
                    result = obj
            except ValueError:
                pass
    if proc.returncode == 0 and result is not None:
        print(json.dumps(result), flush=True)
    else:
        sys.stderr.write(proc.stdout)
    return proc.returncode if proc.returncode != 0 else (0 if result is not None else 1)


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
    # issue slots: VALU wave-instructions per second against what 1024 SIMDs can issue.  22.95 k VALU instructions per lane and env
    # step (PMC SQ_INSTS_VALU, profiles/r01_c); a SIMD issues a wave64 fp32 instruction every 2 cycles at best, and with the two
    # waves this kernel keeps resident every 2.27 cycles MEASURED (profiles/r02_simd_share_issue_rate.txt) at ~2.4 GHz.
    wave_insts_per_s = 22950.0 * (n / (ms * 1e-3)) / 64.0
    slots = 1024 * 2.4e9 / 2.0
    return {"kernel": "so100_step_fused", "envs": n, "kernel_ms": ms, "env_steps_per_s": n / (ms * 1e-3),
            "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "valu_tflops": tf, "valu_frac": tf / VALU_PEAK_TFLOPS,
            "valu_issue_slot_frac": wave_insts_per_s / slots, "valu_issue_slot_frac_of_measured_two_wave_ceiling": wave_insts_per_s / (slots * 2.0 / 2.27),
            "valu_insts_per_lane_env_step": 22950.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--workload", default="env01_free", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the supplementary 1M-env kernel measurement")
    ap.add_argument("--no-sb3-path", action="store_true", help="skip the So100VecEnv numpy round-trip measurement")
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the timed region (0 = as many as fit ~1.5 s, 3..40)")
    ap.add_argument("--policy", default="persistent", choices=["persistent", "fused", "torch"])
    ap.add_argument("--worker", default=None, help=argparse.SUPPRESS)      # script the launcher starts (tests use a stub)
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the ranks ourselves, before anything here initialises the GPU
        argv = [a for a in sys.argv[1:]]
        sys.exit(launch_ranks(args.gpus, argv, worker=args.worker))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
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
    from so100_mujoco_rl_amd import lib as so100lib
    from so100_mujoco_rl_amd.lib import So100Sim
    kind, flag_names, wl_desc = WORKLOADS[args.workload]
    flags = 0
    for fn in flag_names:
        flags |= getattr(so100lib, fn)
    rccl_ranks = 1
    if use_dist:                                             # the communicator really spans `world` ranks
        ones = torch.ones(1, device=dev); dist.all_reduce(ones); rccl_ranks = int(ones.item())
        assert rccl_ranks == world, (rccl_ranks, world)
    n = args.envs
    sim = So100Sim(kind, n, device=dev, flags=flags, solver_iters=2, contact_iters=20, seed=1234 + rank, env_id_offset=rank * n)
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

    kev = []                                                 # (start, end, env-steps) HIP event pairs around the dominant kernel's launches
    evpool = []                                              # event objects are created OUTSIDE the timed regions (only their record() calls are inside)

    def ev_pair():
        if len(evpool) < 2:                                  # (stepwise modes record a pair per step: refill)
            evpool.extend(torch.cuda.Event(enable_timing=True) for _ in range(64))
        return evpool.pop(), evpool.pop()
    evpool.extend(torch.cuda.Event(enable_timing=True) for _ in range(2*(42*((args.steps + ROLLOUT_T - 1)//ROLLOUT_T) + 4)))

    def run(nsteps, record=False):
        """exactly nsteps vectorised env steps, in rollout chunks of at most T steps"""
        ci = 0
        for c0 in range(0, nsteps, T):
            Tc = min(T, nsteps - c0)
            ci ^= (len(chunks) - 1)
            if pending[ci] is not None:                      # buffer about to be overwritten: its gather must be done
                pending[ci].wait(); pending[ci] = None
            chunk = chunks[ci][:Tc]
            if args.policy == "persistent":
                if record:                                   # events on the launch stream (torch's current stream = where the C ABI enqueues)
                    ea, eb = ev_pair(); ea.record()
                sim.rollout(chunk, counter[0]); counter[0] += Tc               # ONE launch for Tc steps
                if record:
                    eb.record(); kev.append((ea, eb, n * Tc))
            else:
                for t in range(Tc):
                    row = chunk[t]
                    if args.policy == "fused":
                        sim.policy_forward(sim.obs, act, counter[0], rollout_row=row)      # obs | action | value | logp -> row
                        if record:
                            ea, eb = ev_pair(); ea.record()
                        sim.step(act, rollout_row=row)                                     # reward | done -> row
                        if record:
                            eb.record(); kev.append((ea, eb, n))
                        counter[0] += 1
                    else:
                        noise.normal_(generator=g)
                        a, value, logp = pol.forward(sim.obs, noise)
                        row[:, :sim.obs_dim] = sim.obs
                        row[:, sim.obs_dim:sim.obs_dim + 6] = a
                        a = a.clamp_(-1.0, 1.0)
                        if record:
                            ea, eb = ev_pair(); ea.record()
                        ob, rew, done, trunc = sim.step(a)
                        if record:
                            eb.record(); kev.append((ea, eb, n))
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

    def timed(record):
        """EXACTLY args.steps steps between two {barrier + device synchronize}; max over ranks"""
        sync()
        t0 = time.perf_counter()
        run(args.steps, record)
        sync()
        dt_own = time.perf_counter() - t0
        dt = dt_own
        if use_dist:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
        return dt, dt_own

    run(args.warmup)
    first, first_own = timed(False)
    # repeats of the timed region: as many as fit ~1.5 s (3..40) unless given; every rank must agree on the count
    reps = args.repeats if args.repeats > 0 else max(3, min(40, int(1.5 / max(first, 1e-6))))
    if use_dist:
        rr = torch.tensor([reps], device=dev); dist.broadcast(rr, src=0); reps = int(rr.item())
    dts, owns = [first], [first_own]
    rec = args.policy == "persistent"                        # one event pair per 64-step launch costs nothing; per-step pairs would
    for _ in range(reps - 1):
        d, o = timed(rec)
        dts.append(d); owns.append(o)
    if len(kev) == 0:                                        # stepwise modes (or a single repeat): one more pass, only for the kernel events
        timed(True)
    torch.cuda.synchronize(dev)
    order = sorted(range(len(dts)), key=lambda i: dts[i])
    med = order[len(order) // 2]
    dt = dts[med]
    kern_total_ms = sum(a.elapsed_time(b) for a, b, _ in kev); kern_units = sum(u for _, _, u in kev)
    units = kev[0][2]                                        # env-steps of one full launch (the first recorded one)
    kern_ms = kern_total_ms / kern_units * units             # average duration of a launch of `units` env-steps
    kernel_name = "so100_rollout_fused" if args.policy == "persistent" else ("so100_step_mw" if n <= 16384 else "so100_step_fused")
    per_rank = None
    if use_dist:                                             # every rank's own rate over the median repeat
        mine = torch.tensor([n * args.steps / owns[med]], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]; dist.all_gather(allr, mine)
        per_rank = [float(x.item()) for x in allr]

    # HBM traffic of the dominant kernel: rocprofv3 PMC counters cannot be collected from inside this process; the
    # committed PMC pass is quoted only when it was taken on THIS kernel source (fingerprint), workload and batch size.
    traffic, traffic_note = None, "no committed PMC pass matches this kernel source / workload / batch size"
    try:
        import glob
        sha = source_sha16()
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
            pmc = json.load(open(f))
            if pmc.get("source_sha16") != sha or pmc.get("bench_workload") != args.workload or pmc.get("envs") != n or pmc.get("policy", "persistent") != args.policy:
                continue
            hit = [v for k, v in pmc["kernels"].items() if kernel_name in k and "hbm_traffic_bytes_per_env_step" in v]
            if hit:
                traffic = hit[0]["hbm_traffic_bytes_per_env_step"]; traffic_note = os.path.basename(f)
    except Exception as ex:                                  # a broken summary file must not break the bench line
        traffic, traffic_note = None, f"PMC summary unreadable: {ex}"

    if rank == 0:
        bytes_per = BYTES_PER_ENV_STEP_BY_KIND.get((kind, bool(flags & CONTACT_BITS)), BYTES_PER_ENV_STEP)
        value = world * n * args.steps / dt
        achieved = bytes_per * units / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env steps/sec at 4096 envs/GPU, Env01 PPO rollout", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": len(dts), "value_min": world * n * args.steps / max(dts), "value_max": world * n * args.steps / min(dts),
            "ms_per_step_min": min(dts) / args.steps * 1e3, "ms_per_step_max": max(dts) / args.steps * 1e3,
            "rccl_ranks": rccl_ranks, "per_rank_env_steps_per_s": per_rank,
            "config": {"workload": f"{args.workload}: Env{kind:02d} x {n} envs/GPU, frame_skip 16, " + wl_desc
                                   + ", SB3-MlpPolicy-shaped rollout (" + args.policy + " policy), randomized resets, staggered episodes",
                       "envs_per_gpu": n, "rollout_chunk": T, "parallelism": f"env-sharded x{world}, RCCL gather per chunk" if world > 1 else "single GPU"},
            # `traffic` and `algorithmic_bytes_per_launch` are both per launch of THIS run (`env_steps_per_launch` env-steps): the PMC pass
            # measures bytes per env-step of the same kernel / workload / batch (its own launches are 64-step chunks) and is scaled to it
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic is None else traffic * units, "traffic_source": traffic_note,
                         "traffic_unit": "bytes per launch of env_steps_per_launch env-steps (FETCH_SIZE+WRITE_SIZE per env-step from the PMC pass in profiles/, x env_steps_per_launch)",
                         "traffic_bytes_per_env_step": traffic, "algorithmic_bytes_per_launch": bytes_per * units, "kernel": kernel_name, "kernel_ms": kern_ms,
                         "kernel_launches_timed": len(kev), "env_steps_per_launch": units, "bytes_per_env_step": bytes_per,
                         "kernel_env_steps_per_s": units / (kern_ms * 1e-3)},
        }
        if args.workload == "env01_free":                    # the measured instruction count (PMC) is for this workload only
            tf = FLOP_PER_ENV_STEP * units / (kern_ms * 1e-3) / 1e12
            out["roofline"]["valu"] = {"flop_per_env_step": FLOP_PER_ENV_STEP, "flop_per_env_step_survey_estimate": FLOP_PER_ENV_STEP_SURVEY,
                                       "achieved_tflops": tf, "peak_tflops": VALU_PEAK_TFLOPS, "frac": tf / VALU_PEAK_TFLOPS}
        if world == 1 and not args.no_sb3_path:
            out["sb3_vecenv_path"] = sb3_vecenv_path(kind, flags, n, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, flags, -1 if (flags & CONTACT_BITS) else 2)     # pad rows: the oracle's Newton (PGS crawls on them)
        if world == 1 and args.workload == "env01_free" and not args.no_large_batch:
            out["roofline"]["large_batch"] = large_batch_roofline(kind, flags, dev)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
