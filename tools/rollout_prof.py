"""Cycle accounting of the persistent rollout kernel's phases (waves 0-3 of workgroup 0).
Build the instrumented side library HERE (no GPU needed), then run on the GPU box:
    python tools/rollout_prof.py build
    gpurun -- python tools/rollout_prof.py [free|arm|nopads|ref|c5] [random|resting|held]
The product library is untouched (the instrumentation is compiled out without -DSO100_ROLLOUT_PROF)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIDE = os.path.join(ROOT, "so100_mujoco_rl_amd", "libso100sim_prof.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    import torch
    from concurrent.futures import ThreadPoolExecutor
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    csrc = os.path.join(ROOT, "so100_mujoco_rl_amd", "csrc")
    odir = os.path.join(ROOT, "gpurun_out", "prof_obj"); os.makedirs(odir, exist_ok=True)
    base = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=fast",
            "-fno-slp-vectorize", "-Wno-unused-function", "-DSO100_ROLLOUT_PROF", "-c"]
    jobs = [(base + ["-o", os.path.join(odir, "sim.o"), os.path.join(csrc, "so100_sim.hip")])]
    jobs += [(base + [f"-DSO100_KIND={k}", "-o", os.path.join(odir, f"kind{k}.o"), os.path.join(csrc, "so100_kind.hip")]) for k in range(1, 7)]
    with ThreadPoolExecutor(7) as ex:
        list(ex.map(subprocess.check_call, jobs))
    objs = [os.path.join(odir, "sim.o")] + [os.path.join(odir, f"kind{k}.o") for k in range(1, 7)]
    subprocess.check_call(["g++", "-shared", "-o", SIDE] + objs + ["-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl, "-Wl,-rpath,/opt/rocm/lib", "-lstdc++", "-lm"])
    print("built", SIDE); sys.exit(0)
os.environ["SO100_LIB"] = SIDE
sys.path.insert(0, ROOT)
import torch
from so100_mujoco_rl_amd import lib
from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED, F_REFERENCE, F_NOPADS, F_CONTACT5, F_FRICTIONLOSS, F_LIMITS
from so100_mujoco_rl_amd.collector import RolloutCollector, SB3_STATE_DICT_KEYS, POLICY_TENSORS
which = sys.argv[1] if len(sys.argv) > 1 else "free"
mode = sys.argv[2] if len(sys.argv) > 2 else "random"
flags = {"free": F_CUBE_PINNED, "arm": F_FRICTIONLOSS | F_LIMITS | F_CUBE_PINNED, "nopads": F_NOPADS, "ref": F_REFERENCE, "c5": F_CONTACT5}[which]
n, T = 4096, 64
sim = So100Sim(1, n, flags=flags, seed=1); sim.reset()
sd = RolloutCollector.random_policy_state(sim.obs_dim, sim.device, seed=0)
if mode in ("resting", "held"):             # zero action: Env01's ctrl = measured angle lets every arm sag onto the floor and rest there
    sd["action_net.weight"].zero_(); sd["action_net.bias"].zero_(); sd["log_std"].fill_(-30.0)
if mode == "held":                          # a constant lifting action on the shoulder: no arm touches the floor (the cost of detection alone)
    sd["action_net.bias"][1] = -1.0
sim.set_policy({k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS})
buf = torch.empty(T, n, sim.obs_dim + 10, device="cuda")
hist = (C.c_ulonglong * 48)()
import numpy as np
sim.L.so100_prof_read_wg.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
prev_env = None; busy_at_start = None
for i in range(4):
    if i == 3:
        torch.cuda.synchronize(); sim.L.so100_prof_read_hist(1, hist, 1)          # statistics of the last launch only
        _wg = (C.c_longlong * 4096)(); _ev = (C.c_int * 32768)()
        if sim.L.so100_prof_read_wg(1, _wg, _ev) == 0:
            prev_env = np.array(_ev[:2*n]).reshape(n, 2).copy()                   # per slot: passes, substeps in contact of launch 2
        busy_at_start = ((sim.get_field("contact_stat", dtype=torch.int32) & 255) > 0).cpu().numpy()
    sim.rollout(buf, i * T)
torch.cuda.synchronize()
sim.L.so100_prof_read_hist(1, hist, 0)
out = (C.c_longlong * 48)()
assert sim.L.so100_prof_read(1, out) == 0
names = ["policy layers (MFMA+tanh, 2 barriers)", "head+noise+row+env_step_pre", "trig + publish", "barrier-1 wait",
         "phase 1: CRBA+factor (w0) / RNEA (w1) / cube_prepare (w2) / FK+narrowphase (w3)", "barrier-2 wait",
         "phase 2: PGS (w0) / cube Newton (w2) / contact Newton (w3)", "step tail: poses, obs, reset, end barrier", "barrier-3 wait", "integrate",
         "  (w3) world FK, before the narrowphase", "  (w3) contact-solve set-up, before the Newton"]
print(f"# tools/rollout_prof.py {which} {mode}   (Env01 x {n}, epw {16 if flags & 4 else 64})")
for w in range(4):
    v = [out[12 * w + i] for i in range(12)]; tot = sum(v)
    print(f"wave {w}: total {tot / T:9.0f} ticks/step")
    for nm, x in zip(names, v):
        per = "  (%.0f / substep)" % (x / T / 16) if nm.split()[0] in ("trig", "barrier-1", "phase", "barrier-2", "barrier-3", "integrate", "(w3)") else ""
        print(f"    {nm:84s} {x / T:9.0f} ticks/step  {100.0 * x / max(tot, 1):5.1f} %{per}")

# per-workgroup totals (the launch ends with its slowest workgroup) and per-env Newton work of the last launch
import numpy as np
wg = (C.c_longlong * 4096)(); envw = (C.c_int * 32768)()
sim.L.so100_prof_read_wg.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
if sim.L.so100_prof_read_wg(1, wg, envw) == 0:
    epw = 16 if flags & 4 else 64
    nwg = (n + epw - 1)//epw
    W = np.array(wg[:4*nwg], dtype=np.float64).reshape(nwg, 4)/T
    E = np.array(envw[:2*n]).reshape(n, 2)
    tot = W[:, 0]
    print(f"# per workgroup ({nwg} x {epw} envs), ticks/step: total min {tot.min():.0f}  median {np.median(tot):.0f}  p90 {np.percentile(tot, 90):.0f}  max {tot.max():.0f}")
    print(f"#   narrowphase (w3): median {np.median(W[:,1]):.0f} max {W[:,1].max():.0f};  contact Newton (w3): median {np.median(W[:,2]):.0f} p90 {np.percentile(W[:,2],90):.0f} max {W[:,2].max():.0f};  w0 barrier-3 wait: median {np.median(W[:,3]):.0f} max {W[:,3].max():.0f}")
    order = np.argsort(-tot)
    for b in list(order[:6]) + list(order[len(order)//2:len(order)//2 + 2]):
        ev = E[b*epw:(b + 1)*epw]
        act = ev[ev[:, 1] > 0]
        print(f"#   wg {b:4d}: total {tot[b]:8.0f}  newton {W[b,2]:8.0f}  envs with contact {len(act):2d}/{epw}  substeps in contact (of {T*16}) {sorted(act[:,1].tolist(), reverse=True)[:6]}  row passes per contact substep {[round(a/max(b_,1),2) for a, b_ in sorted(act.tolist(), key=lambda r: -r[1])[:6]]}")
    insub = E[:, 1].sum(); work = E[:, 0].sum()
    print(f"# all envs: {100.0*insub/(n*T*16):.1f} % of env-substeps in pad contact, {work/max(insub,1):.2f} row passes per contact substep; per-wg sum over envs of row passes: median {np.median(E[:,0].reshape(nwg, epw).sum(1)):.0f} max {E[:,0].reshape(nwg, epw).sum(1).max():.0f};"
          f" per-wg MAX-lane passes: median {np.median(E[:,0].reshape(nwg, epw).max(1)):.0f} max {E[:,0].reshape(nwg, epw).max(1).max():.0f}")

if prev_env is not None and os.environ.get("SO100_BALANCE") == "0":        # (identity slot map: slot == env, so launches can be compared env by env)
    a = prev_env[:, 1].astype(float); b = E[:, 1].astype(float)
    print(f"# contact substeps per env, launch 2 vs launch 3: correlation {np.corrcoef(a, b)[0, 1]:.3f}; envs in contact at the start of launch 3 ({busy_at_start.mean():.3f} of all) "
          f"carry {b[busy_at_start].sum()/max(b.sum(), 1):.3f} of its contact substeps; the top quarter of launch 2 carries {b[np.argsort(-a)[:n//4]].sum()/max(b.sum(), 1):.3f}")
h = list(hist)
if h[39]:
    print(f"# contact Newton per (workgroup, substep) of the last launch: {h[39]} pairs, {100.0*h[38]/h[39]:.1f} % with a solve; envs solving at once: "
          + " ".join(f"{k}:{100.0*h[16+k]/h[39]:.1f}%" for k in range(17) if h[16+k]))
    if h[38]:
        print("#   slowest lane's gradient+Hessian passes: " + " ".join(f"{k}:{100.0*h[k]/h[38]:.1f}%" for k in range(16) if h[k])
              + f";  its estimated instructions: mean {h[37]/h[38]:.0f} (all solving envs: mean {h[40]/max(1, h[33] and sum(h[16+k]*k for k in range(17))):.0f})")
        ns = max(1, sum(h[16+k]*k for k in range(17)))
        print(f"#   passes per solve (all envs): full {h[33]/ns:.2f}  sign {h[34]/ns:.2f}  gradient {h[35]/ns:.2f}  line-search {h[36]/ns:.2f}")
