"""Cycle accounting of the persistent rollout kernel's phases (wave 0 / 1 / 2 of workgroup 0).
Build the instrumented side library HERE (no GPU needed), then run on the GPU box:
    python tools/rollout_prof.py build
    gpurun -- python tools/rollout_prof.py [free|arm|ref]
The product library is untouched (the instrumentation is compiled out without -DSO100_ROLLOUT_PROF)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIDE = os.path.join(ROOT, "so100_mujoco_rl_amd", "libso100sim_prof.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    csrc = os.path.join(ROOT, "so100_mujoco_rl_amd", "csrc")
    obj = os.path.join(ROOT, "gpurun_out", "so100_sim_prof.o"); os.makedirs(os.path.dirname(obj), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=fast",
                           "-fno-slp-vectorize", "-Wno-unused-function", "-DSO100_ROLLOUT_PROF", "-c", "-o", obj, os.path.join(csrc, "so100_sim.hip")])
    subprocess.check_call(["g++", "-shared", "-o", SIDE, obj, "-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl, "-Wl,-rpath,/opt/rocm/lib", "-lstdc++", "-lm"])
    print("built", SIDE); sys.exit(0)
os.environ["SO100_LIB"] = SIDE
sys.path.insert(0, ROOT)
import torch
from so100_mujoco_rl_amd import lib
from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED, F_REFERENCE, F_FRICTIONLOSS, F_LIMITS
from so100_mujoco_rl_amd.collector import RolloutCollector, SB3_STATE_DICT_KEYS, POLICY_TENSORS
which = sys.argv[1] if len(sys.argv) > 1 else "free"
flags = {"free": F_CUBE_PINNED, "arm": F_FRICTIONLOSS | F_LIMITS | F_CUBE_PINNED, "ref": F_REFERENCE}[which]
n, T = 4096, 64
sim = So100Sim(1, n, flags=flags, seed=1); sim.reset()
sd = RolloutCollector.random_policy_state(sim.obs_dim, sim.device, seed=0)
sim.set_policy({k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS})
buf = torch.empty(T, n, sim.obs_dim + 10, device="cuda")
for i in range(3):
    sim.rollout(buf, i * T)
torch.cuda.synchronize()
out = (C.c_longlong * 32)()
assert sim.L.so100_prof_read(out) == 0
names = ["policy layers (MFMA+tanh, 2 barriers)", "head+noise+row+env_step_pre", "trig + publish", "barrier-1 wait", "CRBA+factor (w0) / RNEA (w1)",
         "barrier-2 wait", "solve+integrate (w0)", "step tail: poses, obs, reset, end barrier"]
for w in range(3):
    v = [out[8 * w + i] for i in range(8)]; tot = sum(v)
    print(f"wave {w}: total {tot / T:9.0f} ticks/step")
    for nm, x in zip(names, v):
        per = "  (%.0f / substep)" % (x / T / 16) if nm.split()[0] in ("trig", "barrier-1", "CRBA+factor", "barrier-2", "solve+integrate") else ""
        print(f"    {nm:42s} {x / T:9.0f} ticks/step  {100.0 * x / max(tot, 1):5.1f} %{per}")
