"""Statistics of the pad-contact Newton under the bench's workload, on the HOST instantiation of the device code (fp32,
tests/_hostcheck; no GPU): Env01, reference physics, random actions, staggered episodes.  Per solve: histograms of gradient +
Hessian passes, sign passes, gradient passes, line-search passes.  Used to design the solver; the GPU numbers are in profiles/.
    python tools/host_newton_stats.py [envs] [steps]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "_hostcheck"), "-s"])
H = C.CDLL(os.path.join(ROOT, "tests", "_hostcheck", os.environ.get("HC_LIB", "libhostcheck.so")))
H.hc_env_new.restype = C.c_void_p
for f in ("hc_cdbg_passes", "hc_cdbg_signpasses", "hc_cdbg_gradpasses", "hc_cdbg_lastiter"):
    getattr(H, f).restype = C.c_long
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
FLAGS = 1 | 2 | 4 | 16
rs = np.random.RandomState(0)
P = lambda a: a.ctypes.data_as(C.c_void_p)
envs = [C.c_void_p(H.hc_env_new(1)) for _ in range(n)]
obs = np.zeros(15, np.float32); tobs = np.zeros(15, np.float32); rew = C.c_float(); done = C.c_int(); trunc = C.c_int()
for e in envs:
    inj = rs.random_sample(16).astype(np.float32); H.hc_env_reset(e, 1, P(inj), P(obs))
hist = (C.c_long*64)(); H.hc_cdbg_hist(hist, 1)
stat = np.zeros(2); touching = 0
for t in range(steps):
    for e in envs:
        a = np.clip(rs.randn(6), -1, 1).astype(np.float32); inj = rs.random_sample(16).astype(np.float32)
        H.hc_env_step(e, 1, FLAGS, 2, 20, 4000, P(a), P(inj), P(obs), P(tobs), C.byref(rew), C.byref(done), C.byref(trunc))
        H.hc_env_stats(e, P(stat)); touching += (int(stat[0]) & 255) > 0
    if t == steps//4:
        H.hc_cdbg_hist(hist, 1)                              # discard the start-up transient
H.hc_cdbg_hist(hist, 0)
h = np.array(list(hist)).reshape(4, 16); solves = h[0].sum()
print(f"{n} envs x {steps} steps; env-steps with a pad contact {touching/(n*steps):.3f}; solves (after the transient) {solves}")
for name, row in zip(("grad+Hessian", "sign", "gradient", "line-search"), h):
    print(f"  {name:13s} passes per solve: mean {(row*np.arange(16)).sum()/max(solves,1):.2f}   " + " ".join(f"{k}:{100.0*v/max(solves,1):.1f}%" for k, v in enumerate(row) if v))
cost = 970*(h[0]*np.arange(16)).sum() + 140*(h[1]*np.arange(16)).sum() + 520*(h[2]*np.arange(16)).sum() + 300*(h[3]*np.arange(16)).sum()
print(f"  estimated instructions per solve: {cost/max(solves,1):.0f};  solves with >= 12 / >= 15 gradient + Hessian passes: {h[0][12:].sum()} / {h[0][15]};  solves that reached their last allowed iteration (whole run): {H.hc_cdbg_lastiter()}")
