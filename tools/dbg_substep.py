import sys, numpy as np, torch, ctypes as C
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import substep_harness as SH
from oracle import so100_oracle as O
import test_gpu_substep_parity as G
L, M = SH.L, SH.M
m, nsub = 96, 24
for n, flags in ((96, 23), (8192, 23), (96, 21)):
    qpos, qvel, act = SH.floor_batch(m, 0)
    dev = G.HipDevice(n, m, flags)
    ds = SH.oracle_states(qpos, qvel)
    for s in range(nsub):
        for d in ds: SH.round_state_to_fp32(d)
        q32 = np.stack([O.arr(d.qpos).copy() for d in ds]); v32 = np.stack([O.arr(d.qvel).copy() for d in ds])
        gq, gv, gcount, gsig, gres = dev(q32, v32, act)
        for i, d in enumerate(ds):
            O.arr(d.ctrl)[:] = (q32[i,:6].astype(np.float32) + act[i].astype(np.float32)*SH.JS).astype(np.float64)
            L.so100o_step(C.byref(M), C.byref(d), flags, -1, 1)
            dvo = O.arr(d.qvel)[:6] - v32[i,:6]; dvg = gv[i,:6] - v32[i,:6]
            err = np.abs(dvg-dvo).max()
            if gres[i] > 1e-3 or err > 5e-5:
                print(f"n={n} flags={flags} substep {s} env {i}: residual {gres[i]:.3e} err {err:.3e} count {gcount[i]} oracle ncon {d.ncon} |dvo| {np.abs(dvo).max():.3e}")
                print("   q32", repr(q32[i].tolist())); print("   v32", repr(v32[i].tolist())); print("   act", repr(act[i].tolist()))
    print(f"n={n} flags={flags} done")
