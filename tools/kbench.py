"""Kernel-only timing of so100_step_fused for a few (env kind, physics flags, N) points (needs a GPU).
    python tools/kbench.py [reps]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED, F_REFERENCE, F_FRICTIONLOSS, F_LIMITS

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pts = [(1, F_CUBE_PINNED, 4096, "env01 free"), (1, F_FRICTIONLOSS | F_LIMITS | F_CUBE_PINNED, 4096, "env01 fric+lim"),
       (1, F_REFERENCE, 4096, "env01 reference"), (2, F_REFERENCE, 16384, "env02 reference"), (5, F_REFERENCE, 8192, "env05 reference"),
       (1, F_CUBE_PINNED, 65536, "env01 free"), (1, F_CUBE_PINNED, 262144, "env01 free"), (1, F_CUBE_PINNED, 1048576, "env01 free"),
       (1, F_REFERENCE, 262144, "env01 reference")]
for kind, flags, n, name in pts:
    sim = So100Sim(kind, n, flags=flags, seed=1)
    sim.reset()
    act = torch.rand(n, 6, device="cuda") * 2 - 1
    for _ in range(10):
        sim.step(act)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        sim.step(act)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:18s} N={n:8d}  {ms*1e3:9.1f} us/step  {n/ms/1e3:9.2f} M env-steps/s  ({n*16/ms/1e6:7.2f} G substeps/s)", flush=True)
    sim.close()
