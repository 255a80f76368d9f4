"""so100_step at large batch (the one-wave throughput kernel): env-steps/s for the library selected by SO100_LIB.
    [SO100_LIB=...] [PHYS=free|nopads|ref|c5] [KIND=1..6] python tools/kbench_large.py [envs ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED, F_NOPADS, F_REFERENCE, F_CONTACT5, LIB_PATH
PHYS = os.environ.get('PHYS', 'free'); FLAGS = {'free': F_CUBE_PINNED, 'nopads': F_NOPADS, 'ref': F_REFERENCE, 'c5': F_CONTACT5}[PHYS]
for n in [int(a) for a in sys.argv[1:]] or [65536, 262144, 1 << 20]:
    sim = So100Sim(int(os.environ.get('KIND', '1')), n, flags=FLAGS, seed=99); sim.reset()
    a = (torch.rand(n, 6, device="cuda")*2 - 1).contiguous()
    for _ in range(40 if PHYS != 'free' else 3): sim.step(a)          # (constrained variants: let the arms settle / sag first)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): sim.step(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print(f"{os.path.basename(LIB_PATH)} kind {os.environ.get('KIND', '1')} {PHYS} N={n}: {ms*1e3:.1f} us per step, {n/ms/1e6:.3f} G env-steps/s", flush=True)
    sim.close()
