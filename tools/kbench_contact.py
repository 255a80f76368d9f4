import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_REFERENCE, F_FRICTIONLOSS, F_LIMITS, F_FLOOR
n = 4096
for flags, nm in ((F_REFERENCE, "fric+lim+floor"), (F_FLOOR, "floor only")):
  for ci in (1, 2, 4, 8):
    sim = So100Sim(1, n, flags=flags, contact_iters=ci, seed=1); sim.reset()
    act = torch.zeros(n, 6, device="cuda")
    out = []
    for blk in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): sim.step(act)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10 * 1e3)
    z = sim.get_field("cube_z"); vz = sim.get_field("cube_vz")
    print(f"{nm:15s} contact_iters={ci}: us/step per 10-step block {['%.0f' % o for o in out]}  z={z[0].item():.6f} vz={vz[0].item():.2e} zspread={(z.max()-z.min()).item():.1e}", flush=True)
    sim.close()
