"""profiling target: so100_step (4-wave kernel) with every env resting on its pads (sustained pad/floor contact)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_NOPADS, F_REFERENCE
n = 4096
flags = F_REFERENCE if (len(sys.argv) < 2 or sys.argv[1] == "ref") else F_NOPADS
sim = So100Sim(1, n, flags=flags, seed=1, max_episode_steps=0)
sim.reset()
qp = torch.zeros(13, n, device="cuda"); qp[9] = 1.0; qp[6] = 0.2; qp[7] = -0.2; qp[8] = 0.0099
for i, v in enumerate([0.0, -1.6, 1.9, 1.5, 0.0, 0.3]): qp[i] = v
sim.set_state(qp, torch.zeros(12, n, device="cuda"))
a = torch.zeros(n, 6, device="cuda"); a[:, 1] = 1.0
for t in range(60): sim.step(a)
a.zero_()
for t in range(40): sim.step(a)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for t in range(50): sim.step(a)
e1.record(); torch.cuda.synchronize()
cs = sim.get_field("contact_stat", dtype=torch.int32)
print(f"flags {flags}: so100_step {e0.elapsed_time(e1)/50*1e3:.1f} us; envs in contact {float(((cs & 255) > 0).float().mean()):.3f} max contacts {int((cs&255).max())} residual {float(sim.get_field('solver_residual').max()):.2e}")
