import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from so100_mujoco_rl_amd.lib import So100Sim, F_NOPADS, F_REFERENCE
for n in (64, 4096, 16384+64):
    sim = So100Sim(1, n, flags=F_REFERENCE, seed=1, max_episode_steps=0)
    sim.reset()
    qp = torch.zeros(13, n, device="cuda"); qp[9] = 1.0; qp[6] = 0.2; qp[7] = -0.2; qp[8] = 0.0099
    for i, v in enumerate([0.0, -1.7, 1.2, 0.3, 0.0, 0.3]): qp[i] = v
    sim.set_state(qp, torch.zeros(12, n, device="cuda"))
    a = torch.zeros(n, 6, device="cuda")
    for t in range(3):
        sim.step(a)
        cs = sim.get_field("contact_stat", dtype=torch.int32); q, v = sim.get_state()
        print(n, t, "cstat", cs[:4].tolist(), "max", int((cs & 255).max()), "q", [round(x, 4) for x in q[:6, 0].tolist()], "res", float(sim.get_field("solver_residual").max()))
