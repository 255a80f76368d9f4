import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oracle import so100_oracle as O
from so100_mujoco_rl_amd.lib import So100Sim
ARM = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_CUBE_PINNED
n, steps = 16, 1000
for iters in (1, 2, 3):
    rs = np.random.RandomState(3)
    sim = So100Sim(1, n, flags=ARM, solver_iters=iters, max_episode_steps=0, seed=2)
    orc = [O.OracleEnv(1, flags=ARM, iters=0, seed=2, env_id=i) for i in range(n)]
    for e in orc: e.e.max_episode_steps = 0
    inj = rs.random_sample((n, 16)).astype(np.float32)
    sim.reset(inject=torch.from_numpy(inj).cuda()); [e.reset(inject=inj[i]) for i, e in enumerate(orc)]
    a = np.zeros((n, 6), np.float32); wq = wv = 0.0
    for t in range(steps):
        a = np.clip(a + rs.uniform(-0.2, 0.2, (n, 6)), -1, 1).astype(np.float32)
        sim.step(torch.from_numpy(a).cuda())
        for i, e in enumerate(orc): e.step(a[i])
        if t % 50 == 49:
            qpos, qvel = sim.get_state()
            qo = np.stack([O.arr(e.d.qpos)[:6].copy() for e in orc]); vo = np.stack([O.arr(e.d.qvel)[:6].copy() for e in orc])
            wq = max(wq, np.abs(qpos[:6].cpu().numpy().T - qo).max()); wv = max(wv, np.abs(qvel[:6].cpu().numpy().T - vo).max())
    print(f"solver_iters={iters}: max |dq| {wq:.2e} rad, max |dqvel| {wv:.2e} rad/s", flush=True)
