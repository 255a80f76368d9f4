"""How far from the converged constraint solve are 2 block-PGS sweeps under a wild / the bench's random policy?  One env step from identical states with
solver_iters = 2, 3, 4 against solver_iters = 32 (converged), NOPADS physics (no contact Newton involved), 4096 envs."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_NOPADS, F_FRICTIONLOSS, F_LIMITS, F_CUBE_PINNED
N = 4096
for pol in ("uniform", "gauss_clipped", "random_walk"):
    base = So100Sim(1, N, flags=F_NOPADS, seed=3, max_episode_steps=0); base.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    a_prev = torch.zeros(N, 6, device="cuda")
    def draw():
        global a_prev
        if pol == "uniform": return (torch.rand(N, 6, device="cuda", generator=g)*2 - 1).contiguous()
        if pol == "gauss_clipped": return torch.randn(N, 6, device="cuda", generator=g).clamp(-1, 1).contiguous()
        a_prev = (a_prev + 0.2*torch.randn(N, 6, device="cuda", generator=g)).clamp(-1, 1); return a_prev.contiguous()
    for _ in range(60): base.step(draw())
    q0, v0 = base.get_state(); q0 = q0.clone(); v0 = v0.clone()
    a = draw()
    out = {}
    for it in (2, 3, 4, 32):
        s = So100Sim(1, N, flags=F_NOPADS, seed=3, max_episode_steps=0, solver_iters=it); s.reset(); s.set_state(q0, v0)
        s.set_field("elapsed_steps", base.get_field("elapsed_steps", dtype=torch.int32)) if False else None
        s.step(a); q, v = s.get_state(); out[it] = (q.clone(), v.clone(), s.get_field("solver_residual").clone())
    for it in (2, 3, 4):
        dq = (out[it][0][:6] - out[32][0][:6]).abs().max(0).values; dv = (out[it][1][:6] - out[32][1][:6]).abs().max(0).values
        r = out[it][2]
        print(f"{pol:14s} sweeps {it}: |dq| median {dq.median():.2e} p99 {dq.quantile(0.99):.2e} max {dq.max():.2e};  |dv| median {dv.median():.2e} p99 {dv.quantile(0.99):.2e} max {dv.max():.2e};  "
              f"residual row > 1e-2 in {float((r > 1e-2).float().mean())*100:.1f} % of envs")
