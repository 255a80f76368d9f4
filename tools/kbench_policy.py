import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sim = So100Sim(1, n, flags=F_CUBE_PINNED); sim.reset()
g = torch.Generator(device="cuda"); g.manual_seed(0)
r = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.2).contiguous()
sim.set_policy({"pi_w0": r(64, 15), "pi_b0": r(64), "pi_w1": r(64, 64), "pi_b1": r(64), "mu_w": r(6, 64), "mu_b": r(6), "log_std": r(6),
                "vf_w0": r(64, 15), "vf_b0": r(64), "vf_w1": r(64, 64), "vf_b1": r(64), "v_w": r(1, 64), "v_b": r(1)})
act = torch.zeros(n, 6, device="cuda"); row = torch.zeros(n, 25, device="cuda")
for _ in range(10): sim.policy_forward(sim.obs, act, 0, rollout_row=row)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for i in range(200): sim.policy_forward(sim.obs, act, i, rollout_row=row)
e1.record(); torch.cuda.synchronize()
print(f"N={n}: policy kernel (matrix-core) {e0.elapsed_time(e1)/200*1e3:.1f} us")
