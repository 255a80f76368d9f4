"""Cost of the in-kernel auto-reset path: rollout step time vs episode length (shorter episodes => more steps in which some lane of
a wavefront resets).   python tools/reset_cost.py"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_REFERENCE, F_CUBE_PINNED
from so100_mujoco_rl_amd.collector import RolloutCollector, SB3_STATE_DICT_KEYS, POLICY_TENSORS
T, n = 64, 4096
for kind, flags, name in ((1, F_CUBE_PINNED, "Env01 contact disabled"), (1, F_REFERENCE, "Env01 reference"), (5, F_REFERENCE, "Env05 reference")):
    for L in (0, 4000, 400, 64, 16, 4):
        sim = So100Sim(kind, n, flags=flags, seed=1, max_episode_steps=L); sim.reset()
        if L:
            sim.set_field("elapsed_steps", torch.randint(0, L, (n,), device="cuda", dtype=torch.int32))
        sd = RolloutCollector.random_policy_state(sim.obs_dim, sim.device, seed=0)
        sim.set_policy({k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS})
        buf = torch.empty(T, n, sim.obs_dim + 10, device="cuda")
        for i in range(2): sim.rollout(buf, i * T)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(6): sim.rollout(buf, (2 + i) * T)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:24s} max_episode_steps {L:5d}: {e0.elapsed_time(e1)/6/T*1e3:7.2f} us/step   dones/step {float((buf[..., -3] > 0).float().sum())/T:7.1f}", flush=True)
        sim.close(); del sim, buf; gc.collect(); torch.cuda.synchronize()
