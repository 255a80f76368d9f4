"""Per-step cost model of the persistent rollout kernel: time so100_rollout at several frame_skip values and fit
t(step) = a + b * frame_skip  (a = policy + task layer + rollout-row traffic, b = one physics substep).  Needs a GPU.
    [SO100_LIB=...] python tools/kbench_rollout.py [envs]
"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED, F_REFERENCE, F_FRICTIONLOSS, F_LIMITS
from so100_mujoco_rl_amd.collector import RolloutCollector, SB3_STATE_DICT_KEYS, POLICY_TENSORS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 64
for kind, flags, name in [(1, F_CUBE_PINNED, "env01 free"), (1, F_FRICTIONLOSS | F_LIMITS | F_CUBE_PINNED, "env01 arm rows"), (1, F_REFERENCE, "env01 reference"),
                          (5, F_REFERENCE, "env05 reference")]:
    xs, ys = [], []
    for fs in (1, 4, 8, 16, 32):
        sim = So100Sim(kind, n, flags=flags, seed=1, frame_skip=fs)
        sim.reset()
        sd = RolloutCollector.random_policy_state(sim.obs_dim, sim.device, seed=0)
        sim.set_policy({k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS})
        buf = torch.empty(T, n, sim.obs_dim + 10, device="cuda")
        for i in range(3):
            sim.rollout(buf, i * T)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        reps = 8
        for i in range(reps):
            sim.rollout(buf, (3 + i) * T)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps / T * 1e3
        xs.append(fs); ys.append(us)
        sim.close(); del sim, buf; gc.collect(); torch.cuda.synchronize()      # freeing buffers later would stall a timed loop
    b, a = np.polyfit(xs, ys, 1)
    print(f"{name:18s} N={n}  " + "  ".join(f"fs{f}:{y:6.2f}" for f, y in zip(xs, ys)) + f"  us/step   fit: a = {a:5.2f} us, b = {b:5.3f} us/substep", flush=True)
