import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED, F_REFERENCE
from so100_mujoco_rl_amd.collector import RolloutCollector, SB3_STATE_DICT_KEYS, POLICY_TENSORS
T = 64
for flags, name in ((F_CUBE_PINNED, "free"), (F_REFERENCE, "ref")):
    for n in (4096, 6144, 8192, 10240, 12288, 14336, 16384):
        sim = So100Sim(1, n, flags=flags, seed=1); sim.reset()
        sd = RolloutCollector.random_policy_state(sim.obs_dim, sim.device, seed=0)
        sim.set_policy({k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS})
        buf = torch.empty(T, n, sim.obs_dim + 10, device="cuda")
        for i in range(2): sim.rollout(buf, i * T)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(4): sim.rollout(buf, (2 + i) * T)
        e1.record(); torch.cuda.synchronize()
        print(f"{name} N={n:6d} WGs={n//64:4d}  {e0.elapsed_time(e1)/4/T*1e3:7.1f} us/step", flush=True)
        sim.close()
