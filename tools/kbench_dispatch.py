"""Which step kernel for which batch size: so100_step at large batches with the 4-wave kernel so100_step_mw
forced on (SO100_MW_MAX_ENVS = 2^30) and forced off (= 0 -> so100_step_fused), per physics flag set.
    python tools/kbench_dispatch.py [envs ...]          -> table on stdout (profiles/r03_large_batch_dispatch.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.lib import So100Sim, F_CUBE_PINNED, F_NOPADS, F_REFERENCE, F_CONTACT5

sizes = [int(a) for a in sys.argv[1:]] or [32768, 65536, 131072, 262144]
print(f"# {torch.cuda.get_device_name(0)}; so100_step, random actions after 40 settling steps, 20 timed launches; G env-steps/s")
print(f"{'kind':>4} {'flags':>8} {'envs':>8} {'step_mw':>9} {'step_fused':>11}  winner")
for kind, name, flags in ((1, "free", F_CUBE_PINNED), (1, "nopads", F_NOPADS), (1, "ref", F_REFERENCE), (1, "c5", F_CONTACT5), (2, "ref", F_REFERENCE), (5, "ref", F_REFERENCE)):
    for n in sizes:
        rate = {}
        for which, mw in (("mw", 1 << 30), ("fused", 0)):
            os.environ["SO100_MW_MAX_ENVS"] = str(mw)
            sim = So100Sim(kind, n, flags=flags, seed=99); sim.reset()
            a = (torch.rand(n, 6, device="cuda")*2 - 1).contiguous()
            for _ in range(40 if name != "free" else 3):
                sim.step(a)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20):
                sim.step(a)
            e1.record(); torch.cuda.synchronize()
            rate[which] = n/(e0.elapsed_time(e1)/20)/1e6
            sim.close()
        print(f"{kind:>4} {name:>8} {n:>8} {rate['mw']:>9.3f} {rate['fused']:>11.3f}  {'mw' if rate['mw'] > rate['fused'] else 'fused'}", flush=True)
del os.environ["SO100_MW_MAX_ENVS"]
