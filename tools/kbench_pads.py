"""Cost of the pad-contact variants: per-step time of the persistent rollout kernel and of so100_step at 4096 envs for
NOPADS / REFERENCE (pad/floor) / CONTACT5 (+ pad/cube), (a) "held": every arm lifted by a constant shoulder action (no contact occurs: the price of detection alone), (b) "raised":
zero action from a raised pose (Env01's ctrl = measured angle lets the arms sag onto the floor and rest there) and (c) under the
random-init policy from reset (arms hit the floor all the time).
    [SO100_LIB=...] python tools/kbench_pads.py [envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so100_mujoco_rl_amd.lib import So100Sim, F_NOPADS, F_REFERENCE, F_CONTACT5, POLICY_TENSORS, SB3_STATE_DICT_KEYS
from so100_mujoco_rl_amd.collector import RolloutCollector

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 64
for flags, name in ((F_NOPADS, "nopads"), (F_REFERENCE, "reference"), (F_CONTACT5, "contact5"))[int(os.environ.get("SKIP", "0")):]:
    for mode in ("held", "raised", "random"):
        sim = So100Sim(1, n, flags=flags, seed=1, contact_iters=int(os.environ.get('CIT', '6')))
        sd = RolloutCollector.random_policy_state(15, sim.device, seed=0)
        if mode in ("raised", "held"):
            sd["action_net.weight"].zero_(); sd["action_net.bias"].zero_(); sd["log_std"].fill_(-30.0)
        if mode == "held":                      # a constant lifting action on the shoulder: the arm is driven against its upper stop and stays off the floor
            sd["action_net.bias"][1] = float(os.environ.get("HOLD", "-1.0"))
        sim.set_policy({k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS})
        sim.reset()
        if mode in ("raised", "held"):
            qp = torch.zeros(13, n, device="cuda"); qp[9] = 1.0; qp[6] = 0.2; qp[7] = -0.2; qp[8] = 0.0099
            for i, v in enumerate([0.0, -1.7, 1.2, 0.3, 0.0, 0.3]): qp[i] = v
            sim.set_state(qp, torch.zeros(12, n, device="cuda"))
        else:
            g = torch.Generator(device="cuda"); g.manual_seed(1)
            sim.set_field("elapsed_steps", torch.randint(0, 4000, (n,), device="cuda", generator=g, dtype=torch.int32))
        buf = torch.zeros(T, n, 25, device="cuda")
        for i in range(3): sim.rollout(buf, i*T)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(6): sim.rollout(buf, (3 + i)*T)
        e1.record(); torch.cuda.synchronize()
        roll = e0.elapsed_time(e1)/6/T*1e3
        cs = sim.get_field("contact_stat", dtype=torch.int32); res = sim.get_field("solver_residual")
        a = torch.zeros(n, 6, device="cuda") if mode in ("raised", "held") else (torch.rand(n, 6, device="cuda")*2 - 1)
        if mode == "held": a[:, 1] = float(os.environ.get("HOLD", "-1.0"))
        for i in range(10): sim.step(a)
        torch.cuda.synchronize(); e0.record()
        for i in range(50): sim.step(a)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:10s} {mode:7s} N={n}: rollout {roll:8.1f} us/step   so100_step {e0.elapsed_time(e1)/50*1e3:8.1f} us   envs in contact {float(((cs & 255) > 0).float().mean()):.3f}  "
              f"max contacts {int((cs & 255).max())}  dropped {int((cs >> 8).max())}  max residual {float(res.max()):.2e}", flush=True)
        sim.close()
