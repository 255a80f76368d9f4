"""Long soak of the default rollout path: reference physics, staggered episodes, random policy; checks every chunk for
non-finite values, the non-finite guard's latch, counters and device memory growth.   [PHYS=ref|c5|nopads|links|links_c5|proxies|proxies_c5] python tools/soak.py [seconds] [envs] [env_id]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so100_mujoco_rl_amd.vec_env import So100VecEnv
from so100_mujoco_rl_amd.collector import RolloutCollector
from so100_mujoco_rl_amd.lib import F_REFERENCE, F_CONTACT5, F_NOPADS, F_REFERENCE_LINKS, F_REFERENCE_PROXIES, F_PADS_CUBE

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
env_id = sys.argv[3] if len(sys.argv) > 3 else "Env05-v1"
flags = {"ref": F_REFERENCE, "c5": F_CONTACT5, "nopads": F_NOPADS, "links": F_REFERENCE_LINKS, "links_c5": F_REFERENCE_LINKS | F_PADS_CUBE,
         "proxies": F_REFERENCE_PROXIES, "proxies_c5": F_REFERENCE_PROXIES | F_PADS_CUBE}[os.environ.get("PHYS", "ref")]
env = So100VecEnv(env_id, n, flags=flags, seed=0, stagger_episodes=True)
sd = RolloutCollector.random_policy_state(env.sim.obs_dim, env.device)
sd["log_std"] = sd["log_std"] + 0.5                          # a wilder policy than the initial one
col = RolloutCollector(env, sd, T=64)
col.collect(); torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated(); t0 = time.time(); steps = 0; episodes = 0; chunks = 0; last = t0
while time.time() - t0 < secs:
    b = col.collect(); chunks += 1; steps += 64 * n
    ok = bool(torch.isfinite(b["obs"]).all() & torch.isfinite(b["rewards"]).all() & torch.isfinite(b["values"]).all() & torch.isfinite(b["log_probs"]).all())
    episodes += int((b["dones"] > 0).sum())
    assert ok, f"non-finite value in chunk {chunks}"
    if time.time() - last > 15:
        last = time.time(); print(f"  {time.time()-t0:5.0f} s  {steps/1e9:6.2f} G env-steps  {episodes} episodes", flush=True)
del b                                                        # (the last chunk's derived tensors -- done / truncated masks -- are live allocations)
grow = torch.cuda.memory_allocated() - mem0                 # (before the queries below allocate their outputs)
qpos, qvel = env.sim.get_state()
res = env.sim.get_field("solver_residual"); cst = env.sim.get_field("contact_stat", dtype=torch.int32)
print(f"  solver residual of the last step: max {float(res.max()):.3g}; pad contacts: max per substep {int((cst & 255).max())}, dropped over budget {int((cst >> 8).max())}")
bad = int(env.sim.bad_state_mask().sum())
print(f"{env_id} x {n}: {steps/1e9:.2f} G env-steps in {time.time()-t0:.0f} s ({steps/(time.time()-t0)/1e6:.0f} M/s), {episodes} episodes, bad-state envs {bad}, "
      f"state finite {bool(torch.isfinite(qpos).all() & torch.isfinite(qvel).all())}, max |qvel| {float(qvel.abs().max()):.1f}, device memory growth {grow} B")
assert bad == 0 and grow == 0
