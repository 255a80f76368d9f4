"""What a two-kernel split of so100_step could reach for the pad-contact variants at large batches: the one-wave kernel's
rate (a) on a batch in which no arm is near the table (a constant lifting action: the contact Newton never runs), (b) on a batch in which EVERY arm
starts within a few centimetres of / on the table and moves under random actions (what a compacted "contact-prone" sub-batch looks like), and (c) today's
mixed batch from reset under random actions, with the share of envs whose pads are within MARGIN of the table.  Prediction for a split that sends the
contact-prone share f to its own launch: 1 / ((1 - f) / r_a + f / r_b).  (Round 3 BUILT that split -- a per-env "within 3 cm of the table" flag written at the
end of every step, a compaction kernel, two launches of so100_step_fused -- and measured 0.155 -> 0.172 G at 5 % of the envs in contact, 0.100 G at 19 %: the
compacted launch runs at rate (b), and with a 3 cm margin it receives a third of the batch.  Removed again; profiles/r03_large_batch_split_probe.txt.)
    python tools/kbench_split.py [envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so100_mujoco_rl_amd.lib import So100Sim, F_REFERENCE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
os.environ["SO100_MW_MAX_ENVS"] = "0"                         # the one-wave kernel

def rate(sim, act_fn, warm, reps):
    for _ in range(warm): sim.step(act_fn())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): sim.step(act_fn())
    e1.record(); torch.cuda.synchronize()
    cs = sim.get_field("contact_stat", dtype=torch.int32)
    return n/(e0.elapsed_time(e1)/reps)/1e6, float(((cs & 255) > 0).float().mean())

g = torch.Generator(device="cuda"); g.manual_seed(0)
rnd = lambda: (torch.rand(n, 6, device="cuda", generator=g)*2 - 1).contiguous()
# (a) held: shoulder lifted, nothing near the table
sim = So100Sim(1, n, flags=F_REFERENCE, seed=1); sim.reset()
qp = torch.zeros(13, n, device="cuda"); qp[9] = 1.0; qp[6] = 0.2; qp[7] = -0.2; qp[8] = 0.0099
for i, v in enumerate([0.0, -1.7, 1.2, 0.3, 0.0, 0.3]): qp[i] = v
sim.set_state(qp, torch.zeros(12, n, device="cuda"))
def held():
    a = rnd(); a[:, 1] = -1.0; return a
ra, fa = rate(sim, held, 10, 20); sim.close()
print(f"(a) held off the table      : {ra:.3f} G env-steps/s, envs in contact {fa:.3f}")
# (b) every arm at the table: poses with the lowest pad corner within +-1 cm of the floor, random actions, timed over the first 12 steps
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_oracle_contacts import floor_poses
poses = np.array(floor_poses(512, 1, band=0.01)); rs = np.random.RandomState(0)
sim = So100Sim(1, n, flags=F_REFERENCE, seed=2, max_episode_steps=0); sim.reset()
qp = torch.zeros(13, n, device="cuda"); qp[9] = 1.0; qp[6] = 0.2; qp[7] = -0.2; qp[8] = 0.0099
qp[:6] = torch.from_numpy(poses[rs.randint(0, len(poses), n)].T.astype(np.float32)).cuda()
sim.set_state(qp, torch.zeros(12, n, device="cuda"))
rb, fb = rate(sim, rnd, 2, 12); sim.close()
print(f"(b) every arm at the table  : {rb:.3f} G env-steps/s, envs in contact {fb:.3f}")
# (c) today: from reset, random actions, arms settled
sim = So100Sim(1, n, flags=F_REFERENCE, seed=3); sim.reset()
rc, fc = rate(sim, rnd, 40, 20)
print(f"(c) mixed batch (today)     : {rc:.3f} G env-steps/s, envs in contact {fc:.3f}")
for f in (fc, 1.5*fc, 2*fc, 0.3):
    print(f"    split prediction with a contact-prone share of {f:.2f}: {1.0/((1 - f)/ra + f/rb):.3f} G env-steps/s")
