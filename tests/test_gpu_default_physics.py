"""The kernels users get BY DEFAULT (So100VecEnv / So100Sim / main.py default to F_REFERENCE = pad/floor contacts on; the collector
defaults to the persistent rollout kernel so100_rollout_fused<K, 23 | 55, 4>) against the stepwise kernels and the fp64 oracle, from
contact-rich starts: arms injected with their finger pads at the table (and, for F_CONTACT5, a cube between the closing jaws), policy
biased to servo the shoulder further down, raised log_std.  TimeLimit resets inside the chunk.

Per env STEP (16 fused substeps) fp32 and fp64 may see a corner make / break contact a substep apart (~1 m/s impact moved by 2 ms: physics,
see tests/test_gpu_contacts.py) -- the per-SUBSTEP test (tests/test_gpu_substep_parity.py) has no such effect and is exact on every env.
Here every (env, step) pair is classified and the class shares are printed and bounded; nothing is dropped from the comparison.
"parity unpinned (physics)": the oracle restates MuJoCo's algorithm; MuJoCo is not available."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import substep_harness as SH                              # noqa: E402
from oracle import so100_oracle as O                      # noqa: E402  (the checker)

REFP = O.F_REFERENCE
C5 = O.F_CONTACT5


def _contact_rich_state(n, flags, seed):
    qpos, qvel, _ = SH.floor_batch(n, seed)
    if flags & O.F_PADS_CUBE:                               # second half: the closing-jaw grasp of BASELINE.json configs[4]
        gq, gv, _ = SH.grasp_batch(n - n//2, seed + 1)
        qpos[n//2:] = gq; qvel[n//2:] = gv
    return qpos.astype(np.float32), qvel.astype(np.float32)


def _start(env, col, qpos, qvel, kind):
    sim = env.sim
    sim.reset()
    sim.set_state(torch.from_numpy(np.ascontiguousarray(qpos.T)).cuda(), torch.from_numpy(np.ascontiguousarray(qvel.T)).cuda())
    if kind in (3, 4, 5):                                    # look-at kinds servo to their COMMANDED angles (env_base_02.py:85-86)
        for i in range(6):
            sim.set_field(f"cmd{i}", torch.from_numpy(np.ascontiguousarray(qpos[:, i])).cuda())
    col._started = True


PROXIES = O.F_LINKS_FLOOR | O.F_LINKS_CUBE                 # every capsule proxy pair: the run-time-flags kernels so100_rollout_fused / so100_step_mw<K, -1>


@pytest.mark.parametrize("kind,flags", [(1, REFP), (2, C5), (5, REFP), (6, REFP), (3, REFP), (1, REFP | PROXIES), (2, C5 | PROXIES)])
def test_default_rollout_kernel_vs_stepwise_and_oracle(kind, flags):
    from so100_mujoco_rl_amd.vec_env import So100VecEnv
    from so100_mujoco_rl_amd.collector import RolloutCollector
    n, T, T2, tl = 200, 10, 5, 12                            # tail workgroup partially filled; a second, shorter chunk; TimeLimit hits inside it
    qpos, qvel = _contact_rich_state(n, flags, 10 + kind)
    runs = []
    for persistent in (True, False):
        env = So100VecEnv(kind, n, flags=flags, seed=4, max_episode_steps=tl, solver_iters=4, contact_iters=30)
        sd = RolloutCollector.random_policy_state(env.sim.obs_dim, env.device, seed=2)
        sd["action_net.bias"][1] = 0.5                       # shoulder down: the pads stay on / are pressed into the table
        if flags & O.F_PADS_CUBE:
            sd["action_net.bias"][5] = -0.5                  # ... and the jaw closes
        sd["log_std"] = sd["log_std"] + 0.3                  # raised: contact-rich AND irregular
        col = RolloutCollector(env, sd, T=T, persistent=persistent, bootstrap_truncated=False)
        _start(env, col, qpos, qvel, kind)
        b = {k: v.clone() for k, v in col.collect().items()}
        cs1 = env.sim.get_field("contact_stat", dtype=torch.int32).clone()      # (before the TimeLimit sends every arm back to its start pose)
        b2 = {k: v.clone() for k, v in col.collect(T2).items()}
        for k in ("obs", "actions", "rewards", "dones", "values", "log_probs"):
            b[k] = torch.cat([b[k], b2[k]], 0)
        b["last_obs"] = b2["last_obs"]
        q, v = env.sim.get_state()
        runs.append((b, q.clone(), v.clone(), cs1))
    T = T + T2
    (a, aq, av, acs), (s, sq, sv, scs) = runs
    # ---- persistent kernel vs stepwise kernels: the same actions (the policy noise stream is shared), so rows can be compared one by one
    assert torch.equal(a["dones"], s["dones"]) and a["dones"].sum() > 0
    d_obs = (a["obs"] - s["obs"]).abs().amax(-1); d_last = (a["last_obs"] - s["last_obs"]).abs().amax(-1)
    d_all = torch.cat([d_obs.flatten(), d_last.flatten()])
    tight = float((d_all < 1e-5).float().mean()); loose = float((d_all < 2e-2).float().mean())
    # ---- 32 sampled envs against the oracle's env step (its Newton), replaying the persistent run's actions
    sample = list(range(0, n, n//32))[:32]
    act = a["actions"].clamp(-1, 1).cpu().numpy(); obs = a["obs"].cpu().numpy(); rew = a["rewards"].cpu().numpy(); last = a["last_obs"].cpu().numpy()
    classes = {"tight": 0, "event": 0, "fail": 0}; worst_tight = 0.0; errs = []
    for i in sample:
        e = O.OracleEnv(kind, flags=flags, iters=-1, seed=4, env_id=i); e.e.max_episode_steps = tl
        e.reset()
        O.arr(e.d.qpos)[:] = qpos[i].astype(np.float64); O.arr(e.d.qvel)[:] = qvel[i].astype(np.float64)
        if kind in (3, 4, 5):
            O.arr(e.e.cmd)[:] = qpos[i, :6].astype(np.float64)
        for t in range(T):
            o, r = e.step(act[t, i], autoreset=True)[:2]
            og = obs[t + 1, i] if t + 1 < T else last[i]
            scale = 5.0 if kind in (3, 4, 5) else 1.0        # look-at observations carry 5 * (pixel centre) in their last two entries
            err = max(np.abs(og[:6] - o[:6]).max(), np.abs(og[6:] - o[6:]).max()/scale, abs(rew[t, i] - r)/20.0)
            errs.append(err)
            if err < 2e-5:
                classes["tight"] += 1; worst_tight = max(worst_tight, err)
            elif err < 2e-2:
                classes["event"] += 1
            else:
                classes["fail"] += 1
    touched = float(((acs & 255) > 0).float().mean())
    print(f"[default rollout kernel, kind {kind} flags {flags}] persistent vs stepwise: {tight:.3f} of rows within 1e-5, {loose:.3f} within 2e-2; "
          f"vs oracle ({len(sample)} envs x {T} steps): {classes}, median err {np.median(errs):.2e}; envs with a pad contact in step 10 {touched:.2f}")
    assert torch.isfinite(a["obs"]).all() and torch.isfinite(a["rewards"]).all() and torch.isfinite(aq).all() and torch.isfinite(av).all()
    assert touched > 0.2                                     # the contact path really ran
    assert (acs >> 8).max() == 0 and (scs >> 8).max() == 0   # nothing over the contact budget
    assert loose == 1.0 and tight > 0.5
    assert classes["fail"] == 0 and classes["tight"] >= 0.5*len(errs) and np.median(errs) < 2e-5


def test_contact5_determinism_and_shard_invariance_4096():
    """F_CONTACT5 at the BASELINE per-GPU batch: 4096 closing-jaw grasps (cube between the pads, coupled 12-unknown solves) are
    bitwise reproducible and unchanged by sharding the batch over two handles (env_id_offset), like F_REFERENCE in
    test_pads_keep_the_gripper_above_the_floor_at_full_size."""
    from so100_mujoco_rl_amd.lib import So100Sim
    n = 4096
    qpos, qvel, act = SH.grasp_batch(n, 7)
    rs = np.random.RandomState(1)
    act = act + rs.uniform(-0.3, 0.3, act.shape).astype(np.float32); act[:, 5] = -1.0

    def run(n_envs, off, sl):
        sim = So100Sim(1, n_envs, flags=C5, seed=9, env_id_offset=off, max_episode_steps=0)     # (Env02 would re-randomise the cube on its first step: Q1)
        sim.reset()
        sim.set_state(torch.from_numpy(np.ascontiguousarray(qpos[sl].T, np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(qvel[sl].T, np.float32)).cuda())
        a = torch.from_numpy(np.ascontiguousarray(act[sl])).cuda(); coupled = 0
        outs = []
        for t in range(8):
            ob, r, d, _ = sim.step(a)
            outs.append(torch.cat([ob, r[:, None]], 1).clone())
            coupled = max(coupled, int((sim.get_field("contact_stat", dtype=torch.int32) & 255).max()))
        q, v = sim.get_state()
        return torch.cat(outs, 1), q.clone(), v.clone(), coupled
    full, q, v, coupled = run(n, 0, slice(0, n))
    again, q2, v2, _ = run(n, 0, slice(0, n))
    assert torch.equal(full, again) and torch.equal(q, q2) and torch.equal(v, v2)
    lo, qa, va, _ = run(n//2, 0, slice(0, n//2)); hi, qb, vb, _ = run(n//2, n//2, slice(n//2, n))
    assert torch.equal(torch.cat([lo, hi], 0), full) and torch.equal(torch.cat([qa, qb], 1), q) and torch.equal(torch.cat([va, vb], 1), v)
    assert torch.isfinite(full).all() and coupled >= 8       # face-face manifolds on both sides of the cube


def test_workgroup_balancing_leaves_every_result_bit_identical():
    """so100_rollout deals the envs that ended the previous chunk in pad contact out over the workgroups (csrc/so100_balance.hpp): which
    lane computes an env must not change a single bit of its results.  Half of the batch starts on the table, half parked in the air, so
    the map differs from the identity; three chunks, TimeLimit resets inside."""
    import os
    from so100_mujoco_rl_amd.vec_env import So100VecEnv
    from so100_mujoco_rl_amd.collector import RolloutCollector
    n, T = 1000, 6                                           # 63 workgroups of 16 envs, the last one partially filled
    qpos, qvel = _contact_rich_state(n, REFP, 3)
    qpos[::2, :6] = [0.0, -1.9, 1.6, 0.3, 1.5708, 0.1]; qvel[::2] = 0.0      # every other arm folded up 28 cm above the table
    outs = []
    for bal in ("1", "0"):
        os.environ["SO100_BALANCE"] = bal
        try:
            env = So100VecEnv(1, n, flags=REFP, seed=8, max_episode_steps=11)
        finally:
            del os.environ["SO100_BALANCE"]
        sd = RolloutCollector.random_policy_state(15, env.device, seed=5); sd["action_net.bias"][1] = 0.4
        col = RolloutCollector(env, sd, T=T, persistent=True, bootstrap_truncated=False)
        _start(env, col, qpos, qvel, 1)
        chunks = []
        for c in range(3):
            chunks.append({k: v.clone() for k, v in col.collect().items()})
            if c == 0:                                       # (contact_stat at the end of a chunk is what the next launch's map is dealt from)
                cs = env.sim.get_field("contact_stat", dtype=torch.int32).clone()
        q, v = env.sim.get_state()
        outs.append((chunks, q.clone(), v.clone(), cs, env.sim.ep_length.clone()))
        env.close()
    (ca, qa, va, csa, la), (cb, qb, vb, csb, lb) = outs
    assert ((csa & 255) > 0).float().mean() > 0.1            # the contact path ran, so the map was not the identity
    for a, b in zip(ca, cb):
        for k in ("obs", "actions", "rewards", "dones", "values", "log_probs", "last_obs"):
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(qa, qb) and torch.equal(va, vb) and torch.equal(csa, csb) and torch.equal(la, lb)


def test_link_proxies_keep_the_arm_above_the_table_on_the_gpu():
    """SO100_F_LINKS_FLOOR (stand-in capsules for the arm's collision meshes): 512 arms that reach the table wrist / forearm first are
    servoed further down for 40 env steps.  With the proxies every capsule end rests above -1 mm (< 3 mm dip during the impact), as in
    the oracle; with the reference physics alone (pads only) the same links end centimetres under the table.  Run-time-flags kernels."""
    import ctypes as C
    from so100_mujoco_rl_amd.lib import So100Sim, F_REFERENCE_LINKS
    from test_oracle_contacts import _wrist_first_poses, proxy_bottoms, fresh, L, M
    n = 512
    poses = np.array(_wrist_first_poses(64, 11))
    rs = np.random.RandomState(2)
    QP = np.zeros((n, 13)); QP[:, 9] = 1.0; QP[:, 6:9] = [0.2, -0.3, 0.0099]
    QP[:, :6] = poses[rs.randint(0, len(poses), n)] + rs.uniform(-1, 1, (n, 6))*0.002
    act = np.zeros((n, 6), np.float32); act[:, 1] = 1.0

    def bottoms(qp):
        out = []
        for i in range(0, len(qp), 8):
            d = fresh(qp[i, :6]); L.so100o_kinematics(C.byref(M), C.byref(d)); out.append(proxy_bottoms(d).min())
        return np.array(out)

    def run(flags):
        sim = So100Sim(1, n, flags=flags, seed=9, max_episode_steps=0)
        sim.reset()
        sim.set_state(torch.from_numpy(np.ascontiguousarray(QP.T, np.float32)).cuda(), torch.zeros(12, n, device="cuda"))
        a = torch.from_numpy(act).cuda(); low = 1.0
        for t in range(40):
            sim.step(a)
            if t % 4 == 3:
                low = min(low, bottoms(sim.get_state()[0].cpu().numpy().T).min())
        q, v = sim.get_state()
        return q.cpu().numpy().T, v.cpu().numpy().T, low, sim.get_field("contact_stat", dtype=torch.int32).cpu().numpy(), sim.get_field("solver_residual").cpu().numpy()
    q, v, low, cs, res = run(F_REFERENCE_LINKS)
    final = bottoms(q)
    print(f"[link proxies, GPU] lowest capsule end during the run {low*1e3:.2f} mm, at the end {final.min()*1e3:.2f} mm; max |qvel| {np.abs(v[:, :6]).max():.3f}; "
          f"envs in contact {float(((cs & 255) > 0).mean()):.2f}; worst residual {res.max():.1e}")
    assert np.isfinite(q).all() and np.isfinite(v).all() and (cs >> 8).max() == 0
    assert low > -0.003 and final.min() > -0.001 and ((cs & 255) > 0).mean() > 0.9
    q0, _, _, _, _ = run(REFP)
    assert np.median(bottoms(q0)) < -0.005                   # pads alone do not hold these poses up
    # the oracle from the same states, same actions: same resting configuration (a sample of 16 envs)
    worst = 0.0
    for i in range(0, n, n//16):
        d = fresh(); O.arr(d.qpos)[:] = QP[i].astype(np.float32)
        for t in range(40):
            O.arr(d.ctrl)[:] = (O.arr(d.qpos)[:6].astype(np.float32) + act[i]*np.float32(0.075)).astype(np.float64)
            L.so100o_step(C.byref(M), C.byref(d), F_REFERENCE_LINKS, -1, 16)
        worst = max(worst, np.abs(q[i, :6] - O.arr(d.qpos)[:6]).max())
    assert worst < 2e-2                                       # through the impact (a make / break a substep apart), see tests/test_gpu_contacts.py


def test_link_cube_proxies_push_the_cube_and_survive_whole_rollouts_on_the_gpu():
    """SO100_F_LINKS_CUBE (Rotation_Pitch / Upper_Arm capsules vs the cube) over whole env steps on the GPU.
    (a) 64 cubes placed against a capsule (0.2 - 3 mm deep), arm servoed to hold its pose: after 8 env steps with the flag every cube has been pushed out along
        the contact normal (no overlap deeper than 0.5 mm left) and nothing is non-finite; without it the same cubes fall freely through the link -- and 16 sampled
        envs follow the oracle's env steps.
    (b) every proxy pair on (SO100_F_REFERENCE_PROXIES) under a wild random policy, 1024 envs x 128 steps of Env01 and of Env03 (whose cube is moved around the
        arm's workspace): finite states, contact budget never exceeded.  (No bound on the solver_residual row here: for contact-free substeps it holds the
        acceleration change of the LAST block-PGS sweep, which under a wild policy exceeds 1e-2 rad/s^2 in ~10 % of the envs although the state after the step is
        within 1e-6 rad / 1e-4 rad/s of the converged solve -- tools/pgs_probe.py)"""
    import ctypes as C
    from so100_mujoco_rl_amd.lib import So100Sim, F_REFERENCE_LINKS, F_REFERENCE_PROXIES
    from test_oracle_contacts import link_cube_states, capsule_box, fresh, L, M
    n = 64
    states = link_cube_states(n, 21)
    QP = np.zeros((n, 13))
    for i, (q, c, qc) in enumerate(states):
        QP[i, :6] = q; QP[i, 6:9] = c; QP[i, 9:13] = qc
    act = np.zeros((n, 6), np.float32)

    def overlap(qp):
        """deepest capsule / cube penetration per env (0 when apart), from the oracle's kinematics"""
        out = np.zeros(len(qp))
        for i in range(len(qp)):
            d = fresh(qp[i, :6]); O.arr(d.qpos)[6:13] = qp[i, 6:13]; L.so100o_kinematics(C.byref(M), C.byref(d))
            xp = O.arr(d.xpos); xm = O.arr(d.xmat)
            for k in range(O.NCPROX):
                b = M.cprox_body[k]
                kk, _, _, dist = capsule_box(xp[b], xp[b + 1], M.cprox_radius[k], xp[8], xm[8].reshape(3, 3), np.full(3, 0.01))
                if kk: out[i] = min(out[i], dist)
        return out

    def run(flags, steps=8):
        sim = So100Sim(1, n, flags=flags, seed=4, max_episode_steps=0)
        sim.reset()
        sim.set_state(torch.from_numpy(np.ascontiguousarray(QP.T, np.float32)).cuda(), torch.zeros(12, n, device="cuda"))
        a = torch.from_numpy(act).cuda()
        for _ in range(steps):
            sim.step(a)
        q, v = sim.get_state()
        return q.cpu().numpy().T.astype(np.float64), v.cpu().numpy().T.astype(np.float64), sim.get_field("solver_residual").cpu().numpy()
    assert overlap(QP).max() < -1e-4                         # every start state overlaps
    q, v, res = run(F_REFERENCE_PROXIES)
    ov = overlap(q)
    print(f"[link/cube proxies, GPU] deepest overlap left after 8 env steps {ov.min()*1e3:.3f} mm (start {overlap(QP).min()*1e3:.2f} mm); worst residual {res.max():.1e}")
    assert np.isfinite(q).all() and np.isfinite(v).all() and ov.min() > -5e-4 and res.max() < 1e-2
    q0, v0, _ = run(F_REFERENCE_LINKS)
    fell = QP[:, 8] - q0[:, 8]
    free = 0.5*9.81*(8*16*0.002)**2                           # free fall over 8 env steps (cubes that reach the table stop there)
    assert np.median(np.minimum(fell, free)) > 0.6*min(free, np.median(QP[:, 8]) - 0.01)
    worst = 0.0
    for i in range(0, n, 4):                                  # the oracle's env steps from the same states
        d = fresh(); O.arr(d.qpos)[:] = QP[i].astype(np.float32)
        for t in range(8):
            O.arr(d.ctrl)[:] = O.arr(d.qpos)[:6].astype(np.float32).astype(np.float64)
            L.so100o_step(C.byref(M), C.byref(d), F_REFERENCE_PROXIES, -1, 16)
        worst = max(worst, np.abs(q[i, :9] - O.arr(d.qpos)[:9]).max())
    print(f"[link/cube proxies, GPU] 16 envs x 8 env steps vs the oracle: worst |dq| {worst:.2e}")
    assert worst < 5e-3                                       # (through the push-out: a make / break a substep apart, see tests/test_gpu_contacts.py)
    # (b) soak with every proxy pair on
    for kind in (1, 3):
        sim = So100Sim(kind, 1024, flags=F_REFERENCE_PROXIES, seed=11)
        sim.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        dropped = 0
        for t in range(128):
            sim.step((torch.rand(1024, 6, device="cuda", generator=g)*2 - 1).contiguous())
            if t % 16 == 15:
                dropped = max(dropped, int(sim.contacts_dropped().max().item()))
        q, v = sim.get_state()
        assert torch.isfinite(q).all() and torch.isfinite(v).all() and torch.isfinite(sim.obs).all()
        assert dropped == 0, (kind, dropped)
