"""GPU parity: the HIP path (through the C ABI) vs the fp64 CPU oracle on identical seeded inputs, vs the golden
trajectories recorded from the reference's Python, and size-independent properties at BASELINE.json sizes.

Tolerances (fp32 device arithmetic vs fp64 oracle), stated per check:
  * joint angles / cube position after up to 60 env steps (960 substeps): 2e-5 abs (north star: 1e-5 rel of
    angles of magnitude ~1-3 rad); velocities 5e-4 abs
  * observations 2e-5 abs, rewards 1e-4 abs
  * Env05 pixel centre: exact, except when the fp64 sub-pixel coordinate is within 2e-3 px of an integer
    boundary (then +-1 px = 9.3e-4 / 5.2e-4 in normalised units) -- expressed as a 6e-3 tolerance on 5*cx.
"""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import so100_oracle as O                      # noqa: E402  (the checker)


def _sim(*a, **k):
    from so100_mujoco_rl_amd.lib import So100Sim
    return So100Sim(*a, **k)


NOPADS = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR         # friction loss + limits + cube/floor, no finger-pad contacts (round 1's "reference")
REFP = O.F_REFERENCE                                       # + the 8 finger pads vs the floor: what So100Sim / So100VecEnv / main.py run by default
ARM = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_CUBE_PINNED
FREE = O.F_CUBE_PINNED


def _run_pair(kind, flags, n, steps, seed, action_scale=1.0, solver_iters=4, contact_iters=6, max_steps=0, inject=True):
    """Step n envs on the GPU and in the oracle with identical actions / uniforms; yield per-step results.
    Oracle solver: PGS on the dual to 1e-15 -- or its primal Newton when pad rows are simulated (PGS needs ~1e4 sweeps on them)."""
    rs = np.random.RandomState(seed)
    pads = (flags & (O.F_PADS_FLOOR | O.F_PADS_CUBE)) != 0
    sim = _sim(kind, n, flags=flags, solver_iters=solver_iters, contact_iters=30 if pads else contact_iters, max_episode_steps=max_steps, seed=seed)
    orc = [O.OracleEnv(kind, flags=flags, iters=-1 if pads else 0, seed=seed, env_id=i) for i in range(n)]
    for e in orc:
        e.e.max_episode_steps = max_steps
    inj = rs.random_sample((n, 16)).astype(np.float32)
    obs_g = sim.reset(inject=torch.from_numpy(inj).cuda() if inject else None).cpu().numpy().copy()
    obs_o = np.stack([e.reset(inject=inj[i] if inject else None) for i, e in enumerate(orc)])
    yield -1, sim, orc, obs_g, obs_o, None, None, None, None
    for t in range(steps):
        a = np.clip(rs.uniform(-1, 1, (n, 6)) * action_scale, -1, 1).astype(np.float32)
        inj = rs.random_sample((n, 16)).astype(np.float32)
        og, rg, dg, tg = sim.step(torch.from_numpy(a).cuda(), inject=torch.from_numpy(inj).cuda() if inject else None)
        res = [e.step(a[i], inject=inj[i] if inject else None, autoreset=True) for i, e in enumerate(orc)]
        oo = np.stack([r[0] for r in res]); ro = np.array([r[1] for r in res])
        do = np.array([r[2] or r[3] for r in res]); to = np.array([r[3] and not r[2] for r in res])
        yield t, sim, orc, og.cpu().numpy().copy(), oo, (rg.cpu().numpy().copy(), ro), (dg.cpu().numpy().copy(), do), (tg.cpu().numpy().copy(), to), res


def _state_err(sim, orc):
    qpos, qvel = sim.get_state()
    qpos = qpos.cpu().numpy().T; qvel = qvel.cpu().numpy().T
    qo = np.stack([O.arr(e.d.qpos).copy() for e in orc]); vo = np.stack([O.arr(e.d.qvel).copy() for e in orc])
    return np.abs(qpos - qo).max(), np.abs(qvel - vo).max()


@pytest.mark.parametrize("flags,name", [(FREE, "constraint-free"), (ARM, "friction+limits"), (NOPADS, "reference")])
def test_env01_vs_oracle(flags, name):
    n, steps = (48, 40) if flags == NOPADS else (96, 60)
    worst_o = worst_r = 0.0
    for t, sim, orc, og, oo, rew, done, trunc, _ in _run_pair(1, flags, n, steps, seed=7):
        worst_o = max(worst_o, np.abs(og - oo).max())
        if rew is not None:
            worst_r = max(worst_r, np.abs(rew[0] - rew[1]).max())
            assert not done[0].any() and not done[1].any()
    eq, ev = _state_err(sim, orc)
    print(f"[{name}] obs {worst_o:.2e} reward {worst_r:.2e} qpos {eq:.2e} qvel {ev:.2e}")
    assert worst_o < 2e-5 and worst_r < 1e-4
    assert eq < 2e-5 and ev < 5e-4


def test_env02_vs_oracle_with_reach_branch():
    n, steps = 64, 30
    hits = 0
    gen = _run_pair(2, ARM, n, steps, seed=11)
    for t, sim, orc, og, oo, rew, done, trunc, _ in gen:
        np.testing.assert_allclose(og, oo, rtol=0, atol=2e-5)
        if rew is not None:
            np.testing.assert_allclose(rew[0], rew[1], rtol=0, atol=1e-4)
            hits += int((rew[1] > 1.0).sum())
        if t in (3, 9, 15):
            # force the reach branch (< 3 cm): teleport every 4th cube onto the stale end effector, on both sides
            ee = torch.stack([sim.get_field(f"ee_{c}") for c in "xyz"])
            for c, k in zip("xyz", range(3)):
                cx = sim.get_field(f"cx_{c}"); cx[::4] = ee[k][::4]; sim.set_field(f"cx_{c}", cx)
            for i in range(0, n, 4):
                d = orc[i].d
                e3 = np.zeros(3); O.lib().so100o_end_effector(d.xpos[6], d.xmat[6], e3.ctypes.data_as(O.C.c_void_p))
                O.arr(d.xpos)[8] = e3
    assert hits > 0                            # the reach branch fired (also pinned by golden trajectory 5)
    eq, ev = _state_err(sim, orc)
    assert eq < 2e-5 and ev < 5e-4


def test_env06_vs_oracle_with_gripper_term():
    """Env06 (ref: env06_v1.py, env_base_06.py:149-162): the reach bonus + gripper sigmoid fire every step the cube is
    within 3 cm and the cube is NOT re-randomised; TimeLimit resets inside the run exercise the block memory."""
    n, steps = 64, 36
    hits = 0
    for t, sim, orc, og, oo, rew, done, trunc, _ in _run_pair(6, ARM, n, steps, seed=13, max_steps=14):
        np.testing.assert_allclose(og, oo, rtol=0, atol=2e-5)
        if rew is not None:
            np.testing.assert_allclose(rew[0], rew[1], rtol=0, atol=2e-3)      # d(gripper)/d(jaw) <= 114 / rad
            np.testing.assert_array_equal(done[0].astype(bool), done[1]); np.testing.assert_array_equal(trunc[0].astype(bool), trunc[1])
            hits += int((rew[1] > 5.0).sum())
        if t in (3, 4, 5, 20):
            ee = torch.stack([sim.get_field(f"ee_{c}") for c in "xyz"])
            for c, k in zip("xyz", range(3)):
                cx = sim.get_field(f"cx_{c}"); cx[::4] = ee[k][::4]; sim.set_field(f"cx_{c}", cx)
            for i in range(0, n, 4):
                d = orc[i].d
                e3 = np.zeros(3); O.lib().so100o_end_effector(d.xpos[6], d.xmat[6], e3.ctypes.data_as(O.C.c_void_p))
                O.arr(d.xpos)[8] = e3
    assert hits >= 4 * (n // 4)                # forced reaches + the first step after every reset (all poses zero, Q1)
    eq, ev = _state_err(sim, orc)
    assert eq < 2e-5 and ev < 5e-4


@pytest.mark.parametrize("kind,flags", [(5, NOPADS), (3, NOPADS), (4, NOPADS), (5, REFP), (3, REFP), (4, REFP)])
def test_lookat_envs_vs_oracle(kind, flags):
    """REFP = the default physics of So100VecEnv / main.py for these kinds (so100_step_mw<K, 23>): the look-at arms stay above the
    table, so the pad narrowphase runs and finds nothing -- the results must be the NOPADS ones to the same tolerance"""
    n, steps = 64, 60
    n_px = n_px_bad = 0
    for t, sim, orc, og, oo, rew, done, trunc, _ in _run_pair(kind, flags, n, steps, seed=20 + kind, action_scale=0.6):
        np.testing.assert_allclose(og[:, :6], oo[:, :6], rtol=0, atol=1e-6)
        d = np.abs(og[:, 6:] - oo[:, 6:])
        n_px += d.size; n_px_bad += int((d > 1e-4).sum())
        assert d.max() < 6e-3, (t, d.max())
        if rew is not None:
            # one pixel (9.3e-4 of the frame) moves exp(-10 d) by up to ~1e-2 in Env04's reward
            np.testing.assert_allclose(rew[0], rew[1], rtol=0, atol=1.2e-2 if kind == 4 else 2e-3)
            np.testing.assert_array_equal(done[0].astype(bool), done[1])
    assert n_px_bad <= 0.01 * n_px               # off-by-one pixels are rare
    eq, ev = _state_err(sim, orc)
    assert eq < 3e-5 and ev < 5e-4


def test_env05_termination_and_autoreset():
    """Rotate the base away until the cube is lost for > 30 steps: terminated, terminal obs, auto-reset."""
    n = 64
    sim = _sim(5, n, flags=NOPADS, contact_iters=6, max_episode_steps=0, seed=3)
    orc = [O.OracleEnv(5, flags=NOPADS, iters=0, seed=3, env_id=i) for i in range(n)]
    for e in orc:
        e.e.max_episode_steps = 0
    rs = np.random.RandomState(5)
    sim.reset(inject=torch.zeros(n, 16).cuda())
    [e.reset(inject=np.zeros(16, np.float32)) for e in orc]
    n_done = 0
    for t in range(60):
        a = np.zeros((n, 6), np.float32); a[:, 0] = 1.0 if t < 28 else 0.0
        a[n // 2:, 0] = 0.0                                   # half of the envs keep looking at the cube
        inj = rs.random_sample((n, 16)).astype(np.float32)
        og, rg, dg, tg = sim.step(torch.from_numpy(a).cuda(), inject=torch.from_numpy(inj).cuda())
        res = [e.step(a[i], inject=inj[i], autoreset=True) for i, e in enumerate(orc)]
        do = np.array([r[2] for r in res])
        np.testing.assert_array_equal(dg.cpu().numpy().astype(bool), do)
        assert not tg.any()
        if do.any():
            n_done += int(do.sum())
            tob = sim.terminal_obs.cpu().numpy()
            for i in np.nonzero(do)[0]:
                np.testing.assert_allclose(tob[i], res[i][4], rtol=0, atol=6e-3)
                np.testing.assert_allclose(og.cpu().numpy()[i], res[i][0], rtol=0, atol=1e-6)   # reset obs
                assert sim.ep_length.cpu().numpy()[i] == t + 1
    assert n_done == n // 2


def test_timelimit_truncation_and_episode_stats():
    n = 128
    sim = _sim(1, n, flags=FREE, max_episode_steps=7, seed=1)
    sim.reset()
    el = torch.arange(n, dtype=torch.int32).cuda() % 7
    sim.set_field("elapsed_steps", el)
    ret = np.zeros(n); length = np.zeros(n, int)
    for t in range(10):
        a = torch.zeros(n, 6).cuda()
        ob, r, d, tr = sim.step(a)
        elapsed = (el.cpu().numpy() + t + 1)
        want = (elapsed % 7) == 0
        np.testing.assert_array_equal(d.cpu().numpy().astype(bool), want)
        np.testing.assert_array_equal(tr.cpu().numpy().astype(bool), want)      # truncated, never terminated
        ret += r.cpu().numpy(); length += 1
        idx = np.nonzero(want)[0]
        np.testing.assert_allclose(sim.ep_return.cpu().numpy()[idx], ret[idx], rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(sim.ep_length.cpu().numpy()[idx], length[idx])
        ret[idx] = 0; length[idx] = 0
        # reset observation (all poses zero) replaces the terminal one
        assert np.all(ob.cpu().numpy()[idx, 6:] == 0)
        assert np.all(sim.terminal_obs.cpu().numpy()[idx, 12:] != 0)


def test_device_rng_matches_oracle_philox():
    """No injection: the device Philox stream equals the oracle's, so resets agree (up to fp32 sin/cos)."""
    n = 256
    for kind in (1, 2):
        sim = _sim(kind, n, flags=FREE, seed=0xDEADBEEF12345, env_id_offset=1000)
        og = sim.reset().cpu().numpy()
        for i in range(0, n, 17):
            e = O.OracleEnv(kind, flags=FREE, seed=0xDEADBEEF12345, env_id=1000 + i)
            np.testing.assert_allclose(og[i], e.reset(), rtol=0, atol=1e-6)
        qpos, _ = sim.get_state()
        e = O.OracleEnv(kind, flags=FREE, seed=0xDEADBEEF12345, env_id=1000 + 5); e.reset()
        np.testing.assert_allclose(qpos[:, 5].cpu().numpy(), O.arr(e.d.qpos), rtol=0, atol=1e-6)


@pytest.mark.parametrize("idx", range(15))
def test_golden_trajectories_on_gpu(golden_dir, idx):
    """The trajectories recorded from the reference's own Python (over oracle physics) replayed on the HIP path."""
    tr = json.load(open(os.path.join(golden_dir, "trajectories.json")))[idx]
    n = 64                                              # 64 identical lanes; lane 0 and lane 63 are checked
    sim = _sim(tr["kind"], n, flags=tr["flags"], solver_iters=4, contact_iters=6, max_episode_steps=0)
    rep = lambda v: torch.tensor(np.tile(np.array(v, np.float32), (n, 1))).cuda()
    ob = sim.reset(inject=rep(tr["reset_inject"])).cpu().numpy()
    np.testing.assert_allclose(ob[0], np.array(tr["reset_obs"], np.float32), rtol=0, atol=1e-6)
    lookat = tr["kind"] in (3, 4, 5)
    rtol_r = 2e-3 if (lookat or tr["kind"] == 6) else 1e-4            # Env06: d(gripper term)/d(jaw angle) <= 114 / rad
    for k, s in enumerate(tr["steps"]):
        if s.get("pre_teleport"):
            for c, j in zip("xyz", range(3)):
                sim.set_field(f"cube_{c}", torch.full((n,), s["pre_teleport"]["cube_qpos"][j]).cuda())
                sim.set_field(f"cx_{c}", torch.full((n,), s["pre_teleport"]["cube_xpos"][j]).cuda())
        ob, r, d, trc = sim.step(rep(s["action"]), inject=rep(s["inject"]))
        ob = ob.cpu().numpy(); want = np.array(s["obs"], np.float32)
        got = sim.terminal_obs.cpu().numpy() if s["terminated"] else ob
        for lane in (0, n - 1):
            if lookat:
                np.testing.assert_allclose(got[lane][:6], want[:6], rtol=0, atol=1e-6, err_msg=f"step {k}")
                np.testing.assert_allclose(got[lane][6:], want[6:], rtol=0, atol=6e-3, err_msg=f"step {k}")
            else:
                np.testing.assert_allclose(got[lane], want, rtol=0, atol=2e-5, err_msg=f"step {k}")
            assert abs(r.cpu().numpy()[lane] - s["reward"]) < rtol_r, f"step {k}"
            assert bool(d.cpu().numpy()[lane]) == s["terminated"]
        if s["terminated"]:
            np.testing.assert_allclose(ob[0], np.array(s["reset_obs"], np.float32), rtol=0, atol=1e-6)
        elif s["reset_after"]:
            ob2 = sim.reset(inject=rep(s["inject"])).cpu().numpy()
            np.testing.assert_allclose(ob2[0], np.array(s["reset_obs"], np.float32), rtol=0, atol=1e-6)
    qpos, qvel = sim.get_state()
    if not tr["steps"][-1]["reset_after"]:
        np.testing.assert_allclose(qpos[:, 0].cpu().numpy(), tr["steps"][-1]["qpos"], rtol=0, atol=3e-5)


@pytest.mark.parametrize("kind", [1, 5])
def test_non_finite_action_ends_only_that_episode(kind):
    """NaN / inf actions: that env's episode ends (done, reward 0, zero terminal obs, latched flag) and it is reset; every other
    env -- and the poisoned envs after their reset -- keep matching the oracle, which carries the same guard."""
    n, bad = 96, {5: float("nan"), 9: float("inf"), 70: float("-inf")}
    hit = False
    for t, sim, orc, og, oo, rew, done, trunc, res in _run_pair_with(kind, NOPADS, n, 12, seed=3, poison=(4, bad)):
        assert np.isfinite(og).all() and np.isfinite(oo).all()
        np.testing.assert_allclose(og, oo, rtol=0, atol=2e-5 if kind == 1 else 6e-3)
        if rew is None:
            continue
        assert np.isfinite(rew[0]).all()
        np.testing.assert_array_equal(done[0].astype(bool), done[1])
        if t == 4:
            hit = True
            for i in bad:
                assert done[0][i] == 1 and trunc[0][i] == 0 and rew[0][i] == 0.0
                np.testing.assert_array_equal(sim.terminal_obs[i].cpu().numpy(), 0.0)
            assert done[0].sum() == len(bad)
    assert hit
    m = sim.bad_state_mask().cpu().numpy()
    assert sorted(np.nonzero(m)[0].tolist()) == sorted(bad) and all(orc[i].e.bad_state == 1 for i in bad)
    qpos, qvel = sim.get_state()
    assert torch.isfinite(qpos).all() and torch.isfinite(qvel).all()


def _run_pair_with(kind, flags, n, steps, seed, poison):
    """_run_pair with the actions of step poison[0] overwritten per env by poison[1] (same values on both sides)."""
    rs = np.random.RandomState(seed)
    sim = _sim(kind, n, flags=flags, solver_iters=4, contact_iters=6, max_episode_steps=0, seed=seed)
    orc = [O.OracleEnv(kind, flags=flags, iters=0, seed=seed, env_id=i) for i in range(n)]
    for e in orc:
        e.e.max_episode_steps = 0
    inj = rs.random_sample((n, 16)).astype(np.float32)
    obs_g = sim.reset(inject=torch.from_numpy(inj).cuda()).cpu().numpy().copy()
    obs_o = np.stack([e.reset(inject=inj[i]) for i, e in enumerate(orc)])
    yield -1, sim, orc, obs_g, obs_o, None, None, None, None
    for t in range(steps):
        a = np.clip(rs.uniform(-1, 1, (n, 6)) * 0.5, -1, 1).astype(np.float32)
        if t == poison[0]:
            for i, v in poison[1].items():
                a[i, i % 6] = v
        inj = rs.random_sample((n, 16)).astype(np.float32)
        og, rg, dg, tg = sim.step(torch.from_numpy(a).cuda(), inject=torch.from_numpy(inj).cuda())
        res = [e.step(a[i], inject=inj[i], autoreset=True) for i, e in enumerate(orc)]
        oo = np.stack([r[0] for r in res]); ro = np.array([r[1] for r in res])
        do = np.array([r[2] or r[3] for r in res]); to = np.array([r[3] and not r[2] for r in res])
        yield t, sim, orc, og.cpu().numpy().copy(), oo, (rg.cpu().numpy().copy(), ro), (dg.cpu().numpy().copy(), do), (tg.cpu().numpy().copy(), to), res


def test_state_roundtrip_and_errors():
    from so100_mujoco_rl_amd import lib
    sim = _sim(1, 100, flags=FREE)
    sim.reset()
    qpos, qvel = sim.get_state()
    qpos2 = qpos + 0.01; qvel2 = qvel - 0.5
    sim.set_state(qpos2.contiguous(), qvel2.contiguous())
    a, b = sim.get_state()
    assert torch.equal(a, qpos2) and torch.equal(b, qvel2)
    with pytest.raises(lib.So100Error):
        sim.step(torch.zeros(99, 6).cuda())               # wrong batch size
    with pytest.raises(lib.So100Error):
        lib.So100Sim(9, 10)                               # bad env kind
    with pytest.raises(lib.So100Error):
        lib.So100Sim(1, 10, flags=lib.F_FLOOR | lib.F_CUBE_PINNED)


# ---- size-independent properties at BASELINE.json's full sizes ---------------------------------------------------
@pytest.mark.parametrize("kind,n,flags", [(1, 4096, FREE), (1, 4096, NOPADS), (2, 16384, NOPADS), (5, 8192, NOPADS),
                                          (1, 4096, REFP), (2, 16384, REFP), (5, 8192, REFP), (3, 4096, REFP), (4, 4096, REFP), (6, 4096, REFP)])
def test_full_size_properties(kind, n, flags):
    """determinism, shard invariance (env_id_offset), joint limits, finiteness, obs-space bounds"""
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    acts = [torch.rand(n, 6, device="cuda", generator=g) * 2 - 1 for _ in range(12)]

    # shards of another size than the full batch must be told the full batch's envs-per-workgroup to agree with it bit for bit (the pad-contact
    # solve sums over the 64 / epw lanes of an env in that order); the library's own choice for `n`, on the 256-CU MI355X:
    epw = 64
    if flags & (O.F_FLOOR | O.F_PADS_FLOOR | O.F_PADS_CUBE):
        while epw > 16 and (n + epw//2 - 1)//(epw//2) <= 256:
            epw //= 2

    def rollout(n_envs, offset, acts_slice):
        sim = _sim(kind, n_envs, flags=flags, seed=99, env_id_offset=offset, envs_per_workgroup=epw)
        outs = [sim.reset().clone()]
        for a in acts:
            ob, r, d, tr = sim.step(a[acts_slice].contiguous())
            outs.append(torch.cat([ob, r[:, None], d[:, None].float()], 1).clone())
        q, v = sim.get_state()
        return torch.cat(outs, 1), q, v
    full, q, v = rollout(n, 0, slice(0, n))
    again, _, _ = rollout(n, 0, slice(0, n))
    assert torch.equal(full, again)                                         # bitwise deterministic
    half = n // 2
    lo, _, _ = rollout(half, 0, slice(0, half)); hi, _, _ = rollout(half, half, slice(half, n))
    assert torch.equal(torch.cat([lo, hi], 0), full)                        # sharding leaves every env unchanged
    assert torch.isfinite(full).all() and torch.isfinite(q).all() and torch.isfinite(v).all()
    from so100_mujoco_rl_amd.constants import JOINT_RANGES
    for i, (a, b) in enumerate(JOINT_RANGES):
        if flags & O.F_LIMITS:
            assert q[i].min() > a - 0.05 and q[i].max() < b + 0.05          # soft limits hold
    nq = torch.linalg.vector_norm(q[9:13], dim=0)
    assert (nq - 1).abs().max() < 1e-5                                      # cube quaternion stays normalised


# ---- rollout-side fused policy kernel vs a plain PyTorch fp32 reference of the same op -----------------------------
def _torch_policy(t, obs, noise):
    h = torch.tanh(obs @ t["pi_w0"].T + t["pi_b0"]); h = torch.tanh(h @ t["pi_w1"].T + t["pi_b1"])
    mean = h @ t["mu_w"].T + t["mu_b"]
    g = torch.tanh(obs @ t["vf_w0"].T + t["vf_b0"]); g = torch.tanh(g @ t["vf_w1"].T + t["vf_b1"])
    value = (g @ t["v_w"].T + t["v_b"]).squeeze(1)
    act = mean + t["log_std"].exp() * noise
    logp = (-0.5 * noise ** 2 - t["log_std"] - 0.9189385332046727).sum(1)
    return act, value, logp


@pytest.mark.parametrize("kind", [1, 5])
def test_policy_kernel_vs_torch(kind):
    from so100_mujoco_rl_amd import lib
    n = 1000                                                   # not a multiple of 64: tail lanes are masked
    sim = _sim(kind, n, flags=FREE)
    od = sim.obs_dim
    g = torch.Generator(device="cuda"); g.manual_seed(kind)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    t = {"pi_w0": rnd(64, od) * 0.3, "pi_b0": rnd(64) * 0.1, "pi_w1": rnd(64, 64) * 0.2, "pi_b1": rnd(64) * 0.1,
         "mu_w": rnd(6, 64) * 0.2, "mu_b": rnd(6) * 0.1, "log_std": rnd(6) * 0.3,
         "vf_w0": rnd(64, od) * 0.3, "vf_b0": rnd(64) * 0.1, "vf_w1": rnd(64, 64) * 0.2, "vf_b1": rnd(64) * 0.1,
         "v_w": rnd(1, 64) * 0.2, "v_b": rnd(1)}
    t = {k: v.contiguous() for k, v in t.items()}
    sim.set_policy(t)
    obs = rnd(n, od).contiguous(); noise = rnd(n, 6).contiguous()
    act_env = torch.zeros(n, 6, device="cuda"); act_raw = torch.zeros_like(act_env)
    value = torch.zeros(n, device="cuda"); logp = torch.zeros(n, device="cuda"); row = torch.zeros(n, od + 10, device="cuda")
    sim.policy_forward(obs, act_env, 0, noise=noise, act_raw=act_raw, value=value, logp=logp, rollout_row=row)
    a, v, lp = _torch_policy(t, obs, noise)
    # tolerance: fp32 accumulation order + exp2-based tanh (abs err < 2e-7 per activation)
    assert (act_raw - a).abs().max() < 2e-5 and (value - v).abs().max() < 2e-5 and (logp - lp).abs().max() < 2e-5
    assert torch.equal(act_env, act_raw.clamp(-1, 1))
    assert torch.equal(row[:, :od], obs) and torch.equal(row[:, od:od + 6], act_raw)
    assert torch.equal(row[:, od + 8], value) and torch.equal(row[:, od + 9], logp)
    # device RNG: deterministic, depends on the step counter, standard normal
    sim2 = _sim(kind, 65536, flags=FREE, seed=5); sim2.set_policy(t)
    o2 = torch.zeros(65536, od, device="cuda"); a1 = torch.zeros(65536, 6, device="cuda"); a2 = torch.zeros_like(a1); a3 = torch.zeros_like(a1)
    e1 = torch.zeros_like(a1)
    sim2.policy_forward(o2, e1, 7, act_raw=a1); sim2.policy_forward(o2, e1, 7, act_raw=a2); sim2.policy_forward(o2, e1, 8, act_raw=a3)
    assert torch.equal(a1, a2) and not torch.equal(a1, a3)
    mean0, _, _ = _torch_policy(t, o2[:1], torch.zeros(1, 6, device="cuda"))
    eps = (a1 - mean0) / t["log_std"].exp()
    assert eps.mean().abs() < 0.01 and (eps.std() - 1).abs() < 0.01 and (eps ** 4).mean().sub(3).abs() < 0.1
    # the env step fills reward / done into the same rollout row
    ob, r, d, tr = sim.step(act_env, rollout_row=row)
    assert torch.equal(row[:, od + 6], r) and torch.equal(row[:, od + 7], d.float())


def test_north_star_1000_steps():
    """BASELINE.json north star: qpos/qvel within 1e-5 (relative to the joint-angle / velocity scale) of the fp64 CPU
    physics over 1000 env steps = 16000 substeps on identical seeds.  Smooth bounded actions (a random walk), the
    reference-faithful arm (friction loss + limits) and the constraint-free configuration."""
    n, steps = 16, 1000
    for flags in (ARM, FREE, NOPADS):
        rs = np.random.RandomState(3)
        sim = _sim(1, n, flags=flags, solver_iters=2, max_episode_steps=0, seed=2)     # 2 sweeps = the product default
        orc = [O.OracleEnv(1, flags=flags, iters=0, seed=2, env_id=i) for i in range(n)]
        for e in orc:
            e.e.max_episode_steps = 0
        inj = rs.random_sample((n, 16)).astype(np.float32)
        sim.reset(inject=torch.from_numpy(inj).cuda()); [e.reset(inject=inj[i]) for i, e in enumerate(orc)]
        a = np.zeros((n, 6), np.float32); worst_q = worst_v = worst_c = 0.0
        for t in range(steps):
            a = np.clip(a + rs.uniform(-0.2, 0.2, (n, 6)), -1, 1).astype(np.float32)
            sim.step(torch.from_numpy(a).cuda())
            for i, e in enumerate(orc):
                e.step(a[i])
            if t % 50 == 49 or t == steps - 1:
                qpos, qvel = sim.get_state()
                qo = np.stack([O.arr(e.d.qpos)[:6].copy() for e in orc]); vo = np.stack([O.arr(e.d.qvel)[:6].copy() for e in orc])
                worst_q = max(worst_q, np.abs(qpos[:6].cpu().numpy().T - qo).max())
                worst_v = max(worst_v, np.abs(qvel[:6].cpu().numpy().T - vo).max())
                if flags == NOPADS:                             # the cube: settles out of the floor after the reset, then rests
                    co = np.stack([O.arr(e.d.qpos)[6:13].copy() for e in orc])
                    worst_c = max(worst_c, np.abs(qpos[6:13].cpu().numpy().T - co).max())
        print(f"flags={flags}: 1000 steps, max |dq| = {worst_q:.2e} rad (scale pi), max |dqvel| = {worst_v:.2e} rad/s, cube pose {worst_c:.2e}")
        # north-star bound: 1e-5 relative to the angle scale pi / the velocity scale 40 rad/s.  The asserted bounds are 3x
        # the drift MEASURED on MI355X (round 1: 1.8e-6 rad / 5.9e-6 rad/s with arm rows and with the full reference
        # physics, 3.7e-6 / 8.6e-6 constraint-free, 4.7e-8 on the cube pose), so a 10x regression of the solver or of the
        # integration fails here long before it reaches the north-star bound; solver_iters = 2 is the shipped default, so
        # this is also the check that 2 block-PGS sweeps never leave 1e-5.
        bq, bv = (1.2e-5, 3e-5) if flags == FREE else (6e-6, 3e-5)
        assert worst_c < 2e-7                               # metres / quaternion components
        assert worst_q < bq and worst_q < 1e-5 * np.pi
        assert worst_v < bv and worst_v < 1e-5 * 40


def test_rollout_collector_and_vecenv():
    """The packaged fast path (collector) and the SB3 VecEnv adapter agree with stepping the handle by hand."""
    from so100_mujoco_rl_amd.vec_env import So100VecEnv
    from so100_mujoco_rl_amd.collector import RolloutCollector
    n, T = 256, 8
    env = So100VecEnv("Env01-v1", n, flags=ARM, seed=9, max_episode_steps=5)
    sd = RolloutCollector.random_policy_state(15, env.device, seed=1)
    col = RolloutCollector(env, sd, T=T, bootstrap_truncated=False)      # raw env rewards: compared with the VecEnv below
    b = col.collect()
    assert b["obs"].shape == (T, n, 15) and b["actions"].shape == (T, n, 6) and b["rewards"].shape == (T, n)
    # the same rollout with SB3's TimeLimit bootstrap (the default): rewards differ exactly by gamma * V(terminal_observation)
    # on the truncated steps (all 256 episode ends here are truncations), nowhere else
    env_b = So100VecEnv("Env01-v1", n, flags=ARM, seed=9, max_episode_steps=5)
    col_b = RolloutCollector(env_b, sd, T=T, gamma=0.97)
    bb = col_b.collect()
    assert torch.equal(bb["dones"], b["dones"]) and torch.equal(bb["truncated"], b["dones"] > 0) and int(bb["truncated"].sum()) == n
    tv = col_b._value(col_b.tobs[:T].reshape(-1, 15)).reshape(T, n)
    torch.testing.assert_close(bb["rewards"], b["rewards"] + torch.where(bb["truncated"], 0.97 * tv, torch.zeros_like(tv)), rtol=0, atol=1e-6)
    assert (bb["rewards"] - b["rewards"])[~bb["truncated"]].abs().max() == 0
    # replay by hand: same seeds, same policy noise stream (step counter), same actions
    env2 = So100VecEnv("Env01-v1", n, flags=ARM, seed=9, max_episode_steps=5)
    ob = env2.reset()
    np.testing.assert_array_equal(ob, b["obs"][0].cpu().numpy())
    n_done = 0
    for t in range(T):
        a = b["actions"][t].cpu().numpy()
        ob, r, d, infos = env2.step(np.clip(a, -1, 1))
        # the persistent kernel and so100_step_fused are separate compilations of the same code: results agree to an ulp
        np.testing.assert_allclose(r, b["rewards"][t].cpu().numpy(), rtol=0, atol=2e-5)       # the reward multiplies ulp-level angle differences by 10-20
        np.testing.assert_array_equal(d.astype(np.float32), b["dones"][t].cpu().numpy())
        if t + 1 < T:
            np.testing.assert_allclose(ob, b["obs"][t + 1].cpu().numpy(), rtol=0, atol=1e-6)
        for i in np.nonzero(d)[0]:
            assert infos[i]["TimeLimit.truncated"] is True and infos[i]["terminal_observation"].shape == (15,)
            assert infos[i]["episode"]["l"] == 5
            n_done += 1
    assert n_done == n                                       # every env hit its 5-step TimeLimit exactly once
    assert len(infos) == n and env2.observation_space.shape == (15,) and env2.action_space.shape == (6,)
    assert env2.env_is_wrapped(object) == [False] * n and env2.get_attr("num_envs", [0, 1]) == [n, n]


@pytest.mark.parametrize("kind,flags", [(1, FREE), (1, NOPADS), (2, NOPADS), (5, NOPADS), (1, ARM), (6, NOPADS)])
def test_persistent_rollout_equals_stepwise(kind, flags):
    """so100_rollout (one launch for T steps) == T x (so100_policy_forward + so100_step), buffer row by row."""
    from so100_mujoco_rl_amd.vec_env import So100VecEnv
    from so100_mujoco_rl_amd.collector import RolloutCollector
    n, T = 200, 12                                           # tail workgroup partially filled; TimeLimit hits inside the chunk
    outs = []
    for persistent in (True, False):
        env = So100VecEnv(kind, n, flags=flags, seed=4, max_episode_steps=7)
        sd = RolloutCollector.random_policy_state(env.sim.obs_dim, env.device, seed=2)
        sd["log_std"] = sd["log_std"] - 0.5
        col = RolloutCollector(env, sd, T=T, persistent=persistent)
        b1 = {k: v.clone() for k, v in col.collect().items()}
        b2 = {k: v.clone() for k, v in col.collect(5).items()}          # a second, shorter chunk continues the episode
        q, v = env.sim.get_state()
        # terminal observations: the collector's per-chunk buffer (entries are only written where an episode ended)
        tob = torch.where((b2["dones"] > 0)[..., None], col.tobs[:5], torch.zeros_like(col.tobs[:5]))
        outs.append((b1, b2, q, v, tob, env.sim.ep_length.clone()))
    (a1, a2, aq, av, at, al), (s1, s2, sq, sv, st_, sl) = outs
    for a, s in ((a1, s1), (a2, s2)):
        for k in ("obs", "actions", "rewards", "dones", "values", "log_probs", "last_obs"):
            # separate compilations of the same code agree to an ulp or two of the angles (1e-6); the reward multiplies
            # angle overshoots by 10 and end-effector heights by 20 (env_base_01.py:153-163, 207-211)
            assert torch.allclose(a[k], s[k], rtol=0, atol=2e-5 if k == "rewards" else 1e-6), k
    assert torch.allclose(aq, sq, atol=1e-6) and torch.allclose(av, sv, atol=1e-5)
    assert a1["dones"].sum() > 0 and torch.equal(al, sl) and torch.allclose(at, st_, atol=1e-6)


def test_c_abi_from_plain_cpp(tmp_path):
    """examples/abi_demo.cpp drives libso100sim.so from C++ with raw hipMalloc'ed buffers (no Python, no torch);
    the same run through the Python binding gives the same checksums."""
    import subprocess, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_demo")
    libdir = os.path.join(root, "so100_mujoco_rl_amd")
    import so100_mujoco_rl_amd.lib as lib
    hipdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-o", exe, os.path.join(root, "examples", "abi_demo.cpp"), "-I" + os.path.join(root, "include"),
                           "-L" + libdir, "-lso100sim", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hipdir])
    n, steps = 300, 20
    out = subprocess.check_output([exe, str(n), str(steps)], text=True)
    m = re.search(r"obs_checksum (\S+)\s+reward_sum (\S+)", out)
    assert m, out
    sim = _sim(1, n, flags=lib.F_REFERENCE, solver_iters=3, contact_iters=4, seed=42)     # SO100_F_REFERENCE in abi_demo.cpp
    sim.reset()
    idx = np.arange(6 * n, dtype=np.uint64)
    a = ((idx * np.uint64(2654435761)) % np.uint64(2001)).astype(np.float32) / np.float32(1000.0) - np.float32(1.0)   # as abi_demo.cpp (64-bit product)
    act = torch.from_numpy(a.reshape(n, 6)).cuda()
    for _ in range(steps):
        ob, r, d, tr = sim.step(act)
    assert abs(float(m.group(1)) - ob.double().sum().item()) < 1e-3
    assert abs(float(m.group(2)) - r.double().sum().item()) < 1e-3


def test_single_env_gym_view():
    """So100Env: the N = 1 Gymnasium-style view (reset(seed) -> (obs, info); step -> 5-tuple; TimeLimit -> truncated)."""
    from so100_mujoco_rl_amd.envs import So100Env
    env = So100Env(1, flags=ARM, seed=3)
    env.sim.cfg  # handle exists
    ob, info = env.reset(seed=11)
    assert ob.shape == (15,) and ob.dtype == np.float32 and info == {} and np.all(ob[6:] == 0)
    ob2, _ = So100Env(1, flags=ARM, seed=11).reset()
    np.testing.assert_array_equal(ob, ob2)                     # seeding is effective (unlike the reference)
    total = 0.0
    for t in range(5):
        ob, r, term, trunc, info = env.step(env.action_space.sample() if hasattr(env.action_space, "sample") else np.zeros(6))
        assert ob.shape == (15,) and isinstance(r, float) and term is False and trunc is False
        total += r
    assert np.isfinite(total) and env.observation_space.shape == (15,)
    env.close()
    # TimeLimit: the 4th step of a 4-step episode is truncated (not terminated) and reports the terminal observation
    env = So100Env(1, flags=ARM, seed=5, max_episode_steps=4)
    env.reset()
    import time
    for t in range(4):
        ob, r, term, trunc, info = env.step(np.full(6, 0.3, np.float32))
        assert term is False and trunc is (t == 3)
    assert np.any(ob[6:] != 0)                                  # the terminal observation, not the all-zero-poses reset observation
    rc0 = int(env.sim.get_field("rng_counter", dtype=torch.int32)[0])
    ob0, _ = env.reset()
    assert np.all(ob0[6:] == 0)
    # the fused step had already reset the env when the episode ended: reset() hands out that episode's first observation and
    # does NOT reset a second time (no extra RNG draw, elapsed_steps still 0) ...
    assert int(env.sim.get_field("rng_counter", dtype=torch.int32)[0]) == rc0 and int(env.sim.get_field("elapsed_steps", dtype=torch.int32)[0]) == 0
    np.testing.assert_array_equal(ob0[:6], env.sim.get_state()[0][:6, 0].cpu().numpy())
    # ... and stepping on without reset() (the reference's viewer loop, main.py:118-124) continues in that next episode
    for t in range(4):
        ob, r, term, trunc, info = env.step(np.full(6, 0.3, np.float32))
    assert trunc is True and int(env.sim.get_field("elapsed_steps", dtype=torch.int32)[0]) == 0
    ob, r, term, trunc, info = env.step(np.zeros(6, np.float32))
    assert trunc is False and int(env.sim.get_field("elapsed_steps", dtype=torch.int32)[0]) == 1
    t0 = time.perf_counter()
    for _ in range(300):
        env.step(np.zeros(6, np.float32))
    print(f"So100Env single-env step: {(time.perf_counter() - t0) / 300 * 1e6:.0f} us")
    env.close()


@pytest.mark.parametrize("kind,flags,n", [(1, FREE, 4096), (2, NOPADS, 4096), (5, NOPADS, 2048), (6, NOPADS, 2048), (1, REFP, 4096), (5, REFP, 2048), (2, O.F_CONTACT5, 2048)])
def test_soak_full_batch(kind, flags, n):
    """Thousands of vectorised steps through the default (persistent) collector with short staggered episodes: every env
    resets many times; state stays finite, joint limits hold, counters are consistent, no pipeline faults."""
    from so100_mujoco_rl_amd.vec_env import So100VecEnv
    from so100_mujoco_rl_amd.collector import RolloutCollector
    from so100_mujoco_rl_amd.constants import JOINT_RANGES
    env = So100VecEnv(kind, n, flags=flags, seed=123, max_episode_steps=37, stagger_episodes=True)
    sd = RolloutCollector.random_policy_state(env.sim.obs_dim, env.device, seed=3)
    sd["log_std"] = sd["log_std"] + 0.5                     # energetic exploration: joints run into their limits
    col = RolloutCollector(env, sd, T=64)
    assert col.persistent
    dones = 0; steps = 0
    for it in range(40 if flags == FREE else 12):
        b = col.collect()
        assert torch.isfinite(b["obs"]).all() and torch.isfinite(b["rewards"]).all() and torch.isfinite(b["values"]).all()
        dones += int(b["dones"].sum().item()); steps += 64
    q, v = env.sim.get_state()
    assert torch.isfinite(q).all() and torch.isfinite(v).all()
    expected = n * steps / 37
    assert abs(dones - expected) < 0.05 * expected + n       # TimeLimit resets happen at the right rate
    el = env.sim.get_field("elapsed_steps", dtype=torch.int32)
    assert int(el.min()) >= 0 and int(el.max()) < 37
    if flags & O.F_LIMITS:
        for i, (a, bnd) in enumerate(JOINT_RANGES):
            assert q[i].min() > a - 0.1 and q[i].max() < bnd + 0.1
    if not (flags & O.F_CUBE_PINNED):
        assert q[8].min() > -0.02                             # the cube never falls through the floor
        assert (torch.linalg.vector_norm(q[9:13], dim=0) - 1).abs().max() < 1e-5


def test_stepwise_calls_are_graph_capturable():
    """so100_policy_forward + so100_step enqueue on the caller's stream without allocating or synchronising, so a
    rollout step can be captured into a hipGraph (torch.cuda.CUDAGraph) and replayed; replay == eager."""
    from so100_mujoco_rl_amd.collector import RolloutCollector
    n, T = 512, 6
    outs = []
    for use_graph in (False, True):
        sim = _sim(1, n, flags=ARM, seed=8)
        sd = RolloutCollector.random_policy_state(15, sim.device, seed=4)
        from so100_mujoco_rl_amd.lib import POLICY_TENSORS, SB3_STATE_DICT_KEYS
        sim.set_policy({k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS})
        sim.reset()
        act = torch.zeros(n, 6, device="cuda"); row = torch.zeros(n, 25, device="cuda")
        rows = []
        if use_graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                # the policy-noise counter is a launch argument: one graph per counter value would be needed to vary it,
                # so this check replays a fixed-noise step (the env's own RNG counter lives in device state and advances)
                with torch.cuda.graph(g, stream=s):
                    sim.policy_forward(sim.obs, act, 0, rollout_row=row)
                    sim.step(act, rollout_row=row)
            torch.cuda.current_stream().wait_stream(s)
            for t in range(T):
                g.replay(); rows.append(row.clone())
        else:
            for t in range(T):
                sim.policy_forward(sim.obs, act, 0, rollout_row=row)
                sim.step(act, rollout_row=row); rows.append(row.clone())
        torch.cuda.synchronize()
        outs.append(torch.stack(rows))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("kind,flags", [(1, FREE), (1, NOPADS), (5, NOPADS), (2, ARM)])
def test_multiwave_step_kernel_equals_throughput_kernel(kind, flags):
    """so100_step picks the 4-wave latency kernel (so100_step_mw) for N <= 16384 and the one-wave throughput kernel
    (so100_step_fused) above.  Envs are independent and their RNG is keyed by env id, so the first 200 envs of a 16 576-env
    handle (throughput kernel) must evolve exactly like a 200-env handle (latency kernel) under the same actions: same
    arithmetic, split over waves -- results agree to the last bit or two, including TimeLimit resets and the tail workgroup."""
    n, big, steps = 200, 16384 + 192, 40
    a_mw = _sim(kind, n, flags=flags, seed=9, max_episode_steps=15)
    a_sw = _sim(kind, big, flags=flags, seed=9, max_episode_steps=15)
    o1 = a_mw.reset().clone(); o2 = a_sw.reset()[:n].clone()
    assert torch.equal(o1, o2)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    lookat = kind >= 3 and kind <= 5
    for t in range(steps):
        act_big = torch.rand(big, 6, device="cuda", generator=g) * 2 - 1
        act = act_big[:n].contiguous()
        r1 = [x.clone() for x in a_mw.step(act)]; r2 = [x[:n].clone() for x in a_sw.step(act_big)]
        torch.testing.assert_close(r1[0][:, :6], r2[0][:, :6], rtol=0, atol=2e-6)
        torch.testing.assert_close(r1[0], r2[0], rtol=0, atol=6e-3 if lookat else 2e-6)     # look-at obs: integer pixel centres
        torch.testing.assert_close(r1[1], r2[1], rtol=0, atol=2e-3 if lookat else 1e-5)
        assert torch.equal(r1[2], r2[2]) and torch.equal(r1[3], r2[3])
    q1, v1 = a_mw.get_state(); q2, v2 = a_sw.get_state()
    torch.testing.assert_close(q1, q2[:, :n], rtol=0, atol=2e-6); torch.testing.assert_close(v1, v2[:, :n], rtol=0, atol=2e-5)


def test_dense_throughput_kernel_131072():
    """N >= 131 072 with friction / limit / cube-floor physics: so100_step takes the 2-waves-per-SIMD build of the throughput kernel
    (256 registers + scratch).  Same source, so the results must be BIT-IDENTICAL to the 1-wave-per-SIMD build that two half-size
    handles take (65 536 envs each, env_id_offset keeps the Philox streams), and 32 sampled envs are held to the fp64 oracle."""
    n, steps = 131072, 5
    rs = np.random.RandomState(12)
    sample = np.sort(rs.choice(n, 32, replace=False)); sample[0] = 0; sample[-1] = n - 1
    sim = _sim(1, n, flags=NOPADS, solver_iters=4, seed=23, max_episode_steps=3)
    orc = [O.OracleEnv(1, flags=NOPADS, iters=0, seed=23, env_id=int(i)) for i in sample]
    for e in orc:
        e.e.max_episode_steps = 3
    sim.reset(); [e.reset() for e in orc]
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    acts = [torch.rand(n, 6, device="cuda", generator=g) * 2 - 1 for _ in range(steps)]
    trace = []
    for t in range(steps):
        ob, r, d, tr = sim.step(acts[t])
        trace.append(torch.cat([ob, r[:, None], d[:, None].float()], 1).clone())
        a_h = acts[t][sample].cpu().numpy()
        res = [e.step(a_h[j], autoreset=True) for j, e in enumerate(orc)]
        np.testing.assert_allclose(ob[sample].cpu().numpy(), np.stack([x[0] for x in res]), rtol=0, atol=2e-5)
        np.testing.assert_allclose(r[sample].cpu().numpy(), np.array([x[1] for x in res]), rtol=0, atol=1e-4)
        np.testing.assert_array_equal(d[sample].cpu().numpy().astype(bool), np.array([x[2] or x[3] for x in res]))
    full = torch.cat(trace, 1); sim.close()
    halves = []
    for off in (0, n // 2):
        s2 = _sim(1, n // 2, flags=NOPADS, solver_iters=4, seed=23, max_episode_steps=3, env_id_offset=off); s2.reset(); tr2 = []
        for t in range(steps):
            ob, r, d, _ = s2.step(acts[t][off:off + n // 2].contiguous())
            tr2.append(torch.cat([ob, r[:, None], d[:, None].float()], 1).clone())
        halves.append(torch.cat(tr2, 1)); s2.close()
    assert torch.equal(torch.cat(halves, 0), full)


def test_large_batch_dispatch_65536_vs_oracle():
    """N = 65 536: so100_step takes the production throughput kernel (so100_step_fused, one wave per 64 envs, the
    __launch_bounds__(64, 2) variant for the contact-free flags) and so100_policy_forward walks 1024 tiles with a 512-block
    grid (grid-stride path).  64 sampled envs are held to the fp64 oracle (device Philox = oracle Philox, no injection), the
    policy outputs of the same envs to a plain PyTorch fp32 reference, and the whole batch to determinism and shard invariance."""
    n, steps = 65536, 6
    rs = np.random.RandomState(11)
    sample = np.sort(rs.choice(n, 64, replace=False)); sample[0] = 0; sample[-1] = n - 1
    for kind, flags in ((1, FREE), (1, NOPADS)):
        sim = _sim(kind, n, flags=flags, solver_iters=4, seed=21, max_episode_steps=4)       # TimeLimit resets inside the run
        orc = [O.OracleEnv(kind, flags=flags, iters=0, seed=21, env_id=int(i)) for i in sample]
        for e in orc:
            e.e.max_episode_steps = 4
        og = sim.reset().clone()
        oo = np.stack([e.reset() for e in orc])
        np.testing.assert_allclose(og[sample].cpu().numpy(), oo, rtol=0, atol=1e-6)
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        trace = []
        for t in range(steps):
            act = torch.rand(n, 6, device="cuda", generator=g) * 2 - 1
            ob, r, d, tr = sim.step(act)
            trace.append(torch.cat([ob, r[:, None], d[:, None].float()], 1).clone())
            a_h = act[sample].cpu().numpy()
            res = [e.step(a_h[j], autoreset=True) for j, e in enumerate(orc)]
            np.testing.assert_allclose(ob[sample].cpu().numpy(), np.stack([x[0] for x in res]), rtol=0, atol=2e-5)
            np.testing.assert_allclose(r[sample].cpu().numpy(), np.array([x[1] for x in res]), rtol=0, atol=1e-4)
            np.testing.assert_array_equal(d[sample].cpu().numpy().astype(bool), np.array([x[2] or x[3] for x in res]))
        assert sum(int(x[:, -1].sum()) for x in trace) == n                                  # every env hit its 4-step limit once
        full = torch.cat(trace, 1)
        # determinism + shard invariance of the throughput kernel (two half-size handles, both still above the 16 384 switch)
        halves = []
        for off in (0, n // 2):
            s2 = _sim(kind, n // 2, flags=flags, solver_iters=4, seed=21, max_episode_steps=4, env_id_offset=off)
            s2.reset()
            g2 = torch.Generator(device="cuda"); g2.manual_seed(5); tr2 = []
            for t in range(steps):
                act = torch.rand(n, 6, device="cuda", generator=g2) * 2 - 1
                ob, r, d, _ = s2.step(act[off:off + n // 2].contiguous())
                tr2.append(torch.cat([ob, r[:, None], d[:, None].float()], 1).clone())
            halves.append(torch.cat(tr2, 1)); s2.close()
        assert torch.equal(torch.cat(halves, 0), full)
        if flags == FREE:                                    # the policy kernel at the same batch (grid-stride over 1024 tiles)
            from so100_mujoco_rl_amd.collector import RolloutCollector
            sd = RolloutCollector.random_policy_state(15, sim.device, seed=3)
            from so100_mujoco_rl_amd.lib import POLICY_TENSORS, SB3_STATE_DICT_KEYS
            tens = {k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS}
            sim.set_policy(tens)
            noise = torch.randn(n, 6, device="cuda", generator=g)
            act_env = torch.empty(n, 6, device="cuda"); raw = torch.empty(n, 6, device="cuda")
            val = torch.empty(n, device="cuda"); lp = torch.empty(n, device="cuda")
            sim.policy_forward(sim.obs, act_env, 0, noise=noise, act_raw=raw, value=val, logp=lp)
            ra, rv, rl = _torch_policy(tens, sim.obs, noise)
            torch.testing.assert_close(raw, ra, rtol=0, atol=2e-5); torch.testing.assert_close(val, rv, rtol=0, atol=2e-5)
            torch.testing.assert_close(lp, rl, rtol=0, atol=1e-4)
            torch.testing.assert_close(act_env, ra.clamp(-1, 1), rtol=0, atol=2e-5)
        sim.close()


def test_policy_saturation_is_finite():
    """A saturated hidden unit (pre-activation far beyond +-44, where exp(2x) overflows) must give tanh = +-1, not NaN: a trained
    or loaded SB3 policy with large weights / observations would otherwise write NaN actions and values into the rollout buffer."""
    from so100_mujoco_rl_amd.collector import RolloutCollector
    from so100_mujoco_rl_amd.lib import POLICY_TENSORS, SB3_STATE_DICT_KEYS
    n = 256
    sim = _sim(1, n, flags=FREE, seed=1)
    sim.reset()
    sd = RolloutCollector.random_policy_state(15, sim.device, seed=3)
    tens = {k: sd[SB3_STATE_DICT_KEYS[k]].clone().contiguous() for k in POLICY_TENSORS}
    for k in ("pi_w0", "vf_w0", "pi_w1", "vf_w1"):
        tens[k] *= 400.0                                     # pre-activations of several hundred in both layers
    sim.set_policy(tens)
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    obs = (torch.rand(n, 15, device="cuda", generator=g) * 2 - 1) * 3
    noise = torch.randn(n, 6, device="cuda", generator=g)
    act_env = torch.empty(n, 6, device="cuda"); raw = torch.empty(n, 6, device="cuda")
    val = torch.empty(n, device="cuda"); lp = torch.empty(n, device="cuda")
    sim.policy_forward(obs, act_env, 0, noise=noise, act_raw=raw, value=val, logp=lp)
    for x in (act_env, raw, val, lp):
        assert torch.isfinite(x).all()
    ra, rv, rl = _torch_policy(tens, obs, noise)
    torch.testing.assert_close(raw, ra, rtol=0, atol=1e-4); torch.testing.assert_close(val, rv, rtol=1e-5, atol=1e-4)
    # and through the persistent rollout kernel (its own copy of the policy phase)
    buf = torch.zeros(3, n, 25, device="cuda")
    sim.obs.copy_(obs); sim.rollout(buf, 0)
    assert torch.isfinite(buf).all()


def test_vecenv_zero_copy_round_trip_equals_staged():
    """So100VecEnv's numpy round trip lets the step kernel read / write pinned host memory directly (one launch); it must return
    exactly what the staged path (H2D, kernel on device buffers, D2H) returns, infos included, across TimeLimit resets."""
    from so100_mujoco_rl_amd.vec_env import So100VecEnv
    n = 300
    eg = So100VecEnv("Env05-v1", n, flags=NOPADS, seed=4, max_episode_steps=6, use_graph=True)
    ee = So100VecEnv("Env05-v1", n, flags=NOPADS, seed=4, max_episode_steps=6, use_graph=False)
    np.testing.assert_array_equal(eg.reset(), ee.reset())
    rs = np.random.RandomState(0)
    for t in range(20):
        a = rs.uniform(-1, 1, (n, 6)).astype(np.float32)
        og, rg, dg, ig = eg.step(a); oe, re_, de, ie = ee.step(a)
        np.testing.assert_array_equal(og, oe); np.testing.assert_array_equal(rg, re_); np.testing.assert_array_equal(dg, de)
        assert og.dtype == np.float32 and rg.dtype == np.float32 and dg.dtype == bool and og.shape == (n, 8)
        for i in np.nonzero(dg)[0]:
            assert ig[i]["TimeLimit.truncated"] == ie[i]["TimeLimit.truncated"] and ig[i]["episode"]["l"] == ie[i]["episode"]["l"] == 6
            np.testing.assert_array_equal(ig[i]["terminal_observation"], ie[i]["terminal_observation"])
            assert ig[i]["episode"]["r"] == ie[i]["episode"]["r"]
        assert all(ig[i] == {} for i in np.nonzero(~dg)[0][:10])
    assert eg._zero_copy and not ee._zero_copy
    # mixing in the tensor API and a reset does not disturb the round trip
    ot, _, _, _ = eg.step_tensor(torch.zeros(n, 6, device=eg.device)); ee.step_tensor(torch.zeros(n, 6, device=ee.device))
    np.testing.assert_array_equal(eg.reset(), ee.reset())
    a = rs.uniform(-1, 1, (n, 6)).astype(np.float32)
    np.testing.assert_array_equal(eg.step(a)[0], ee.step(a)[0])
    eg.close(); ee.close()


def test_device_outputs_stay_current_after_numpy_steps(tmp_path):
    """So100VecEnv's zero-copy numpy path writes its results to pinned host memory only; the handle's device tensors (sim.obs, ...)
    that a checkpoint, the rollout collector or policy_forward read must still be the CURRENT ones afterwards."""
    from so100_mujoco_rl_amd.vec_env import So100VecEnv
    from so100_mujoco_rl_amd.collector import RolloutCollector
    n = 130
    env = So100VecEnv("Env01-v1", n, flags=ARM, seed=6, max_episode_steps=5)
    env.reset()
    rs = np.random.RandomState(0)
    for t in range(7):                                       # crosses a TimeLimit reset
        ob, rw, dn, infos = env.step(rs.uniform(-1, 1, (n, 6)).astype(np.float32))
    assert env._zero_copy
    np.testing.assert_array_equal(env.sim.obs.cpu().numpy(), ob)
    np.testing.assert_array_equal(env.sim.rew.cpu().numpy(), rw); np.testing.assert_array_equal(env.sim.done.cpu().numpy().astype(bool), dn)
    # checkpoint -> restore into a fresh handle: the stored observation is the last returned one, not the reset observation
    path = str(tmp_path / "sim.npz"); env.sim.save_state(path)
    env2 = So100VecEnv("Env01-v1", n, flags=ARM, seed=6, max_episode_steps=5); env2.reset(); env2.sim.load_state(path)
    np.testing.assert_array_equal(env2.sim.obs.cpu().numpy(), ob)
    # numpy steps, then the on-device collector: its first policy forward sees the current observation
    ob2 = env.step(rs.uniform(-1, 1, (n, 6)).astype(np.float32))[0]
    col = RolloutCollector(env, RolloutCollector.random_policy_state(15, env.device, seed=1), T=4, persistent=True); col._started = True
    b = col.collect()
    np.testing.assert_array_equal(b["obs"][0].cpu().numpy(), ob2)
    env.close(); env2.close()


@pytest.mark.parametrize("n,frame_skip", [(1, 16), (63, 5), (65, 1), (130, 7)])
def test_odd_batch_sizes_and_frame_skips_vs_oracle(n, frame_skip):
    """Tail workgroups (N not a multiple of 64, N = 1) and frame_skip values other than the reference's 16, against the oracle."""
    rs = np.random.RandomState(n)
    sim = _sim(2, n, flags=NOPADS, solver_iters=4, contact_iters=6, max_episode_steps=9, seed=5, frame_skip=frame_skip)
    orc = [O.OracleEnv(2, flags=NOPADS, iters=0, seed=5, env_id=i) for i in range(n)]
    for e in orc:
        e.e.max_episode_steps = 9; e.e.frame_skip = frame_skip
    inj = rs.random_sample((n, 16)).astype(np.float32)
    og = sim.reset(inject=torch.from_numpy(inj).cuda()).cpu().numpy()
    oo = np.stack([e.reset(inject=inj[i]) for i, e in enumerate(orc)])
    np.testing.assert_allclose(og, oo, rtol=0, atol=1e-6)
    for t in range(20):
        a = rs.uniform(-1, 1, (n, 6)).astype(np.float32); inj = rs.random_sample((n, 16)).astype(np.float32)
        og, rg, dg, tg = sim.step(torch.from_numpy(a).cuda(), inject=torch.from_numpy(inj).cuda())
        res = [e.step(a[i], inject=inj[i], autoreset=True) for i, e in enumerate(orc)]
        np.testing.assert_allclose(og.cpu().numpy(), np.stack([r[0] for r in res]), rtol=0, atol=2e-5, err_msg=f"step {t}")
        np.testing.assert_allclose(rg.cpu().numpy(), np.array([r[1] for r in res]), rtol=0, atol=1e-4)
        np.testing.assert_array_equal(dg.cpu().numpy().astype(bool), np.array([r[2] or r[3] for r in res]))
    # and the rollout kernel on the same odd sizes: finite, right shape, TimeLimit fires
    from so100_mujoco_rl_amd.collector import RolloutCollector, SB3_STATE_DICT_KEYS, POLICY_TENSORS
    sd = RolloutCollector.random_policy_state(sim.obs_dim, sim.device, seed=0)
    sim.set_policy({k: sd[SB3_STATE_DICT_KEYS[k]].contiguous() for k in POLICY_TENSORS})
    buf = torch.zeros(12, n, sim.obs_dim + 10, device="cuda")
    sim.rollout(buf, 0)
    assert torch.isfinite(buf).all() and float(buf[..., -3].sum()) >= n
