"""Model-independent identities for the oracle's physics restatement (oracle/so100_oracle.c part 1).

MuJoCo itself is not available, so the restatement is checked against
mechanics, each identity using an INDEPENDENT computation path:
  * FK vs a scipy.spatial.transform chain built straight from the raw model numbers,
  * CRB mass matrix vs RNE (two different algorithms) and vs kinetic energy from finite-differenced FK,
  * gravity bias vs the gradient of potential energy, Coriolis bias vs the Lagrange equations,
  * constraint solve vs the KKT conditions of the dual box-QP,
  * free fall / resting contact / servo steady state closed forms, energy conservation.
CPU only."""
import ctypes as C

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

from oracle import so100_oracle as O

L = O.lib()
M = O.model()
NV = 12


def fresh(q=None, v=None):
    d = O.Data()
    L.so100o_reset_data(C.byref(M), C.byref(d))
    if q is not None:
        O.arr(d.qpos)[:len(q)] = q
    if v is not None:
        O.arr(d.qvel)[:len(v)] = v
    return d


def rand_state(rs, vel=True):
    q = np.zeros(13); v = np.zeros(12)
    rng = np.array(M.jnt_range)
    q[:6] = rs.uniform(rng[:, 0], rng[:, 1])
    q[6:9] = rs.uniform(-0.3, 0.3, 3); q[8] = rs.uniform(0.05, 0.3)
    quat = R.random(random_state=rs).as_quat()          # x,y,z,w
    q[9:13] = [quat[3], quat[0], quat[1], quat[2]]
    if vel:
        v[:6] = rs.uniform(-3, 3, 6); v[6:9] = rs.uniform(-1, 1, 3); v[9:12] = rs.uniform(-5, 5, 3)
    return q, v


def quat_wxyz_to_R(q):
    return R.from_quat([q[1], q[2], q[3], q[0]])


def scipy_fk(q):
    """FK straight from so100_model_def.h's raw numbers (via the oracle model struct), with scipy."""
    pos = np.zeros(3); rot = R.identity()
    out = []
    for k in range(6):
        b = k + 2
        pos = pos + rot.apply(np.array(M.body_pos[b]))
        rot = rot * quat_wxyz_to_R(np.array(M.body_quat[b]))
        rot = rot * R.from_rotvec(np.array(M.jnt_axis[b]) * q[k])
        out.append((pos.copy(), rot))
    return out


def test_model_constants():
    assert sum(M.body_mass[2:8]) == pytest.approx(0.6089654, abs=1e-12)    # hand sum of arm:73-113 (0.608966)
    assert M.body_mass[8] == pytest.approx(0.008)
    assert M.body_inertia[8][0] == pytest.approx(0.008 * (0.02**2 + 0.02**2) / 12)
    # euler "1.57079 0 0" -> rotation about x
    np.testing.assert_allclose(np.array(M.body_quat[3]), [np.cos(1.57079 / 2), np.sin(1.57079 / 2), 0, 0], atol=1e-15)
    # camera euler (4.974, 0, 3.142) intrinsic xyz = Rx(4.974) Rz(3.142)
    want = (R.from_euler("X", 4.974) * R.from_euler("Z", 3.142)).as_matrix()
    np.testing.assert_allclose(quat_wxyz_to_R(np.array(M.cam_quat)).as_matrix(), want, atol=1e-14)
    # default solref/solimp -> K, B
    assert 1 / (0.95**2 * 0.02**2) == pytest.approx(2770.08, abs=0.01)
    assert 2 / (0.95 * 0.02) == pytest.approx(105.263, abs=1e-3)
    # kv = 2 sqrt(kp M0), armature inside M0
    np.testing.assert_allclose(np.array(M.kv), 2 * np.sqrt(50 * np.array(M.dof_M0)[:6]), rtol=1e-15)
    assert all(m0 > 0.1 for m0 in np.array(M.dof_M0)[:6])


def test_fk_matches_scipy_chain():
    rs = np.random.RandomState(0)
    for _ in range(50):
        q, _ = rand_state(rs, vel=False)
        d = fresh(q)
        L.so100o_kinematics(C.byref(M), C.byref(d))
        ref = scipy_fk(q)
        for k in range(6):
            np.testing.assert_allclose(O.arr(d.xpos)[k + 2], ref[k][0], atol=1e-14)
            np.testing.assert_allclose(O.arr(d.xmat)[k + 2].reshape(3, 3), ref[k][1].as_matrix(), atol=1e-14)
        cam_p = ref[4][0] + ref[4][1].apply(np.array(M.cam_pos))
        cam_R = (ref[4][1] * quat_wxyz_to_R(np.array(M.cam_quat))).as_matrix()
        np.testing.assert_allclose(O.arr(d.cam_xpos), cam_p, atol=1e-14)
        np.testing.assert_allclose(O.arr(d.cam_xmat).reshape(3, 3), cam_R, atol=1e-14)
        np.testing.assert_allclose(O.arr(d.xpos)[8], q[6:9], atol=0)


def mass_matrix(q):
    d = fresh(q)
    L.so100o_forward(C.byref(M), C.byref(d), 0, 0)
    return O.arr(d.M).reshape(NV, NV).copy()


def bias(q, v):
    d = fresh(q, v)
    out = np.zeros(NV)
    L.so100o_rne(C.byref(M), C.byref(d), None, out.ctypes.data_as(C.c_void_p))
    return out


def test_crb_equals_rne_columns():
    """M from composite-rigid-body == columns of RNE(q, 0, e_i) - RNE(q, 0, 0)."""
    rs = np.random.RandomState(1)
    for _ in range(10):
        q, _ = rand_state(rs, vel=False)
        Mq = mass_matrix(q)
        np.testing.assert_allclose(Mq, Mq.T, atol=1e-18)
        assert np.all(np.linalg.eigvalsh(Mq) > 0)
        b0 = bias(q, np.zeros(NV))
        for i in range(NV):
            d = fresh(q)
            e = np.zeros(NV); e[i] = 1.0; out = np.zeros(NV)
            L.so100o_rne(C.byref(M), C.byref(d), e.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
            np.testing.assert_allclose(out - b0, Mq[:, i], atol=1e-15)


def body_energy(q, v, eps=1e-7):
    """Kinetic and potential energy of all bodies from FK alone (central differences along v)."""
    def frames(qq):
        d = fresh(qq); L.so100o_kinematics(C.byref(M), C.byref(d))
        return O.arr(d.xipos).copy(), O.arr(d.ximat).copy().reshape(9, 3, 3)

    def advance(qq, h):
        out = qq.copy(); out[:6] += h * v[:6]; out[6:9] += h * v[6:9]
        rot = quat_wxyz_to_R(qq[9:13]) * R.from_rotvec(h * v[9:12])     # body-frame angular velocity
        x = rot.as_quat(); out[9:13] = [x[3], x[0], x[1], x[2]]
        return out
    p0, R0 = frames(q)
    pp, Rp = frames(advance(q, eps)); pm, Rm = frames(advance(q, -eps))
    T = 0.0; V = 0.0
    for b in range(2, 9):
        m = M.body_mass[b]; I = np.diag(np.array(M.body_inertia[b]))
        vc = (pp[b] - pm[b]) / (2 * eps)
        dR = (Rp[b] - Rm[b]) / (2 * eps)
        W = R0[b].T @ dR                       # [omega_local]x
        w = np.array([W[2, 1], W[0, 2], W[1, 0]])
        T += 0.5 * m * vc @ vc + 0.5 * w @ I @ w
        V += m * 9.81 * p0[b][2]
    return T, V


def test_mass_matrix_is_kinetic_energy():
    rs = np.random.RandomState(2)
    for _ in range(10):
        q, v = rand_state(rs)
        Mq = mass_matrix(q)
        T, _ = body_energy(q, v)
        arm = 0.5 * SO_ARM * (v[:6] @ v[:6])
        assert 0.5 * v @ Mq @ v - arm == pytest.approx(T, rel=2e-7)


SO_ARM = 0.1


def test_gravity_bias_is_potential_gradient():
    rs = np.random.RandomState(3)
    for _ in range(5):
        q, _ = rand_state(rs, vel=False)
        g = bias(q, np.zeros(NV))
        eps = 1e-6
        for i in range(6):
            qp = q.copy(); qm = q.copy(); qp[i] += eps; qm[i] -= eps
            dV = (body_energy(qp, np.zeros(NV))[1] - body_energy(qm, np.zeros(NV))[1]) / (2 * eps)
            assert g[i] == pytest.approx(dV, abs=1e-8)
        np.testing.assert_allclose(g[6:9], [0, 0, 0.008 * 9.81], atol=1e-15)
        np.testing.assert_allclose(g[9:12], 0, atol=1e-15)


def test_coriolis_bias_is_lagrange():
    """arm: c(q,v) = Mdot v - 1/2 d(v'Mv)/dq  (finite differences of the CRB matrix)."""
    rs = np.random.RandomState(4)
    for _ in range(5):
        q, v = rand_state(rs)
        v[6:] = 0
        c = bias(q, v) - bias(q, np.zeros(NV))
        eps = 1e-6
        dM = []
        for i in range(6):
            qp = q.copy(); qm = q.copy(); qp[i] += eps; qm[i] -= eps
            dM.append((mass_matrix(qp)[:6, :6] - mass_matrix(qm)[:6, :6]) / (2 * eps))
        Mdot = sum(dM[i] * v[i] for i in range(6))
        want = Mdot @ v[:6] - 0.5 * np.array([v[:6] @ dM[i] @ v[:6] for i in range(6)])
        np.testing.assert_allclose(c[:6], want, atol=2e-8)


def test_cube_gyroscopic_and_free_fall():
    q, v = rand_state(np.random.RandomState(5))
    b = bias(q, v)
    np.testing.assert_allclose(b[9:12], 0, atol=1e-15)            # isotropic inertia: w x Iw = 0
    d = fresh(); O.arr(d.qpos)[8] = 1.0
    n = 100
    L.so100o_step(C.byref(M), C.byref(d), 0, 0, n)
    h = 0.002
    assert O.arr(d.qvel)[8] == pytest.approx(-9.81 * h * n, rel=1e-12)
    assert O.arr(d.qpos)[8] == pytest.approx(1.0 - 9.81 * h * h * n * (n + 1) / 2, rel=1e-12)   # semi-implicit Euler


def kkt_check(d, tol=1e-9):
    n = d.nefc
    J = O.arr(d.efc_J)[:n]; f = O.arr(d.efc_force)[:n]; Rr = O.arr(d.efc_R)[:n]
    aref = O.arr(d.efc_aref)[:n]; typ = O.arr(d.efc_type)[:n]; fl = O.arr(d.efc_floss)[:n]
    qacc = O.arr(d.qacc)
    Mq = O.arr(d.M).reshape(NV, NV)
    # qacc = qacc_smooth + Minv J' f
    np.testing.assert_allclose(Mq @ (qacc - O.arr(d.qacc_smooth)), J.T @ f, atol=1e-9)
    res = J @ qacc - aref + Rr * f          # gradient of the dual cost
    for i in range(n):
        if typ[i] == 0:
            if abs(f[i]) < fl[i] - 1e-12: assert abs(res[i]) < tol
            elif f[i] > 0: assert res[i] < tol
            else: assert res[i] > -tol
        else:
            assert f[i] >= 0
            if f[i] > 1e-12: assert abs(res[i]) < tol
            else: assert res[i] > -tol


def test_constraint_solve_kkt():
    rs = np.random.RandomState(6)
    flags = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR
    rng = np.array(M.jnt_range)
    seen_limit = seen_contact = 0
    for t in range(40):
        q, v = rand_state(rs)
        if t % 2 == 0:      # push some joints beyond their limits
            k = rs.randint(0, 6); q[k] = rng[k, rs.randint(0, 2)] + rs.uniform(-0.05, 0.05)
        if t % 3 == 0:      # cube in / near the floor, tilted
            q[8] = rs.uniform(0.0, 0.015)
            if t % 2: q[9:13] = [1, 0, 0, 0]
        d = fresh(q, v * 0.3)
        O.arr(d.ctrl)[:] = rs.uniform(-3, 3, 6)
        L.so100o_forward(C.byref(M), C.byref(d), flags, 0)
        assert d.solver_last_change < 1e-12
        kkt_check(d)
        seen_limit += np.sum(O.arr(d.efc_type)[:d.nefc] == 1); seen_contact += d.ncon
    assert seen_limit > 5 and seen_contact > 10


def test_cube_rests_on_floor():
    d = fresh()                                  # cube at z = 0: 1 cm inside the floor
    flags = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR
    L.so100o_step(C.byref(M), C.byref(d), flags, 0, 1000)
    z = O.arr(d.qpos)[8]
    # steady state: 16 pyramid edges share m g; f = -K imp dist / R  
    # closed form: per edge f = m g / 16 and K imp(dist) dist + R(dist) f = 0 with R = 4 (1-imp)/imp / m
    dist = z - 0.01
    x = abs(dist) / 0.001; imp = 0.9 + 0.05 * (2 * x * x if x <= 0.5 else 1 - 2 * (1 - x)**2)
    Rr = 4 * 125.0 * (1 - imp) / imp
    assert 2770.0831 * imp * dist + Rr * 0.008 * 9.81 / 16 == pytest.approx(0.0, abs=1e-6)
    assert 0.0098 < z < 0.01
    assert abs(O.arr(d.qvel)[8]) < 1e-9
    np.testing.assert_allclose(O.arr(d.qpos)[9:13], [1, 0, 0, 0], atol=1e-12)
    f = O.arr(d.efc_force)[6:6 + 16] if d.nefc >= 22 else None
    assert d.ncon == 4 and f is not None
    assert f.sum() == pytest.approx(0.008 * 9.81, rel=1e-6)      # each edge has unit normal component


def test_servo_steady_state_and_friction_deadband():
    """No velocity: the servo holds ctrl up to gravity sag and friction-loss dead band:
    |kp (ctrl - q) - g(q)| <= frictionloss at rest."""
    d = fresh()
    target = np.array([0.3, -1.5, 1.2, 0.4, -0.5, 0.3])
    O.arr(d.qpos)[:6] = target; O.arr(d.ctrl)[:] = target
    O.arr(d.qpos)[8] = 0.01
    flags = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR
    L.so100o_step(C.byref(M), C.byref(d), flags, 0, 3000)
    q = O.arr(d.qpos)[:6].copy()
    # friction loss is a SOFT constraint (R > 0): a slow creep of O(1e-4 rad/s) remains, as in MuJoCo
    assert np.max(np.abs(O.arr(d.qvel)[:6])) < 1e-3
    g = bias(np.r_[q, O.arr(d.qpos)[6:]], np.zeros(NV))[:6]
    resid = 50 * (target - q) - g
    assert np.all(np.abs(resid) <= 0.1 + 1e-3)


def test_energy_conservation_passive_arm():
    """kp = kv = 0, no friction: total energy drifts only at O(h) over 2 s of swinging."""
    import copy
    m2 = copy.copy(M)
    m2.kp = 0.0
    for i in range(6): m2.kv[i] = 0.0
    d = fresh(); O.arr(d.qpos)[:6] = [0.2, -1.0, 1.0, 0.3, 0.1, 0.2]
    O.arr(d.qpos)[8] = 0.5

    def energy():
        qq = O.arr(d.qpos).copy(); vv = O.arr(d.qvel).copy(); vv[6:] = 0
        T, V = body_energy(qq, vv)
        return T + 0.5 * 0.1 * vv[:6] @ vv[:6] + V
    E0 = energy(); Es = []
    for _ in range(100):
        L.so100o_step(C.byref(m2), C.byref(d), O.F_CUBE_PINNED, 0, 10)
        Es.append(energy())
    swing = max(abs(O.arr(d.qvel)[:6]))
    assert swing > 0.1                          # it actually moves
    assert max(abs(np.array(Es) - E0)) < 2e-3 * abs(E0) + 2e-3


def test_non_finite_guard_ends_episode():
    """Product behaviour mirrored in the oracle (oracle/so100_oracle.c, so100_task.hpp::env_step_finish): a NaN action ends
    the episode with reward 0 and a zero terminal observation, and the env is usable again after its auto-reset."""
    for kind in (1, 2, 5, 6):
        e = O.OracleEnv(kind, flags=O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR, iters=0, seed=1)
        e.reset()
        a = np.zeros(6, np.float32)
        for _ in range(3):
            ob, r, term, trunc, tob = e.step(a, autoreset=True)
            assert not term and np.isfinite(ob).all()
        a[2] = np.nan
        ob, r, term, trunc, tob = e.step(a, autoreset=True)
        assert term and not trunc and r == 0.0 and e.e.bad_state == 1
        assert np.isfinite(ob).all() and (tob == 0).all()
        ob, r, term, trunc, tob = e.step(np.zeros(6, np.float32), autoreset=True)
        assert not term and np.isfinite(ob).all() and np.isfinite(r)
        assert np.isfinite(O.arr(e.d.qpos)).all() and np.isfinite(O.arr(e.d.qvel)).all()
