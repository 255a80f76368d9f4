"""The DEVICE contact code (csrc/so100_contact.hpp) instantiated on the host in fp64 and fp32 (tests/_hostcheck) against the
oracle: same narrowphase slot order, same rows, primal Newton.  In fp64 the two implementations must agree to round-off over
whole trajectories (they are independent formulations: link-frame dynamics + wrench-projected Jacobians vs the oracle's dense
world-frame rows); in fp32 per env step (16 substeps) from injected states -- across contact make/break events fp32 and fp64
trajectories separate (a corner touching one substep earlier changes the velocity by ~0.1 rad/s), which is physics, not error."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import so100_oracle as O
from test_oracle_contacts import C5, REF, M, L, floor_poses, fresh, rot, box_box, _grasp_state

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def H():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "_hostcheck"), "-s"])
    return C.CDLL(os.path.join(HERE, "_hostcheck", "libhostcheck.so"))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def pack(q, v, cube, cq, cvel=None):
    st = np.zeros(49); st[:6] = q; st[6:12] = v[:6]; st[30:33] = cube; st[33:37] = cq
    if cvel is not None:
        st[37:43] = cvel
    return st


def host_steps(fn, st, ctrl, flags, n, iters=4, citers=20):
    stat = np.zeros(5, np.int32); ct = np.ascontiguousarray(ctrl, np.float64); ap = np.zeros(3)
    fn(P(st), P(ct), P(ap), flags, iters, citers, n, P(stat))
    return stat


def err(st, d):
    qo = np.concatenate([O.arr(d.qpos)[:6], O.arr(d.qpos)[6:9]]); vo = np.concatenate([O.arr(d.qvel)[:6], O.arr(d.qvel)[6:9]])
    return np.abs(np.concatenate([st[:6], st[30:33]]) - qo).max(), np.abs(np.concatenate([st[6:12], st[37:40]]) - vo).max()


def test_box_box_device_code_equals_oracle(H):
    rs = np.random.RandomState(8); hits = 0; same32 = 0
    for _ in range(3000):
        RA, _ = rot(rs); RB, _ = rot(rs)
        hA = np.array([0.001, 0.005 + 0.005*rs.rand(), 0.004 + 0.004*rs.rand()]); hB = np.full(3, 0.01)
        cB = rs.randn(3)*0.1; cA = rs.randn(3); cA = cB + cA*(0.004 + 0.012*rs.rand())/np.linalg.norm(cA)
        k, pos, n, dist = box_box(cA, RA, hA, cB, RB, hB)
        out = {}
        for name, fn in (("d", H.hc_boxbox_d), ("f", H.hc_boxbox_f)):
            p2 = np.zeros((8, 3)); n2 = np.zeros(3); d2 = np.zeros(8)
            args = [np.ascontiguousarray(a, np.float64) for a in (cA, RA, hA, cB, RB, hB)]
            k2 = fn(*[P(a) for a in args], P(p2), P(n2), P(d2))
            out[name] = (k2, p2[:k2], n2, d2[:k2])
        k2, p2, n2, d2 = out["d"]
        assert k2 == k
        if k:
            hits += 1
            assert np.allclose(p2, pos, atol=1e-12) and np.allclose(n2, n, atol=1e-12) and np.allclose(d2, dist, atol=1e-12)
            k3, p3, n3, d3 = out["f"]
            if k3 == k and np.allclose(n3, n, atol=1e-4):    # fp32: same manifold unless two axes tie to within round-off
                same32 += 1
                assert np.allclose(p3, pos, atol=2e-6) and np.allclose(d3, dist, atol=2e-6)
    assert hits > 500 and same32 > 0.97*hits


def test_pad_floor_device_code_vs_oracle(H):
    rs = np.random.RandomState(0); worst64 = worst32 = 0.0; touched = 0; compared = 0
    for q in floor_poses(10, 21):
        v = rs.randn(6)*0.5; ctrl = q + rs.randn(6)*0.05
        cube = np.array([0.15, -0.2, 0.0099]); cq = np.array([1.0, 0, 0, 0])
        d = fresh(q, np.concatenate([v, np.zeros(6)]), cube, cq); O.arr(d.ctrl)[:] = ctrl
        s64 = pack(q, v, cube, cq); s32 = pack(q, v, cube, cq)
        n64 = 0; same_sets = True
        for sub in range(96):                                # 96 substeps in fp64; fp32 for as long as it sees the same contact sets
            L.so100o_step(C.byref(M), C.byref(d), REF, -1, 1)
            st = host_steps(H.hc_csub_d, s64, ctrl, REF, 1); n64 = max(n64, st[0])
            assert st[0] == sum(1 for i in range(d.ncon) if d.con[i].kind == 1)
            e = err(s64, d); worst64 = max(worst64, e[0], e[1]*1e-2)
            if same_sets:
                st32 = host_steps(H.hc_csub_f, s32, ctrl, REF, 1)
                same_sets = st32[0] == st[0]                 # a corner made / broke contact a substep apart: the runs separate here
                if same_sets:
                    e = err(s32, d); worst32 = max(worst32, e[0], e[1]*1e-2); compared += 1
        touched += n64 > 0
    assert touched >= 8 and compared > 300
    assert worst64 < 1e-11                                   # positions 1e-11, velocities 1e-9: the same algorithm to round-off
    assert worst32 < 2e-6                                    # fp32 on identical contact sets: angles 2e-6 rad, velocities 2e-4 rad/s


def test_grasp_device_code_vs_oracle(H):
    """the coupled 12-dof solve (pad/cube box-box + cube/floor + friction-loss + limits) through the closing-jaw scenario"""
    q, centre, cq = _grasp_state()
    ctrl = q.copy(); ctrl[5] = -0.2
    d = fresh(q, cube=centre, cquat=cq); O.arr(d.ctrl)[:] = ctrl
    s64 = pack(q, np.zeros(6), centre, cq); s32 = pack(q, np.zeros(6), centre, cq)
    coupled_seen = 0; ncon_max = 0; w32 = 0.0
    for s in range(160):
        L.so100o_step(C.byref(M), C.byref(d), C5, -1, 1)
        st = host_steps(H.hc_csub_d, s64, ctrl, C5, 1)
        coupled_seen += st[1]; ncon_max = max(ncon_max, st[0])
        assert st[0] == d.ncon and st[2] == 0
        e = err(s64, d)
        assert e[0] < 1e-11 and e[1] < 1e-9, (s, e)
        if s < 30:                                           # fp32 through the first impact (contacts appear at substep 13)
            host_steps(H.hc_csub_f, s32, ctrl, C5, 1)
            e32 = err(s32, d); w32 = max(w32, e32[0], e32[1]*1e-2)
    assert coupled_seen > 100 and ncon_max >= 8
    assert w32 < 1e-5                                        # 1e-5 rad / m, 1e-3 rad/s / m/s while the cube is being hit at ~1 m/s
    cnt = np.zeros(3, np.int64); H.hc_cdbg_counters(P(cnt))
    assert cnt[1] < 4*cnt[0]                                 # Newton stays at a few iterations per substep


def test_contact_budget_is_counted_not_exceeded(H):
    """jaw lying flat on the floor: more than 16 pad corners touch; the device keeps the first 16 in pad order and counts the rest
    exactly as the oracle's model.max_contacts does"""
    rs = np.random.RandomState(3)
    found = 0
    for trial in range(4000):
        q = np.array([-2.2, -3.14158, 0, -2.0, -3.14158, -0.2]) + np.array([4.4, 3.34158, 3.14158, 3.8, 6.28316, 2.2])*rs.rand(6)
        d = fresh(q, cube=[0.2, -0.3, 0.0099]); L.so100o_forward(C.byref(M), C.byref(d), REF, -1)
        if d.ncon_dropped == 0:
            continue
        found += 1
        s64 = pack(q, np.zeros(6), [0.2, -0.3, 0.0099], [1, 0, 0, 0])
        st = host_steps(H.hc_csub_d, s64, q, REF, 1)
        npad = sum(1 for i in range(d.ncon) if d.con[i].kind == 1)
        assert npad == 16 and st[0] == 16 and st[2] == d.ncon_dropped
        d2 = fresh(q, cube=[0.2, -0.3, 0.0099]); O.arr(d2.ctrl)[:] = q
        L.so100o_step(C.byref(M), C.byref(d2), REF, -1, 1)
        e = err(s64, d2)
        assert e[0] < 1e-10 and e[1] < 1e-7
        if found >= 3:
            break
    assert found >= 1


def test_active_set_memory_over_whole_env_steps(H):
    """the kernels' calling pattern: 16 substeps per call, the Newton's active-set memory (arm-row zones, per-contact edge masks)
    carried from substep to substep.  Closing-jaw grasps with slightly turned cubes, 6 env steps through the impact, fp32 device
    code vs the oracle for as long as both see the same contact counts (this is the scenario of tests/test_gpu_contacts.py on the
    host; the first version of the memory stopped at a kink of the piecewise-quadratic cost and was 3e-3 off here)."""
    rs = np.random.RandomState(1)
    q0, centre, cq = _grasp_state()
    worst = 0.0; compared = 0; passes0 = None
    H.hc_cdbg_passes.restype = C.c_long
    c0 = np.zeros(3, np.int64); H.hc_cdbg_counters(P(c0)); p0 = H.hc_cdbg_passes()
    for i in range(12):
        q = q0.copy(); q[5] = 0.065 + rs.uniform(0.0, 0.01)
        cube = centre + rs.uniform(-1, 1, 3)*np.array([0.0004, 0.002, 0.002])
        w = rs.randn(3)*0.03; ang = np.linalg.norm(w); ax = w/ang
        a, b = cq, np.array([np.cos(ang/2), *(np.sin(ang/2)*ax)])
        quat = np.array([a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3], a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
                         a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1], a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]])
        q32 = q.astype(np.float32).astype(np.float64); c32 = cube.astype(np.float32).astype(np.float64); qq32 = quat.astype(np.float32).astype(np.float64)
        d = fresh(q32, cube=c32, cquat=qq32)
        st = pack(q32, np.zeros(6), c32, qq32)
        for t in range(6):
            ctrl = st[:6].copy(); ctrl[5] -= 0.075
            O.arr(d.ctrl)[:] = O.arr(d.qpos)[:6]; O.arr(d.ctrl)[5] -= 0.075
            nmax = 0
            for s in range(16):
                L.so100o_step(C.byref(M), C.byref(d), C5, -1, 1); nmax = max(nmax, d.ncon)
            stat = host_steps(H.hc_csub_f, st, ctrl, C5, 16, citers=30)
            if stat[0] != nmax:
                break
            e = err(st, d); worst = max(worst, e[0], e[1]*1e-2); compared += 1
    c1 = np.zeros(3, np.int64); H.hc_cdbg_counters(P(c1)); p1 = H.hc_cdbg_passes()
    assert compared > 40
    assert worst < 2e-5                                      # measured 1e-6: 2e-5 rad / m, 2e-3 per second
    assert (p1 - p0) < 3.5*(c1 - c0)[0]                      # row passes per solve (7.2 before the memory; 2.9 with it)


def test_feature_hash_of_the_active_set_memory(H):
    """the one-byte memory entry keeps a 4-bit hash of the contact's feature id (pad/floor: 8 * pad + corner): the same corner of
    different pads must hash differently (two pads of a jaw lying flat touch with the same corners), and so must the corners of
    one pad; collisions that remain (different corners of different pads) only cost a worse first guess"""
    h = lambda i: H.hc_contact_id_hash(i)
    for corner in range(8):
        assert len({h(8*g + corner) for g in range(8)}) == 8
    for g in range(8):
        assert len({h(8*g + c) for c in range(8)}) == 8
    assert all(0 <= h(i) < 16 for i in range(256))


def test_capsule_box_device_code_equals_oracle(H):
    """the device's capsule-box stand-in (so100_contact.hpp: capsule_box) against oracle/so100_oracle.c: so100o_capsule_box: fp64 to round-off,
    fp32 to 2e-6 m in distance, 5e-5 m in position, 5e-3 in the normal on the same hit-or-miss decision except within round-off of touching"""
    from test_oracle_contacts import capsule_box
    H.hc_capbox_d.argtypes = H.hc_capbox_f.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rs = np.random.RandomState(11); hits = 0; grazing = 0
    for _ in range(3000):
        R, _ = rot(rs); c = rs.randn(3)*0.1; h = np.full(3, 0.01); r = 0.02 + 0.004*rs.rand()
        u = rs.randn(3); u /= np.linalg.norm(u)
        mid = c + u*(0.01 + r)*(0.5 + 1.0*rs.rand()); w = rs.randn(3); w /= np.linalg.norm(w)
        ln = 0.1 + 0.02*rs.rand(); t0 = rs.rand()
        a = mid - w*ln*t0; b = mid + w*ln*(1 - t0)
        k, pos, n, dist = capsule_box(a, b, r, c, R, h)
        args = [np.ascontiguousarray(x, np.float64) for x in (a, b, c, R, h)]
        for name, fn, tol in (("d", H.hc_capbox_d, 1e-12), ("f", H.hc_capbox_f, 2e-6)):
            p2 = np.zeros(3); n2 = np.zeros(3); d2 = np.zeros(1)
            k2 = fn(P(args[0]), P(args[1]), float(r), P(args[2]), P(args[3]), P(args[4]), P(p2), P(n2), P(d2))
            if k2 != k:
                assert name == "f" and abs(dist if k else d2[0]) < 2e-6       # fp32 may disagree only within round-off of touching
                grazing += 1
                continue
            if k:
                # (the nearest point slides along directions in which the distance is flat -- a segment nearly parallel to a face: fp32 position 25 x looser)
                assert np.allclose(p2, pos, atol=tol*25) and abs(d2[0] - dist) < tol and np.allclose(n2, n, atol=tol*2500)
        hits += k
    assert hits > 800 and grazing < 5
