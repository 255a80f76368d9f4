"""Oracle checks for the finger-pad contacts (pad/floor live in the reference scene; pad/cube = BASELINE.json configs[4]):
narrowphase geometry, contact Jacobians, the primal Newton solver against PGS-on-the-dual and the KKT conditions, and
end-to-end behaviour (an arm driven into the floor stops on its pads; a closing jaw holds the cube against gravity).
MuJoCo is not available: these are model-independent identities, not parity pins ("parity unpinned (physics)")."""
import ctypes as C

import numpy as np
import pytest

from oracle import so100_oracle as O

L = O.lib()
M = O.model()
ARMROWS = O.F_FRICTIONLOSS | O.F_LIMITS
REF = O.F_REFERENCE
C5 = O.F_CONTACT5
LO = np.array([-2.2, -3.14158, 0, -2.0, -3.14158, -0.2]); HI = np.array([2.2, 0.2, 3.14158, 1.8, 3.14158, 2.0])


def fresh(q=None, v=None, cube=None, cquat=None):
    d = O.Data()
    L.so100o_reset_data(C.byref(M), C.byref(d))
    if q is not None:
        O.arr(d.qpos)[:6] = q
    if v is not None:
        O.arr(d.qvel)[:len(v)] = v
    if cube is not None:
        O.arr(d.qpos)[6:9] = cube
    if cquat is not None:
        O.arr(d.qpos)[9:13] = np.asarray(cquat) / np.linalg.norm(cquat)
    return d


def fwd(d, flags, iters):
    L.so100o_forward(C.byref(M), C.byref(d), flags, iters)


def pad_frames(d):
    """world centre and rotation of the 8 pad boxes"""
    xp = O.arr(d.xpos); xm = O.arr(d.xmat)
    out = []
    for g in range(8):
        b = M.pad_body[g]; R = xm[b].reshape(3, 3)
        out.append((xp[b] + R @ np.array(M.pad_pos[g][:]), R.copy(), np.array(M.pad_size[g][:])))
    return out


def floor_poses(n, seed, band=0.003):
    """random arm poses whose lowest pad corner is within `band` below .. above the floor"""
    rs = np.random.RandomState(seed); out = []
    while len(out) < n:
        q = LO + (HI - LO) * rs.rand(6)
        d = fresh(q); L.so100o_kinematics(C.byref(M), C.byref(d))
        z = min(c[2] - (np.abs(R[2]) * h).sum() for c, R, h in pad_frames(d))
        if -band < z < 0.0005 and O.arr(d.xpos)[5][2] > 0.03:
            out.append(q)
    return out


def rot(rs):
    q = rs.randn(4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y)], [2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x)],
                     [2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y)]]), q


def box_box(cA, RA, hA, cB, RB, hB):
    pos = np.zeros((8, 3)); n = np.zeros(3); dist = np.zeros(8)
    p = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)
    cA, RA, hA, cB, RB, hB = [np.ascontiguousarray(a, np.float64) for a in (cA, RA, hA, cB, RB, hB)]
    k = L.so100o_box_box(p(cA), p(RA), p(hA), p(cB), p(RB), p(hB), pos.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p))
    return k, pos[:k].copy(), n.copy(), dist[:k].copy()


def inside(p, c, R, h, tol):
    return bool(np.all(np.abs(R.T @ (p - c)) <= h + tol))


# ---- narrowphase -------------------------------------------------------------------------------------------------
def test_box_box_known_configurations():
    I = np.eye(3); hc = np.full(3, 0.01); hp = np.array([0.001, 0.01, 0.008])
    # separated
    assert box_box([0, 0, 0], I, hp, [0.02, 0, 0], I, hc)[0] == 0
    # pad flat against the cube's -x face, 0.4 mm deep: 4 face contacts at the pad's corners, normal +x, midway points
    k, pos, n, dist = box_box([-0.0106, 0.002, 0.001], I, hp, [0, 0, 0], I, hc)
    assert k == 4 and np.allclose(n, [1, 0, 0]) and np.allclose(dist, -0.0004)
    assert np.allclose(sorted(pos[:, 1]), [-0.008, -0.008, 0.01, 0.01]) and np.allclose(sorted(pos[:, 2]), [-0.007, -0.007, 0.009, 0.009])
    assert np.allclose(pos[:, 0], -0.0098)                   # between the pad face (-0.0096) and the cube face (-0.01)
    # pad hanging over the cube's edge: the overlap rectangle, clipped by the cube face
    k, pos, n, dist = box_box([-0.0106, 0.015, 0.0], I, hp, [0, 0, 0], I, hc)
    assert k == 4 and np.allclose(sorted(pos[:, 1]), [0.005, 0.005, 0.01, 0.01])
    # pad rotated 45 degrees about the face normal, centred: an octagon is impossible here (pad smaller than the face) -> 4 points
    c, s = np.cos(np.pi/4), np.sin(np.pi/4)
    Rx = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    k, pos, n, dist = box_box([-0.0106, 0, 0], Rx, [0.001, 0.006, 0.005], [0, 0, 0], I, hc)
    assert k == 4 and np.allclose(dist, -0.0004)
    # a big square rotated 45 degrees over the cube face: the clipped polygon is an octagon -> 8 contacts
    k, pos, n, dist = box_box([-0.0106, 0, 0], Rx, [0.001, 0.013, 0.013], [0, 0, 0], I, hc)
    assert k == 8 and np.allclose(dist, -0.0004)
    # edge-edge: two crossed bars touching along their edges
    Ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])       # bar A rotated about y: an edge points down the x-z diagonal
    Rz45x = Rx                                               # bar B rotated about x
    k, pos, n, dist = box_box([0, 0, 0.0275], Ry @ np.diag([1, 1, 1]), [0.01, 0.05, 0.01], [0, 0, 0], Rz45x.T @ np.diag([1, 1, 1]) , [0.05, 0.01, 0.01])
    assert k == 1 and abs(abs(n[2]) - 1) < 1e-9 and dist[0] < 0 and abs(dist[0] + (2*0.01*np.sqrt(2) - 0.0275)) < 1e-9


def test_box_box_random_fuzz_invariants():
    rs = np.random.RandomState(4); hits = 0; faces = 0
    for trial in range(3000):
        RA, _ = rot(rs); RB, _ = rot(rs)
        hA = np.array([0.001, 0.005 + 0.005*rs.rand(), 0.004 + 0.004*rs.rand()]); hB = np.full(3, 0.01)
        cB = np.zeros(3); cA = rs.randn(3); cA *= (0.004 + 0.012*rs.rand()) / np.linalg.norm(cA)
        k, pos, n, dist = box_box(cA, RA, hA, cB, RB, hB)
        if k == 0:
            continue
        hits += 1; faces += k > 1
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and n @ (cB - cA) > 0          # unit normal from A to B
        assert np.all(dist <= 1e-15)
        for p, ds in zip(pos, dist):                                               # midway points lie in both boxes (to |dist|/2)
            assert inside(p, cA, RA, hA, -ds/2 + 1e-9) and inside(p, cB, RB, hB, -ds/2 + 1e-9)
        # swapping the arguments: same contact set, opposite normal
        k2, pos2, n2, dist2 = box_box(cB, RB, hB, cA, RA, hA)
        assert k2 == k and np.allclose(n2, -n, atol=1e-12)
        o1 = np.lexsort(np.round(pos, 9).T); o2 = np.lexsort(np.round(pos2, 9).T)
        assert np.allclose(pos[o1], pos2[o2], atol=1e-10) and np.allclose(dist[o1], dist2[o2], atol=1e-10)
        # rigid-motion invariance
        Q, _ = rot(rs); t = rs.randn(3)*0.1
        k3, pos3, n3, dist3 = box_box(Q @ cA + t, Q @ RA, hA, Q @ cB + t, Q @ RB, hB)
        assert k3 == k and np.allclose(n3, Q @ n, atol=1e-9) and np.allclose(pos3, pos @ Q.T + t, atol=1e-9) and np.allclose(dist3, dist, atol=1e-9)
        # the deepest contact is the SAT depth along the normal (support functions)
        depth = (np.abs(RA.T @ n) @ hA + np.abs(RB.T @ n) @ hB) - n @ (cB - cA)
        assert depth > -1e-12 and (-dist.min() <= depth + 1e-9)
    assert hits > 500 and faces > 300


def test_plane_box_matches_the_cube_case_and_the_geometry():
    rs = np.random.RandomState(1)
    for _ in range(200):
        R, _ = rot(rs); h = np.array([0.001, 0.01, 0.008]); c = np.array([rs.randn()*0.1, rs.randn()*0.1, rs.rand()*0.012 - 0.002])
        pos = np.zeros((4, 3)); dist = np.zeros(4)
        k = L.so100o_plane_box(c.ctypes.data_as(C.c_void_p), np.ascontiguousarray(R).ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p),
                               pos.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p))
        corners = np.array([c + R @ (np.array([sx, sy, sz])*h) for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)])
        want = [p for p in corners if p[2] <= 0 and p[2] <= c[2]][:4]
        assert k == len(want)
        for p, w, ds in zip(pos[:k], want, dist[:k]):
            assert np.allclose(p[:2], w[:2]) and abs(ds - w[2]) < 1e-15 and abs(p[2] - w[2]/2) < 1e-15


# ---- rows ----------------------------------------------------------------------------------------------------------
def test_contact_parameters_mixing():
    q = floor_poses(1, 3)[0]
    d = fresh(q, cube=[0.0, -0.3, 0.0099]); fwd(d, C5, -1)
    assert d.ncon > 0
    kinds = [d.con[i].kind for i in range(d.ncon)]
    assert 0 in kinds and 1 in kinds
    for i in range(d.ncon):
        c = d.con[i]
        if c.kind == 0:                                      # cube/floor: MuJoCo defaults
            assert list(c.solref) == [0.02, 1.0] and np.allclose(list(c.solimp), [0.9, 0.95, 0.001, 0.5, 2]) and c.mu == 1.0
        else:                                                # a pad involved: 1:1 mix with arm:61, then the clamp
            assert np.allclose(list(c.solref), [0.015, 1.0]) and np.allclose(list(c.solimp), [0.9999, 0.975, 0.0055, 0.5, 2]) and c.mu == 1.0
        f = np.array(c.frame[:]).reshape(3, 3)
        assert np.allclose(f @ f.T, np.eye(3), atol=1e-12) and np.linalg.det(f) > 0.999


def test_contact_jacobian_is_the_material_point_velocity():
    """efc_J of a contact row times qvel = relative velocity of the two bodies' material points at the contact, along the edge
    direction: checked against finite differences of the body-fixed points through the kinematics."""
    rs = np.random.RandomState(0)
    for q in floor_poses(6, 5):
        cube = None
        d = fresh(q, cube=[0.1, -0.25, 0.0099]); fwd(d, REF, -1)
        assert d.ncon > 0
        xp = O.arr(d.xpos).copy(); xm = O.arr(d.xmat).copy(); qpos0 = O.arr(d.qpos).copy()
        for ci in range(d.ncon):
            c = d.con[ci]
            if c.kind != 1:
                continue
            b = c.b2; p = np.array(c.pos[:]); f = np.array(c.frame[:]).reshape(3, 3)
            ploc = xm[b].reshape(3, 3).T @ (p - xp[b])
            Jnum = np.zeros((3, 6)); eps = 1e-6
            for j in range(6):
                pp = []
                for sg in (1, -1):
                    dd = fresh(qpos0[:6] + sg*eps*np.eye(6)[j]); L.so100o_kinematics(C.byref(M), C.byref(dd))
                    pp.append(O.arr(dd.xpos)[b] + O.arr(dd.xmat)[b].reshape(3, 3) @ ploc)
                Jnum[:, j] = (pp[0] - pp[1]) / (2*eps)
            for e, (k, sgn) in enumerate(((0, 1), (0, -1), (1, 1), (1, -1))):
                dirv = f[0] + sgn*c.mu*f[1 + k]
                row = O.arr(d.efc_J)[c.efc0 + e]
                assert np.allclose(row[:6], dirv @ Jnum, atol=1e-8) and np.all(row[6:] == 0)


def _kkt(d, nvs, tol):
    n = d.nefc
    J = O.arr(d.efc_J)[:n, :nvs]; f = O.arr(d.efc_force)[:n]; R = O.arr(d.efc_R)[:n]; aref = O.arr(d.efc_aref)[:n]
    Mm = O.arr(d.M).reshape(12, 12)[:nvs, :nvs]
    qacc = O.arr(d.qacc)[:nvs]; a0 = O.arr(d.qacc_smooth)[:nvs]
    assert np.allclose(Mm @ (qacc - a0), J.T @ f, atol=tol * max(1.0, np.abs(J.T @ f).max()))     # stationarity
    jar = J @ qacc - aref
    for r in range(n):
        t = d.efc_type[r]
        if t == 0:
            fl = d.efc_floss[r]
            assert abs(f[r]) <= fl + 1e-12
            if abs(f[r]) < fl - 1e-9:
                assert abs(f[r] + jar[r]/R[r]) < tol*max(1, abs(f[r]))
            else:
                assert np.sign(f[r]) == -np.sign(jar[r]) and abs(jar[r]) >= R[r]*fl - 1e-9
        else:
            assert f[r] >= 0 and abs(f[r] - max(0.0, -jar[r]/R[r])) < tol*max(1, abs(f[r]))


def test_newton_equals_pgs_without_pad_contacts():
    rs = np.random.RandomState(2)
    for _ in range(20):
        q = LO + (HI - LO)*rs.rand(6)
        if rs.rand() < 0.5:
            q[rs.randint(6)] = (HI if rs.rand() < 0.5 else LO)[0] * 0 + (HI + 0.01)[rs.randint(6)]   # push some joint past a limit
            q = np.clip(q, LO - 0.02, HI + 0.02)
        v = rs.randn(12)*np.array([2]*6 + [0.1]*3 + [1]*3)
        cz = 0.0095 + 0.0004*rs.rand()
        cq = np.array([1, 0.02*rs.randn(), 0.02*rs.randn(), 0.3*rs.randn()])
        a = fresh(q, v, [0.1, -0.3, cz], cq); b = fresh(q, v, [0.1, -0.3, cz], cq)
        O.arr(a.ctrl)[:] = q + rs.randn(6)*0.1; O.arr(b.ctrl)[:] = O.arr(a.ctrl)
        fwd(a, ARMROWS | O.F_FLOOR, 0); fwd(b, ARMROWS | O.F_FLOOR, -1)
        assert a.nefc == b.nefc and a.nefc >= 6
        assert np.allclose(O.arr(a.qacc), O.arr(b.qacc), rtol=1e-7, atol=1e-7)
        _kkt(b, 12, 1e-8)


def test_newton_kkt_with_pad_contacts_and_against_pgs():
    rs = np.random.RandomState(7)
    checked = 0
    for q in floor_poses(12, 11):
        v = np.concatenate([rs.randn(6)*1.0, np.zeros(6)])
        d = fresh(q, v, [0.15, -0.2, 0.0099]); O.arr(d.ctrl)[:] = q + rs.randn(6)*0.05
        fwd(d, REF, -1)
        npad = sum(1 for i in range(d.ncon) if d.con[i].kind == 1)
        if npad == 0:
            continue
        _kkt(d, 12, 1e-7)
        if d.nefc <= 40:                                     # PGS on the dual reaches the same point (slowly)
            e = fresh(q, v, [0.15, -0.2, 0.0099]); O.arr(e.ctrl)[:] = O.arr(d.ctrl)
            fwd(e, REF, 20000)
            assert np.allclose(O.arr(e.qacc), O.arr(d.qacc), rtol=2e-4, atol=2e-4 * max(1.0, np.abs(O.arr(d.qacc)).max()))
        checked += 1
    assert checked >= 6


# ---- behaviour -----------------------------------------------------------------------------------------------------
def _step(d, flags, n):
    L.so100o_step(C.byref(M), C.byref(d), flags, -1, n)


def test_arm_driven_into_the_floor_stops_on_its_pads():
    """VERDICT f-2, first slice: with the pad/floor contacts the gripper cannot pass through the floor (the situation the
    reward's `end_pos[2] < 0.02` term punishes, env_base_01.py:207-211); without them it does."""
    q0 = np.array([0.0, -1.6, 1.9, 1.5, 0.0, 0.3])          # gripper pointing down, lowest pad corner 28 mm above the floor
    lows = {}
    for flags in (ARMROWS | O.F_FLOOR, REF):
        d = fresh(q0, cube=[0.2, -0.2, 0.0099])
        ctrl = q0.copy(); ctrl[1] += 0.5                     # servo the shoulder 0.5 rad further down
        O.arr(d.ctrl)[:] = ctrl
        lowest = 1.0
        for _ in range(60):
            _step(d, flags, 16)
            L.so100o_kinematics(C.byref(M), C.byref(d))
            lowest = min(lowest, min(c[2] - (np.abs(R[2]) * h).sum() for c, R, h in pad_frames(d)))
        lows[flags] = lowest
        final = min(c[2] - (np.abs(R[2]) * h).sum() for c, R, h in pad_frames(d))
        assert np.all(np.isfinite(O.arr(d.qpos))) and np.abs(O.arr(d.qvel)[:6]).max() < 0.05      # at rest either way
    assert lows[ARMROWS | O.F_FLOOR] < -0.02                 # no arm contact: straight through the floor
    assert lows[REF] > -0.002                                # the impact at ~0.5 m/s dips < 2 mm into the (soft) contact ...
    assert -0.0003 < final < 0.0                             # ... and the gripper comes to rest on its pads, < 0.3 mm deep


def _grasp_state():
    """gripper horizontal 28 cm above the floor, closing direction along world x, gravity along the pads' short side; the cube
    floats between the jaws: 0.5 mm from the fixed jaw's pads, ~1.5 mm from the moving jaw's (jaw angle 0.1 rad)"""
    q = np.array([0.0, -1.9, 1.6, 0.3, 1.5708, 0.1])
    d = fresh(q); L.so100o_kinematics(C.byref(M), C.byref(d))
    xp = O.arr(d.xpos)[6].copy(); R = O.arr(d.xmat)[6].reshape(3, 3).copy()
    centre = xp + R @ np.array([-0.0026, -0.088, 0.0])
    # cube axes = jaw axes (a 180-degree turn about y here: w ~ 0, so the quaternion is taken from the largest diagonal term)
    cq = np.array([0.0, 0.0, 1.0, 0.0])
    Rq = np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1.0]])
    assert np.allclose(R, Rq, atol=1e-4)
    return q, centre, cq


def test_closing_jaw_holds_the_cube_against_gravity():
    """BASELINE.json configs[4]: pad/cube box-box contact, arm and cube dofs coupled in one solve.  The cube starts in mid-air
    between the open jaws; the jaw closes on it.  With the pad/cube contacts it is clamped (friction 1, pyramidal) and stays in
    the gripper; without them it falls to the floor."""
    q, centre, cq = _grasp_state()
    assert centre[2] > 0.03
    res = {}
    for flags in (REF, C5):
        d = fresh(q, cube=centre, cquat=cq)
        ctrl = q.copy(); ctrl[5] = -0.2                      # close the jaw
        O.arr(d.ctrl)[:] = ctrl
        maxcon = 0
        for _ in range(40):
            _step(d, flags, 16)
            maxcon = max(maxcon, sum(1 for i in range(d.ncon) if d.con[i].kind == 2))
        res[flags] = (O.arr(d.qpos)[6:9].copy(), O.arr(d.qvel).copy(), O.arr(d.qpos)[5], maxcon)
        assert np.all(np.isfinite(O.arr(d.qpos)))
    assert res[REF][0][2] < 0.0105 and res[REF][3] == 0      # fell to the floor
    pos, vel, jaw, ncon = res[C5]
    assert ncon >= 8                                          # face-face manifolds on both sides
    # clamped after sliding ~1 cm while the jaw closed; what remains is the slow creep of soft pyramidal friction (< 5 mm/s)
    assert pos[2] > centre[2] - 0.015 and np.abs(vel[6:9]).max() < 5e-3 and np.abs(vel[:6]).max() < 5e-2
    assert jaw > -0.15                                        # the jaw is stopped by the cube, not by its joint limit
    # lift: the cube follows the gripper
    d = fresh(q, cube=centre, cquat=cq); ctrl = q.copy(); ctrl[5] = -0.2; O.arr(d.ctrl)[:] = ctrl
    for _ in range(40):
        _step(d, C5, 16)
    z0 = O.arr(d.qpos)[8]
    ctrl[1] -= 0.25; O.arr(d.ctrl)[:] = ctrl                 # raise the shoulder
    for _ in range(40):
        _step(d, C5, 16)
    assert O.arr(d.qpos)[8] > z0 + 0.02


# ---- link proxies (stand-in capsules for the arm's collision meshes, SO100_F_LINKS_FLOOR) -------------------------------------
LINKS = O.F_REFERENCE | O.F_LINKS_FLOOR


def proxy_bottoms(d):
    """world z of the lowest point of the two end spheres of every link proxy (after so100o_kinematics)"""
    xp = O.arr(d.xpos); xm = O.arr(d.xmat); out = []
    for k in range(O.NPROX):
        b = M.prox_body[k]; R = xm[b].reshape(3, 3)
        for e in range(2):
            out.append((xp[b] + R @ np.array(M.prox_p[k][e][:]))[2] - M.prox_radius[k])
    return np.array(out)


def test_link_proxy_geometry_follows_its_rule():
    """segments run from a link's joint origin to its child's (links 1-3) or to the far end of the jaw's pads (links 4-5); radii
    from the links' inertia boxes, jaws capped (csrc/so100_model_def.h)"""
    d = fresh(np.array([0.3, -1.2, 1.0, 0.4, 0.2, 0.5])); L.so100o_kinematics(C.byref(M), C.byref(d))
    xp = O.arr(d.xpos); xm = O.arr(d.xmat)
    for k in range(3):
        b = M.prox_body[k]; assert b == k + 3
        assert np.allclose(np.array(M.prox_p[k][0][:]), 0)
        assert np.allclose(xp[b] + xm[b].reshape(3, 3) @ np.array(M.prox_p[k][1][:]), xp[b + 1], atol=1e-12)      # ends at the child's joint origin
        assert 0.015 < M.prox_radius[k] < 0.03
    for k in (3, 4):
        far = np.array(M.prox_p[k][1][:])
        pads = [g for g in range(8) if M.pad_body[g] == M.prox_body[k]]
        assert abs(far[1]) >= max(abs(M.pad_pos[g][1]) for g in pads) and M.prox_radius[k] == pytest.approx(0.008)


def _wrist_first_poses(n, seed):
    """arm poses whose lowest point is a LINK proxy (not a finger pad), within 3 mm above the floor"""
    rs = np.random.RandomState(seed); out = []
    while len(out) < n:
        q = LO + (HI - LO)*rs.rand(6)
        d = fresh(q); L.so100o_kinematics(C.byref(M), C.byref(d))
        pb = proxy_bottoms(d)
        zpad = min(c[2] - (np.abs(R[2])*h).sum() for c, R, h in pad_frames(d))
        if 0.0 < pb[:6].min() < 0.003 and zpad > pb[:6].min() + 0.02:
            out.append(q)
    return out


def test_arm_driven_wrist_first_into_the_table_stops_on_its_link_proxies():
    """with the proxies the arm commanded downwards rests with every proxy sphere above -1 mm (a capsule end
    dips < 3 mm during the impact); without them (reference physics: pads only) the same links sink centimetres into the table."""
    sunk = []
    for q0 in _wrist_first_poses(6, 3):
        res = {}
        for flags in (LINKS, REF):
            d = fresh(q0)
            low = 1.0
            for t in range(40):
                O.arr(d.ctrl)[:] = O.arr(d.qpos)[:6]; O.arr(d.ctrl)[1] += 0.075      # shoulder down, Env01's relative servo
                L.so100o_step(C.byref(M), C.byref(d), flags, -1, 16)
                L.so100o_kinematics(C.byref(M), C.byref(d)); low = min(low, proxy_bottoms(d).min())
            res[flags] = (low, proxy_bottoms(d).min(), np.abs(O.arr(d.qvel)[:6]).max(), d.ncon)
            assert np.all(np.isfinite(O.arr(d.qpos)))
        low, final, vmax, ncon = res[LINKS]
        assert low > -0.003 and final > -0.001 and vmax < 0.5
        sunk.append(res[REF][1])
    assert np.median(sunk) < -0.005 and min(sunk) < -0.01     # without the proxies the same links end up to centimetres under the table


def test_link_proxy_rows_satisfy_kkt_and_newton_equals_pgs():
    """the proxy contacts are ordinary pyramidal rows on links 1-5: the Newton solution satisfies the KKT conditions row by row"""
    for q0 in _wrist_first_poses(4, 5):
        q = q0.copy(); q[1] += 0.02                          # push the pose a little into the table
        d = fresh(q, v=np.random.RandomState(1).randn(6)*0.3)
        fwd(d, LINKS, -1)
        assert any(d.con[i].kind == 3 for i in range(d.ncon))
        n = d.nefc
        J = O.arr(d.efc_J)[:n]; f = O.arr(d.efc_force)[:n]; aref = O.arr(d.efc_aref)[:n]; Rr = O.arr(d.efc_R)[:n]; ty = np.ctypeslib.as_array(d.efc_type)[:n]
        jar = J @ O.arr(d.qacc) - aref
        for r in range(n):
            if ty[r] != 0:                                    # one-sided rows: f >= 0, f = max(0, -jar / R)
                assert f[r] >= -1e-12 and abs(f[r] - max(0.0, -jar[r]/Rr[r])) < 1e-6*(1 + abs(f[r]))
        Mq = O.arr(d.M).reshape(12, 12)
        assert np.allclose(Mq[:6, :6] @ (O.arr(d.qacc) - O.arr(d.qacc_smooth))[:6], (J.T @ f)[:6], atol=1e-8)


# ---- link proxies against the cube (SO100_F_LINKS_CUBE: Rotation_Pitch / Upper_Arm, the pairs scene:44-48 leaves live) ---
LCUBE = O.F_REFERENCE | O.F_LINKS_FLOOR | O.F_LINKS_CUBE


def capsule_box(a, b, r, c, R, h):
    p = lambda x: np.ascontiguousarray(x, np.float64).ctypes.data_as(C.c_void_p)
    pos = np.zeros(3); n = np.zeros(3); dist = C.c_double(0)
    L.so100o_capsule_box.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    k = L.so100o_capsule_box(p(a), p(b), float(r), p(c), p(R), p(h), p(pos), p(n), C.byref(dist))
    return k, pos, n, dist.value


def _segment_box_distance(a, b, c, R, h, samples=4001):
    t = np.linspace(0, 1, samples)[:, None]
    s = ((a + t*(b - a)) - c) @ R                            # box frame (R: world <- box, row-major)
    e = s - np.clip(s, -h, h)
    return np.sqrt((e*e).sum(1)).min()


def test_capsule_box_known_configurations():
    I = np.eye(3); h = np.full(3, 0.01); r = 0.02
    # segment parallel to the top face, 1 mm of penetration: one contact, normal capsule -> box = straight down, midway between the surfaces
    k, pos, n, dist = capsule_box([-0.05, 0.002, 0.029], [0.05, 0.002, 0.029], r, [0, 0, 0], I, h)
    assert k == 1 and dist == pytest.approx(-0.001, abs=1e-12) and np.allclose(n, [0, 0, -1]) and pos[2] == pytest.approx(0.0095, abs=1e-12)
    assert abs(pos[0]) <= 0.01 + 1e-12 and pos[1] == pytest.approx(0.002)
    # end-on: the segment points at the +x face from outside
    k, pos, n, dist = capsule_box([0.0295, 0, 0], [0.2, 0, 0], r, [0, 0, 0], I, h)
    assert k == 1 and dist == pytest.approx(-0.0005, abs=1e-12) and np.allclose(n, [-1, 0, 0])
    # separated
    assert capsule_box([0.031, 0, 0], [0.2, 0, 0], r, [0, 0, 0], I, h)[0] == 0
    # the axis passes through the box: leaves through the nearest face (here +z), depth = face distance + radius
    k, pos, n, dist = capsule_box([-0.05, 0, 0.006], [0.05, 0, 0.006], r, [0, 0, 0], I, h)
    assert k == 1 and np.allclose(n, [0, 0, -1]) and dist == pytest.approx(-(0.004 + r), abs=1e-12)


def test_capsule_box_fuzz_invariants():
    """unit normal; rigid-motion equivariance; the distance equals the true (densely sampled) segment-box distance minus the radius"""
    rs = np.random.RandomState(4); hits = 0; worst = 0.0
    for _ in range(3000):
        R, _ = rot(rs); c = rs.randn(3)*0.1; h = np.full(3, 0.01); r = 0.02 + 0.004*rs.rand()
        u = rs.randn(3); u /= np.linalg.norm(u)
        mid = c + u*(0.01 + r)*(0.6 + 0.9*rs.rand()); w = rs.randn(3); w /= np.linalg.norm(w)
        ln = 0.1 + 0.02*rs.rand(); t0 = rs.rand()
        a = mid - w*ln*t0; b = mid + w*ln*(1 - t0)
        k, pos, n, dist = capsule_box(a, b, r, c, R, h)
        true = _segment_box_distance(a, b, c, R, h) - r
        if k == 0:
            assert true > -2e-4
            continue
        hits += 1
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and dist <= 0
        if true > -r + 1e-4:                                  # (axis outside the box)
            assert dist >= true - 1e-6                        # (the sampled minimum is itself only good to ~1e-7)
            worst = max(worst, dist - true)
        Q, _ = rot(rs); t = rs.randn(3)
        k2, pos2, n2, dist2 = capsule_box(Q @ a + t, Q @ b + t, r, Q @ c + t, Q @ R, h)
        assert k2 == 1 and np.allclose(pos2, Q @ pos + t, atol=1e-10) and np.allclose(n2, Q @ n, atol=1e-10) and dist2 == pytest.approx(dist, abs=1e-10)
    assert hits > 1000
    assert worst < 1e-6, worst                                # the bracketed Newton lands on the minimiser (the sampled reference is good to ~1e-7)


def test_cube_proxy_geometry_follows_its_rule():
    d = fresh(np.array([0.3, -1.2, 1.0, 0.4, 0.2, 0.5])); L.so100o_kinematics(C.byref(M), C.byref(d))
    xp = O.arr(d.xpos); xm = O.arr(d.xmat)
    for k in range(O.NCPROX):
        b = M.cprox_body[k]; assert b == k + 2               # Rotation_Pitch, Upper_Arm
        assert np.allclose(np.array(M.cprox_p[k][0][:]), 0)
        assert np.allclose(xp[b] + xm[b].reshape(3, 3) @ np.array(M.cprox_p[k][1][:]), xp[b + 1], atol=1e-12)
        assert 0.015 < M.cprox_radius[k] < 0.03
    assert M.cprox_radius[1] == M.prox_radius[0]             # the Upper_Arm capsule is the one the floor pairs use


def link_cube_states(n, seed, pen=(0.0002, 0.003)):
    """arm poses with the cube placed against the Rotation_Pitch (even i) or Upper_Arm (odd i) capsule, penetrating by pen[0]..pen[1] metres"""
    rs = np.random.RandomState(seed); out = []
    hs = np.full(3, 0.01)
    while len(out) < n:
        k = len(out) % 2
        q = LO + (HI - LO)*rs.rand(6)
        d = fresh(q); L.so100o_kinematics(C.byref(M), C.byref(d))
        xp = O.arr(d.xpos); b = M.cprox_body[k]
        a_, b_ = xp[b].copy(), xp[b + 1].copy()
        s = a_ + (0.15 + 0.85*rs.rand())*(b_ - a_)
        u = rs.randn(3); u /= np.linalg.norm(u)
        Rc, qc = rot(rs)
        want = -(pen[0] + (pen[1] - pen[0])*rs.rand())
        lo_, hi_ = 0.0, 0.08                                 # bisection on the cube's offset along u for the wanted penetration
        for _ in range(40):
            mid = 0.5*(lo_ + hi_)
            kk, _, _, dist = capsule_box(a_, b_, M.cprox_radius[k], s + u*mid, Rc, hs)
            if kk and dist < want: lo_ = mid
            else: hi_ = mid
        c = s + u*lo_
        kk, _, _, dist = capsule_box(a_, b_, M.cprox_radius[k], c, Rc, hs)
        if not kk or abs(dist - want) > 2e-4 or c[2] < 0.012 + 0.0174:      # (cube clear of the floor: the pair under test alone)
            continue
        # no other proxy / pad may touch anything in this pose
        dd = fresh(q); O.arr(dd.qpos)[6:9] = c; O.arr(dd.qpos)[9:13] = qc
        fwd(dd, LCUBE | O.F_PADS_CUBE, -1)
        if dd.ncon != 1 or dd.con[0].kind != 4:
            continue
        out.append((q, c, qc))
    return out


def test_link_cube_contact_pushes_the_cube_away_and_satisfies_kkt():
    """with SO100_F_LINKS_CUBE a cube that overlaps the Upper_Arm / Rotation_Pitch capsule is pushed out along the contact normal and the
    rows satisfy the KKT conditions; without the flag (F_REFERENCE_LINKS) the same cube falls freely through the link"""
    for q, c, qc in link_cube_states(6, 2):
        d = fresh(q); O.arr(d.qpos)[6:9] = c; O.arr(d.qpos)[9:13] = qc
        fwd(d, LCUBE, -1)
        assert d.ncon == 1 and d.con[0].kind == 4 and d.con[0].feat in (160, 161) and d.con[0].b2 == 8
        nrm = np.array(d.con[0].frame[:3])
        n = d.nefc
        J = O.arr(d.efc_J)[:n]; f = O.arr(d.efc_force)[:n]; aref = O.arr(d.efc_aref)[:n]; Rr = O.arr(d.efc_R)[:n]; ty = np.ctypeslib.as_array(d.efc_type)[:n]
        jar = J @ O.arr(d.qacc) - aref
        for r in range(n):
            if ty[r] != 0:
                assert f[r] >= -1e-12 and abs(f[r] - max(0.0, -jar[r]/Rr[r])) < 1e-6*(1 + abs(f[r]))
        Mq = O.arr(d.M).reshape(12, 12)
        assert np.allclose(Mq @ (O.arr(d.qacc) - O.arr(d.qacc_smooth)), J.T @ f, atol=1e-7)
        assert f[ty == 2].sum() > 0                           # the pair carries force
        acc_cube = O.arr(d.qacc)[6:9] - np.array([0, 0, -9.81])
        assert acc_cube @ nrm > 1.0                           # ... that pushes the cube along the normal (capsule -> cube)
        # without the flag: free fall
        d2 = fresh(q); O.arr(d2.qpos)[6:9] = c; O.arr(d2.qpos)[9:13] = qc
        fwd(d2, LINKS, -1)
        assert d2.ncon == 0 and np.allclose(O.arr(d2.qacc)[6:9], [0, 0, -9.81], atol=1e-9)
        # 20 substeps on: the penetration has not grown, nothing blew up
        O.arr(d.ctrl)[:] = q
        L.so100o_step(C.byref(M), C.byref(d), LCUBE, -1, 20)
        assert np.all(np.isfinite(O.arr(d.qpos))) and np.abs(O.arr(d.qvel)[:9]).max() < 2.0      # (the 8 g cube may spin: an off-centre push with friction)
        L.so100o_forward(C.byref(M), C.byref(d), LCUBE, -1)
        assert d.ncon == 0 or d.con[0].dist > -0.004
