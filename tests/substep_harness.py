"""Per-substep contact parity harness (shared by the CPU host-instantiation test and the GPU test).

Every env follows the ORACLE's trajectory: before each substep the oracle's state is rounded to fp32 and handed to the device
code (host instantiation, or the HIP kernels created with frame_skip = 1), both sides take ONE physics substep from that same
state, and EVERY env is compared in EVERY substep -- no survivor filter decides what is looked at:
  * the same number of contact records and the same set of contact features (pad corner / manifold slot ids: the oracle's
    so100o_contact.feat against the device's contact_sig state row),
  * the acceleration the substep applied, h * qacc = v_after - v_before, within the stated fp32 bound,
  * the solver residual.
A pose whose contact set is decided inside fp32 round-off (a corner within ~1e-7 m of the plane, two separating axes tied) is
classified as `knife_edge` -- by re-running the ORACLE on the state nudged by a few fp32 ulps, for EVERY pose, never by looking at
the device's answer -- and counted; on all other poses count and set must match exactly, and the caller bounds the knife-edge share.
"parity unpinned (physics)": the oracle restates MuJoCo's algorithm (oracle/so100_oracle.c), MuJoCo itself is not available."""
import ctypes as C

import numpy as np

from oracle import so100_oracle as O

L = O.lib()
M = O.model()
JS = np.float32(0.075)


def feature_mix(fid):
    """csrc/so100_contact.hpp: contact_id_mix"""
    h = ((fid + 1)*0x9E3779B1) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h*0x85EBCA77) & 0xFFFFFFFF
    h ^= h >> 13
    return h


def oracle_contact_summary(d):
    """(records the device keeps, signature of the pad-contact set, pad contacts, coupled) of the oracle's last forward pass"""
    pads = [d.con[i] for i in range(d.ncon) if d.con[i].kind != 0]
    coupled = any(c.kind in (2, 4) for c in pads)                # pad/cube or link proxy/cube: arm and cube solved together
    n = len(pads) + (sum(1 for i in range(d.ncon) if d.con[i].kind == 0) if coupled else 0)
    sig = sum(feature_mix(c.feat) for c in pads) & 0xFFFFFFFF
    return n, (sig - (1 << 32) if sig >= (1 << 31) else sig), len(pads), coupled


def round_state_to_fp32(d):
    q = O.arr(d.qpos); v = O.arr(d.qvel)
    q[:] = q.astype(np.float32).astype(np.float64); v[:] = v.astype(np.float32).astype(np.float64)


def oracle_substep(d, act, flags):
    """ctrl of Env01 (env01_v1.py:18-24: measured angle + 0.075 a), one mj_step with the oracle's Newton"""
    O.arr(d.ctrl)[:] = (O.arr(d.qpos)[:6].astype(np.float32) + act.astype(np.float32)*JS).astype(np.float64)
    L.so100o_step(C.byref(M), C.byref(d), flags, -1, 1)


def knife_edge(d0, flags, ref, rs, probes=8, ulps=8):
    """is the contact SET of this pose decided inside fp32 round-off?  The oracle's own forward pass is repeated on the state
    nudged by up to `ulps` fp32 ulps per coordinate (what fp32 kinematics loses along the 6-link chain: ~1e-7 m at the pads) in
    `probes` random directions; a pose is a knife edge if any of them changes the oracle's (count, feature set).
    d0: the pre-step oracle state (ctrl already set).  Independent of the device's answer."""
    q0 = O.arr(d0.qpos).astype(np.float32)
    for _ in range(probes):
        d = copy_data(d0)
        k = rs.randint(-ulps, ulps + 1, size=q0.shape)
        q = q0.copy()
        for _ in range(ulps):
            step = np.sign(k).astype(np.float32); k = k - np.sign(k)
            q = np.where(step > 0, np.nextafter(q, np.float32(np.inf)), np.where(step < 0, np.nextafter(q, np.float32(-np.inf)), q))
        O.arr(d.qpos)[:] = q.astype(np.float64)
        L.so100o_forward(C.byref(M), C.byref(d), flags, -1)
        if oracle_contact_summary(d)[:2] != ref[:2]:
            return True
    return False


def copy_data(d):
    c = O.Data(); C.memmove(C.byref(c), C.byref(d), C.sizeof(O.Data))
    return c


class Tally:
    """what happened to every (env, substep) pair; printed so the test log carries the class counts"""
    def __init__(self):
        self.pairs = self.contact = self.coupled = self.knife = self.set_mismatch = self.count_mismatch = 0
        self.worst_dv = self.worst_dv_contact = self.worst_res = 0.0
        self.worst_rel = 0.0

    def line(self, name):
        return (f"[{name}] env-substeps {self.pairs}  in pad contact {self.contact}  coupled {self.coupled}  knife-edge poses {self.knife}  "
                f"count mismatches {self.count_mismatch}  set mismatches {self.set_mismatch}  worst |d(h qacc)| free {self.worst_dv:.2e} contact {self.worst_dv_contact:.2e}  "
                f"worst relative {self.worst_rel:.2e}  worst residual {self.worst_res:.2e}")


# ---- injected contact states (the same generators the 16-substep GPU tests use) ------------------------------------------------
def floor_batch(n, seed, band=0.002):
    """arm poses with the lowest pad corner within `band` of the floor, moderate joint velocities, cube resting on the floor"""
    from test_oracle_contacts import floor_poses
    rs = np.random.RandomState(seed)
    poses = floor_poses(n, seed + 100, band=band)
    qpos = np.zeros((n, 13)); qvel = np.zeros((n, 12))
    for i, q in enumerate(poses):
        qpos[i, :6] = q; qpos[i, 6:9] = [0.15 + 0.02*rs.randn(), -0.25, 0.0099]; qpos[i, 9] = 1.0
        qvel[i, :6] = rs.randn(6)*0.3
    act = rs.uniform(-1, 1, (n, 6)).astype(np.float32)
    return qpos, qvel, act


def grasp_batch(n, seed):
    """the jaw closing on a cube that floats between the pads (BASELINE.json configs[4]): generic small cube rotations"""
    from test_oracle_contacts import _grasp_state
    rs = np.random.RandomState(seed)
    q, centre, cq = _grasp_state()
    qpos = np.zeros((n, 13)); qvel = np.zeros((n, 12))
    qpos[:, :6] = q; qpos[:, 5] = 0.065 + rs.uniform(0.0, 0.01, n)
    qpos[:, 6:9] = centre + rs.uniform(-1, 1, (n, 3))*np.array([0.0004, 0.002, 0.002])
    for i in range(n):
        w = rs.randn(3)*0.03; ang = np.linalg.norm(w); ax = w/ang
        dq = np.array([np.cos(ang/2), *(np.sin(ang/2)*ax)])
        a, b = cq, dq
        qpos[i, 9:13] = [a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3], a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
                         a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1], a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]]
    act = np.zeros((n, 6), np.float32); act[:, 5] = -1.0
    return qpos, qvel, act


def oracle_states(qpos, qvel):
    from test_oracle_contacts import fresh
    ds = []
    for i in range(len(qpos)):
        d = fresh(); O.arr(d.qpos)[:] = qpos[i]; O.arr(d.qvel)[:] = qvel[i]
        ds.append(d)
    return ds


def run_substep_parity(device_substep, qpos, qvel, act, flags, nsub, name, seed=0):
    """device_substep(q32 [n,13], v32 [n,12], act [n,6]) -> (qpos [n,13], qvel [n,12], count [n], sig [n], residual [n]) after ONE
    substep from exactly that state.  Returns the Tally; raises on the first non-knife-edge contact-set mismatch."""
    n = len(qpos)
    ds = oracle_states(qpos, qvel)
    rs = np.random.RandomState(seed)
    T = Tally()
    h = M.timestep
    for s in range(nsub):
        for d in ds:
            round_state_to_fp32(d)
        q32 = np.stack([O.arr(d.qpos).copy() for d in ds]); v32 = np.stack([O.arr(d.qvel).copy() for d in ds])
        gq, gv, gcount, gsig, gres = device_substep(q32, v32, act)
        assert np.isfinite(gq).all() and np.isfinite(gv).all(), (name, s)
        for i, d in enumerate(ds):
            O.arr(d.ctrl)[:] = (q32[i, :6].astype(np.float32) + act[i].astype(np.float32)*JS).astype(np.float64)
            d0 = copy_data(d)
            L.so100o_step(C.byref(M), C.byref(d), flags, -1, 1)
            ref = oracle_contact_summary(d)
            T.pairs += 1; T.contact += ref[2] > 0; T.coupled += bool(ref[3])
            near = ref[2] > 0 or gcount[i] > 0
            ke = near and knife_edge(d0, flags, ref, rs)
            T.knife += bool(ke)
            if not ke:
                if gcount[i] != ref[0]:
                    T.count_mismatch += 1
                elif gsig[i] != ref[1]:
                    T.set_mismatch += 1
                assert gcount[i] == ref[0] and gsig[i] == ref[1], (name, "substep", s, "env", i, "device", int(gcount[i]), int(gsig[i]), "oracle", ref)
                # h * qacc of the arm (and of the cube's translation): what the substep did to the velocities
                dvo = np.concatenate([O.arr(d.qvel)[:6] - v32[i, :6], O.arr(d.qvel)[6:9] - v32[i, 6:9]])
                dvg = np.concatenate([gv[i, :6] - v32[i, :6], gv[i, 6:9] - v32[i, 6:9]])
                err = np.abs(dvg - dvo).max(); scale = np.abs(dvo).max()
                if ref[2] > 0:
                    T.worst_dv_contact = max(T.worst_dv_contact, err); T.worst_rel = max(T.worst_rel, err/(1e-3 + scale))
                else:
                    T.worst_dv = max(T.worst_dv, err)
                T.worst_res = max(T.worst_res, float(gres[i]))
    print(T.line(name))
    return T
