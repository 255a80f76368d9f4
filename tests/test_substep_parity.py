"""Per-substep contact parity of the DEVICE code instantiated on the host in fp32 (tests/_hostcheck: the one-wave kernel's
substep_with_pads) against the fp64 oracle -- every env, every substep, no survivor filter (tests/substep_harness.py).  The
same harness drives the HIP kernels through the C ABI in tests/test_gpu_substep_parity.py.

Stated fp32 bound on h * qacc (the velocity change of one substep, h = 2 ms) from IDENTICAL fp32-rounded states:
  * env-substeps without pad contact:   2e-6 rad/s  (dual block PGS, 4 sweeps; measured 5e-7)
  * env-substeps with pad contact:      5e-5 rad/s (m/s for the cube) and 1e-2 of |h qacc| + 1e-3 -- the pad rows are stiff (1/R ~ 3e3
    against M ~ 0.1, condition ~1e5), so fp32 carries 3-4 significant digits of a contact force (measured: 5e-6 pad/floor,
    1.6e-5 in the coupled grasp, relative 3e-3).
"parity unpinned (physics)": the oracle restates MuJoCo's published algorithm; MuJoCo itself is not available."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import substep_harness as SH
from oracle import so100_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def H():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "_hostcheck"), "-s"])
    return C.CDLL(os.path.join(HERE, "_hostcheck", "libhostcheck.so"))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


class HostDevice:
    """one fp32 substep of the device code per env; warm starts (friction / limit forces, previous acceleration, cube block)
    carried from substep to substep like the state rows of the handle"""
    def __init__(self, H, n, flags):
        self.H, self.flags = H, flags
        self.warm = np.zeros((n, 49))

    def __call__(self, q32, v32, act):
        n = len(q32)
        gq = np.zeros((n, 13)); gv = np.zeros((n, 12)); cnt = np.zeros(n, np.int64); sig = np.zeros(n, np.int64); res = np.zeros(n)
        ap = np.zeros(3)
        for i in range(n):
            st = self.warm[i]
            st[:6] = q32[i, :6]; st[6:12] = v32[i, :6]; st[30:33] = q32[i, 6:9]; st[33:37] = q32[i, 9:13]; st[37:43] = v32[i, 6:12]
            ctrl = (q32[i, :6].astype(np.float32) + act[i].astype(np.float32)*SH.JS).astype(np.float64)
            stat = np.zeros(5, np.int32)
            self.H.hc_csub_f(P(st), P(ctrl), P(ap), self.flags, 4, 30, 1, P(stat))
            gq[i, :6] = st[:6]; gq[i, 6:9] = st[30:33]; gq[i, 9:13] = st[33:37]; gv[i, :6] = st[6:12]; gv[i, 6:12] = st[37:43]
            cnt[i] = stat[0]; sig[i] = stat[4]; res[i] = stat[3]*1e-9
        return gq, gv, cnt, sig, res


def _check(T, n_pairs_min_contact, coupled_min=0):
    assert T.contact >= n_pairs_min_contact and T.coupled >= coupled_min          # the batch did exercise the contact path
    assert T.knife <= 0.02*T.pairs                                                 # poses decided inside fp32 round-off are rare
    assert T.count_mismatch == 0 and T.set_mismatch == 0
    assert T.worst_dv < 2e-6
    assert T.worst_dv_contact < 5e-5 and T.worst_rel < 1e-2
    assert T.worst_res < 1e-2


def test_feature_signature_matches_the_oracle_ids(H):
    """the device's id numbering (pad/floor 8 pad + corner; pad/cube 64 + 8 pad + slot) and its mix function, on single substeps"""
    assert SH.feature_mix(0) != SH.feature_mix(1)
    qpos, qvel, act = SH.floor_batch(24, 3)
    dev = HostDevice(H, 24, O.F_REFERENCE)
    T = SH.run_substep_parity(dev, qpos, qvel, act, O.F_REFERENCE, 1, "host fp32, floor, 1 substep")
    assert T.contact >= 10 and T.count_mismatch == 0 and T.set_mismatch == 0


def test_pad_floor_per_substep_host_fp32(H):
    n = 64
    qpos, qvel, act = SH.floor_batch(n, 0)
    T = SH.run_substep_parity(HostDevice(H, n, O.F_REFERENCE), qpos, qvel, act, O.F_REFERENCE, 32, "host fp32, pad/floor")
    _check(T, n_pairs_min_contact=n*32//3)


def test_pad_cube_grasp_per_substep_host_fp32(H):
    n = 32
    qpos, qvel, act = SH.grasp_batch(n, 1)
    T = SH.run_substep_parity(HostDevice(H, n, O.F_CONTACT5), qpos, qvel, act, O.F_CONTACT5, 48, "host fp32, grasp")
    _check(T, n_pairs_min_contact=n*48//3, coupled_min=n*48//4)


# ---- link proxies (SO100_F_LINKS_FLOOR: stand-in capsules for the arm's collision meshes, contacts on ANY link) ---------------
LINKS = O.F_REFERENCE | O.F_LINKS_FLOOR


def wrist_first_batch(n, seed):
    """poses whose lowest point is a link proxy (wrist / forearm first), pushed 1e-2 rad into the table; random joint velocities"""
    from test_oracle_contacts import _wrist_first_poses
    rs = np.random.RandomState(seed)
    qpos = np.zeros((n, 13)); qvel = np.zeros((n, 12))
    for i, q in enumerate(_wrist_first_poses(n, seed + 7)):
        qpos[i, :6] = q; qpos[i, 1] += 0.01; qpos[i, 6:9] = [0.15, -0.25, 0.0099]; qpos[i, 9] = 1.0; qvel[i, :6] = rs.randn(6)*0.3
    act = rs.uniform(-1, 1, (n, 6)).astype(np.float32); act[:, 1] = 0.5
    return qpos, qvel, act


def test_link_proxies_per_substep_host_fp32(H):
    n = 48
    qpos, qvel, act = wrist_first_batch(n, 0)
    T = SH.run_substep_parity(HostDevice(H, n, LINKS), qpos, qvel, act, LINKS, 32, "host fp32, link proxies, wrist first")
    _check(T, n_pairs_min_contact=n*32//3)
    qpos, qvel, act = SH.floor_batch(n, 0)                    # pads AND proxies on the table
    T = SH.run_substep_parity(HostDevice(H, n, LINKS), qpos, qvel, act, LINKS, 24, "host fp32, link proxies + pads")
    _check(T, n_pairs_min_contact=n*24//2)


def test_link_proxies_with_the_coupled_grasp_per_substep_host_fp32(H):
    n, flags = 24, LINKS | O.F_PADS_CUBE
    qpos, qvel, act = SH.grasp_batch(n, 1)
    T = SH.run_substep_parity(HostDevice(H, n, flags), qpos, qvel, act, flags, 40, "host fp32, link proxies + grasp")
    _check(T, n_pairs_min_contact=n*40//4, coupled_min=n*40//5)


# ---- link proxies against the cube (SO100_F_LINKS_CUBE: Rotation_Pitch / Upper_Arm vs block_a) ---------------------------
LCUBE = O.F_REFERENCE | O.F_LINKS_FLOOR | O.F_LINKS_CUBE


def link_cube_batch(n, seed):
    """the cube placed against the Rotation_Pitch / Upper_Arm capsule (alternating), 0.2-3 mm deep, random joint and cube velocities"""
    from test_oracle_contacts import link_cube_states
    rs = np.random.RandomState(seed)
    qpos = np.zeros((n, 13)); qvel = np.zeros((n, 12))
    for i, (q, c, qc) in enumerate(link_cube_states(n, seed + 3)):
        qpos[i, :6] = q; qpos[i, 6:9] = c; qpos[i, 9:13] = qc
        qvel[i, :6] = rs.randn(6)*0.3; qvel[i, 6:9] = rs.randn(3)*0.02
    act = rs.uniform(-1, 1, (n, 6)).astype(np.float32)
    return qpos, qvel, act


def test_link_cube_per_substep_host_fp32(H):
    n = 32
    qpos, qvel, act = link_cube_batch(n, 0)
    T = SH.run_substep_parity(HostDevice(H, n, LCUBE), qpos, qvel, act, LCUBE, 12, "host fp32, link proxies vs cube")
    _check(T, n_pairs_min_contact=n*12//3, coupled_min=n*12//3)
    flags = LCUBE | O.F_PADS_CUBE                            # the closing-jaw grasp with every proxy pair switched on as well
    qpos, qvel, act = SH.grasp_batch(16, 2)
    T = SH.run_substep_parity(HostDevice(H, 16, flags), qpos, qvel, act, flags, 32, "host fp32, all proxies + grasp")
    _check(T, n_pairs_min_contact=16*32//4, coupled_min=16*32//5)
