"""Per-substep contact parity of the HIP kernels (through the C ABI, handle created with frame_skip = 1) against the fp64
oracle: EVERY env in EVERY substep -- same contact count, same contact features (contact_sig state row vs the oracle's
so100o_contact.feat), h * qacc within the stated fp32 bound, solver residual (tests/substep_harness.py; the host-instantiation
twin of this test is tests/test_substep_parity.py).  All four step kernels are covered through the batch size: 4-wave latency
kernel with 16 / 32 / 64 envs per workgroup (4 / 2 / 1 cooperating contact lanes per env) and the one-wave throughput kernel; plus
a flag set WITHOUT a compile-time instantiation (run-time-flags kernels so100_step_mw / so100_step_fused<K, -1>).

Stated fp32 bound on h * qacc from identical fp32-rounded states (h = 2 ms): 2e-6 rad/s without pad contact, 5e-5 rad/s (m/s for
the cube) and 1e-2 relative with it (stiff pad rows: condition ~1e5).  "parity unpinned (physics)": MuJoCo is not available."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import substep_harness as SH                              # noqa: E402
from oracle import so100_oracle as O                      # noqa: E402  (the checker)

REFP = O.F_REFERENCE
C5 = O.F_CONTACT5
UNINSTANTIATED = O.F_FRICTIONLOSS | O.F_FLOOR | O.F_PADS_FLOOR      # 21: no limits -> KindOps::step falls through to <K, -1>


class HipDevice:
    def __init__(self, n, m, flags):
        from so100_mujoco_rl_amd.lib import So100Sim
        self.sim = So100Sim(1, n, flags=flags, solver_iters=4, contact_iters=30, frame_skip=1, max_episode_steps=0, seed=3)
        self.n, self.m = n, m
        self.sim.reset()
        QP = np.zeros((n, 13)); QP[:, 9] = 1.0; QP[:, 6:9] = [0.2, -0.2, 0.0099]; QP[:, :6] = [0, -1.5, 1.5, 0.5, 0, 0.2]
        self.QP = QP; self.QV = np.zeros((n, 12)); self.A = np.zeros((n, 6), np.float32)

    def __call__(self, q32, v32, act):
        m, sim = self.m, self.sim
        self.QP[:m] = q32; self.QV[:m] = v32; self.A[:m] = act
        sim.set_state(torch.from_numpy(np.ascontiguousarray(self.QP.T, np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(self.QV.T, np.float32)).cuda())
        sim.step(torch.from_numpy(self.A).cuda())
        gq, gv = sim.get_state()
        cs = sim.get_field("contact_stat", dtype=torch.int32).cpu().numpy()[:m]
        assert (cs >> 8).max() == 0                       # nothing over the contact budget
        return (gq.cpu().numpy().T[:m].astype(np.float64), gv.cpu().numpy().T[:m].astype(np.float64), cs & 255,
                sim.get_field("contact_sig", dtype=torch.int32).cpu().numpy()[:m], sim.get_field("solver_residual").cpu().numpy()[:m])


def _check(T, min_contact, min_coupled=0):
    assert T.contact >= min_contact and T.coupled >= min_coupled
    assert T.knife <= 0.02*T.pairs
    assert T.count_mismatch == 0 and T.set_mismatch == 0
    assert T.worst_dv < 2e-6
    assert T.worst_dv_contact < 5e-5 and T.worst_rel < 1e-2
    assert T.worst_res < 1e-2


@pytest.mark.parametrize("n,flags", [(96, REFP), (8192, REFP), (16384, REFP), (16384 + 96, REFP), (96, UNINSTANTIATED), (16384 + 96, UNINSTANTIATED)])
def test_pad_floor_per_substep(n, flags):
    m, nsub = 96, 24
    qpos, qvel, act = SH.floor_batch(m, 0)
    T = SH.run_substep_parity(HipDevice(n, m, flags), qpos, qvel, act, flags, nsub, f"HIP n={n} flags={flags} pad/floor")
    _check(T, m*nsub//3)


@pytest.mark.parametrize("n", [48, 8192, 16384 + 48])
def test_link_proxies_per_substep(n):
    """SO100_F_LINKS_FLOOR (capsule proxies of the arm's collision meshes: contacts on ANY link, the general form of the solver), through
    the run-time-flags kernels: poses that reach the table wrist / forearm first"""
    from test_substep_parity import wrist_first_batch, LINKS
    m, nsub = 48, 24
    qpos, qvel, act = wrist_first_batch(m, 0)
    T = SH.run_substep_parity(HipDevice(n, m, LINKS), qpos, qvel, act, LINKS, nsub, f"HIP n={n} link proxies")
    _check(T, m*nsub//3)


@pytest.mark.parametrize("n", [64, 8192, 16384, 16384 + 64])
def test_pad_cube_grasp_per_substep(n):
    m, nsub = 64, 40
    qpos, qvel, act = SH.grasp_batch(m, 1)
    T = SH.run_substep_parity(HipDevice(n, m, C5), qpos, qvel, act, C5, nsub, f"HIP n={n} grasp")
    _check(T, m*nsub//3, m*nsub//4)


@pytest.mark.parametrize("n", [32, 8192, 16384 + 32])
def test_link_cube_per_substep(n):
    """SO100_F_LINKS_CUBE (Rotation_Pitch / Upper_Arm capsules vs the cube: the pairs the reference scene leaves live) through the
    run-time-flags kernels -- 4 / 2 / 1 contact lanes per env and the one-wave kernel: the cube placed against either capsule, arm and cube
    solved together (12 unknowns, records on links 0 / 1); then the closing-jaw grasp with every proxy pair switched on as well"""
    from test_substep_parity import link_cube_batch, LCUBE
    m, nsub = 32, 12
    qpos, qvel, act = link_cube_batch(m, 0)
    T = SH.run_substep_parity(HipDevice(n, m, LCUBE), qpos, qvel, act, LCUBE, nsub, f"HIP n={n} link proxies vs cube")
    _check(T, m*nsub//3, m*nsub//3)
    if n == 32:
        flags = LCUBE | O.F_PADS_CUBE
        qpos, qvel, act = SH.grasp_batch(m, 2)
        T = SH.run_substep_parity(HipDevice(n, m, flags), qpos, qvel, act, flags, 32, f"HIP n={n} all proxies + grasp")
        _check(T, m*32//4, m*32//5)
