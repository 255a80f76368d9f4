"""The DEVICE physics / task headers (csrc/so100_physics.hpp, so100_cube.hpp, so100_task.hpp) instantiated on the host
(tests/_hostcheck, fp64 and fp32) against the oracle.  This is how the kernel's formulation -- link-frame RNEA + CRBA,
per-joint block PGS, Newton for the cube/floor block, the fp32 task layer -- is checked without a GPU.  CPU only.

The two sides share NO code: the oracle is MuJoCo-structured (world-frame spatial algebra, dense rows, scalar PGS to
convergence); the device headers use a different formulation throughout.  Agreement to 1e-14 in fp64 is the evidence
that both are right; the fp32 numbers below are the device's expected round-off."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

from oracle import so100_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def H():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "_hostcheck"), "-s"])
    h = C.CDLL(os.path.join(HERE, "_hostcheck", "libhostcheck.so"))
    h.hc_env_new.restype = C.c_void_p
    h.hc_env_free.argtypes = [C.c_void_p]
    h.hc_env_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    h.hc_env_step.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
    h.hc_env_qpos.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return h


L = O.lib(); M = O.model()
RNG = np.array(M.jnt_range)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def fresh(q6=None, v6=None):
    d = O.Data(); L.so100o_reset_data(C.byref(M), C.byref(d))
    if q6 is not None: O.arr(d.qpos)[:6] = q6
    if v6 is not None: O.arr(d.qvel)[:6] = v6
    return d


def test_sincos_fp32(H):
    xs = np.linspace(-7.5, 7.5, 200001).astype(np.float32)
    s = C.c_float(); c = C.c_float(); worst = 0
    for x in xs[::97]:
        H.hc_sincos_f(C.c_float(float(x)), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert worst < 1.5e-7


def test_dynamics_and_poses_match_oracle(H):
    rs = np.random.RandomState(0)
    w = dict(M=0, b=0, Mf=0, bf=0, pose=0, posef=0)
    for _ in range(100):
        q = rs.uniform(RNG[:, 0], RNG[:, 1]); v = rs.uniform(-4, 4, 6)
        d = fresh(q, v); L.so100o_forward(C.byref(M), C.byref(d), 0, 0)
        Mo = O.arr(d.M).reshape(12, 12)[:6, :6]; bo = O.arr(d.qfrc_bias)[:6]
        Mh = np.zeros(36); bh = np.zeros(6)
        H.hc_dyn_d(p(q), p(v), p(Mh), p(bh))
        w["M"] = max(w["M"], np.abs(Mh.reshape(6, 6) - Mo).max()); w["b"] = max(w["b"], np.abs(bh - bo).max())
        H.hc_dyn_f(p(q), p(v), p(Mh), p(bh))
        w["Mf"] = max(w["Mf"], np.abs(Mh.reshape(6, 6) - Mo).max()); w["bf"] = max(w["bf"], np.abs(bh - bo).max())
        out = np.zeros(27)
        ref = np.r_[O.arr(d.xpos)[5], O.arr(d.xpos)[6], O.arr(d.xmat)[6], O.arr(d.cam_xpos), O.arr(d.cam_xmat)]
        H.hc_poses_d(p(q), p(out)); w["pose"] = max(w["pose"], np.abs(out - ref).max())
        H.hc_poses_f(p(q), p(out)); w["posef"] = max(w["posef"], np.abs(out - ref).max())
    assert w["M"] < 1e-15 and w["b"] < 1e-13 and w["pose"] < 1e-14          # fp64: same mechanics, different algorithm
    assert w["Mf"] < 1e-7 and w["bf"] < 2e-6 and w["posef"] < 2e-6          # fp32 round-off


def _arm_traj(H, fn, iters_o, iters_h, flags, nenv, nstep, seed):
    rs = np.random.RandomState(seed); wq = wv = 0
    for _ in range(nenv):
        q = rs.uniform(RNG[:, 0] - 0.02, RNG[:, 1] + 0.02); v = rs.uniform(-2, 2, 6)
        d = fresh(q, v); qh = q.copy(); vh = v.copy(); ff = np.zeros(6); fl = np.zeros(6)
        for _ in range(nstep):
            ctrl = O.arr(d.qpos)[:6] + rs.uniform(-1, 1, 6) * 0.075
            O.arr(d.ctrl)[:] = ctrl
            L.so100o_step(C.byref(M), C.byref(d), flags, iters_o, 16)
            fn(p(qh), p(vh), p(ctrl.copy()), p(ff), p(fl), flags, iters_h, 16)
            wq = max(wq, np.abs(qh - O.arr(d.qpos)[:6]).max()); wv = max(wv, np.abs(vh - O.arr(d.qvel)[:6]).max())
    return wq, wv


ARM = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_CUBE_PINNED


def test_block_pgs_reaches_the_oracle_optimum(H):
    """Per-joint block PGS (device) vs scalar PGS to convergence (oracle): same optimum, 3 sweeps reach fp32 level."""
    e2 = _arm_traj(H, H.hc_sub_d, 0, 2, ARM, 3, 60, 1)
    e3 = _arm_traj(H, H.hc_sub_d, 0, 3, ARM, 3, 60, 1)
    e6 = _arm_traj(H, H.hc_sub_d, 0, 6, ARM, 3, 60, 1)
    assert e6[0] < 1e-13 and e6[1] < 1e-12
    assert e3[0] < 1e-8 and e3[1] < 1e-6
    assert e2[0] < 1e-6


def test_arm_fp32_drift_is_within_the_stated_tolerance(H):
    """fp32 device arithmetic vs fp64 oracle over 100 env steps = 1600 substeps: the north star's 1e-5 relative."""
    wq, wv = _arm_traj(H, H.hc_sub_f, 0, 3, ARM, 6, 100, 2)
    assert wq < 1e-5 and wv < 1e-4
    wq, wv = _arm_traj(H, H.hc_sub_f, 0, 0, O.F_CUBE_PINNED, 6, 100, 3)
    assert wq < 1e-5 and wv < 1e-4


def _cube_traj(H, fn, iters_h, tilt, seed, nsub=400):
    rs = np.random.RandomState(seed); w = wv = 0
    for _ in range(4):
        d = fresh()
        pos = np.array([rs.uniform(-.3, .3), rs.uniform(-.3, .3), rs.uniform(0.0, 0.02)]); quat = np.array([1., 0, 0, 0])
        if tilt:
            x = R.from_rotvec(rs.uniform(-0.3, 0.3, 3)).as_quat(); quat = np.array([x[3], x[0], x[1], x[2]])
        vel = np.r_[rs.uniform(-.1, .1, 3), rs.uniform(-1, 1, 3) * tilt]
        O.arr(d.qpos)[6:9] = pos; O.arr(d.qpos)[9:13] = quat; O.arr(d.qvel)[6:] = vel
        ph = pos.copy(); qh = quat.copy(); vh = vel.copy(); wh = np.zeros(16); ap = np.zeros(3)
        for _ in range(nsub // 8):
            L.so100o_step(C.byref(M), C.byref(d), O.F_FLOOR, 0, 8)
            fn(p(ph), p(qh), p(vh), p(wh), p(ap), O.F_FLOOR, iters_h, 8)
            w = max(w, np.abs(ph - O.arr(d.qpos)[6:9]).max(), np.abs(qh - O.arr(d.qpos)[9:13]).max())
            wv = max(wv, np.abs(vh - O.arr(d.qvel)[6:]).max())
    return w, wv


def test_cube_newton_reaches_the_oracle_optimum(H):
    """Primal Newton + exact line search (device) vs dual PGS to convergence (oracle) on the 16 pyramid rows."""
    w, wv = _cube_traj(H, H.hc_cube_d, 6, 0, 5)
    assert w < 1e-12 and wv < 1e-10                              # flat cube incl. impacts from random heights / velocities: <= 6 iterations
    w, wv = _cube_traj(H, H.hc_cube_d, 8, 1, 6)
    assert w < 1e-11 and wv < 1e-9                               # tumbling cube
    w, wv = _cube_traj(H, H.hc_cube_f, 6, 0, 7)
    assert w < 5e-5 and wv < 2e-3                                # fp32 (bouncing transient included)


@pytest.mark.parametrize("kind,flags", [(1, O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR), (2, ARM), (5, O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR),
                                        (3, ARM), (4, O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR), (6, O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR)])
def test_task_layer_fp32_vs_oracle(H, kind, flags):
    """so100_task.hpp (what the kernel runs per lane) on the host in fp32 vs the oracle's C task layer."""
    rs = np.random.RandomState(40 + kind)
    reach = kind in (1, 2, 6)
    od = 15 if reach else 8
    for env in range(3):
        e = O.OracleEnv(kind, flags=flags, iters=0); e.e.max_episode_steps = 25
        h = H.hc_env_new(kind)
        inj = rs.random_sample(16).astype(np.float32)
        oo = e.reset(inject=inj); oh = np.zeros(od, np.float32); H.hc_env_reset(h, kind, p(inj), p(oh))
        np.testing.assert_allclose(oh, oo, rtol=0, atol=1e-6)
        for t in range(60):
            a = np.clip(rs.uniform(-1, 1, 6) * 0.7, -1, 1).astype(np.float32); inj = rs.random_sample(16).astype(np.float32)
            oo, ro, to, tro, tobo = e.step(a, inject=inj, autoreset=True)
            oh = np.zeros(od, np.float32); th = np.zeros(od, np.float32); rh = C.c_float(); dh = C.c_int(); trh = C.c_int()
            H.hc_env_step(h, kind, flags, 4, 6, 25, p(a), p(inj), p(oh), p(th), C.byref(rh), C.byref(dh), C.byref(trh))
            tol = 2e-5 if reach else 6e-3
            np.testing.assert_allclose(oh[:6], oo[:6], rtol=0, atol=1e-5 if reach else 2e-6)
            np.testing.assert_allclose(oh, oo, rtol=0, atol=tol, err_msg=f"kind {kind} step {t}")
            assert abs(rh.value - ro) < (1e-4 if kind <= 2 else 2e-3 if kind == 6 else 1.2e-2)
            assert bool(dh.value) == (to or tro) and bool(trh.value) == (tro and not to)
            if to or tro:
                np.testing.assert_allclose(th, tobo, rtol=0, atol=tol)
        H.hc_env_free(h)
