"""The escape hatch for SURVEY.md A.1's inertia assumption: the build assumes the arm keeps its explicit <inertial>
elements although the scene says inertiafromgeom="true" (env01.xml:2; the attached arm model has its own compiler settings).  If MuJoCo
derives the arm's inertials from its mesh geoms instead, the numbers are known only to who can run MuJoCo -- they go in as a table:
`make -C so100_mujoco_rl_amd/csrc gen INERTIALS=file` (device constants) and so100o_model_init_with_inertials (oracle).  This test runs that
path with a made-up table: a second generated header, the device headers instantiated on it on the host (fp64), against the oracle
initialised from the same table -- same bar as the shipped model (tests/test_hostcheck.py); and both differ from the shipped model."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import so100_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "so100_mujoco_rl_amd", "csrc")
L = O.lib(); M0 = O.model()


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def _table():
    """made-up inertials: heavier links, shifted centres of mass, rotated principal axes (what a mesh-derived set would look like)"""
    rs = np.random.RandomState(5)
    t = np.zeros((6, 11))
    for k in range(6):
        b = k + 2
        t[k, 0] = M0.body_mass[b]*(1.2 + 0.3*rs.rand())
        t[k, 1:4] = np.array(M0.body_ipos[b][:]) + rs.randn(3)*0.004
        q = np.array(M0.body_iquat[b][:]) + rs.randn(4)*0.2; t[k, 4:8] = q/np.linalg.norm(q)
        t[k, 8:11] = np.array(M0.body_inertia[b][:])*(0.8 + 0.6*rs.rand())      # (one factor per link: keeps the triangle inequality MuJoCo's compiler enforces)
    return t


@pytest.fixture(scope="module")
def alt(tmp_path_factory):
    d = tmp_path_factory.mktemp("inert")
    tab = _table()
    f = d / "inertials.txt"
    f.write_text("# mass ipos(3) iquat(4) diaginertia(3): MjModel.body_mass / body_ipos / body_iquat / body_inertia of the six arm bodies\n"
                 + "\n".join(" ".join(repr(float(x)) for x in row) for row in tab) + "\n")
    hdr = d / "so100_model_gen_alt.h"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(d / "gen"), os.path.join(CSRC, "gen_model.cpp")])
    subprocess.check_call([str(d / "gen"), str(hdr), str(f)])
    assert "INERTIALS_OVERRIDDEN = true" in hdr.read_text()
    so = d / "libhostcheck_alt.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           f'-DSO100_MODEL_GEN_HEADER="{hdr}"', "-o", str(so), os.path.join(HERE, "_hostcheck", "hostcheck.cpp")])
    m = O.Model()
    L.so100o_model_init_with_inertials(C.byref(m), P(np.ascontiguousarray(tab.reshape(-1))))
    return C.CDLL(str(so)), m, tab


def test_override_reaches_both_sides_and_they_agree(alt):
    H, M1, tab = alt
    assert abs(M1.body_mass[3] - tab[1, 0]) < 1e-15 and abs(M1.body_mass[3] - M0.body_mass[3]) > 0.01
    assert abs(M1.kv[1] - M0.kv[1]) > 1e-3 and abs(M1.prox_radius[0] - M0.prox_radius[0]) > 1e-5       # derived quantities follow
    rs = np.random.RandomState(0); RNG = np.array(M0.jnt_range)
    wM = wb = dM = 0
    for _ in range(50):
        q = rs.uniform(RNG[:, 0], RNG[:, 1]); v = rs.uniform(-4, 4, 6)
        ref = {}
        for name, m in (("alt", M1), ("shipped", M0)):
            d = O.Data(); L.so100o_reset_data(C.byref(m), C.byref(d)); O.arr(d.qpos)[:6] = q; O.arr(d.qvel)[:6] = v
            L.so100o_forward(C.byref(m), C.byref(d), 0, 0)
            ref[name] = (O.arr(d.M).reshape(12, 12)[:6, :6].copy(), O.arr(d.qfrc_bias)[:6].copy())
        Mh = np.zeros(36); bh = np.zeros(6)
        H.hc_dyn_d(P(q), P(v), P(Mh), P(bh))
        wM = max(wM, np.abs(Mh.reshape(6, 6) - ref["alt"][0]).max()); wb = max(wb, np.abs(bh - ref["alt"][1]).max())
        dM = max(dM, np.abs(ref["alt"][0] - ref["shipped"][0]).max())
    assert wM < 1e-15 and wb < 1e-13 and dM > 1e-3


def test_whole_substeps_with_contacts_agree_under_the_override(alt):
    """32 substeps of the reference physics (friction loss, limits, pad / floor contacts, cube) from poses at the table: device code on the
    alternative header (fp64) against the oracle on the same table; the regulariser of the pad rows (body_invweight0) and the servo's kv
    are derived from the inertials, so this exercises everything the table feeds"""
    from test_oracle_contacts import floor_poses
    H, M1, _ = alt
    flags = O.F_REFERENCE
    wq = wv = 0; contacts = 0
    for q0 in floor_poses(6, 11):
        d = O.Data(); L.so100o_reset_data(C.byref(M1), C.byref(d))
        O.arr(d.qpos)[:6] = q0; O.arr(d.qpos)[6:9] = [0.15, -0.25, 0.0099]
        ctrl = q0.copy(); ctrl[1] += 0.05
        O.arr(d.ctrl)[:] = ctrl
        st = np.zeros(49); st[:6] = q0; st[30:33] = [0.15, -0.25, 0.0099]; st[33] = 1.0
        stat = np.zeros(5, np.int32); ap = np.zeros(3)
        H.hc_csub_d(P(st), P(ctrl), P(ap), flags, 60, 60, 32, P(stat))
        L.so100o_step(C.byref(M1), C.byref(d), flags, -1, 32)
        contacts += int(stat[0] > 0)
        wq = max(wq, np.abs(st[:6] - O.arr(d.qpos)[:6]).max()); wv = max(wv, np.abs(st[6:12] - O.arr(d.qvel)[:6]).max())
    assert contacts >= 4
    assert wq < 1e-9 and wv < 1e-7, (wq, wv)
