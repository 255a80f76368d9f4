"""ctypes binding of the CPU oracle (oracle/so100_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (so100_mujoco_rl_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NB, NQ, NV, MAXEFC, NINJECT, NPAD, MAXCON, NPROX, NCPROX = 9, 13, 12, 412, 16, 8, 100, 5, 2
F_FRICTIONLOSS, F_LIMITS, F_FLOOR, F_CUBE_PINNED, F_PADS_FLOOR, F_PADS_CUBE, F_LINKS_FLOOR, F_LINKS_CUBE = 1, 2, 4, 8, 16, 32, 64, 128
F_REFERENCE = F_FRICTIONLOSS | F_LIMITS | F_FLOOR | F_PADS_FLOOR        # what the reference scene simulates (minus the mesh geoms)
F_CONTACT5 = F_REFERENCE | F_PADS_CUBE                                   # BASELINE.json configs[4]

d_ = C.c_double


class Model(C.Structure):
    _fields_ = [
        ("body_parent", C.c_int * NB), ("body_jnt", C.c_int * NB),
        ("body_dofadr", C.c_int * NB), ("body_qposadr", C.c_int * NB),
        ("body_pos", d_ * 3 * NB), ("body_quat", d_ * 4 * NB),
        ("body_ipos", d_ * 3 * NB), ("body_iquat", d_ * 4 * NB),
        ("body_mass", d_ * NB), ("body_inertia", d_ * 3 * NB),
        ("jnt_axis", d_ * 3 * NB), ("jnt_range", d_ * 2 * 6),
        ("cam_pos", d_ * 3), ("cam_quat", d_ * 4),
        ("armature", d_ * NV), ("frictionloss", d_ * NV),
        ("dof_M0", d_ * NV), ("dof_invweight0", d_ * NV),
        ("body_invweight0", d_ * 2 * NB),
        ("kp", d_), ("kv", d_ * 6),
        ("timestep", d_), ("gravity", d_ * 3),
        ("qpos0", d_ * NQ),
        ("pad_body", C.c_int * NPAD), ("pad_pos", d_ * 3 * NPAD), ("pad_size", d_ * 3 * NPAD),
        ("pad_solref", d_ * 2), ("pad_solimp", d_ * 5), ("pad_friction", d_),
        ("def_solref", d_ * 2), ("def_solimp", d_ * 5), ("def_friction", d_),
        ("max_contacts", C.c_int),
        ("prox_body", C.c_int * NPROX), ("prox_p", d_ * 3 * 2 * NPROX), ("prox_radius", d_ * NPROX),
        ("cprox_body", C.c_int * NCPROX), ("cprox_p", d_ * 3 * 2 * NCPROX), ("cprox_radius", d_ * NCPROX),
    ]


class Contact(C.Structure):
    _fields_ = [("b1", C.c_int), ("b2", C.c_int), ("kind", C.c_int), ("geom", C.c_int), ("feat", C.c_int),
                ("pos", d_ * 3), ("frame", d_ * 9), ("dist", d_), ("mu", d_), ("solref", d_ * 2), ("solimp", d_ * 5),
                ("efc0", C.c_int)]


class Data(C.Structure):
    _fields_ = [
        ("qpos", d_ * NQ), ("qvel", d_ * NV), ("qacc_warmstart", d_ * NV),
        ("ctrl", d_ * 6), ("qfrc_applied", d_ * NV), ("time", d_),
        ("xpos", d_ * 3 * NB), ("xquat", d_ * 4 * NB), ("xmat", d_ * 9 * NB),
        ("xipos", d_ * 3 * NB), ("ximat", d_ * 9 * NB),
        ("xaxis", d_ * 3 * NV), ("xanchor", d_ * 3 * NV),
        ("cam_xpos", d_ * 3), ("cam_xmat", d_ * 9),
        ("cdof", d_ * 6 * NV), ("cdof_dot", d_ * 6 * NV), ("cvel", d_ * 6 * NB),
        ("cinert", d_ * 36 * NB), ("crb", d_ * 36 * NB),
        ("M", d_ * (NV * NV)), ("L", d_ * (NV * NV)),
        ("qfrc_bias", d_ * NV), ("qfrc_actuator", d_ * NV), ("qfrc_smooth", d_ * NV),
        ("qacc_smooth", d_ * NV), ("qfrc_constraint", d_ * NV), ("qacc", d_ * NV),
        ("nefc", C.c_int), ("ncon", C.c_int),
        ("efc_type", C.c_int * MAXEFC), ("efc_id", C.c_int * MAXEFC),
        ("efc_J", d_ * NV * MAXEFC), ("efc_pos", d_ * MAXEFC), ("efc_aref", d_ * MAXEFC),
        ("efc_R", d_ * MAXEFC), ("efc_force", d_ * MAXEFC), ("efc_floss", d_ * MAXEFC),
        ("efc_b", d_ * MAXEFC),
        ("warm_fric", d_ * 6), ("warm_limit", d_ * 6), ("warm_contact", d_ * 16),
        ("solver_iter_used", C.c_int), ("solver_last_change", d_),
        ("con", Contact * MAXCON), ("ncon_dropped", C.c_int),
    ]


class Env(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("flags", C.c_uint), ("iters", C.c_int), ("frame_skip", C.c_int),
        ("max_episode_steps", C.c_int), ("seed", C.c_uint64), ("env_id", C.c_uint32),
        ("rng_counter", C.c_uint32),
        ("d", Data),
        ("has_prev", C.c_int), ("elapsed_steps", C.c_int),
        ("episode_return", d_), ("episode_length", C.c_int),
        ("have_block_pos", C.c_int), ("have_last_block_pos", C.c_int),
        ("block_pos", d_ * 3), ("last_block_pos", d_ * 3),
        ("cmd", d_ * 6),
        ("have_center", C.c_int), ("last_center", d_ * 2), ("lost_count", C.c_int),
        ("space_min", d_ * 3), ("space_max", d_ * 3), ("block_speed", d_),
        ("block_target", d_ * 3), ("target_dt", d_), ("target_time", d_),
        ("have_last_angvel", C.c_int), ("last_angvel", d_ * 6),
        ("bad_state", C.c_int),
        ("block_position_updated", C.c_int),
    ]


def build(force=False):
    """Compile oracle/libso100oracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "libso100oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "so100_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        assert L.so100o_sizeof(0) == C.sizeof(Model), (L.so100o_sizeof(0), C.sizeof(Model))
        assert L.so100o_sizeof(1) == C.sizeof(Data), (L.so100o_sizeof(1), C.sizeof(Data))
        assert L.so100o_sizeof(2) == C.sizeof(Env), (L.so100o_sizeof(2), C.sizeof(Env))
        L.so100o_joint_penalty.restype = d_
        L.so100o_joint_penalty.argtypes = [d_, d_, d_]
        L.so100o_reward_base.restype = d_
        L.so100o_envs_alloc.restype = C.POINTER(Env)
        L.so100o_envs_at.restype = C.POINTER(Env)
        L.so100o_envs_at.argtypes = [C.POINTER(Env), C.c_int]
        L.so100o_envs_free.argtypes = [C.POINTER(Env)]
        L.so100o_env_init.argtypes = [C.POINTER(Model), C.POINTER(Env), C.c_int, C.c_uint, C.c_int, C.c_uint64, C.c_uint32]
        _lib = L
    return _lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_model = None


def model():
    global _model
    if _model is None:
        _model = Model()
        lib().so100o_model_init(C.byref(_model))
    return _model


def arr(cfield):
    """numpy view (no copy) of a ctypes array field."""
    return np.ctypeslib.as_array(cfield)


class OracleEnv:
    """One reference-semantics env (kind 1..6) on the CPU."""

    def __init__(self, kind, flags=F_FRICTIONLOSS | F_LIMITS | F_FLOOR, iters=0, seed=0, env_id=0):
        self.L = lib(); self.m = model(); self.e = Env()
        self.L.so100o_env_init(C.byref(self.m), C.byref(self.e), kind, flags, iters, seed, env_id)
        self.kind = kind
        self.obs_dim = self.L.so100o_obs_dim(kind)

    @property
    def d(self):
        return self.e.d

    def reset(self, inject=None):
        obs = np.zeros(self.obs_dim, np.float32)
        inj = None if inject is None else np.ascontiguousarray(inject, np.float32)
        assert inj is None or inj.size == NINJECT
        self.L.so100o_env_reset(C.byref(self.m), C.byref(self.e), _fp(inj), _fp(obs))
        return obs

    def step(self, action, inject=None, autoreset=False):
        obs = np.zeros(self.obs_dim, np.float32); tobs = np.zeros(self.obs_dim, np.float32)
        a = np.ascontiguousarray(action, np.float32)
        inj = None if inject is None else np.ascontiguousarray(inject, np.float32)
        r = d_(0); t = C.c_int(0); tr = C.c_int(0)
        self.L.so100o_env_step(C.byref(self.m), C.byref(self.e), _fp(a), _fp(inj), int(autoreset),
                               _fp(obs), C.byref(r), C.byref(t), C.byref(tr), _fp(tobs))
        return obs, r.value, bool(t.value), bool(tr.value), tobs


class OracleBatch:
    """N envs stepped on the host, optionally over several threads (ctypes releases the GIL).
    This is bench.py's cpu_baseline ("port")."""

    def __init__(self, kind, n, flags, iters, seed=0, env_id0=0):
        self.L = lib(); self.m = model(); self.n = n; self.kind = kind
        self.obs_dim = self.L.so100o_obs_dim(kind)
        self.envs = self.L.so100o_envs_alloc(n)
        for i in range(n):
            self.L.so100o_env_init(C.byref(self.m), self.L.so100o_envs_at(self.envs, i), kind, flags, iters, seed, env_id0 + i)
        self.obs = np.zeros((n, self.obs_dim), np.float32); self.tobs = np.zeros_like(self.obs)
        self.rew = np.zeros(n, np.float32); self.term = np.zeros(n, np.uint8); self.trunc = np.zeros(n, np.uint8)

    def env(self, i):
        return self.L.so100o_envs_at(self.envs, i).contents

    def reset(self, inject=None):
        for i in range(self.n):
            inj = None if inject is None else np.ascontiguousarray(inject[i], np.float32)
            self.L.so100o_env_reset(C.byref(self.m), self.L.so100o_envs_at(self.envs, i), _fp(inj), _fp(self.obs[i]))
        return self.obs

    def step_range(self, actions, begin, end, autoreset=True):
        self.L.so100o_envs_step_range(C.byref(self.m), self.envs, begin, end, _fp(actions), int(autoreset),
                                      _fp(self.obs), _fp(self.rew), _fp(self.term), _fp(self.trunc), _fp(self.tobs))

    def step(self, actions, threads=1, autoreset=True, native_threads=False):
        actions = np.ascontiguousarray(actions, np.float32)
        if threads <= 1:
            self.step_range(actions, 0, self.n, autoreset)
        elif native_threads:                                 # pthreads inside the C library: no per-slice Python dispatch (bench.py's all-core baseline)
            rc = self.L.so100o_envs_step_threads(C.byref(self.m), self.envs, self.n, int(threads), _fp(actions), int(autoreset),
                                                 _fp(self.obs), _fp(self.rew), _fp(self.term), _fp(self.trunc), _fp(self.tobs))
            assert rc >= 0
        else:
            import concurrent.futures as cf
            if not hasattr(self, "_pool") or self._pool._max_workers != threads:
                self._pool = cf.ThreadPoolExecutor(threads)
            cuts = np.linspace(0, self.n, threads + 1).astype(int)
            list(self._pool.map(lambda k: self.step_range(actions, int(cuts[k]), int(cuts[k + 1]), autoreset), range(threads)))
        return self.obs, self.rew, self.term, self.trunc

    def __del__(self):
        try:
            self.L.so100o_envs_free(self.envs)
        except Exception:
            pass
