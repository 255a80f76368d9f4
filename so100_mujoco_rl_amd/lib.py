"""ctypes binding of libso100sim.so (C ABI: include/so100_sim.h).

PyTorch is plumbing here: it owns device memory and streams; every tensor crosses the boundary as a raw
device pointer (`tensor.data_ptr()`) plus the current HIP stream handle.  There is NO CPU fallback: if the
shared object is missing or no HIP device is usable, loading / `So100Sim()` raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch  # imported BEFORE the .so so that both bind to the same libamdhip64 (see csrc/Makefile)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SO100_LIB", os.path.join(_HERE, "libso100sim.so"))   # SO100_LIB: A/B builds of the same ABI (tools/)

ENV01, ENV02, ENV03, ENV04, ENV05, ENV06 = 1, 2, 3, 4, 5, 6
F_FRICTIONLOSS, F_LIMITS, F_FLOOR, F_CUBE_PINNED, F_PADS_FLOOR, F_PADS_CUBE, F_LINKS_FLOOR, F_LINKS_CUBE = 1, 2, 4, 8, 16, 32, 64, 128
F_REFERENCE = F_FRICTIONLOSS | F_LIMITS | F_FLOOR | F_PADS_FLOOR     # what the reference scene simulates (minus its mesh geoms)
F_NOPADS = F_FRICTIONLOSS | F_LIMITS | F_FLOOR                         # round-1 "reference": no arm contact at all
F_CONTACT5 = F_REFERENCE | F_PADS_CUBE                                  # BASELINE.json configs[4]: finger pads vs cube, coupled solve
F_REFERENCE_LINKS = F_REFERENCE | F_LINKS_FLOOR                         # + capsule proxies of the arm's collision meshes vs the floor (a documented stand-in)
F_REFERENCE_PROXIES = F_REFERENCE_LINKS | F_LINKS_CUBE                  # + Rotation_Pitch / Upper_Arm capsules vs the cube (needs a dynamic cube)
B_BAD_STATE = 128            # bit of the `bits` state row latched when a non-finite state ended an episode (csrc/so100_task.hpp)
NINJECT = 16
ABI_VERSION = 3              # include/so100_sim.h: SO100_ABI_VERSION


class So100Error(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("env_kind", C.c_int32), ("num_envs", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("solver_iters", C.c_int32), ("contact_iters", C.c_int32), ("frame_skip", C.c_int32),
                ("max_episode_steps", C.c_int32), ("seed", C.c_uint64), ("env_id_offset", C.c_uint32),
                ("envs_per_workgroup", C.c_uint32)]


class StepIO(C.Structure):
    _fields_ = [("act_dev", C.c_void_p), ("obs_dev", C.c_void_p), ("rew_dev", C.c_void_p), ("done_dev", C.c_void_p),
                ("trunc_dev", C.c_void_p), ("terminal_obs_dev", C.c_void_p), ("ep_return_dev", C.c_void_p),
                ("ep_length_dev", C.c_void_p), ("inject_dev", C.c_void_p), ("rollout_row_dev", C.c_void_p)]


POLICY_TENSORS = ["pi_w0", "pi_b0", "pi_w1", "pi_b1", "mu_w", "mu_b", "log_std", "vf_w0", "vf_b0", "vf_w1", "vf_b1", "v_w", "v_b"]
# the SB3 ActorCriticPolicy state_dict entry behind each pointer
SB3_STATE_DICT_KEYS = {"pi_w0": "mlp_extractor.policy_net.0.weight", "pi_b0": "mlp_extractor.policy_net.0.bias",
                       "pi_w1": "mlp_extractor.policy_net.2.weight", "pi_b1": "mlp_extractor.policy_net.2.bias",
                       "mu_w": "action_net.weight", "mu_b": "action_net.bias", "log_std": "log_std",
                       "vf_w0": "mlp_extractor.value_net.0.weight", "vf_b0": "mlp_extractor.value_net.0.bias",
                       "vf_w1": "mlp_extractor.value_net.2.weight", "vf_b1": "mlp_extractor.value_net.2.bias",
                       "v_w": "value_net.weight", "v_b": "value_net.bias"}


class PolicyWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in POLICY_TENSORS]


class RolloutIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rollout_dev", "obs_dev", "rew_dev", "done_dev", "trunc_dev", "terminal_obs_dev",
                                          "ep_return_dev", "ep_length_dev", "terminal_obs_chunk_dev")]


class PolicyIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs_dev", "noise_dev", "act_env_dev", "act_raw_dev", "value_dev", "logp_dev", "rollout_row_dev")]


EXPORTS = ["so100_abi_version", "so100_obs_dim", "so100_num_state_fields", "so100_state_field_index", "so100_state_field_name", "so100_create",
           "so100_envs_per_workgroup", "so100_destroy", "so100_reset", "so100_step", "so100_get_state", "so100_set_state", "so100_get_field",
           "so100_set_field", "so100_last_error", "so100_policy_forward", "so100_rollout"]


def build(verbose=False):
    """Compile libso100sim.so for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-j7", "-C", os.path.join(_HERE, "csrc")], capture_output=True, text=True)
    if out.returncode != 0:
        raise So100Error("building libso100sim.so failed:\n" + out.stdout[-4000:] + out.stderr[-4000:])
    if verbose:
        print(out.stdout[-3000:])
    return LIB_PATH


_lib = None


def load():
    """dlopen the shared object (built in-tree; never falls back to anything else)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise So100Error(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.so100_last_error.restype = C.c_char_p
        L.so100_state_field_index.argtypes = [C.c_char_p]
        L.so100_state_field_name.argtypes = [C.c_int32]; L.so100_state_field_name.restype = C.c_char_p
        L.so100_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.so100_envs_per_workgroup.argtypes = [C.c_void_p]
        L.so100_destroy.argtypes = [C.c_void_p]
        L.so100_destroy.restype = None
        L.so100_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.so100_step.argtypes = [C.c_void_p, C.POINTER(StepIO), C.c_void_p]
        L.so100_policy_forward.argtypes = [C.c_void_p, C.POINTER(PolicyWeights), C.POINTER(PolicyIO), C.c_uint32, C.c_void_p]
        L.so100_rollout.argtypes = [C.c_void_p, C.POINTER(PolicyWeights), C.POINTER(RolloutIO), C.c_int32, C.c_uint32, C.c_void_p]
        L.so100_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.so100_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.so100_get_field.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.so100_set_field.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        if L.so100_abi_version() != ABI_VERSION:
            raise So100Error("libso100sim.so ABI version mismatch")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise So100Error(f"{what} failed ({rc}): {load().so100_last_error().decode()}")


def _hptr(t, dtype, shape):
    """pointer of a PINNED host tensor (hipHostMalloc memory is mapped into the GPU's address space at the same address)"""
    if t is None:
        return None
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.device.type != "cpu" or not t.is_pinned():
        raise So100Error(f"bad tensor: want {dtype} {tuple(shape)} contiguous in pinned host memory, got {t.dtype} {tuple(t.shape)} on {t.device}")
    return t.data_ptr()


def _ptr(t, dtype, shape, device):
    if t is None:
        return None
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.device != device:
        raise So100Error(f"bad tensor: want {dtype} {tuple(shape)} contiguous on {device}, got {t.dtype} {tuple(t.shape)} on {t.device}")
    return t.data_ptr()


class So100Sim:
    """One batched simulator handle on one GPU (one per process per device)."""

    def __init__(self, env_kind, num_envs, device=None, flags=F_REFERENCE, solver_iters=2, contact_iters=20,
                 frame_skip=16, max_episode_steps=None, seed=0, env_id_offset=0, envs_per_workgroup=0):
        self.L = load()
        if not torch.cuda.is_available():
            raise So100Error("no HIP device visible to PyTorch: so100 has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if max_episode_steps is None:
            max_episode_steps = 4000 if env_kind == ENV01 else 6000       # ref: so100_mujoco_rl/__init__.py:5-45
        self.cfg = Config(env_kind, num_envs, self.device.index, flags, solver_iters, contact_iters, frame_skip,
                          max_episode_steps, seed, env_id_offset, envs_per_workgroup)
        h = C.c_void_p()
        _check(self.L.so100_create(C.byref(self.cfg), C.byref(h)), "so100_create")
        self.h = h
        self.envs_per_workgroup = self.L.so100_envs_per_workgroup(h)     # in use (chosen by the library when 0 was passed): part of a checkpoint's configuration
        self.n = num_envs
        self.kind = env_kind
        self.obs_dim = self.L.so100_obs_dim(env_kind)
        kw = dict(device=self.device)
        # the handle's device-side outputs (read through the properties below: a step_host() leaves them stale until then)
        self._obs = torch.zeros(num_envs, self.obs_dim, dtype=torch.float32, **kw)
        self._rew = torch.zeros(num_envs, dtype=torch.float32, **kw)
        self._done = torch.zeros(num_envs, dtype=torch.uint8, **kw)
        self._trunc = torch.zeros(num_envs, dtype=torch.uint8, **kw)
        self._terminal_obs = torch.zeros(num_envs, self.obs_dim, dtype=torch.float32, **kw)
        self._ep_return = torch.zeros(num_envs, dtype=torch.float32, **kw)
        self._ep_length = torch.zeros(num_envs, dtype=torch.int32, **kw)
        self._host_out = None        # pinned host buffers of the last step_host() while they are newer than the device tensors
        self._io = StepIO()

    # obs / rew / done / trunc / terminal_obs / ep_return / ep_length: device tensors, always current.  step_host() writes its
    # results to the caller's pinned host buffers only; the first read of any of these afterwards (a checkpoint, the rollout
    # collector, policy_forward on sim.obs ...) mirrors them back to the device, on the current stream, once.
    def _refresh(self):
        h = self._host_out
        if h is not None:
            self._host_out = None
            # obs / rew / done / trunc are rewritten for every env by every step; terminal_obs / ep_return / ep_length only where an episode ended --
            # the caller's host buffers persist from step to step, so they hold the latest value of every env: whole-buffer copies are right for all seven
            for dst, src in zip((self._obs, self._rew, self._done, self._trunc, self._terminal_obs, self._ep_return, self._ep_length), h):
                if src is not None:
                    dst.copy_(src, non_blocking=True)

    obs = property(lambda self: (self._refresh(), self._obs)[1])
    rew = property(lambda self: (self._refresh(), self._rew)[1])
    done = property(lambda self: (self._refresh(), self._done)[1])
    trunc = property(lambda self: (self._refresh(), self._trunc)[1])
    terminal_obs = property(lambda self: (self._refresh(), self._terminal_obs)[1])
    ep_return = property(lambda self: (self._refresh(), self._ep_return)[1])
    ep_length = property(lambda self: (self._refresh(), self._ep_length)[1])

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "h", None):
            self.L.so100_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self, mask=None, inject=None):
        """Reset all envs (mask None) or those with mask != 0.  Returns the (persistent) obs tensor."""
        m = _ptr(mask, torch.uint8, (self.n,), self.device)
        i = _ptr(inject, torch.float32, (self.n, NINJECT), self.device)
        _check(self.L.so100_reset(self.h, m, i, self.obs.data_ptr(), self._stream()), "so100_reset")
        return self.obs

    def step(self, actions, inject=None, rollout_row=None, terminal_obs=None):
        """actions: float32 [N,6] on the device.  Returns views of the handle's output tensors.
        rollout_row: optional float32 [N, obs_dim+10] row of a rollout buffer; reward/done are written into it.
        terminal_obs: optional float32 [N, obs_dim] destination of the terminal observations (default: self.terminal_obs)."""
        io = self._io
        io.rollout_row_dev = _ptr(rollout_row, torch.float32, (self.n, self.obs_dim + 10), self.device)
        io.act_dev = _ptr(actions, torch.float32, (self.n, 6), self.device)
        io.obs_dev = self.obs.data_ptr(); io.rew_dev = self.rew.data_ptr()
        io.done_dev = self.done.data_ptr(); io.trunc_dev = self.trunc.data_ptr()
        io.terminal_obs_dev = self.terminal_obs.data_ptr() if terminal_obs is None else _ptr(terminal_obs, torch.float32, (self.n, self.obs_dim), self.device)
        io.ep_return_dev = self.ep_return.data_ptr(); io.ep_length_dev = self.ep_length.data_ptr()
        io.inject_dev = _ptr(inject, torch.float32, (self.n, NINJECT), self.device)
        _check(self.L.so100_step(self.h, C.byref(io), self._stream()), "so100_step")
        return self.obs, self.rew, self.done, self.trunc

    def step_host(self, act, obs, rew, done, trunc, terminal_obs=None, ep_return=None, ep_length=None):
        """The same env step with every boundary buffer in PINNED HOST memory: the kernel reads the actions and writes its results
        over the host link itself -- one launch, no copy nodes (So100VecEnv's numpy path: 62 instead of 94 us per step at 4096 envs).
        The results are valid after the stream is synchronised; the handle's device-side obs / rew / done ... tensors are refreshed
        from these buffers lazily, on their next read (`_refresh`) -- so the buffers must stay alive and unmodified until then
        (So100VecEnv owns them; a following step_host() simply supersedes a mirror that nobody asked for: the numpy loop pays nothing)."""
        io = StepIO(_hptr(act, torch.float32, (self.n, 6)), _hptr(obs, torch.float32, (self.n, self.obs_dim)), _hptr(rew, torch.float32, (self.n,)),
                    _hptr(done, torch.uint8, (self.n,)), _hptr(trunc, torch.uint8, (self.n,)),
                    _hptr(terminal_obs, torch.float32, (self.n, self.obs_dim)), _hptr(ep_return, torch.float32, (self.n,)),
                    _hptr(ep_length, torch.int32, (self.n,)), None, None)
        _check(self.L.so100_step(self.h, C.byref(io), self._stream()), "so100_step")
        self._host_out = (obs, rew, done, trunc, terminal_obs, ep_return, ep_length)

    def set_policy(self, tensors):
        """tensors: dict name -> float32 device tensor (names: POLICY_TENSORS; SB3 keys: SB3_STATE_DICT_KEYS)."""
        od = self.obs_dim
        shapes = {"pi_w0": (64, od), "pi_b0": (64,), "pi_w1": (64, 64), "pi_b1": (64,), "mu_w": (6, 64), "mu_b": (6,), "log_std": (6,),
                  "vf_w0": (64, od), "vf_b0": (64,), "vf_w1": (64, 64), "vf_b1": (64,), "v_w": (1, 64), "v_b": (1,)}
        self._policy_tensors = {k: tensors[k] for k in POLICY_TENSORS}          # keep them alive
        self._pw = PolicyWeights(*[_ptr(tensors[k], torch.float32, shapes[k], self.device) for k in POLICY_TENSORS])
        self._pio = PolicyIO()

    def policy_forward(self, obs, act_env, step_counter, noise=None, act_raw=None, value=None, logp=None, rollout_row=None):
        """Fused SB3-MlpPolicy forward + sample + clip (+ rollout-row write): one launch on the current stream."""
        io = self._pio
        io.obs_dev = _ptr(obs, torch.float32, (self.n, self.obs_dim), self.device)
        io.noise_dev = _ptr(noise, torch.float32, (self.n, 6), self.device)
        io.act_env_dev = _ptr(act_env, torch.float32, (self.n, 6), self.device)
        io.act_raw_dev = _ptr(act_raw, torch.float32, (self.n, 6), self.device)
        io.value_dev = _ptr(value, torch.float32, (self.n,), self.device)
        io.logp_dev = _ptr(logp, torch.float32, (self.n,), self.device)
        io.rollout_row_dev = _ptr(rollout_row, torch.float32, (self.n, self.obs_dim + 10), self.device)
        _check(self.L.so100_policy_forward(self.h, C.byref(self._pw), C.byref(io), int(step_counter) & 0xFFFFFFFF, self._stream()), "so100_policy_forward")

    def rollout(self, rollout_buf, step_counter0, terminal_obs_chunk=None):
        """T = rollout_buf.shape[0] steps of {policy, sample, env step, buffer write} in one launch (persistent kernel).
        rollout_buf: float32 [T, N, obs_dim+10].  Uses / updates the handle's obs, rew, done, ... tensors.
        terminal_obs_chunk: optional float32 [T, N, obs_dim], receives the terminal observation wherever an episode ended."""
        T = rollout_buf.shape[0]
        io = RolloutIO(_ptr(rollout_buf, torch.float32, (T, self.n, self.obs_dim + 10), self.device), self.obs.data_ptr(), self.rew.data_ptr(),
                       self.done.data_ptr(), self.trunc.data_ptr(), self.terminal_obs.data_ptr(), self.ep_return.data_ptr(), self.ep_length.data_ptr(),
                       _ptr(terminal_obs_chunk, torch.float32, (T, self.n, self.obs_dim), self.device))
        _check(self.L.so100_rollout(self.h, C.byref(self._pw), C.byref(io), T, int(step_counter0) & 0xFFFFFFFF, self._stream()), "so100_rollout")

    def get_state(self):
        qpos = torch.empty(13, self.n, dtype=torch.float32, device=self.device)
        qvel = torch.empty(12, self.n, dtype=torch.float32, device=self.device)
        _check(self.L.so100_get_state(self.h, qpos.data_ptr(), qvel.data_ptr(), self._stream()), "so100_get_state")
        return qpos, qvel

    def set_state(self, qpos, qvel):
        _check(self.L.so100_set_state(self.h, _ptr(qpos, torch.float32, (13, self.n), self.device),
                                      _ptr(qvel, torch.float32, (12, self.n), self.device), self._stream()), "so100_set_state")

    def field_index(self, name):
        i = self.L.so100_state_field_index(name.encode())
        if i < 0:
            raise So100Error(f"unknown state field {name!r}")
        return i

    def field_names(self):
        return [self.L.so100_state_field_name(i).decode() for i in range(self.L.so100_num_state_fields())]

    # ---- sim checkpoint: the whole [field][N] state matrix as raw 32-bit words + the current
    # observation, beside the learner's own checkpoint (ref: main.py:227-232 saves only the SB3 zip).  Resuming is
    # bit exact: the Philox counters and the solver warm starts are rows of the matrix.
    def save_state(self, path):
        names = self.field_names()
        words = torch.stack([self.get_field(n, dtype=torch.int32) for n in names]).cpu().numpy()
        c = self.cfg
        np.savez(path, words=words, names=np.array(names), obs=self.obs.cpu().numpy(),
                 config=np.array([c.env_kind, c.num_envs, c.flags, c.solver_iters, c.contact_iters, c.frame_skip,
                                  c.max_episode_steps, c.seed, c.env_id_offset, self.envs_per_workgroup], dtype=np.int64))

    def load_state(self, path, allow_config_mismatch=False):
        with np.load(path, allow_pickle=False) as z:
            words, names, obs, conf = z["words"], [str(n) for n in z["names"]], z["obs"], z["config"]
        if int(conf[0]) != self.cfg.env_kind or int(conf[1]) != self.n:
            raise So100Error(f"checkpoint is for env kind {int(conf[0])} x {int(conf[1])} envs, this sim is kind {self.cfg.env_kind} x {self.n}")
        # resume is bit exact only under the configuration the checkpoint was taken with: refuse anything else
        c = self.cfg
        mine = [c.env_kind, c.num_envs, c.flags, c.solver_iters, c.contact_iters, c.frame_skip, c.max_episode_steps, c.seed, c.env_id_offset,
                self.envs_per_workgroup]
        labels = ["env_kind", "num_envs", "flags", "solver_iters", "contact_iters", "frame_skip", "max_episode_steps", "seed", "env_id_offset",
                  "envs_per_workgroup"]
        # (envs_per_workgroup follows from N and the device's CU count unless pinned: the pad-contact solve's summation order depends on it,
        #  so a resume is bit exact only under the same value; checkpoints of earlier versions do not carry it)
        diff = [f"{l}: checkpoint {int(a)} != sim {int(b)}" for l, a, b in zip(labels, conf, mine) if int(a) != int(b)]
        if len(conf) < len(mine):
            diff.append("envs_per_workgroup: not recorded in this (older) checkpoint")
        if diff and not allow_config_mismatch:
            raise So100Error("checkpoint was taken under a different configuration (" + "; ".join(diff) + "); pass allow_config_mismatch=True to load it anyway")
        missing = set(self.field_names()) - set(names)
        if missing and not allow_config_mismatch:
            raise So100Error(f"checkpoint lacks state fields {sorted(missing)} (written by an older version?); allow_config_mismatch=True zero-fills them")
        for n in sorted(missing):                           # older checkpoint, loaded on request: solver memory / statistics rows start from zero
            self.set_field(n, torch.zeros(self.n, dtype=torch.int32, device=self.device))
        for row, n in zip(words, names):
            if self.L.so100_state_field_index(n.encode()) >= 0:
                self.set_field(n, torch.from_numpy(row).to(self.device))
        self._host_out = None
        self._obs.copy_(torch.from_numpy(obs))

    def contacts_dropped(self):
        """int32 [N]: contacts dropped over the 16-record budget in the last env step (0 where the contact flags are off).  MuJoCo would keep
        them: an env that reports > 0 deviates from the reference model in that step (a jaw lying flat on the table)."""
        if (self.cfg.flags & (F_PADS_FLOOR | F_PADS_CUBE | F_LINKS_FLOOR | F_LINKS_CUBE)) == 0:
            return torch.zeros(self.n, dtype=torch.int32, device=self.device)
        return self.get_field("contact_stat", dtype=torch.int32) >> 8

    def bad_state_mask(self):
        """bool [N]: envs whose episode was ever ended by the non-finite state guard (NaN / inf action or state)."""
        return (self.get_field("bits", dtype=torch.int32) & B_BAD_STATE) != 0

    def get_field(self, name, dtype=torch.float32):
        out = torch.empty(self.n, dtype=dtype, device=self.device)
        _check(self.L.so100_get_field(self.h, self.field_index(name), out.data_ptr(), self._stream()), "so100_get_field")
        return out

    def set_field(self, name, value):
        assert value.element_size() == 4
        _check(self.L.so100_set_field(self.h, self.field_index(name), _ptr(value, value.dtype, (self.n,), self.device),
                                      self._stream()), "so100_set_field")
