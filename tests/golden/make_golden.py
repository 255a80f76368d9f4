"""Generate golden fixtures by running the REFERENCE's own Python task code in this container.

Run (build container only; /root/reference does not exist on the GPU box):
    python3 -B tests/golden/make_golden.py

How: the reference's env classes (env01_v1.Env01, env02_v1.Env02, env03_v1.Env03, env04_v1.Env04, env05_v1.Env05,
env06_v1.Env06) are imported
unmodified from /root/reference/src with the third-party modules that are not installed here
(mujoco, gymnasium, glfw, cv2, ultralytics, PIL) replaced by stubs in sys.modules:
  * `MjData` is a thin named-accessor view over the oracle's fp64 `so100o_data` struct,
  * `mujoco.mj_step` calls the oracle's restatement of mj_step, `mj_resetData` its reset,
  * `ultralytics.YOLO.track` (Env03 / Env04 only) returns ONE box: the reference's own get_projected_cube_bounding_box(),
    over the black frame the stubbed offscreen renderer produces (the product's documented detector substitution),
  * `np.random.uniform/randint` are scripted so every draw is a recorded uniform u in [0,1)
    (numpy computes low + (high-low)*u, which the stub reproduces).
So every line of reward / obs / ctrl / reset / curriculum / reprojection logic that executes is the
reference's; only the physics underneath is ours.  The recorded trajectories (inputs: actions +
uniforms; outputs: obs, reward, terminated, qpos/qvel) are what tests/test_golden_task.py replays
through the oracle's C task layer.  Pure-function fixtures (joint penalty, base reward, pinhole
projection incl. a hand-checkable example) are emitted too.

Only DATA is written (tests/golden/*.json); no reference source or bytecode is copied.
`env_base_02.py` uses CAMERA_NAME without importing it (NameError upstream); the harness injects
utils.CAMERA_NAME, the unambiguous intent.
"""
import json
import math
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import so100_oracle as O  # noqa: E402

REF_SRC = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))

JOINT_NAMES = ["Rotation", "Pitch", "Elbow", "Wrist_Pitch", "Wrist_Roll", "Jaw"]
BODY_IDS = {"world": 0, "so100_Base": 1, "so100_Rotation_Pitch": 2, "so100_Upper_Arm": 3, "so100_Lower_Arm": 4,
            "so100_Wrist_Pitch_Roll": 5, "so100_Fixed_Jaw": 6, "so100_Moving_Jaw": 7, "block_a": 8}
CAMERA = "so100_end_point_camera"
PHYS_FLAGS = O.F_FRICTIONLOSS | O.F_LIMITS | O.F_FLOOR


# ----------------------------------------------------------------------------------------------
# scripted RNG
# ----------------------------------------------------------------------------------------------
class ScriptedRandom:
    """Maps each np.random call of the reference to a slot of the 16-float inject vector."""

    def __init__(self):
        self.inject = None
        self.phase = 0
        self.n_generic = 0
        self.n_noise = 0
        self.rs = np.random.RandomState(1234)

    def begin(self, phase_step_inject):
        self.inject = phase_step_inject
        self.n_generic = [0, 0]
        self.n_noise = [0, 0]

    def _u(self, slot):
        return float(self.inject[8 * self.phase + slot])

    def uniform(self, low=0.0, high=1.0, size=None):
        assert size is None
        p = self.phase
        caller = sys._getframe(1).f_code.co_name      # which reference function is drawing
        if caller == "_update_block_target" and (low, high) == (1.2, 5.1):
            slot = 3
        elif caller == "_get_obs":                    # Env05 detection noise
            assert (low, high) == (-0.05, 0.05)
            slot = 4 + self.n_noise[p]; self.n_noise[p] += 1
        else:
            slot = self.n_generic[p]; self.n_generic[p] += 1
        assert slot < 8
        return low + (high - low) * self._u(slot)

    def randint(self, low, high=None, size=None):
        assert size is None
        slot = 3
        return low + int(self._u(slot) * (high - low))


RNG = ScriptedRandom()


# ----------------------------------------------------------------------------------------------
# stub modules
# ----------------------------------------------------------------------------------------------
class _Acc:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class FakeData:
    """Named-accessor view over oracle so100o_data (what mujoco.MjData offers the reference)."""

    def __init__(self, d):
        object.__setattr__(self, "_d", d)
        self.qpos = O.arr(d.qpos); self.qvel = O.arr(d.qvel)
        self.ctrl = O.arr(d.ctrl); self.qfrc_applied = O.arr(d.qfrc_applied)
        self.xpos = O.arr(d.xpos); self.xmat = O.arr(d.xmat)
        self.cam_xpos = O.arr(d.cam_xpos); self.cam_xmat = O.arr(d.cam_xmat)

    @property
    def time(self):
        return float(self._d.time)

    def joint(self, name):
        if name == "block_a_joint":
            return _Acc(qpos=self.qpos[6:13], qvel=self.qvel[6:12], qfrc_applied=self.qfrc_applied[6:12])
        i = JOINT_NAMES.index(name[len("so100_"):])
        return _Acc(qpos=self.qpos[i:i + 1], qvel=self.qvel[i:i + 1])

    def body(self, name):
        b = BODY_IDS[name]
        return _Acc(xpos=self.xpos[b], xmat=self.xmat[b])

    def actuator(self, name):
        i = JOINT_NAMES.index(name[len("so100_"):])
        ctrl = self.ctrl

        class A:
            def __setattr__(s, k, v):
                assert k == "ctrl"
                ctrl[i] = v
        return A()

    def camera(self, name):
        assert name == CAMERA
        return _Acc(xpos=self.cam_xpos, xmat=self.cam_xmat)


class FakeModel:
    def __init__(self, m):
        self._m = m
        self.njnt = 7
        self.jnt_range = np.array([list(r) for r in m.jnt_range] + [[0.0, 0.0]])
        self.opt = _Acc(timestep=float(m.timestep), gravity=np.array(list(m.gravity)))
        self.body_mass = np.array(list(m.body_mass))
        self.cam_fovy = np.array([120.0])
        self.vis = _Acc(global_=_Acc(offwidth=640, offheight=480))

    def body(self, name):
        return _Acc(id=BODY_IDS[name])

    def camera(self, name):
        assert name == CAMERA
        return _Acc(id=0)


def install_stubs():
    mj = types.ModuleType("mujoco")
    names = ["so100_" + n for n in JOINT_NAMES] + ["block_a_joint"]

    class MjModel:
        @staticmethod
        def from_xml_path(path):
            assert path.endswith(("env01.xml", "env06.xml")), path     # env06.xml == env01.xml (byte-identical scene)
            return FakeModel(O.model())
    mj.MjModel = MjModel
    mj.mj_id2name = lambda model, typ, i: names[i]
    mj.mjtObj = _Acc(mjOBJ_JOINT=3)
    mj.mjtCamera = _Acc(mjCAMERA_FIXED=2)
    mj.mjtGridPos = _Acc(mjGRID_TOPRIGHT=1)
    mj.mjtCatBit = _Acc(mjCAT_ALL=7)
    mj.MjvCamera = lambda: _Acc(type=0, fixedcamid=-1)

    def mj_step(model, data, nstep=1):
        O.lib().so100o_step(O.C.byref(model._m), O.C.byref(data._d), PHYS_FLAGS, 0, nstep)
    mj.mj_step = mj_step
    mj.mj_rnePostConstraint = lambda model, data: None     # results never read
    mj.mjv_updateScene = mj.mjr_render = mj.mjr_readPixels = lambda *a, **k: None      # the offscreen image stays black
    mj.mj_resetData = lambda model, data: O.lib().so100o_reset_data(O.C.byref(model._m), O.C.byref(data._d))
    sys.modules["mujoco"] = mj

    gym = types.ModuleType("gymnasium")
    gym_utils = types.ModuleType("gymnasium.utils")

    class EzPickle:
        def __init__(self, *a, **k):
            pass
    gym_utils.EzPickle = EzPickle
    spaces = types.ModuleType("gymnasium.spaces")

    class Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low = np.asarray(low, dtype); self.high = np.asarray(high, dtype); self.dtype = dtype
            self.shape = self.low.shape
    spaces.Box = Box
    envs = types.ModuleType("gymnasium.envs")
    envs_mj = types.ModuleType("gymnasium.envs.mujoco")

    class MujocoEnv:
        """The slice of gymnasium 1.1.1's MujocoEnv the reference relies on."""

        def __init__(self, model_path, frame_skip, observation_space, render_mode=None, **kw):
            self.model = mj.MjModel.from_xml_path(model_path)
            self._od = O.Data()
            O.lib().so100o_reset_data(O.C.byref(self.model._m), O.C.byref(self._od))
            self.data = FakeData(self._od)
            self.frame_skip = frame_skip
            self.observation_space = observation_space
            self.render_mode = render_mode
            self._set_action_space()
            self.mujoco_renderer = _Acc(viewer=None)

        def reset(self, *, seed=None, options=None):
            mj.mj_resetData(self.model, self.data)
            ob = self.reset_model()
            return ob, {}

        def render(self):
            return None
    envs_mj.MujocoEnv = MujocoEnv
    rendering = types.ModuleType("gymnasium.envs.mujoco.mujoco_rendering")

    class OffScreenViewer:
        def __init__(self, model, data, width, height):
            self.model = model; self.data = data
            self.viewport = _Acc(width=width, height=height); self.vopt = self.scn = self.con = None

        def make_context_current(self):
            pass
    rendering.OffScreenViewer = OffScreenViewer
    registration = types.ModuleType("gymnasium.envs.registration")
    REGISTRY = []
    registration.register = lambda **kw: REGISTRY.append(kw)
    registration.make = registration.pprint_registry = registration.spec = lambda *a, **k: None
    registration.registry = {}
    gym.utils = gym_utils; gym.spaces = spaces; gym.envs = envs
    envs.mujoco = envs_mj; envs.registration = registration; envs_mj.mujoco_rendering = rendering
    for name, mod in [("gymnasium", gym), ("gymnasium.utils", gym_utils), ("gymnasium.spaces", spaces),
                      ("gymnasium.envs", envs), ("gymnasium.envs.mujoco", envs_mj),
                      ("gymnasium.envs.mujoco.mujoco_rendering", rendering),
                      ("gymnasium.envs.registration", registration)]:
        sys.modules[name] = mod

    for name in ["glfw", "cv2", "PIL", "PIL.Image", "ultralytics"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    glfw = sys.modules["glfw"]; glfw.get_current_context = lambda: None; glfw.make_context_current = lambda c: None
    cv2 = sys.modules["cv2"]
    cv2.line = cv2.rectangle = cv2.putText = lambda img, *a, **k: img
    cv2.FONT_HERSHEY_SIMPLEX = 0; cv2.INTER_LINEAR = 1

    class Detector:
        """Stands in for ultralytics.YOLO in Env03 / Env04 (env_base_02.py:186-222): ONE box per frame, the reference's own
        get_projected_cube_bounding_box() (env_base_02.py:154-176), confidence 0.9, track id 1, class 1; no box when that is None.
        This is the documented detector substitution of the product; everything else in _get_obs / step is the reference's."""
        names = {0: "other", 1: "cube"}
        env = None

        def __init__(self, path):
            pass

        def track(self, img, **kw):
            assert img.shape == (1920, 1080, 3)
            bb = self.env.get_projected_cube_bounding_box()
            boxes = [] if bb is None else [_Acc(conf=[0.9], id=[1], cls=[1], xyxy=[[bb[0][0], bb[0][1], bb[1][0], bb[1][1]]])]
            return [_Acc(boxes=boxes)]
    sys.modules["ultralytics"].YOLO = Detector
    return REGISTRY


def reference_cli_surface():
    """The click surface of the reference's main.py (group options, commands, command options): imported with
    stable_baselines3 stubbed (none of its classes is touched at import time)."""
    import click
    sb3 = types.ModuleType("stable_baselines3")
    mods = {"stable_baselines3": sb3}
    for sub, names in (("common", []), ("common.base_class", ["BaseAlgorithm"]),
                       ("common.callbacks", ["StopTrainingOnNoModelImprovement", "StopTrainingOnRewardThreshold", "EvalCallback", "CheckpointCallback", "CallbackList"]),
                       ("common.monitor", ["Monitor"]), ("common.noise", ["NormalActionNoise"]), ("common.vec_env", ["VecVideoRecorder", "DummyVecEnv"])):
        m = types.ModuleType("stable_baselines3." + sub)
        for n in names:
            setattr(m, n, type(n, (), {}))
        mods["stable_baselines3." + sub] = m
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    gym = sys.modules["gymnasium"]                       # (the stub of install_stubs) annotations / wrappers main.py names at import time
    for n in ("Env", "Wrapper"):
        if not hasattr(gym, n): setattr(gym, n, type(n, (), {}))
    if not hasattr(gym, "wrappers"): gym.wrappers = types.SimpleNamespace(RecordVideo=type("RecordVideo", (), {}))
    try:
        import importlib
        ref_main = importlib.import_module("so100_mujoco_rl.main")
    finally:
        for k, v in saved.items():
            if v is None: sys.modules.pop(k, None)
            else: sys.modules[k] = v

    def params(cmd):
        return [{"opts": sorted(p.opts), "required": bool(p.required), "is_flag": bool(getattr(p, "is_flag", False)),
                 "default": p.default if isinstance(p.default, (str, int, float, bool, type(None))) else str(p.default)}
                for p in cmd.params if isinstance(p, click.Option)]
    return {"group": params(ref_main.cli), "commands": {name: params(c) for name, c in sorted(ref_main.cli.commands.items())},
            "dirs": [ref_main.MODEL_DIR, ref_main.LOG_DIR, ref_main.RECORDING_DIR]}


def f(x):
    """json-able float/array"""
    if isinstance(x, (list, tuple)):
        return [f(v) for v in x]
    a = np.asarray(x)
    if a.ndim == 0:
        return float(a)
    return [f(v) for v in a]


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as fh:
        json.dump(obj, fh, indent=0, separators=(",", ":"))
    print("wrote", name)


def main():
    registry = install_stubs()
    sys.path.insert(0, REF_SRC)
    os.chdir(tempfile.mkdtemp())          # env_base_02 creates ./images in the cwd
    np.random.uniform = RNG.uniform
    np.random.randint = RNG.randint

    import so100_mujoco_rl  # noqa: F401  (runs the register() calls)
    from so100_mujoco_rl.envs import utils as ref_utils
    from so100_mujoco_rl.envs import env_base_02
    env_base_02.CAMERA_NAME = ref_utils.CAMERA_NAME
    from so100_mujoco_rl.envs.env01_v1 import Env01
    from so100_mujoco_rl.envs.env02_v1 import Env02
    from so100_mujoco_rl.envs.env05_v1 import Env05
    from so100_mujoco_rl.envs.env06_v1 import Env06
    from so100_mujoco_rl.envs.env03_v1 import Env03
    from so100_mujoco_rl.envs.env04_v1 import Env04
    from so100_mujoco_rl.envs import env03_v1

    rs = np.random.RandomState(20240801)

    # ---- registry / constants / spaces ------------------------------------------------------
    e1 = Env01(); e5 = Env05()
    meta = {
        "registry": [{k: v for k, v in r.items()} for r in registry],
        "JOINT_STEP_SCALE": ref_utils.JOINT_STEP_SCALE,
        "REST_POSITION": ref_utils.REST_POSITION,
        "START_POSITION": env03_v1.START_POSITION,
        "VALID_START_POSITIONS": ref_utils.VALID_START_POSITIONS,
        "joint_names": [j.name for j in e1.joints],
        "joint_ranges": [list(j.range) for j in e1.joints],
        "obs_space_15": {"low": f(e1.observation_space.low), "high": f(e1.observation_space.high)},
        "obs_space_8": {"low": f(e5.observation_space.low), "high": f(e5.observation_space.high)},
        "action_space": {"low": f(e1.action_space.low), "high": f(e1.action_space.high)},
        "frame_skip": e1.frame_skip, "render_fps": Env01.metadata["render_fps"],
        "cli": reference_cli_surface(),
        "spaces": {str(k): {"obs_low": f(e.observation_space.low), "obs_high": f(e.observation_space.high),
                            "act_low": f(e.action_space.low), "act_high": f(e.action_space.high)}
                   for k, e in ((1, e1), (2, Env02()), (3, Env03()), (4, Env04()), (5, e5), (6, Env06()))},
    }
    dump("meta.json", meta)

    # ---- pure functions ----------------------------------------------------------------------
    pen = []
    for _ in range(200):
        lo, hi = sorted(rs.uniform(-3.2, 3.2, 2)); a = rs.uniform(lo - 0.5, hi + 0.5)
        pen.append({"a": a, "lo": lo, "hi": hi, "out": e1._calculate_joint_penalty(a, (lo, hi))})
    for j in e1.joints:     # exactly at the thresholds / bounds
        for a in (j.range[0], j.range[1], j.range[0] + 0.05 * (j.range[1] - j.range[0]), 0.0):
            pen.append({"a": a, "lo": j.range[0], "hi": j.range[1], "out": e1._calculate_joint_penalty(a, j.range)})
    rew = []
    for case in range(300):
        env = Env01()
        d = env.data
        q = np.array([rs.uniform(j.range[0] - 0.2, j.range[1] + 0.2) for j in env.joints])
        d.qpos[0:6] = q
        d.xpos[8] = rs.uniform(-0.45, 0.45, 3) * [1, 1, 0.05]
        d.xpos[5] = rs.uniform(-0.3, 0.3, 3); d.xpos[5][2] = rs.uniform(-0.02, 0.2)
        d.xpos[6] = rs.uniform(-0.3, 0.3, 3); d.xpos[6][2] = rs.uniform(-0.02, 0.2)
        from scipy.spatial.transform import Rotation
        d.xmat[6] = Rotation.random(random_state=rs).as_matrix().reshape(9)
        if case % 7 == 0:      # make the cube very close to the end effector (distance term = 0)
            d.xpos[8] = env.get_end_effector_pos() + rs.uniform(-0.005, 0.005, 3)
        ee = env.get_end_effector_pos()
        r_first = env._get_reward()      # first call after construction: last_* is None (has_prev False)
        r_next = env._get_reward()       # every later call: has_prev True
        for hp, out in ((False, r_first), (True, r_next)):
            rew.append({"q": f(q), "block": f(d.xpos[8]), "jaw_xpos": f(d.xpos[6]), "jaw_xmat": f(d.xmat[6]),
                        "wrist": f(d.xpos[5]), "has_prev": hp, "ee": f(ee), "reward": float(out),
                        "obs": f(env._get_obs())})
    proj = []
    env = Env05()
    # hand-checkable example: camera (0,-0.25,0.25), R=diag(1,-1,-1), p=(0.05,-0.3,0.01) -> (425,845)
    cases = [((0, -0.25, 0.25), np.diag([1.0, -1.0, -1.0]).reshape(9), (0.05, -0.3, 0.01))]
    from scipy.spatial.transform import Rotation
    for _ in range(400):
        cam = rs.uniform(-0.3, 0.3, 3); cam[2] = rs.uniform(0.05, 0.4)
        R = Rotation.random(random_state=rs).as_matrix()
        if rs.rand() < 0.6:      # look roughly at the point so that many cases land inside the frame
            p = cam + R @ (np.array([rs.uniform(-0.2, 0.2), rs.uniform(-0.3, 0.3), -1.0]) * rs.uniform(0.05, 0.5))
        else:
            p = rs.uniform(-0.45, 0.45, 3)
        cases.append((cam, R.reshape(9), p))
    cases.append(((0, 0, 0), np.zeros(9), (0.0, -0.35, 0.01)))      # state after mj_resetData: NaN -> None
    for cam, R, p in cases:
        env.data.cam_xpos[:] = cam; env.data.cam_xmat[:] = R
        out = env._get_projected_position(np.array(p, dtype=float))
        proj.append({"cam_xpos": f(cam), "cam_xmat": f(R), "p": f(p), "uv": None if out is None else [int(out[0]), int(out[1])]})
    # 8-corner bounding box of the cube (env_base_02.py:129-176, unused upstream; the product's stand-in for YOLO in Env03/04)
    bbox = []
    for cam, R, p in cases:
        env.data.cam_xpos[:] = cam; env.data.cam_xmat[:] = R
        env.data.qpos[6:9] = p
        out = env.get_projected_cube_bounding_box()
        bbox.append({"cam_xpos": f(cam), "cam_xmat": f(R), "p": f(p),
                     "box": None if out is None else [int(out[0][0]), int(out[0][1]), int(out[1][0]), int(out[1][1])]})
    dump("pure.json", {"joint_penalty": pen, "reward_base": rew, "projection": proj, "bbox": bbox})

    # ---- trajectories: reference task code over oracle physics --------------------------------
    def run(EnvCls, kind, n_steps, seed, action_scale, episodes_reset_every=None, tweak=None):
        r2 = np.random.RandomState(seed)
        env = EnvCls()
        if getattr(env, "yolo_model", None) is not None:
            env.yolo_model.env = env
        steps = []
        inj = r2.random_sample(16).astype(np.float32)
        RNG.begin(inj); RNG.phase = 1
        ob, _ = env.reset()
        rec = {"kind": kind, "flags": PHYS_FLAGS, "reset_inject": f(inj), "reset_obs": f(ob), "steps": steps}
        for t in range(n_steps):
            a = (r2.uniform(-1, 1, 6) * action_scale).astype(np.float32)
            a = np.clip(a, -1, 1)
            if tweak is not None:
                a = tweak(t, env, a)
            inj = r2.random_sample(16).astype(np.float32)
            RNG.begin(inj); RNG.phase = 0
            ob, rew_, term, trunc, info = env.step(a)
            st = {"action": f(a), "inject": f(inj), "obs": f(ob), "reward": float(rew_), "terminated": bool(term),
                  "qpos": f(env.data.qpos), "qvel": f(env.data.qvel), "time": env.data.time, "reset_after": False}
            if term or (episodes_reset_every and (t + 1) % episodes_reset_every == 0):
                RNG.phase = 1
                ob2, _ = env.reset()
                st["reset_after"] = True; st["reset_obs"] = f(ob2)
            steps.append(st)
        return rec

    def drive_to_cube(t, env, a):
        # Env02: steer the end effector towards the cube so the reach branch (< 3 cm) fires.
        # A crude Jacobian-free heuristic is enough: alternate random exploration with holding.
        return a

    trajs = []
    trajs.append(run(Env01, 1, 60, 11, 1.0))
    trajs.append(run(Env01, 1, 40, 12, 0.3, episodes_reset_every=15))
    trajs.append(run(Env02, 2, 60, 21, 1.0, episodes_reset_every=25))
    trajs.append(run(Env05, 5, 120, 51, 0.5))
    trajs.append(run(Env05, 5, 80, 52, 1.0, episodes_reset_every=30))

    # Env02 reach branch: place the cube right at the (stale) end-effector position before a step
    env = Env02(); r2 = np.random.RandomState(77)
    inj = r2.random_sample(16).astype(np.float32); RNG.begin(inj); RNG.phase = 1
    ob, _ = env.reset()
    rec = {"kind": 2, "flags": PHYS_FLAGS, "reset_inject": f(inj), "reset_obs": f(ob), "steps": []}
    for t in range(30):
        a = np.clip(r2.uniform(-1, 1, 6), -1, 1).astype(np.float32)
        inj = r2.random_sample(16).astype(np.float32); RNG.begin(inj); RNG.phase = 0
        pre = None
        if t in (5, 6, 17):       # teleport the cube (qpos AND stale xpos) onto the stale end effector
            ee = env.get_end_effector_pos()
            env.data.qpos[6:9] = ee; env.data.xpos[8] = ee
            pre = {"cube_qpos": f(env.data.qpos[6:9]), "cube_xpos": f(env.data.xpos[8])}
        ob, rew_, term, trunc, info = env.step(a)
        rec["steps"].append({"action": f(a), "inject": f(inj), "obs": f(ob), "reward": float(rew_), "terminated": bool(term),
                             "qpos": f(env.data.qpos), "qvel": f(env.data.qvel), "time": env.data.time,
                             "reset_after": False, "pre_teleport": pre})
    trajs.append(rec)

    # Env05 termination branch: hold the arm still, turn the camera away from the cube -> >30 lost steps
    env = Env05(); r2 = np.random.RandomState(78)
    inj = r2.random_sample(16).astype(np.float32); RNG.begin(inj); RNG.phase = 1
    ob, _ = env.reset()
    rec = {"kind": 5, "flags": PHYS_FLAGS, "reset_inject": f(inj), "reset_obs": f(ob), "steps": []}
    for t in range(70):
        a = np.zeros(6, np.float32); a[0] = 1.0 if t < 28 else 0.0      # rotate the base away: camera loses the cube
        inj = r2.random_sample(16).astype(np.float32); RNG.begin(inj); RNG.phase = 0
        ob, rew_, term, trunc, info = env.step(a)
        st = {"action": f(a), "inject": f(inj), "obs": f(ob), "reward": float(rew_), "terminated": bool(term),
              "qpos": f(env.data.qpos), "qvel": f(env.data.qvel), "time": env.data.time, "reset_after": False}
        if term:
            RNG.phase = 1
            ob2, _ = env.reset(); st["reset_after"] = True; st["reset_obs"] = f(ob2)
        rec["steps"].append(st)
    trajs.append(rec)
    # Env06 (appended last so the indices above stay put): random run with resets (block memory persists across
    # episodes), then a reach run: the cube is teleported onto the stale end effector and STAYS there (Env06 does not
    # re-randomise), small arm actions + a closing jaw, so the gripper term varies step to step.
    trajs.append(run(Env06, 6, 50, 61, 1.0, episodes_reset_every=20))
    env = Env06(); r2 = np.random.RandomState(79)
    inj = r2.random_sample(16).astype(np.float32); RNG.begin(inj); RNG.phase = 1
    ob, _ = env.reset()
    rec = {"kind": 6, "flags": PHYS_FLAGS, "reset_inject": f(inj), "reset_obs": f(ob), "steps": []}
    for t in range(40):
        a = np.clip(r2.uniform(-1, 1, 6) * (1.0 if t < 8 else 0.02), -1, 1).astype(np.float32)
        if t >= 8:
            a[5] = 1.0 if t < 30 else -1.0                                # close, then open the jaw
        inj = r2.random_sample(16).astype(np.float32); RNG.begin(inj); RNG.phase = 0
        pre = None
        if t in (8, 20):
            ee = env.get_end_effector_pos()
            env.data.qpos[6:9] = ee; env.data.xpos[8] = ee
            pre = {"cube_qpos": f(env.data.qpos[6:9]), "cube_xpos": f(env.data.xpos[8])}
        ob, rew_, term, trunc, info = env.step(a)
        st = {"action": f(a), "inject": f(inj), "obs": f(ob), "reward": float(rew_), "terminated": bool(term),
              "qpos": f(env.data.qpos), "qvel": f(env.data.qvel), "time": env.data.time, "reset_after": False, "pre_teleport": pre}
        if t == 25:                                                       # second episode: last_block_pos != block_pos afterwards
            RNG.phase = 1
            ob2, _ = env.reset(); st["reset_after"] = True; st["reset_obs"] = f(ob2)
        rec["steps"].append(st)
    trajs.append(rec)
    # Env03 / Env04 (appended last): the reference's step / reward / curriculum / _get_obs code with the detector stub above.
    # A still arm (the cube wanders through the frame), a random run with resets, and a run that turns the camera away so
    # that the lost-count branch (Env03: termination after > 30 lost steps; Env04: last centre re-used) fires.
    def look_away(t, env, a):
        a = np.zeros(6, np.float32); a[0] = 1.0 if t < 28 else 0.0
        return a
    trajs.append(run(Env03, 3, 80, 31, 0.0))
    trajs.append(run(Env03, 3, 60, 32, 0.6, episodes_reset_every=25))
    trajs.append(run(Env03, 3, 70, 33, 0.0, tweak=look_away))
    trajs.append(run(Env04, 4, 80, 41, 0.0))
    trajs.append(run(Env04, 4, 60, 42, 0.6, episodes_reset_every=25))
    trajs.append(run(Env04, 4, 70, 43, 0.0, tweak=look_away))
    dump("trajectories.json", trajs)
    n_term = sum(s["terminated"] for tr in trajs for s in tr["steps"])
    n_reach = sum(1 for s in trajs[5]["steps"] if s.get("pre_teleport"))
    print("terminated steps:", n_term, "reach teleports:", n_reach)


if __name__ == "__main__":
    if not os.path.isdir(REF_SRC):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    main()
