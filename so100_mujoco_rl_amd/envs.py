"""Gymnasium single-env view (N = 1) of the batched simulator and the `register()` ids of the reference.

ref: /root/reference/src/so100_mujoco_rl/__init__.py:5-45 registers Env01-v1 .. Env06-v1 with
max_episode_steps 4000/6000 and reward_threshold 6000/8000.  `register_envs()` does the same for this package's entry
point when gymnasium is importable (it is not in the build image; the class works without it).
Step/reset signatures follow Gymnasium: reset(seed, options) -> (obs, info); step(a) -> (obs, reward, terminated,
truncated, info).  TimeLimit is applied by the simulator itself (max_episode_steps), so no wrapper is needed.
"""
import numpy as np
import torch

from . import constants as K
from .lib import F_REFERENCE
from .vec_env import So100VecEnv, make_spaces


class So100Env:
    metadata = {"render_modes": [], "render_fps": K.RENDER_FPS}

    def __init__(self, env_kind=1, device=None, flags=F_REFERENCE, seed=0, render_mode=None, max_episode_steps=None, **kwargs):
        self.kind = env_kind
        self.observation_space, self.action_space = make_spaces(env_kind)
        # a one-env So100VecEnv underneath: its numpy round trip (pinned host buffers the step kernel reads / writes directly, one launch, one sync per step)
        self._mk = lambda s: So100VecEnv(env_kind, 1, device=device, flags=flags, seed=s,
                                         max_episode_steps=K.MAX_EPISODE_STEPS[env_kind] if max_episode_steps is None else max_episode_steps)
        self._v = self._mk(seed)
        self.render_mode = render_mode
        self._after_done = None          # observation of the episode the simulator has already started (see step)

    @property
    def sim(self):
        return self._v.sim

    def reset(self, *, seed=None, options=None):
        if seed is not None:                            # unlike the reference (global np.random) seeding works
            self._v.close(); self._v = self._mk(int(seed)); self._after_done = None
        if self._after_done is not None:                # the fused step already reset this env when the episode ended:
            ob, self._after_done = self._after_done, None   # hand out THAT episode's first observation, do not reset twice
            return ob, {}
        return self._v.reset()[0].copy(), {}

    def step(self, action):
        """Gymnasium semantics on top of a simulator that auto-resets inside the fused step (SB3 VecEnv semantics):
        on done the TERMINAL observation is returned and the next episode's first observation is kept for reset().
        Stepping on without reset() -- what the reference's own viewer loop does for up to 200 steps after an episode
        ends (main.py:118-124) -- simply continues in that next episode; no state is lost or reset twice either way."""
        self._v.step_async(np.asarray(action, np.float32).reshape(1, 6))
        obs, rew, done, infos = self._v.step_wait()
        d = bool(done[0]); tr = d and bool(infos[0].get("TimeLimit.truncated", False))
        ob = infos[0]["terminal_observation"].copy() if d else obs[0]
        self._after_done = obs[0].copy() if d else None
        return ob, float(rew[0]), d and not tr, tr, {}

    def close(self):
        self._v.close()

    def render(self):
        return None


def _entry(kind):
    def make(**kw):
        return So100Env(kind, **kw)
    return make


def register_envs():
    """gymnasium.register the reference's ids; returns the list of ids registered ([] if gymnasium is absent)."""
    try:
        from gymnasium.envs.registration import register
    except Exception:
        return []
    ids = []
    for kind, env_id in K.ENV_IDS.items():
        register(id=env_id, entry_point=_entry(kind), max_episode_steps=None, reward_threshold=K.REWARD_THRESHOLD[kind])
        ids.append(env_id)
    return ids
