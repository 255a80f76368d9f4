"""So100VecEnv -- the Stable-Baselines3 `VecEnv` surface over the batched HIP simulator.

Replaces, for N envs at once, what the reference gets from SB3 wrapping its single Gymnasium env
(ref: /root/reference/src/so100_mujoco_rl/main.py:57-63, 131 -> DummyVecEnv([lambda: Monitor(gym.make(id))])):
`reset() -> obs[N, .]`, `step_async(actions)`, `step_wait() -> (obs, rewards, dones, infos)`, auto-reset on done with
`infos[i]["terminal_observation"]`, `infos[i]["TimeLimit.truncated"]` and Monitor's `infos[i]["episode"] = {"r","l","t"}`.

Everything stays on the GPU between calls; numpy only appears at the SB3 edge (SB3's rollout buffers are numpy).
If stable_baselines3 / gymnasium are importable the class derives from SB3's VecEnv and uses gymnasium.spaces.Box, so
`PPO("MlpPolicy", So100VecEnv("Env01-v1", 4096))` drops in unchanged; otherwise a structural stand-in with the same
methods is used (neither package is installed in the build image).
"""
import time

import numpy as np
import torch

from . import constants as K
from .lib import So100Sim, F_REFERENCE

try:                                                   # optional: real SB3 / gymnasium types when present
    from stable_baselines3.common.vec_env.base_vec_env import VecEnv as _VecEnvBase
    from gymnasium import spaces as _spaces
    _HAVE_SB3 = True
except Exception:                                      # pragma: no cover - not installed in the build image
    _HAVE_SB3 = False

    class _VecEnvBase:                                  # the slice of SB3's VecEnv contract that callers rely on
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs = num_envs; self.observation_space = observation_space; self.action_space = action_space
            self.render_mode = None
            self.reset_infos = [{} for _ in range(num_envs)]
            self._seeds = [None] * num_envs; self._options = [{}] * num_envs

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()

    class _Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low = np.asarray(low, dtype); self.high = np.asarray(high, dtype)
            self.shape = self.low.shape; self.dtype = np.dtype(dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class _spaces:                                      # noqa: N801
        Box = _Box


def make_spaces(env_kind):
    lo, hi = K.observation_space_bounds(env_kind)
    alo, ahi = K.action_space_bounds()
    return _spaces.Box(lo, hi, dtype=np.float32), _spaces.Box(alo, ahi, dtype=np.float32)


def kind_from_id(env_id):
    for k, v in K.ENV_IDS.items():
        if v == env_id:
            return k
    raise KeyError(f"unknown env id {env_id!r}; known: {sorted(K.ENV_IDS.values())}")


class So100VecEnv(_VecEnvBase):
    metadata = {"render_modes": [], "render_fps": K.RENDER_FPS}

    def __init__(self, env_id="Env01-v1", num_envs=4096, device=None, flags=F_REFERENCE, seed=0, env_id_offset=0,
                 solver_iters=2, contact_iters=20, max_episode_steps=None, stagger_episodes=False, full_infos=False, use_graph=True,
                 envs_per_workgroup=0):
        self.env_id = env_id
        self.kind = kind_from_id(env_id) if isinstance(env_id, str) else int(env_id)
        obs_space, act_space = make_spaces(self.kind)
        _VecEnvBase.__init__(self, num_envs, obs_space, act_space)
        self.sim = So100Sim(self.kind, num_envs, device=device, flags=flags, solver_iters=solver_iters, contact_iters=contact_iters,
                            max_episode_steps=max_episode_steps, seed=seed, env_id_offset=env_id_offset,
                            envs_per_workgroup=envs_per_workgroup)      # 0 = automatic; pin it when shards of another batch size must agree bit for bit
        self.device = self.sim.device
        self.full_infos = full_infos
        self._stagger = stagger_episodes
        self._actions = torch.zeros(num_envs, 6, dtype=torch.float32, device=self.device)
        self._infos = [{} for _ in range(num_envs)]
        # pinned host buffers: the step kernel reads the actions from and writes its results to them directly (see _round_trip)
        od = self.sim.obs_dim
        self._h_obs = torch.empty(num_envs, od, dtype=torch.float32, pin_memory=True)
        self._h_rew = torch.empty(num_envs, dtype=torch.float32, pin_memory=True)
        self._h_done = torch.empty(num_envs, dtype=torch.uint8, pin_memory=True)
        self._h_trunc = torch.empty(num_envs, dtype=torch.uint8, pin_memory=True)
        self._h_act = torch.zeros(num_envs, 6, dtype=torch.float32, pin_memory=True)
        self._h_tobs = torch.empty(num_envs, od, dtype=torch.float32, pin_memory=True)
        self._h_epr = torch.empty(num_envs, dtype=torch.float32, pin_memory=True)
        self._h_epl = torch.empty(num_envs, dtype=torch.int32, pin_memory=True)
        # use_graph (the historical name of the switch) = True: zero-copy round trip -- ONE kernel launch per step, the kernel itself
        # moves actions / results over the host link (62 us per step at 4096 envs).  False: the staged path -- actions H2D, the step
        # kernel on device buffers, seven D2H copies (94 us even when replayed from one hipGraph); kept as the cross-check.
        self._zero_copy = bool(use_graph)
        self._dirty = []                                # infos filled on the previous step (cleared lazily)
        self._t0 = time.time()
        self.spec = type("Spec", (), {"id": K.ENV_IDS[self.kind], "max_episode_steps": self.sim.cfg.max_episode_steps,
                                      "reward_threshold": K.REWARD_THRESHOLD[self.kind]})()

    # ---- tensor API (no host round trip): what an on-device rollout collector uses --------------------------------
    def reset_tensor(self):
        obs = self.sim.reset()
        if self._stagger:
            g = torch.Generator(device=self.device); g.manual_seed(int(self.sim.cfg.seed) + 7)
            self.sim.set_field("elapsed_steps", torch.randint(0, max(1, self.sim.cfg.max_episode_steps), (self.num_envs,),
                                                               device=self.device, generator=g, dtype=torch.int32))
        return obs

    def step_tensor(self, actions):
        """actions float32 [N,6] on the device -> (obs, rew, done(uint8), trunc(uint8)) device tensors (views)."""
        return self.sim.step(actions)

    # ---- SB3 VecEnv API ------------------------------------------------------------------------------------------------
    def reset(self):
        return self.reset_tensor().cpu().numpy()

    def step_async(self, actions):
        self._h_act.numpy()[...] = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 6)

    def _round_trip(self):
        if self._zero_copy:
            self.sim.step_host(self._h_act, self._h_obs, self._h_rew, self._h_done, self._h_trunc, self._h_tobs, self._h_epr, self._h_epl)
            return
        self._actions.copy_(self._h_act, non_blocking=True)
        obs, rew, done, trunc = self.sim.step(self._actions)
        self._h_obs.copy_(obs, non_blocking=True); self._h_rew.copy_(rew, non_blocking=True)
        self._h_done.copy_(done, non_blocking=True); self._h_trunc.copy_(trunc, non_blocking=True)
        self._h_tobs.copy_(self.sim.terminal_obs, non_blocking=True)
        self._h_epr.copy_(self.sim.ep_return, non_blocking=True); self._h_epl.copy_(self.sim.ep_length, non_blocking=True)

    def step_wait(self):
        self._round_trip()
        torch.cuda.current_stream(self.device).synchronize()
        # fresh arrays every step: SB3 keeps `_last_obs` alive across the next env.step()
        obs_h = self._h_obs.numpy().copy(); rew_h = self._h_rew.numpy().copy()
        done_h = self._h_done.numpy().astype(bool)
        for i in self._dirty:
            self._infos[i] = {}
        self._dirty = []
        if done_h.any():
            idx = np.nonzero(done_h)[0]
            trunc_h = self._h_trunc.numpy().astype(bool)
            tobs = self._h_tobs.numpy()[idx].copy()
            ep_r = self._h_epr.numpy(); ep_l = self._h_epl.numpy()
            t = round(time.time() - self._t0, 6)
            for j, i in enumerate(idx):
                self._infos[i] = {"terminal_observation": tobs[j], "TimeLimit.truncated": bool(trunc_h[i]),
                                  "episode": {"r": float(ep_r[i]), "l": int(ep_l[i]), "t": t}}
            self._dirty = list(idx)
        if self.full_infos:                             # SB3 itself reads infos with .get(..., False); only for strict consumers
            for i in range(self.num_envs):
                self._infos[i].setdefault("TimeLimit.truncated", False)
        return obs_h, rew_h, done_h, self._infos

    def close(self):
        self.sim.close()

    def seed(self, seed=None):
        return [seed] * self.num_envs                  # the device RNG is keyed at construction (Philox seed)

    def get_attr(self, attr_name, indices=None):
        n = len(self._indices(indices))
        return [getattr(self, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return [getattr(self, method_name)(*method_args, **method_kwargs) for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None                                     # rasteriser / viewer are out of scope

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)
