"""so100_mujoco_rl_amd -- MI355X-native batched simulator for the so100 arm (drop-in for the per-env-step hot
path of PieterBecking/so100-mujoco-rl).  See DESIGN.md / INTEGRATION.md at the repo root."""
from .constants import (ENV_IDS, JOINT_NAMES, JOINT_RANGES, JOINT_STEP_SCALE, MAX_EPISODE_STEPS, REST_POSITION,
                        REWARD_THRESHOLD, START_POSITION, action_space_bounds, observation_space_bounds)

__all__ = ["ENV_IDS", "JOINT_NAMES", "JOINT_RANGES", "JOINT_STEP_SCALE", "MAX_EPISODE_STEPS", "REST_POSITION",
           "REWARD_THRESHOLD", "START_POSITION", "action_space_bounds", "observation_space_bounds"]
