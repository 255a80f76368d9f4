"""Task constants mirrored from the reference (no third-party imports; usable without a GPU).

ref = /root/reference/src/so100_mujoco_rl/
"""
import numpy as np

# ref: __init__.py:5-38 (gymnasium.register ids, TimeLimit, reward_threshold)
ENV_IDS = {1: "Env01-v1", 2: "Env02-v1", 3: "Env03-v1", 4: "Env04-v1", 5: "Env05-v1", 6: "Env06-v1"}
MAX_EPISODE_STEPS = {1: 4000, 2: 6000, 3: 6000, 4: 6000, 5: 6000, 6: 6000}
REWARD_THRESHOLD = {1: 6000, 2: 8000, 3: 8000, 4: 8000, 5: 8000, 6: 8000}
RENDER_FPS = 31                                          # ref: envs/env_base_01.py:32
FRAME_SKIP = 16                                          # ref: envs/env_base_01.py:45

JOINT_STEP_SCALE = 0.075                                 # ref: envs/utils.py:9
REST_POSITION = [0.0, -3.141, 3.117, 1.0, 0.0, 0.0]      # ref: envs/utils.py:11
START_POSITION = [0.0, -2.04, 1.19, 1.5, -1.58, 0.5]     # ref: envs/env03_v1.py:10
JOINT_NAMES = ["Rotation", "Pitch", "Elbow", "Wrist_Pitch", "Wrist_Roll", "Jaw"]
# ref: envs/model/so_arm100_camera.xml:35-50 via joints_from_model (envs/utils.py:64-89)
JOINT_RANGES = [(-2.2, 2.2), (-3.14158, 0.2), (0.0, 3.14158), (-2.0, 1.8), (-3.14158, 3.14158), (-0.2, 2.0)]


def observation_space_bounds(env_kind):
    """(low, high) float32 arrays of the observation Box; ref: envs/env_base_01.py:63-75 (15-dim, Env01/02),
    envs/env_base_02.py:56-69 (8-dim, Env03-05)."""
    lo = [r[0] for r in JOINT_RANGES]; hi = [r[1] for r in JOINT_RANGES]
    if env_kind in (1, 2, 6):
        low = lo + [-1.0] * 3 + [-0.5] * 6; high = hi + [1.0] * 3 + [0.5] * 6
    else:
        low = lo + [0.0, 0.0]; high = hi + [5.0, 5.0]
    return np.array(low, np.float32), np.array(high, np.float32)


def action_space_bounds():
    """ref: envs/env_base_01.py:77-83"""
    return -np.ones(6, np.float32), np.ones(6, np.float32)
