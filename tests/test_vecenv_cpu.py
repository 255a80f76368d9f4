"""Host-side mirror of the reference interface (spaces, ids, episode limits, shard layout) against the fixtures
recorded from the reference (tests/golden/meta.json).  CPU only."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def meta(golden_dir):
    return json.load(open(os.path.join(golden_dir, "meta.json")))


def test_spaces_match_reference(meta):
    from so100_mujoco_rl_amd.vec_env import make_spaces
    o15, a = make_spaces(1); o8, _ = make_spaces(5)
    np.testing.assert_array_equal(o15.low, np.array(meta["obs_space_15"]["low"], np.float32))
    np.testing.assert_array_equal(o15.high, np.array(meta["obs_space_15"]["high"], np.float32))
    np.testing.assert_array_equal(o8.low, np.array(meta["obs_space_8"]["low"], np.float32))
    np.testing.assert_array_equal(o8.high, np.array(meta["obs_space_8"]["high"], np.float32))
    np.testing.assert_array_equal(a.low, np.array(meta["action_space"]["low"], np.float32))
    np.testing.assert_array_equal(a.high, np.array(meta["action_space"]["high"], np.float32))
    assert o15.shape == (15,) and o8.shape == (8,) and a.shape == (6,) and o15.dtype == np.float32
    for kind in range(1, 7):                                 # every registered env's own spaces (meta.json "spaces")
        ref = meta["spaces"][str(kind)]; o, a = make_spaces(kind)
        np.testing.assert_array_equal(o.low, np.array(ref["obs_low"], np.float32)); np.testing.assert_array_equal(o.high, np.array(ref["obs_high"], np.float32))
        np.testing.assert_array_equal(a.low, np.array(ref["act_low"], np.float32)); np.testing.assert_array_equal(a.high, np.array(ref["act_high"], np.float32))


def test_ids_limits_constants_match_reference(meta):
    from so100_mujoco_rl_amd import constants as K
    from so100_mujoco_rl_amd.vec_env import kind_from_id
    reg = {r["id"]: r for r in meta["registry"]}
    for kind, env_id in K.ENV_IDS.items():
        assert K.MAX_EPISODE_STEPS[kind] == reg[env_id]["max_episode_steps"]
        assert K.REWARD_THRESHOLD[kind] == reg[env_id]["reward_threshold"]
        assert kind_from_id(env_id) == kind
    with pytest.raises(KeyError):
        kind_from_id("Env99-v1")
    assert K.JOINT_STEP_SCALE == meta["JOINT_STEP_SCALE"] and K.REST_POSITION == meta["REST_POSITION"]
    assert K.START_POSITION == meta["START_POSITION"] and K.JOINT_NAMES == meta["joint_names"]
    np.testing.assert_array_equal(np.array(K.JOINT_RANGES), np.array(meta["joint_ranges"]))
    assert K.FRAME_SKIP == meta["frame_skip"] and K.RENDER_FPS == meta["render_fps"]
    inc = open(os.path.join(os.path.dirname(__file__), "..", "so100_mujoco_rl_amd", "csrc", "so100_start_positions.inc")).read()
    import re
    vals = [float(x) for x in re.findall(r"-?\d+\.\d+(?:e[-+]?\d+)?", inc.split("= {", 1)[1])]
    np.testing.assert_array_equal(np.array(vals).reshape(36, 6), np.array(meta["VALID_START_POSITIONS"]))


def test_shard_ranges():
    from so100_mujoco_rl_amd.rollout import shard_range
    for total, world in [(32768, 8), (65536, 8), (10, 3), (7, 8)]:
        r = [shard_range(total, k, world) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == total
        assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1
