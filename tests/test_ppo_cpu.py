"""CPU tests of the learner-side bookkeeping: TimeLimit truncations must be bootstrapped like SB3's
OnPolicyAlgorithm.collect_rollouts does (rewards += gamma * V(terminal_observation)), then GAE treats the step as an
episode end.  Hand-computed reference, truncation in the middle of a chunk."""
import numpy as np
import torch

from so100_mujoco_rl_amd.ppo import PPO
from so100_mujoco_rl_amd.rollout import RolloutChunk, bootstrap_truncated


def _hand_gae(rew, val, done, last_v, gamma, lam):
    T, N = rew.shape
    adv = np.zeros((T, N))
    for n in range(N):
        g = 0.0
        for t in reversed(range(T)):
            nv = last_v[n] if t == T - 1 else val[t + 1, n]
            nonterm = 1.0 - done[t, n]
            delta = rew[t, n] + gamma * nv * nonterm - val[t, n]
            g = delta + gamma * lam * nonterm * g
            adv[t, n] = g
    return adv


def test_truncation_is_bootstrapped_then_gae_matches_hand_computation():
    T, N, OD = 6, 3, 15
    gamma, lam = 0.99, 0.95
    rs = np.random.RandomState(0)
    chunk = RolloutChunk(T, N, OD, "cpu")
    o = OD
    chunk.buf.copy_(torch.from_numpy(rs.randn(T, N, OD + 10).astype(np.float32)))
    code = np.zeros((T, N), np.float32)
    code[2, 0] = 2.0            # env 0: truncated in the middle of the chunk
    code[4, 1] = 1.0            # env 1: genuinely terminated
    code[5, 2] = 2.0            # env 2: truncated on the last step of the chunk
    chunk.buf[..., o + 7] = torch.from_numpy(code)
    tobs = torch.from_numpy(rs.randn(T, N, OD).astype(np.float32))
    w = torch.from_numpy(rs.randn(OD).astype(np.float32))
    value_fn = lambda x: x @ w + 0.25
    raw_rew = chunk.buf[..., o + 6].clone().numpy().astype(np.float64)
    mask = bootstrap_truncated(chunk.buf[..., o + 6], chunk.buf[..., o + 7], tobs, value_fn, gamma)
    assert mask.sum().item() == 2
    rew = chunk.buf[..., o + 6].numpy().astype(np.float64)
    # only the truncated steps changed, by gamma * V(terminal obs)
    exp = raw_rew.copy()
    for (t, n) in ((2, 0), (5, 2)):
        exp[t, n] += gamma * (float(tobs[t, n] @ w) + 0.25)
    np.testing.assert_allclose(rew, exp, rtol=0, atol=1e-6)

    b = chunk.unpack()
    assert b["truncated"].sum().item() == 2 and b["dones"].sum().item() == 3
    b["last_obs"] = torch.from_numpy(rs.randn(N, OD).astype(np.float32))
    ppo = PPO(OD, "cpu", gamma=gamma, gae_lambda=lam, use_graph=False)
    ppo._alloc(b)
    for k in ("obs", "actions", "rewards", "dones", "values", "log_probs", "last_obs"):
        ppo._s[k].copy_(b[k])
    ppo._gae()
    with torch.no_grad():
        last_v = ppo.net.value(b["last_obs"]).numpy().astype(np.float64)
    hand = _hand_gae(rew, b["values"].numpy().astype(np.float64), b["dones"].numpy().astype(np.float64), last_v, gamma, lam)
    np.testing.assert_allclose(ppo._s["adv"].numpy(), hand, rtol=1e-5, atol=1e-5)
    # the truncated step's TD target really contains the bootstrap: delta = r + gamma V(tobs) - V_t (no V_{t+1} leak)
    d20 = raw_rew[2, 0] + gamma * (float(tobs[2, 0] @ w) + 0.25) - float(b["values"][2, 0])
    np.testing.assert_allclose(hand[2, 0], d20, rtol=1e-6)


def test_reported_mean_reward_is_the_env_reward_not_the_bootstrapped_one():
    """model selection / early stopping in main.py read stats["mean_reward"]: it must be the env's reward, unchanged by the
    gamma * V(terminal_observation) the collector adds on truncated steps (every episode end of Env01/02/06 is a truncation)"""
    T, N, OD = 8, 4, 15
    rs = np.random.RandomState(1)
    chunk = RolloutChunk(T, N, OD, "cpu")
    chunk.buf.copy_(torch.from_numpy(rs.randn(T, N, OD + 10).astype(np.float32)))
    code = np.zeros((T, N), np.float32); code[3, :] = 2.0; code[7, 1] = 2.0
    chunk.buf[..., OD + 7] = torch.from_numpy(code)
    raw_mean = chunk.buf[..., OD + 6].mean().clone()
    tobs = torch.from_numpy(rs.randn(T, N, OD).astype(np.float32))
    bootstrap_truncated(chunk.buf[..., OD + 6], chunk.buf[..., OD + 7], tobs, lambda x: torch.full((x.shape[0],), 50.0), 0.99)   # an over-estimating critic
    b = chunk.unpack(); b["last_obs"] = torch.zeros(N, OD); b["raw_reward_mean"] = raw_mean
    ppo = PPO(OD, "cpu", use_graph=False)
    stats = ppo.update(b)
    assert abs(stats["mean_reward"] - float(raw_mean)) < 1e-6
    assert stats["mean_bootstrapped_reward"] > stats["mean_reward"] + 5.0          # 5 of 32 steps got +49.5
    del b["raw_reward_mean"]                                                         # a caller without the statistic: falls back to the buffer's mean
    assert abs(ppo.update(b)["mean_reward"] - stats["mean_bootstrapped_reward"]) < 1e-5


def test_eval_and_stop_callbacks_follow_sb3_semantics():
    """EvalCallback + StopTrainingOnRewardThreshold + StopTrainingOnNoModelImprovement as the reference wires them (main.py:211-225)"""
    from so100_mujoco_rl_amd.callbacks import EvalCallback, StopTrainingOnRewardThreshold, StopTrainingOnNoModelImprovement
    scores = iter([1.0, 3.0, 2.0, 2.5, 2.9, 2.0, 1.0, 0.5, 0.1, 10.0])
    saved = []
    cb = EvalCallback(lambda: next(scores), lambda: saved.append(cb.best_mean_reward), eval_every=2,
                      on_new_best=StopTrainingOnRewardThreshold(6.0), after_eval=StopTrainingOnNoModelImprovement(max_no_improvement_evals=3, min_evals=2))
    out = [cb.step() for _ in range(14)]
    # evaluations happen on calls 2, 4, 6, ...: bests 1.0 and 3.0 are saved; from the 2nd evaluation on, non-improving evaluations are counted:
    # 2.0, 2.5, 2.9 (3 in a row: still allowed), 2.0 is the 4th -> stop on call 12
    assert saved == [1.0, 3.0] and cb.n_evals == 7 and cb.best_mean_reward == 3.0
    assert out[:11] == [True]*11 and out[11] is False
    cb2 = EvalCallback(lambda: 7.0, lambda: None, eval_every=1, on_new_best=StopTrainingOnRewardThreshold(6.0))
    assert cb2.step() is False                               # a new best at / above the threshold stops the training
    cb3 = EvalCallback(lambda: 1.0, lambda: None, eval_every=1, after_eval=StopTrainingOnNoModelImprovement(5, 10000))
    assert all(cb3.step() for _ in range(50))                # the reference's min_evals = 10000: this rule practically never fires
