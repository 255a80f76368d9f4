"""The reference's training callbacks, restated for the built-in learner (ref: main.py:211-232 builds them from Stable-Baselines3:
EvalCallback(eval_freq=20000, callback_on_new_best=StopTrainingOnRewardThreshold, callback_after_eval=StopTrainingOnNoModelImprovement(
max_no_improvement_evals=5, min_evals=10000), best_model_save_path=...)).  When SB3 is importable main.py hands SB3's own classes to
`model.learn`; these are used by the built-in PPO driver otherwise.  Pure bookkeeping: no torch, no GPU (tests/test_ppo_cpu.py).

Not restated: the reference's DDPG branch (main.py:38-55) -- its NormalActionNoise has dimension 2 (a leftover of another robot) against
this task's 6-dimensional action space, so SB3 raises on the first step; there is no behaviour to reproduce."""


class StopTrainingOnRewardThreshold:
    """SB3 semantics: called when the evaluation found a new best mean reward; training stops once it reaches the threshold."""

    def __init__(self, reward_threshold):
        self.reward_threshold = float(reward_threshold)

    def on_new_best(self, best_mean_reward):
        return not (best_mean_reward >= self.reward_threshold)          # continue_training


class StopTrainingOnNoModelImprovement:
    """SB3 semantics: called after every evaluation; from the `min_evals`-th call on, more than `max_no_improvement_evals`
    consecutive evaluations without a new best stop the training."""

    def __init__(self, max_no_improvement_evals=5, min_evals=10000):
        self.max_no_improvement_evals = int(max_no_improvement_evals); self.min_evals = int(min_evals)
        self.last_best_mean_reward = -float("inf"); self.no_improvement_evals = 0; self.n_calls = 0

    def after_eval(self, best_mean_reward):
        self.n_calls += 1
        continue_training = True
        if self.n_calls >= self.min_evals:
            if best_mean_reward > self.last_best_mean_reward:
                self.no_improvement_evals = 0
            else:
                self.no_improvement_evals += 1
                if self.no_improvement_evals > self.max_no_improvement_evals:
                    continue_training = False
        self.last_best_mean_reward = best_mean_reward
        return continue_training


class EvalCallback:
    """SB3 semantics: every `eval_every` calls, `evaluate()` -> mean episode reward of the DETERMINISTIC policy over `n_eval_episodes`
    episodes; a new best is saved through `save_best()` and reported to `on_new_best`; `after_eval` runs after every evaluation.
    `step()` returns False when training should stop."""

    def __init__(self, evaluate, save_best, eval_every, on_new_best=None, after_eval=None, log=None):
        self.evaluate, self.save_best, self.eval_every = evaluate, save_best, max(1, int(eval_every))
        self.on_new_best, self.after_eval, self.log = on_new_best, after_eval, log
        self.best_mean_reward = -float("inf"); self.last_mean_reward = float("nan"); self.n_calls = 0; self.n_evals = 0

    def step(self):
        self.n_calls += 1
        if self.n_calls % self.eval_every != 0:
            return True
        mean_reward = float(self.evaluate()); self.last_mean_reward = mean_reward; self.n_evals += 1
        continue_training = True
        if self.log:
            self.log(f"Eval num_evals={self.n_evals}, episode_reward={mean_reward:.2f}")
        if mean_reward > self.best_mean_reward:
            if self.log:
                self.log("New best mean reward!")
            self.best_mean_reward = mean_reward
            self.save_best()
            if self.on_new_best is not None:
                continue_training = self.on_new_best.on_new_best(self.best_mean_reward)
        if self.after_eval is not None:
            continue_training = self.after_eval.after_eval(self.best_mean_reward) and continue_training
        return continue_training
