"""Experiment driver with the reference's CLI surface (ref: /root/reference/src/so100_mujoco_rl/main.py:241-284):

    python -m so100_mujoco_rl_amd.main -a PPO [-m MODEL] train  -e Env01-v1 [--envs 4096] [--iters N]
    python -m so100_mujoco_rl_amd.main -a PPO [-m MODEL] test   -e Env01-v1 [--show-io] [--show-i]
    python -m so100_mujoco_rl_amd.main -a PPO [-m MODEL] record -e Env01-v1
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m so100_mujoco_rl_amd.main -a PPO train -e Env01-v1     # 8 x 4096 envs

Same directory layout (models/ logs/ movies/), default model path models/{env}_{algo}/best_model.*, reward thresholds
(6000 / 8000, ref: __init__.py:9,16) and checkpoint naming ({env}_{algo}_cp_*, ref: main.py:227-232).  Differences, all
forced by the environment being a batched GPU simulator: N envs instead of 1; the learner is stable-baselines3 when it
is importable and the built-in PPO (ppo.py) or DDPG (ddpg.py; ref: main.py:38-55), both with SB3-compatible state_dicts, otherwise; `record` writes a state trajectory
(.npz) because there is no rasteriser (viewer / video are out of scope).
"""
import logging
import os
import time

import click
import numpy as np
import torch

from . import constants as K
import torch.distributed as dist

from .callbacks import EvalCallback, StopTrainingOnNoModelImprovement, StopTrainingOnRewardThreshold
from .collector import RolloutCollector
from .ddpg import DDPG, DDPGPolicy
from .lib import F_REFERENCE
from .ppo import PPO, ActorCritic
from .rollout import broadcast_policy
from .vec_env import So100VecEnv, kind_from_id

logging.basicConfig(level=logging.INFO, format="%(message)s")
logger = logging.getLogger("so100")

MODEL_DIR, LOG_DIR, RECORDING_DIR = "models", "logs", "movies"          # ref: main.py:28-30


def _have_sb3():
    try:
        import stable_baselines3  # noqa: F401
        return True
    except Exception:
        return False


def _default_model_path(environment, algorithm):
    return os.path.join(MODEL_DIR, f"{environment}_{algorithm}", "best_model.zip" if _have_sb3() else "best_model.pt")


NATIVE_ALGORITHMS = ("PPO", "DDPG")


def _load_native(path, obs_dim, device, algorithm="PPO"):
    net = (DDPGPolicy(obs_dim) if algorithm == "DDPG" else ActorCritic(obs_dim)).to(device)
    net.load_state_dict(torch.load(path, map_location=device, weights_only=True))
    return net


@click.group()
@click.option("-a", "--algorithm", required=True, type=str, default="PPO", help="algorithm (PPO and DDPG natively; any Stable-Baselines3 name when SB3 is installed)")
@click.option("-m", "--model", default=None, type=click.Path(exists=False), help="Path to model file")
@click.pass_context
def cli(ctx, algorithm, model):
    if not _have_sb3() and algorithm not in NATIVE_ALGORITHMS:
        raise RuntimeError(f"algorithm {algorithm} needs stable-baselines3, which is not installed; the built-in learners are {' and '.join(NATIVE_ALGORITHMS)}")
    ctx.ensure_object(dict)
    ctx.obj["ALGORITHM_NAME"] = algorithm
    ctx.obj["MODEL_PATH"] = model
    for d in (MODEL_DIR, LOG_DIR, RECORDING_DIR):
        os.makedirs(d, exist_ok=True)


@cli.command(name="train", help="Train a model with a given environment")
@click.option("-e", "--environment", required=True, type=str, help="id of the environment (eg; Env01-v1)")
@click.option("--envs", default=4096, type=int, help="envs stepped in parallel on the GPU")
@click.option("--iters", default=0, type=int, help="PPO updates (0 = until the reward threshold / no improvement)")
@click.option("--seed", default=0, type=int)
@click.pass_context
def train(ctx, environment, envs, iters, seed):
    algorithm = ctx.obj["ALGORITHM_NAME"]
    kind = kind_from_id(environment)
    # Multi-GPU (one process per GPU under torchrun): rank r steps global envs [r*envs, (r+1)*envs); once per rollout chunk the
    # packed rollout goes to rank 0 over RCCL (the path's ONE collective), rank 0 learns, the policy is broadcast back.
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("SO100_FORCE_DIST") == "1"      # SO100_FORCE_DIST: the same code path on a one-GPU box
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if _have_sb3():
            raise RuntimeError("multi-GPU training uses the built-in PPO learner; run SB3 single-GPU")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    lead = rank == 0
    env = So100VecEnv(environment, envs, device=torch.device("cuda", local) if distributed else None, flags=F_REFERENCE, seed=seed,
                      env_id_offset=rank * envs, stagger_episodes=True)
    save_dir = os.path.join(MODEL_DIR, f"{environment}_{algorithm}")
    os.makedirs(save_dir, exist_ok=True)
    if lead:
        logger.info("Starting training process"); logger.info(f"Algorithm: {algorithm}"); logger.info(f"Environment: {environment} x {envs}" + (f" x {world} GPUs" if distributed else ""))
    if _have_sb3():                                          # unchanged SB3 learner over the batched VecEnv (ref: main.py:199-238)
        import stable_baselines3
        from stable_baselines3.common.callbacks import CheckpointCallback
        cls = getattr(stable_baselines3, algorithm)
        model_file = ctx.obj["MODEL_PATH"]
        model = cls.load(model_file, env=env, tensorboard_log=LOG_DIR) if model_file else cls("MlpPolicy", env, verbose=1, device="cuda", tensorboard_log=LOG_DIR)
        cb = CheckpointCallback(save_freq=max(1, 40000 // envs), save_path=save_dir, name_prefix=f"{environment}_{algorithm}_cp_", verbose=2)
        model.learn(total_timesteps=int(1e10) if iters == 0 else iters * 64 * envs, tb_log_name=f"{environment}_{algorithm}", callback=cb)
        model.save(os.path.join(save_dir, "best_model"))
        return
    if algorithm == "DDPG":
        if distributed:
            raise RuntimeError("multi-GPU training uses the built-in PPO learner; DDPG runs on one GPU")
        return _train_ddpg(env, environment, ctx.obj["MODEL_PATH"], save_dir, iters, seed, kind)
    learner = PPO(env.sim.obs_dim, env.device, seed=seed)
    if ctx.obj["MODEL_PATH"]:
        if not os.path.isfile(ctx.obj["MODEL_PATH"]):
            raise RuntimeError(f"Model file {ctx.obj['MODEL_PATH']} does not exist")
        learner.net.load_state_dict(torch.load(ctx.obj["MODEL_PATH"], map_location=env.device, weights_only=True))
        logger.info(f"Model: starting with {ctx.obj['MODEL_PATH']}")
    else:
        logger.info("Model: starting with new model")
    if distributed:
        broadcast_policy(list(learner.net.state_dict().values()), src=0)          # every rank starts from rank 0's weights
    col = RolloutCollector(env, learner.net.state_dict(), T=64)
    threshold = K.REWARD_THRESHOLD[kind]                     # StopTrainingOnRewardThreshold (ref: main.py:211; the registered threshold of the env id)
    steps, t0, it = 0, time.time(), 0
    ep_sum = ep_cnt = 0.0
    stop = torch.zeros(1, device=env.device)
    eval_cb = None
    if lead:
        # ref: main.py:211-225 -- EvalCallback(best_model_save_path, callback_on_new_best = StopTrainingOnRewardThreshold, callback_after_eval =
        # StopTrainingOnNoModelImprovement(max_no_improvement_evals=5, min_evals=10000)), restated for the built-in learner (callbacks.py).  The reference
        # evaluates every 20 000 timesteps of ONE env; here an update is 64 x envs timesteps, so an evaluation every EVAL_EVERY updates.
        eval_env = So100VecEnv(environment, N_EVAL_EPISODES, device=env.device, flags=F_REFERENCE, seed=seed + 1000)
        eval_col = RolloutCollector(eval_env, learner.net.state_dict(), T=64, bootstrap_truncated=False)
        eval_cb = EvalCallback(lambda: _evaluate(eval_env, eval_col, learner), lambda: torch.save(learner.net.state_dict(), os.path.join(save_dir, "best_model.pt")),
                               EVAL_EVERY, on_new_best=StopTrainingOnRewardThreshold(threshold), after_eval=StopTrainingOnNoModelImprovement(5, 10000), log=logger.info)
    while True:
        b = col.collect(gather_dst=0 if distributed else None)
        it += 1
        if lead:
            done = (b["dones"] > 0).any(0)[:envs]            # episode statistics from this rank's own envs
            if done.any():                                   # ep_return holds the return of the latest episode that ended in the chunk
                ep_sum += env.sim.ep_return[done].sum().item(); ep_cnt += int(done.sum().item())
            stats = learner.update(b)
            steps += b["rewards"].numel()
            if it % 10 == 0:
                mean_ep = ep_sum / ep_cnt if ep_cnt else float("nan")
                logger.info(f"iter {it:5d}  timesteps {steps/1e6:8.1f} M  reward/step {stats['mean_reward']:+.4f}  ep_rew_mean {mean_ep:9.2f}  "
                            f"value_loss {stats['value_loss']:.4f}  fps {steps/(time.time()-t0)/1e6:.1f} M")
                dropped = int(env.sim.contacts_dropped().max().item())
                if dropped > 0:                              # over the contact budget: this step deviates from the reference model (MuJoCo keeps every contact)
                    logger.warning(f"contact budget exceeded: up to {dropped} contacts dropped in an env of the last step")
                ep_sum = ep_cnt = 0.0
            if not eval_cb.step():
                logger.info(f"Stopping training: best evaluation reward {eval_cb.best_mean_reward:.1f} (threshold {threshold})"); stop.fill_(1.0)
            if it % 40 == 0:                                 # CheckpointCallback (ref: main.py:227-232)
                torch.save(learner.net.state_dict(), os.path.join(save_dir, f"{environment}_{algorithm}_cp__{steps}_steps.pt"))
            if iters and it >= iters:
                stop.fill_(1.0)
        if distributed:
            broadcast_policy(list(learner.net.state_dict().values()) + [stop], src=0)
        col.load_policy(learner.net.state_dict())
        if stop.item() > 0:
            break
    if lead and eval_cb.n_evals == 0:                        # a short run (--iters): one evaluation at the end, so that best_model exists
        eval_cb.eval_every = 1; eval_cb.step()
    if lead:
        torch.save(learner.net.state_dict(), os.path.join(save_dir, "last_model.pt"))
        logger.info(f"done: {steps/1e6:.1f} M timesteps in {time.time()-t0:.1f} s; best evaluation reward {eval_cb.best_mean_reward:.1f} ({eval_cb.n_evals} evaluations); models in {save_dir}")
    if distributed:
        dist.barrier(); dist.destroy_process_group()


def _train_ddpg(env, environment, model_path, save_dir, iters, seed, kind):
    """The reference's DDPG branch (ref: main.py:38-55) on the built-in learner (ddpg.py): off-policy, one vector step at a time through
    the tensor API; an "iter" is 64 vector steps (the PPO driver's chunk), evaluation / stop / checkpoint callbacks as in train()."""
    learner = DDPG(env.sim.obs_dim, env.device, seed=seed, buffer_size=max(1_000_000, 16 * env.num_envs), gradient_steps=DDPG_GRADIENT_STEPS)
    if model_path:
        if not os.path.isfile(model_path):
            raise RuntimeError(f"Model file {model_path} does not exist")
        learner.net.load_state_dict(torch.load(model_path, map_location=env.device, weights_only=True))
        logger.info(f"Model: starting with {model_path}")
    else:
        logger.info("Model: starting with new model")
    threshold = K.REWARD_THRESHOLD[kind]
    eval_env = So100VecEnv(environment, N_EVAL_EPISODES, device=env.device, flags=F_REFERENCE, seed=seed + 1000)
    eval_cb = EvalCallback(lambda: _evaluate_actor(eval_env, learner.net.mean_action), lambda: torch.save(learner.net.state_dict(), os.path.join(save_dir, "best_model.pt")),
                           EVAL_EVERY, on_new_best=StopTrainingOnRewardThreshold(threshold), after_eval=StopTrainingOnNoModelImprovement(5, 10000), log=logger.info)
    obs, steps, it, t0 = None, 0, 0, time.time()
    while True:
        obs, stats = learner.learn_steps(env, 64, obs)
        it += 1; steps += 64 * env.num_envs
        if it % 10 == 0:
            logger.info(f"iter {it:5d}  timesteps {steps/1e6:8.1f} M  reward/step {stats['mean_reward']:+.4f}  critic_loss {stats.get('critic_loss', float('nan')):.4f}  "
                        f"updates {learner.n_updates}  fps {steps/(time.time()-t0)/1e6:.2f} M")
        if not eval_cb.step():
            logger.info(f"Stopping training: best evaluation reward {eval_cb.best_mean_reward:.1f} (threshold {threshold})"); break
        if it % 40 == 0:
            torch.save(learner.net.state_dict(), os.path.join(save_dir, f"{environment}_DDPG_cp__{steps}_steps.pt"))
        if iters and it >= iters:
            break
    if eval_cb.n_evals == 0:
        eval_cb.eval_every = 1; eval_cb.step()
    torch.save(learner.net.state_dict(), os.path.join(save_dir, "last_model.pt"))
    logger.info(f"done: {steps/1e6:.1f} M timesteps in {time.time()-t0:.1f} s; best evaluation reward {eval_cb.best_mean_reward:.1f} ({eval_cb.n_evals} evaluations); models in {save_dir}")


@torch.no_grad()
def _evaluate_actor(eval_env, act_fn):
    """evaluate_policy(deterministic=True) through the stepwise tensor API (any policy: obs -> action): every env's first episode."""
    obs = eval_env.reset_tensor()
    n = eval_env.num_envs
    ret = torch.zeros(n, device=eval_env.device); alive = torch.ones(n, dtype=torch.bool, device=eval_env.device)
    for _ in range(eval_env.sim.cfg.max_episode_steps + 1):
        obs, r, d, _tr = eval_env.step_tensor(act_fn(obs).clamp(-1, 1).contiguous())
        ret += torch.where(alive, r, torch.zeros_like(ret)); alive &= ~d.bool()
        if not bool(alive.any()):
            break
    return float(ret.mean().item())


DDPG_GRADIENT_STEPS = 4   # updates of 256 per VECTOR step (SB3: 1 per step of ONE env; a vector step adds `envs` transitions, see ddpg.py)
N_EVAL_EPISODES = 5       # SB3 EvalCallback's default n_eval_episodes (ref: main.py:217-224 passes none)
EVAL_EVERY = 50           # updates between evaluations (an update is 64 x envs timesteps; the reference's eval_freq is 20 000 timesteps of one env)


@torch.no_grad()
def _evaluate(eval_env, eval_col, learner):
    """SB3 evaluate_policy(deterministic=True) over N_EVAL_EPISODES episodes: one env per episode, the policy's MEAN action (log_std -> -30
    in the copy the kernels read), every env's FIRST episode after a reset; returns the mean episode reward."""
    sd = {k: v.clone() for k, v in learner.net.state_dict().items()}
    sd["log_std"] = torch.full_like(sd["log_std"], -30.0)
    eval_col.load_policy(sd)
    eval_env.reset_tensor(); eval_col._started = True
    n = eval_env.num_envs
    ret = torch.zeros(n, device=eval_env.device); alive = torch.ones(n, dtype=torch.bool, device=eval_env.device)
    for _ in range(eval_env.sim.cfg.max_episode_steps // eval_col.T + 2):
        b = eval_col.collect()
        for t in range(b["rewards"].shape[0]):
            ret += torch.where(alive, b["rewards"][t], torch.zeros_like(ret))
            alive &= ~(b["dones"][t] > 0)
        if not bool(alive.any()):
            break
    return float(ret.mean().item())


def _rollout_policy(environment, algorithm, model_file, n, steps, show_io, show_i, record_path=None):
    env = So100VecEnv(environment, n, flags=F_REFERENCE, seed=1)
    if model_file is None:
        model_file = _default_model_path(environment, algorithm)
    if not os.path.isfile(model_file):
        raise RuntimeError(f"Could not open model file: {model_file}")
    logger.info(f"Algorithm: {algorithm}"); logger.info(f"Environment: {environment}"); logger.info(f"Model: {model_file}")
    if model_file.endswith(".zip"):
        import stable_baselines3
        policy = getattr(stable_baselines3, algorithm).load(model_file, device="cuda").policy
        act_fn = lambda o: policy._predict(o, deterministic=True)
    else:
        net = _load_native(model_file, env.sim.obs_dim, env.device, algorithm)
        act_fn = net.mean_action
    obs = env.reset_tensor()
    total = 0.0; traj = []
    with torch.no_grad():
        for t in range(steps):
            a = act_fn(obs).clamp(-1, 1).contiguous()
            if (show_io or show_i) and t % 30 == 0:           # ref: main.py:110-113
                logger.info(str(obs[0].tolist() + (a[0].tolist() if show_io else [])) + ("," if show_i else ""))
            if record_path is not None:
                q, v = env.sim.get_state()
                traj.append(np.concatenate([q[:, 0].cpu().numpy(), v[:, 0].cpu().numpy(), obs[0].cpu().numpy(), a[0].cpu().numpy()]))
            obs, r, d, tr = env.step_tensor(a)
            total += r.mean().item()
    logger.info(f"mean reward/step over {steps} steps x {n} envs: {total/steps:+.4f}")
    if record_path is not None:
        np.savez(record_path, trajectory=np.stack(traj), layout="qpos[13] qvel[12] obs action[6] per step, env 0")
        logger.info(f"wrote {record_path} (state trajectory; no rasteriser in this build)")
    return total / steps


@cli.command(name="test", help="Test the current model")
@click.option("-e", "--environment", required=True, type=str)
@click.option("--show-io", is_flag=True, default=False, help="log model inputs and outputs")
@click.option("--show-i", is_flag=True, default=False, help="log model inputs in Python array syntax")
@click.option("--envs", default=256, type=int)
@click.option("--steps", default=1000, type=int)
@click.pass_context
def test(ctx, environment, show_io, show_i, envs, steps):
    logger.info("Starting test simulation")
    _rollout_policy(environment, ctx.obj["ALGORITHM_NAME"], ctx.obj["MODEL_PATH"], envs, steps, show_io, show_i)


@cli.command(name="record", help="Record a model with a given environment")
@click.option("-e", "--environment", required=True, type=str)
@click.pass_context
def record(ctx, environment):
    path = os.path.join(RECORDING_DIR, f"{environment}_{ctx.obj['ALGORITHM_NAME']}.npz")
    _rollout_policy(environment, ctx.obj["ALGORITHM_NAME"], ctx.obj["MODEL_PATH"], 1, 3000, False, False, record_path=path)   # 3000 steps, ref: main.py:161


if __name__ == "__main__":
    cli(obj={})
