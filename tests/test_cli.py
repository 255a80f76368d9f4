"""The experiment driver (so100_mujoco_rl_amd/main.py) mirrors the reference CLI (ref: main.py:241-284):
`-a ALGO [-m MODEL] train|test|record -e ENV_ID`, directories models/ logs/ movies/."""
import os

import numpy as np
import pytest
from click.testing import CliRunner

from so100_mujoco_rl_amd import main as drv


def test_cli_surface(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    r = CliRunner().invoke(drv.cli, ["--help"])
    assert r.exit_code == 0
    for word in ("train", "test", "record", "--algorithm", "--model"):
        assert word in r.output
    for cmd in ("train", "test", "record"):
        r = CliRunner().invoke(drv.cli, ["-a", "PPO", cmd, "--help"])
        assert r.exit_code == 0 and "--environment" in r.output
        assert all(os.path.isdir(d) for d in ("models", "logs", "movies"))     # ref: main.py:33-40
    r = CliRunner().invoke(drv.cli, ["-a", "PPO", "test", "--help"])
    assert "--show-io" in r.output and "--show-i" in r.output


def test_cli_covers_the_reference_surface(golden_dir):
    """meta.json["cli"] is the click surface of the reference's main.py, recorded by tests/golden/make_golden.py: every group
    option, command and command option of the reference exists here with the same flags, required-ness and default."""
    import json
    import click
    ref = json.load(open(os.path.join(golden_dir, "meta.json")))["cli"]

    def opts(cmd):
        return {tuple(sorted(p.opts)): p for p in cmd.params if isinstance(p, click.Option)}
    mine = opts(drv.cli)
    for o in ref["group"]:
        p = mine[tuple(o["opts"])]
        assert bool(p.required) == o["required"] and p.default == o["default"]
    assert set(ref["commands"]) <= set(drv.cli.commands)
    for name, ref_opts in ref["commands"].items():
        mine = opts(drv.cli.commands[name])
        for o in ref_opts:
            p = mine[tuple(o["opts"])]
            assert bool(p.required) == o["required"] and bool(p.is_flag) == o["is_flag"]
    assert ref["dirs"] == [drv.MODEL_DIR, drv.LOG_DIR, drv.RECORDING_DIR]


def test_cli_errors(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    if not drv._have_sb3():
        r = CliRunner().invoke(drv.cli, ["-a", "SAC", "train", "-e", "Env01-v1"])
        assert r.exit_code != 0 and "stable-baselines3" in str(r.exception)
        r = CliRunner().invoke(drv.cli, ["-a", "DDPG", "train", "--help"])        # the reference's DDPG branch (main.py:38-55) has a built-in learner
        assert r.exit_code == 0 and drv.NATIVE_ALGORITHMS == ("PPO", "DDPG")
    r = CliRunner().invoke(drv.cli, ["-a", "PPO", "train"])                     # -e is required
    assert r.exit_code == 2
    assert drv._default_model_path("Env01-v1", "PPO").startswith(os.path.join("models", "Env01-v1_PPO", "best_model"))


@pytest.mark.gpu
def test_cli_train_test_record(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    run = CliRunner()
    r = run.invoke(drv.cli, ["-a", "PPO", "train", "-e", "Env01-v1", "--envs", "1024", "--iters", "40"], catch_exceptions=False)
    assert r.exit_code == 0
    d = tmp_path / "models" / "Env01-v1_PPO"
    assert (d / "best_model.pt").is_file() and (d / "last_model.pt").is_file()
    assert any(f.name.startswith("Env01-v1_PPO_cp_") for f in d.iterdir())
    r = run.invoke(drv.cli, ["-a", "PPO", "test", "-e", "Env01-v1", "--envs", "64", "--steps", "64", "--show-io"], catch_exceptions=False)
    assert r.exit_code == 0
    r = run.invoke(drv.cli, ["-a", "PPO", "-m", str(d / "last_model.pt"), "train", "-e", "Env01-v1", "--envs", "256", "--iters", "2"],
                   catch_exceptions=False)                                       # resume from a checkpoint
    assert r.exit_code == 0
    r = run.invoke(drv.cli, ["-a", "PPO", "-m", str(tmp_path / "nope.pt"), "test", "-e", "Env01-v1"])
    assert r.exit_code != 0 and "Could not open model file" in str(r.exception)
    r = run.invoke(drv.cli, ["-a", "PPO", "record", "-e", "Env01-v1"], catch_exceptions=False)
    assert r.exit_code == 0
    traj = np.load(tmp_path / "movies" / "Env01-v1_PPO.npz")["trajectory"]
    assert traj.shape == (3000, 13 + 12 + 15 + 6) and np.isfinite(traj).all()


@pytest.mark.gpu
def test_cli_train_distributed_code_path(tmp_path, monkeypatch):
    """The multi-GPU branch of `train` (RCCL init, rollout gather to rank 0, policy + stop-flag broadcast, barrier) with a
    one-rank group: what can be exercised on a one-GPU box; the N > 1 collective itself is covered over gloo on the CPU."""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SO100_FORCE_DIST", "1"); monkeypatch.setenv("MASTER_PORT", "29577")
    r = CliRunner().invoke(drv.cli, ["-a", "PPO", "train", "-e", "Env02-v1", "--envs", "512", "--iters", "12"], catch_exceptions=False)
    assert r.exit_code == 0
    assert (tmp_path / "models" / "Env02-v1_PPO" / "last_model.pt").is_file()
    import torch.distributed as dist
    assert not dist.is_initialized()


@pytest.mark.gpu
def test_cli_ddpg_train_test(tmp_path, monkeypatch):
    """The reference's DDPG branch (ref: main.py:38-55) on the built-in learner: train -> best / last / checkpoint files -> resume -> test."""
    monkeypatch.chdir(tmp_path)
    run = CliRunner()
    r = run.invoke(drv.cli, ["-a", "DDPG", "train", "-e", "Env01-v1", "--envs", "256", "--iters", "40"], catch_exceptions=False)
    assert r.exit_code == 0
    d = tmp_path / "models" / "Env01-v1_DDPG"
    assert (d / "best_model.pt").is_file() and (d / "last_model.pt").is_file()
    assert any(f.name.startswith("Env01-v1_DDPG_cp_") for f in d.iterdir())
    import torch
    sd = torch.load(d / "last_model.pt", weights_only=True)
    assert tuple(sd["actor.mu.0.weight"].shape) == (300, 15) and tuple(sd["critic.qf0.0.weight"].shape) == (200, 21)
    assert all(torch.isfinite(v).all() for v in sd.values())
    r = run.invoke(drv.cli, ["-a", "DDPG", "test", "-e", "Env01-v1", "--envs", "64", "--steps", "64", "--show-io"], catch_exceptions=False)
    assert r.exit_code == 0
    r = run.invoke(drv.cli, ["-a", "DDPG", "-m", str(d / "last_model.pt"), "train", "-e", "Env01-v1", "--envs", "128", "--iters", "1"], catch_exceptions=False)
    assert r.exit_code == 0
