"""Policy export (obs convention of ref: real.py:207-221) and the sim checkpoint format."""
import numpy as np
import pytest
import torch

from so100_mujoco_rl_amd import export
from so100_mujoco_rl_amd.ppo import ActorCritic


def test_export_policy_roundtrip(tmp_path):
    torch.manual_seed(3)
    net = ActorCritic(8)
    with torch.no_grad():
        net.action_net.weight.mul_(100.0)                               # make the clip to [-1, 1] bite
    p = tmp_path / "policy.pt"
    export.export_policy(net.state_dict(), str(p))
    m = torch.jit.load(str(p))
    obs = torch.stack([export.real_observation([0.1 * i] * 6, (0.4, 0.6)) for i in range(5)] + [export.real_observation([0.0] * 6, (-1, -1))])
    assert obs.shape == (6, 8) and obs[0, 6].item() == pytest.approx(2.0) and obs[-1, 7].item() == -5.0
    want = net.mean_action(obs).clamp(-1, 1)
    assert torch.equal(m(obs), want) and (want.abs() == 1).any()
    q = export.apply_action([0.0] * 6, [1.0, -1.0, 0.5, 0, 0, 0])
    assert q[0] == pytest.approx(0.6 * 0.075) and q[1] == pytest.approx(-0.6 * 0.075) and q[3] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [1, 2, 5])
def test_sim_checkpoint_resume_bit_exact(tmp_path, kind):
    from so100_mujoco_rl_amd.lib import So100Sim, F_REFERENCE
    n = 512
    mk = lambda: So100Sim(kind, n, flags=F_REFERENCE, seed=11, max_episode_steps=40)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    acts = torch.rand(60, n, 6, device="cuda", generator=g) * 2 - 1
    a = mk(); a.reset()
    for t in range(30):
        a.step(acts[t])
    a.save_state(str(tmp_path / "sim.npz"))
    ref = []
    for t in range(30, 60):
        o, r, d, tr = a.step(acts[t]); ref.append((o.clone(), r.clone(), d.clone()))
    b = mk()                                                            # fresh handle, never reset
    b.load_state(str(tmp_path / "sim.npz"))
    for t in range(30, 60):
        o, r, d, tr = b.step(acts[t])
        assert torch.equal(o, ref[t - 30][0]) and torch.equal(r, ref[t - 30][1]) and torch.equal(d, ref[t - 30][2])
    qa, va = a.get_state(); qb, vb = b.get_state()
    assert torch.equal(qa, qb) and torch.equal(va, vb)
    z = np.load(tmp_path / "sim.npz")
    assert z["words"].shape == (len(a.field_names()), n) and z["words"].dtype == np.int32
    c = So100Sim(kind, n // 2, flags=F_REFERENCE)
    with pytest.raises(Exception, match="checkpoint is for"):
        c.load_state(str(tmp_path / "sim.npz"))


