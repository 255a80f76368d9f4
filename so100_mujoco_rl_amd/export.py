"""Policy export for deployment outside the simulator.

The reference's real-robot loop (ref: real.py:207-221) feeds an Env05-trained policy the observation
`[joint command x6, 5*cx, 5*cy]` (cx, cy = detected cube centre as fractions of the frame, -1 when lost) and applies
`joint += action * JOINT_STEP_SCALE`.  `export_policy` writes a self-contained TorchScript module computing the
deterministic action (mean, clipped to the action Box) from that observation; `real_observation` builds the
observation the same way, so a deployment script needs neither the simulator nor a learner library."""
import torch
import torch.nn as nn

from . import constants as K


class DeterministicPolicy(nn.Module):
    def __init__(self, obs_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(obs_dim, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh(), nn.Linear(64, 6))

    def forward(self, obs: torch.Tensor) -> torch.Tensor:
        return self.net(obs).clamp(-1.0, 1.0)


def export_policy(state_dict, path):
    """state_dict: an SB3 ActorCriticPolicy / ppo.ActorCritic state_dict.  Returns the scripted module."""
    w0 = state_dict["mlp_extractor.policy_net.0.weight"]
    m = DeterministicPolicy(w0.shape[1])
    with torch.no_grad():
        for dst, src in ((m.net[0], "mlp_extractor.policy_net.0"), (m.net[2], "mlp_extractor.policy_net.2"), (m.net[4], "action_net")):
            dst.weight.copy_(state_dict[src + ".weight"].detach().cpu()); dst.bias.copy_(state_dict[src + ".bias"].detach().cpu())
    scripted = torch.jit.script(m.eval())
    scripted.save(path)
    return scripted


def real_observation(joint_positions, detection):
    """ref: real.py:207-211.  detection = (cx, cy) in [0,1] or (-1,-1)."""
    return torch.tensor([*joint_positions, detection[0] * 5.0, detection[1] * 5.0], dtype=torch.float32)


def apply_action(joint_positions, action, alpha=0.6):
    """ref: real.py:216-228: step by JOINT_STEP_SCALE, then blend with the previous command (alpha 0.6)."""
    new = [q + float(a) * K.JOINT_STEP_SCALE for q, a in zip(joint_positions, action)]
    return [alpha * n + (1 - alpha) * q for n, q in zip(new, joint_positions)]
