"""Batched analytic-gradient policy optimisation (BPTT through the simulator) — the MI355X counterpart of the reference's
algorithms/gd.py:145-264: there, `num_episodes` episodes run serially through one environment per epoch; here they
run as one batch of environments on each GPU, and ranks (one per GPU) exchange nothing but the flat policy gradient
(tactilesimulation_amd.dist.allreduce_policy_grad_, RCCL over xGMI with the nccl backend).
"""
import torch

from ..dist import allreduce_policy_grad_


class Actor(torch.nn.Module):
    """Deterministic part of utils/model.py:123-151 DiagGaussianActor for gd_tactile.yaml: 393 -> 64 -> 64 -> 3, ELU,
    plus the (unused in deterministic mode) log-std vector: 29 574 parameters, the all-reduce payload of SURVEY.md §8e."""

    def __init__(self, obs_dim=393, act_dim=3, hidden=(64, 64), dtype=torch.float32):
        super().__init__()
        dims = (obs_dim,) + tuple(hidden)
        layers = []
        for a, b in zip(dims[:-1], dims[1:]):
            layers += [torch.nn.Linear(a, b), torch.nn.ELU()]
        layers.append(torch.nn.Linear(dims[-1], act_dim))
        self.mu_net = torch.nn.Sequential(*layers)
        self.logstd = torch.nn.Parameter(torch.full((act_dim,), -1.0))
        self.to(dtype)

    def forward(self, obs):
        return self.mu_net(obs)


def rollout_loss(env, actor, horizon, q0=None, goal=None, disturbances=None):
    """-sum of rewards of all environments over one episode (un-normalised; see train_epoch)."""
    obs = env.reset(q0, goal)
    total = obs.new_zeros(())
    for t in range(horizon):
        u = actor(obs)
        obs, rew, _ = env.step(u, None if disturbances is None else disturbances[t])
        total = total - rew.sum()
    return total


def train_epoch(env, actor, optimizer, horizon, global_episodes, grad_clip=1.0, **rollout_kw):
    """One optimiser step on `global_episodes` episodes (= sum over ranks of env.B): local BPTT, ONE all-reduce of the
    flat gradient, normalisation by the global episode count, then clip-by-global-norm and Adam (gd.py:157-164,258)."""
    optimizer.zero_grad(set_to_none=True)
    loss = rollout_loss(env, actor, horizon, **rollout_kw)
    loss.backward()
    allreduce_policy_grad_(list(actor.parameters()), global_episodes)
    if grad_clip:
        torch.nn.utils.clip_grad_norm_(actor.parameters(), grad_clip)
    optimizer.step()
    return float(loss.detach()) / env.B
