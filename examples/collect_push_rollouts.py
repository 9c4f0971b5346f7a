"""Roll-out collection on the batched TactilePush environment as cfg/ppo_tactile.yaml sets it up (the 393-64-64-3 DiagGaussianActor,
stochastic, observations normalised and clipped; num_steps 1024 per environment = 10 episodes and a bit) — with the policy INSIDE the
simulator's forward launch (FusedPushEpisode.collect: one launch per 100-step episode, no tape), the critic and the log-probabilities
evaluated afterwards in torch over the whole episode at once.  The PPO update itself (externals/pytorch-a2c-ppo-acktr-gail) is not
part of this repository; what it consumes is what this script leaves in `batch`.

    python examples/collect_push_rollouts.py --batch 4096 --episodes 10
    python -m torch.distributed.run --nproc-per-node 8 examples/collect_push_rollouts.py          # 8 x 4096 environments, no collective
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv                      # noqa: E402
from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode                      # noqa: E402
from tactilesimulation_amd.algorithms.batched_gd import Actor                                  # noqa: E402
from tactilesimulation_amd.utils.running_mean_std import RunningMeanStd                        # noqa: E402


def draw_episode_device(gen, B, T, dev, dt):
    """The per-episode draws of tactile_push_env.py:133-190 on the device (box offset, goal, a force on the box with p = 0.5 at every env-step)."""
    U = lambda *shape: torch.rand(*shape, device=dev, dtype=dt, generator=gen)
    q0 = torch.zeros(B, 7, device=dev, dtype=dt); q0[:, 1] = -0.001; q0[:, 4] = U(B) * 0.04 - 0.02
    goal = torch.empty(B, 3, device=dev, dtype=dt)
    goal[:, 0] = 0.15 + 0.1 * U(B); goal[:, 1] = -0.2 + 0.4 * U(B)
    goal[:, 2] = goal[:, 1] * math.pi + (U(B) * 2.0 - 1.0) * (math.pi / 16.0)
    dist = (U(T, B, 1) < 0.5) * (U(T, B, 2) * 2.0 - 1.0)
    return q0, goal, dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--episodes", type=int, default=10)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev, dt = "cuda:%d" % local, (torch.float32 if a.dtype == "f32" else torch.float64)
    B, T = a.batch, a.horizon
    env = BatchedTactilePushEnv(a.model, B, device=dev, dtype=dt, gradient=False, seed=a.seed + rank, tape_steps=1)
    torch.manual_seed(a.seed)
    actor = Actor(dtype=dt).to(dev)
    with torch.no_grad():
        actor.logstd.fill_(0.0)                                     # actor_logstd_init: 0 (ppo_tactile.yaml)
    critic = torch.nn.Sequential(torch.nn.Linear(393, 64), torch.nn.ELU(), torch.nn.Linear(64, 64), torch.nn.ELU(), torch.nn.Linear(64, 1)).to(dev, dt)
    rms = RunningMeanStd(shape=(393,), device=dev)                  # norm_obs: statistics updated BETWEEN episodes, frozen inside one
    ep = FusedPushEpisode(env, actor, T)
    gen = torch.Generator(device=dev).manual_seed(a.seed + rank)
    steps, t_launch = 0, 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for e in range(a.episodes):
        if e == 1:                                                   # the first episode pays the one-off initialisations
            torch.cuda.synchronize(); t0 = time.perf_counter(); steps, t_launch = 0, 0.0
        q0, goal, dist = draw_episode_device(gen, B, T, dev, dt)
        eps = torch.randn(T, B, 3, device=dev, dtype=dt, generator=gen)
        stats = dict(obs_mean=rms.mean.to(dt), obs_var=rms.var.to(dt), obs_clip=10.0) if e > 0 else {}
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        out = ep.collect(q0, goal, dist, eps=eps, **stats)
        ev1.record()
        with torch.no_grad():
            obs = out["obs"]                                        # [T, B, 393] raw
            rms.update(obs.reshape(-1, 393))
            obs_n = torch.clamp((obs - rms.mean.to(dt)) / torch.sqrt(rms.var.to(dt) + 1e-8), -10.0, 10.0) if e > 0 else obs
            value = critic(obs_n).squeeze(-1)                       # [T, B]
            std = torch.exp(actor.logstd)
            logp = (-0.5 * eps ** 2 - actor.logstd - 0.5 * math.log(2.0 * math.pi)).sum(-1)      # log N(action; mean, std), action = mean + std eps
        batch = {"obs": obs_n, "action": out["action"], "reward": out["reward"], "value": value, "logp": logp}
        steps += B * T
        torch.cuda.synchronize(); t_launch += ev0.elapsed_time(ev1) * 1e-3
    el = time.perf_counter() - t0
    print("rank %d: %d environments x %d episodes x %d env-steps in %.2f s = %.2f M env-steps/s (the roll-outs alone: %.2f M; the rest: episode draws, "
          "statistics, critic); mean reward per env-step %.3f, flagged environments in the last episode %d"
          % (rank, B, max(a.episodes - 1, 1), T, el, steps / el / 1e6, steps / max(t_launch, 1e-9) / 1e6, float(batch["reward"].mean()), int((ep.status != 0).sum())))


if __name__ == "__main__":
    main()
