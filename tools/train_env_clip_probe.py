"""Does a per-environment clip of dL/du_t (every env-step's action gradient of every environment, norm <= tau) keep the batched GD training
on the descent where the plain algorithm (global clip only, algorithms/gd.py:157-160) leaves it?  Per-step autograd path, hook on the actions.
   python tools/train_env_clip_probe.py --seed 0 --tau 100"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
from train_tactile_push_gd_batched import draw_episode
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.algorithms.batched_gd import Actor
from tactilesimulation_amd.dist import allreduce_policy_grad_
ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=0); ap.add_argument("--tau", type=float, default=100.0); ap.add_argument("--epochs", type=int, default=300)
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--out", default=None)
a = ap.parse_args()
dev, B, T, dt = "cuda:0", a.batch, 100, torch.float32
env = BatchedTactilePushEnv(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"), B, device=dev, dtype=dt, gradient=True, seed=a.seed, tape_steps=T, observation_type="tactile_flatten")
torch.manual_seed(a.seed)
actor = Actor(obs_dim=env.obs_dim, dtype=dt).to(dev)
opt = torch.optim.Adam(actor.parameters(), lr=5e-3, betas=(0.7, 0.95))
rng = np.random.default_rng(a.seed)
q0, goal, dist_ = draw_episode(rng, B, T, dev, dt, 1)
curve = []
clipped = [0]


def clip(g):
    n = g.norm(dim=1, keepdim=True)
    f = torch.clamp(a.tau / n.clamp_min(1e-30), max=1.0)
    clipped[0] += int((f < 1).sum())
    return torch.nan_to_num(g * f, nan=0.0, posinf=0.0, neginf=0.0)


for epoch in range(a.epochs):
    for g in opt.param_groups:
        g["lr"] = (1e-5 - 5e-3) * float(epoch / a.epochs) + 5e-3
    q0, goal, dist_ = draw_episode(rng, B, T, dev, dt, 1)
    opt.zero_grad(set_to_none=True)
    clipped[0] = 0
    obs = env.reset(q0, goal); acc = None
    for t in range(T):
        u = actor(obs)
        if a.tau > 0:
            u.register_hook(clip)
        obs, rew, _ = env.step(u, dist_[t])
        acc = rew if acc is None else acc + rew
    loss = -acc.sum()
    loss.backward()
    params = list(actor.parameters())
    allreduce_policy_grad_(params, B)
    gn = float(torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0))
    opt.step()
    curve.append({"epoch": epoch, "loss_per_episode": float(loss) / B, "grad_norm_before_clip": gn, "clipped_env_steps": clipped[0]})
    if epoch % 10 == 0 or epoch == a.epochs - 1:
        print(json.dumps(curve[-1]), flush=True)
if a.out:
    json.dump({"seed": a.seed, "tau": a.tau, "curve": curve}, open(a.out, "w"))
