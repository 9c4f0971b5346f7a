"""Long GD training run of the fused closed loop with per-epoch diagnostics: flagged (non-converged) environments, gradient norm before
the clip, largest |u|, loss quantiles over the batch.  python tools/archive/gd_divergence_probe.py [epochs] [lr]"""
import os, sys, json, math
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
from train_tactile_push_gd_batched import draw_episode
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode
from tactilesimulation_amd.algorithms.batched_gd import Actor
from tactilesimulation_amd.workloads import asset
E = int(sys.argv[1]) if len(sys.argv) > 1 else 300
LR = float(sys.argv[2]) if len(sys.argv) > 2 else 5e-3
B, T, dev, dt = 4096, 100, "cuda:0", torch.float32
env = BatchedTactilePushEnv(asset("pusher"), B, device=dev, dtype=dt, gradient=True, seed=0, tape_steps=T)
torch.manual_seed(0)
actor = Actor(dtype=dt).to(dev)
opt = torch.optim.Adam(actor.parameters(), lr=LR, betas=(0.7, 0.95))
rng = np.random.default_rng(0)
ep = FusedPushEpisode(env, actor, T)
rows = []
for epoch in range(E):
    for g in opt.param_groups: g["lr"] = (1e-5 - LR) * float(epoch / E) + LR
    q0, goal, dist = draw_episode(rng, B, T, dev, dt, 1)
    for p in actor.parameters(): p.grad = None
    loss = ep.rollout(q0, goal, dist); ep.backward()
    params = [p for p in actor.parameters() if p.grad is not None]
    for p in params: p.grad.div_(B)
    gn = float(torch.nn.utils.clip_grad_norm_(params, 1.0))
    opt.step()
    flagged = int((ep.status != 0).sum()); gmax = float(torch.tensor(env.sim.last_gnorm()).max())
    # per-environment loss
    g = ep.goal.unsqueeze(0); dp = ep.q[:, :, 3:5] - g[:, :, 0:2]; dr = ep.q[:, :, 6] - g[:, :, 2]; dtt = ep.var[:, :, 0:3] - ep.var[:, :, 3:6]
    k = (36.0 / math.pi) ** 2
    le = ((dp ** 2).sum(2) * 100.0 + dr ** 2 * (0.1 * k) + (dtt ** 2).sum(2) * 2500.0 + (ep.u ** 2).sum(2) * 0.1).sum(0)
    finite = bool(torch.isfinite(ep.dobs_tac).all() and torch.isfinite(ep.g1).all())
    row = dict(epoch=epoch, loss=float(loss) / B, flagged=flagged, gnorm_max=gmax, grad_norm=gn, umax=float(ep.u.abs().max()), qmax=float(ep.q.abs().max()),
               loss_med=float(le.median()), loss_p99=float(le.quantile(0.99)), loss_max=float(le.max()), finite=finite, g1max=float(ep.g1.abs().max()))
    rows.append(row)
    if epoch % 10 == 0 or row["loss"] > 400 and epoch > 60: print(json.dumps(row), flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "gd_divergence_probe.json"), "w"))
