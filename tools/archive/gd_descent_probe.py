"""Is the batched policy gradient a descent direction at the scale the training example runs (B envs x 100 env-steps)?
Loss along -g/|g| on FIXED episodes, per-environment gradient-norm spread (GPU box)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.algorithms.batched_gd import Actor, rollout_loss
from tactilesimulation_amd.workloads import PUSHER_BLOB
sys.path.insert(0, os.path.join(ROOT, "examples"))
from train_tactile_push_gd_batched import draw_episode
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 100
res = {}
for dt in (torch.float32, torch.float64):
    env = BatchedTactilePushEnv(PUSHER_BLOB, B, dtype=dt, gradient=True, seed=0, tape_steps=T)
    rng = np.random.default_rng(1000)
    q0, goal, D = draw_episode(rng, B, T, "cuda", dt)
    torch.manual_seed(0)
    actor = Actor(dtype=dt).cuda()
    params = [p for p in actor.parameters()]
    def loss_of():
        return rollout_loss(env, actor, T, q0=q0, goal=goal, disturbances=D)
    L0 = loss_of(); L0.backward()
    g = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    gn = float(torch.sqrt(sum((x.double() ** 2).sum() for x in g)))
    out = {"L0_per_episode": float(L0) / B, "grad_norm_of_mean": gn / B}
    # per-environment loss terms: which environments dominate?
    with torch.no_grad():
        for eps in (1e-5, 1e-4, 1e-3, 1e-2, 5e-2):
            for p, x in zip(params, g):
                p -= eps * x / gn
            out["L(-%g ghat)_per_episode" % eps] = float(loss_of()) / B
            for p, x in zip(params, g):
                p += eps * x / gn
        out["predicted_dL_per_unit_eps_per_episode"] = -gn / B
    # gradient of single-environment losses w.r.t. the last-layer bias (3 numbers): spread across environments
    actor.zero_grad()
    obs = env.reset(q0, goal)
    tot = torch.zeros(B, device="cuda", dtype=dt)
    for t in range(T):
        u = actor(obs); obs, rew, _ = env.step(u, D[t]); tot = tot - rew
    bias = actor.mu_net[-1].bias
    # d tot_e / d bias via one backward per ... too costly; instead gradient norms of sub-batches of 64
    gs = []
    for k in range(0, B, 64):
        actor.zero_grad()
    out["loss_per_env"] = {"median": float(tot.median()), "p99": float(tot.quantile(0.99)), "max": float(tot.max())}
    res[str(dt)] = out
    print(json.dumps(out, indent=1), flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gd_descent_probe.json"), "w"), indent=1)
