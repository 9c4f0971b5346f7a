"""Round 6: the fused closed-loop GD training (examples/train_tactile_push_gd_batched.py, fp32) leaves the descent on some seeds (seed 0 at epoch 97,
seed 2 at 109) while the per-step graph and the fp64 fused loop train through.  This probe trains fused fp32 and, from --from-epoch on, computes the
policy gradient of the SAME policy on the SAME episode three ways before every optimiser step — fused fp32 (what training uses), per-step autograd
fp32, fused fp64 — and prints norms, cosines and the largest per-parameter differences (GPU box).
   python tools/train_divergence_probe.py --seed 0 --from-epoch 88 --until 100"""
import argparse, copy, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
from train_tactile_push_gd_batched import draw_episode
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.algorithms.batched_gd import Actor, rollout_loss
from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode
from tactilesimulation_amd.dist import allreduce_policy_grad_

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=0); ap.add_argument("--from-epoch", type=int, default=88); ap.add_argument("--until", type=int, default=100)
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--epochs", type=int, default=300); ap.add_argument("--out", default=None)
a = ap.parse_args()
dev, B, T = "cuda:0", a.batch, 100
model = os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz")


def make(dtype):
    env = BatchedTactilePushEnv(model, B, device=dev, dtype=dtype, gradient=True, seed=a.seed, tape_steps=T, observation_type="tactile_flatten")
    torch.manual_seed(a.seed)
    actor = Actor(obs_dim=env.obs_dim, dtype=dtype).to(dev)
    return env, actor


env32, actor = make(torch.float32)
ep32 = FusedPushEpisode(env32, actor, T)
env32b, actor32b = make(torch.float32)            # per-step autograd, fp32
env64, actor64 = make(torch.float64)
ep64 = FusedPushEpisode(env64, actor64, T)
opt = torch.optim.Adam(actor.parameters(), lr=5e-3, betas=(0.7, 0.95))
rng = np.random.default_rng(a.seed)
q0, goal, dist_ = draw_episode(rng, B, T, dev, torch.float32, 1)
rows = []


def flat(params):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().double().reshape(-1) for p in params])


for epoch in range(a.until + 1):
    for g in opt.param_groups:
        g["lr"] = (1e-5 - 5e-3) * float(epoch / a.epochs) + 5e-3
    nq0, ngoal, ndist = draw_episode(rng, B, T, dev, torch.float32, 1)
    q0.copy_(nq0); goal.copy_(ngoal); dist_.copy_(ndist)
    for p in actor.parameters():
        p.grad = None
    loss = ep32.rollout(q0, goal, dist_)
    ep32.backward()
    params = list(actor.parameters())
    allreduce_policy_grad_(params, B)
    if epoch >= a.from_epoch:
        g32 = flat(params)
        st32 = env32.sim.last_evals().max() if hasattr(env32.sim, "last_evals") else None
        # the same policy, the same episode: per-step autograd fp32
        actor32b.load_state_dict(actor.state_dict())
        for p in actor32b.parameters():
            p.grad = None
        # (the loop of rollout_loss, with a hook on every step's action: dL/du_t per environment)
        hooks = []
        obs = env32b.reset(q0, goal); acc = None
        for t in range(T):
            u = actor32b(obs)
            u.register_hook(lambda g, t=t: hooks.append((t, g.detach().double().pow(2).sum(1))))
            obs, rew, _ = env32b.step(u, dist_[t])
            acc = rew if acc is None else acc + rew
        l32b = -acc.sum()
        l32b.backward()
        per_t = torch.stack([g for _, g in sorted(hooks, key=lambda x: x[0])])      # [T, B] squared norms of dL/du_t
        per_env = per_t.sum(0).sqrt()
        srt = torch.sort(per_env, descending=True).values
        du_stats = {"max": float(srt[0]), "top5": [float(x) for x in srt[:5]], "p999": float(srt[int(0.001 * B)]), "p99": float(srt[int(0.01 * B)]), "median": float(srt[B // 2]),
                    "share_of_top5_in_sum_of_squares": float(srt[:5].pow(2).sum() / srt.pow(2).sum()), "envs_over_100x_median": int((per_env > 100 * srt[B // 2]).sum()),
                    "per_step_rms_of_worst_env": [float(x) for x in per_t[:, int(per_env.argmax())].sqrt()[::10]],
                    "per_env_return_of_worst": float(acc[int(per_env.argmax())]), "median_return": float(acc.median())}
        pb = list(actor32b.parameters()); allreduce_policy_grad_(pb, B); g32b = flat(pb)
        # ... fused fp64
        actor64.load_state_dict({k: v.double() for k, v in actor.state_dict().items()})
        for p in actor64.parameters():
            p.grad = None
        l64 = ep64.rollout(q0.double(), goal.double(), dist_.double())
        ep64.backward()
        p64 = list(actor64.parameters()); allreduce_policy_grad_(p64, B); g64 = flat(p64)
        cos = lambda x, y: float((x @ y) / (x.norm() * y.norm()))
        row = {"epoch": epoch, "loss_fused32": float(loss) / B, "loss_step32": float(l32b) / B, "loss_fused64": float(l64) / B,
               "norm_fused32": float(g32.norm()), "norm_step32": float(g32b.norm()), "norm_fused64": float(g64.norm()),
               "cos_fused32_fused64": cos(g32, g64), "cos_step32_fused64": cos(g32b, g64), "cos_fused32_step32": cos(g32, g32b),
               "finite": [bool(torch.isfinite(x).all()) for x in (g32, g32b, g64)],
               "dLdu_per_env": du_stats, "param_absmax": float(max(p.detach().abs().max() for p in params)), "logstd": [float(x) for x in actor.state_dict().get("logstd", torch.zeros(0)).reshape(-1)[:3]]}
        rows.append(row); print(json.dumps(row), flush=True)
    torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0)
    opt.step()
    if epoch < a.from_epoch and epoch % 10 == 0:
        print("epoch", epoch, "loss", float(loss) / B, flush=True)
if a.out:
    json.dump(rows, open(a.out, "w"), indent=1)
