"""Where the fused GD epoch's time goes (envs/push_closed_loop.FusedPushEpisode, B = 4096, T = 100, fp32): event-timed sections of
train_epoch_fused.  Run on the GPU box:  python tools/closed_loop_breakdown.py"""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.workloads import asset
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode
from tactilesimulation_amd.algorithms.batched_gd import Actor
from tactilesimulation_amd.dist import allreduce_policy_grad_

B, T, dev, tdt = 4096, 100, torch.device("cuda:0"), torch.float32
env = BatchedTactilePushEnv(load_model(asset("pusher")), B, device=str(dev), dtype=tdt, gradient=True, seed=0, tape_steps=T)
env.reset()
q0, goal = env.q0.clone(), env.goal.clone()
rng = np.random.default_rng(1)
dist_ = torch.tensor(rng.uniform(-1, 1, size=(T, B, 2)) * (rng.uniform(size=(T, B, 1)) < 0.5), device=dev, dtype=tdt)
torch.manual_seed(0)
actor = Actor(dtype=tdt).to(dev)
opt = torch.optim.Adam(actor.parameters(), lr=5e-3, betas=(0.7, 0.95))
ep = FusedPushEpisode(env, actor, T)
names = ["rollout (reset, read-out, launch, reward + partials)", "backward (launch + weight gradients)", "all-reduce / clip / Adam"]
acc = np.zeros(3); n = 0
for it in range(6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for p in actor.parameters(): p.grad = None
    ev[0].record(); ep.rollout(q0, goal, dist_)
    ev[1].record(); ep.backward()
    ev[2].record()
    params = list(actor.parameters()); allreduce_policy_grad_(params, B)
    torch.nn.utils.clip_grad_norm_(params, 1.0); opt.step()
    ev[3].record(); torch.cuda.synchronize()
    if it >= 2:
        acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]; n += 1
out = {names[i]: acc[i] / n for i in range(3)}
out["total ms"] = float(acc.sum() / n); out["M env-steps/s"] = B * T / (acc.sum() / n) / 1e3
print(json.dumps(out, indent=1))
ev = env.sim.last_evals().astype(float)
print("residual evaluations per env-step in the last fused rollout: mean %.2f, heaviest environment %.2f" % (ev.mean() / T, ev.max() / T))
if os.environ.get("TSIM_HIP_LIB", "").endswith("pptime.so"):
    g = env.sim.last_gnorm()
    print("share of k_forward spent in the policy call (A/B build -DTS_PP_TIME): mean %.3f, max %.3f" % (g.mean(), g.max()))
