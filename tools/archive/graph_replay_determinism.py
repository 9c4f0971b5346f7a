"""Bisect: policy gradient of a graph replay with NEW episode data vs the eager gradient, over batch sizes / horizons / dtypes."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.algorithms.batched_gd import Actor, GraphedRollout, rollout_loss
from tactilesimulation_amd.workloads import PUSHER_BLOB
from train_tactile_push_gd_batched import draw_episode
flat = lambda a: torch.cat([p.grad.reshape(-1) for p in a.parameters() if p.grad is not None]).clone()
rel = lambda x, y: float((x - y).norm() / y.norm())
def actor(dt):
    torch.manual_seed(0); return Actor(dtype=dt).cuda()
cases = [(256, 100), (1024, 100), (2048, 100), (4096, 20), (4096, 50), (4096, 100)] if "--big" not in sys.argv else [(8192, 10), (16384, 10), (32768, 10)]
for dt in (torch.float64, torch.float32):
    for B, T in cases:
        rng = np.random.default_rng(0)
        q0, goal, D = draw_episode(rng, B, T, "cuda", dt)
        mk = lambda b: BatchedTactilePushEnv(PUSHER_BLOB, b, dtype=dt, gradient=True, seed=0, tape_steps=T)
        a1 = actor(dt); gr = GraphedRollout(mk(B), a1, T, q0, goal, D)
        out = {"dtype": str(dt)[6:], "B": B, "T": T, "replay_vs_eager": [], "replay_vs_previous_replay_same_data": []}
        for rep in range(3):
            a, b, c = draw_episode(rng, B, T, "cuda", dt); q0.copy_(a); goal.copy_(b); D.copy_(c)
            a0 = actor(dt); l0 = rollout_loss(mk(B), a0, T, q0=q0, goal=goal, disturbances=D); l0.backward(); ge = flat(a0)
            gr.replay(); torch.cuda.synchronize(); g1 = flat(a1)
            gr.replay(); torch.cuda.synchronize(); g2 = flat(a1)
            out["replay_vs_eager"].append(rel(g1, ge)); out.setdefault("loss_rel", []).append(abs(float(gr.loss.detach()) - float(l0.detach())) / abs(float(l0.detach()))); out["replay_vs_previous_replay_same_data"].append(rel(g2, g1))
        print(json.dumps(out), flush=True)
        del gr, a1; torch.cuda.empty_cache()
