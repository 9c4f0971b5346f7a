"""One graphed GD epoch set (fp32, B = 4096, T = 100, lr 0: the initial policy every epoch) for `rocprofv3 --kernel-trace --stats`:
which kernels the closed loop's time goes to (simulator launches vs the ~40 small policy / observation / reward / autograd kernels
per env-step)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.algorithms.batched_gd import Actor, GraphedRollout, train_epoch_graphed
from tactilesimulation_amd.workloads import PUSHER_BLOB
from train_tactile_push_gd_batched import draw_episode
B, T, dt = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 100, torch.float32
rng = np.random.default_rng(0)
q0, goal, D = draw_episode(rng, B, T, "cuda", dt)
torch.manual_seed(0); actor = Actor(dtype=dt).cuda(); opt = torch.optim.Adam(actor.parameters(), lr=0.0)
gr = GraphedRollout(BatchedTactilePushEnv(PUSHER_BLOB, B, dtype=dt, gradient=True, seed=0, tape_steps=T), actor, T, q0, goal, D)
train_epoch_graphed(gr, opt, B); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): train_epoch_graphed(gr, opt, B)
torch.cuda.synchronize(); s = (time.perf_counter() - t0) / 5
print("ms per epoch %.2f  env-steps/s %.0f" % (s * 1e3, B * T / s))
