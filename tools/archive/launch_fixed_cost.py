"""Fixed cost of a per-env-step launch: k_forward / k_backward on 4096 IDENTICAL TactilePush environments (no stragglers: every wavefront does
the same work), n sub-steps per launch.  Run under `rocprofv3 --kernel-trace --stats`: time(n) = fixed + n * per_sub_step."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT)
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
B = 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
q0, u, _ = push_workload(1, 40, seed=0)
m = load_model(PUSHER_BLOB)
sim = BatchSim(m, B, dtype=torch.float32, tape_capacity=2048)
Q0 = torch.tensor(np.tile(q0, (B, 1)), device="cuda", dtype=torch.float32)
U = torch.tensor(np.tile(u, (B, 1, 1)), device="cuda", dtype=torch.float32)
sim.reset(Q0, None, backward_flag=True)
for t in range(20): sim.step(U[:, t].contiguous(), n)
w = [torch.ones(1, B, d, device="cuda") for d in (7, 6, 390)]
for t in range(20): sim.backward_episode(1, n, *w)
torch.cuda.synchronize()
print("evals per launch", float(sim.last_evals().mean()) if hasattr(sim.last_evals(), "mean") else sim.last_evals()[:4])
