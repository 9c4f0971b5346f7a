"""How much do sub-step-synchronous slots cost?  Per-sub-step evaluation counts of the bench workload (GPU box)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload
B, T, S = 4096, 40, 5
model = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
q0_np, u_np, _ = push_workload(B, T, seed=0)
sim = BatchSim(model, B, dtype=torch.float32, tape_capacity=1)
sim.reset(torch.tensor(q0_np, device="cuda", dtype=torch.float32), None, backward_flag=False)
u = torch.tensor(u_np, device="cuda", dtype=torch.float32).transpose(0, 1).contiguous()
ev = []
for t in range(T):
    for s in range(S):
        sim.step(u[t], 1, want_var=False, want_tactile=False)
        ev.append(sim.last_evals())
ev = np.array(ev)                       # [T*S, B]
res = {"mean_evals_per_substep": float(ev.mean())}
for ns in (1, 2, 4, 8):
    g = ev.reshape(ev.shape[0], B // ns, ns)
    sync = g.max(axis=2).sum(axis=0)                   # rounds a wave needs with sub-step-synchronous slots
    asyn = g.sum(axis=0).max(axis=1)                   # ... with independent slots (episode totals)
    res["slots_%d" % ns] = {"sync_rounds_mean": float(sync.mean()), "sync_rounds_max": int(sync.max()),
                            "async_rounds_mean": float(asyn.mean()), "async_rounds_max": int(asyn.max()),
                            "ideal": float(ev.sum(axis=0).mean())}
hist = np.bincount(ev.reshape(-1), minlength=12)[:40]
res["hist"] = hist.tolist()
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "slot_sync_waste.json"), "w"), indent=1)
