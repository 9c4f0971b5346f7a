"""Is the fp32-vs-fp64 spread of the 100-step episode gradients arithmetic error or the sensitivity of non-smooth contact
dynamics?  fp64 kernels, the same batch twice: exact inputs, and inputs rounded to fp32 (a 6e-8 relative perturbation)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload
B, T, S = 4096, 100, 5
m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
q0, u, _ = push_workload(B, T, seed=0)
dt = torch.float64
def grad(q0_, u_):
    sim = BatchSim(m, B, dtype=dt, tape_capacity=T * S)
    sim.reset(torch.tensor(q0_, device="cuda", dtype=dt), None, backward_flag=True)
    sim.rollout(torch.tensor(u_, device="cuda", dtype=dt).transpose(0, 1).contiguous(), S)
    wq = torch.ones(T, B, sim.ndof_r, device="cuda", dtype=dt); wv = torch.ones(T, B, sim.ndof_var, device="cuda", dtype=dt)
    wt = torch.ones(T, B, sim.ndof_tactile, device="cuda", dtype=dt) * 100
    return sim.backward_episode(T, S, wq, wv, wt).cpu().numpy().sum(0)
g0 = grad(q0, u)
g1 = grad(q0.astype(np.float32).astype(np.float64), u.astype(np.float32).astype(np.float64))
dg = np.abs(g0 - g1).max(1) / np.maximum(np.abs(g0).max(1), 1e-12)
res = {"what": "fp64 kernels, exact inputs vs inputs rounded to fp32; relative difference of the per-environment 100-step episode gradient",
       "median": float(np.median(dg)), "p90": float(np.percentile(dg, 90)), "p99": float(np.percentile(dg, 99)), "max": float(dg.max()),
       "fraction_above_1e-4": float((dg > 1e-4).mean()), "fraction_above_1e-3": float((dg > 1e-3).mean()), "fraction_above_1e-2": float((dg > 1e-2).mean()),
       "batch_gradient_rel_diff": float(np.abs(g0.sum(0) - g1.sum(0)).max() / np.abs(g0.sum(0)).max())}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "grad_sensitivity.json"), "w"), indent=1)
