"""fp32 against fp64 kernels on the whole bench batch (4096 environments x 100 env-steps): how far do the trajectories drift?"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload
B, T, S = 4096, 100, 5
m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
q0, u, _ = push_workload(B, T, seed=0)
out = {}
for dt in (torch.float64, torch.float32):
    sim = BatchSim(m, B, dtype=dt, tape_capacity=T * S)
    sim.reset(torch.tensor(q0, device="cuda", dtype=dt), None, backward_flag=True)
    ro = sim.rollout(torch.tensor(u, device="cuda", dtype=dt).transpose(0, 1).contiguous(), S)
    wq = torch.ones(T, B, sim.ndof_r, device="cuda", dtype=dt); wv = torch.ones(T, B, sim.ndof_var, device="cuda", dtype=dt)
    wt = torch.ones(T, B, sim.ndof_tactile, device="cuda", dtype=dt) * 100
    du = sim.backward_episode(T, S, wq, wv, wt)
    out[dt] = {k: v.double().cpu().numpy() for k, v in (("q", ro["q"]), ("tac", ro["tactile"]), ("du", du))}
    out[dt]["bad"] = int((ro["status"] != 0).sum())
a, b = out[torch.float64], out[torch.float32]
dq = np.abs(a["q"] - b["q"]).max(axis=(0, 2))                       # per env, over time and dofs
dtac = np.abs(a["tac"] - b["tac"]).max(axis=(0, 2)) / max(np.abs(a["tac"]).max(), 1e-12)
g64, g32 = a["du"].sum(0), b["du"].sum(0)                          # dL/du summed over the episode, per env [B, nu]
dg = np.abs(g64 - g32).max(1) / np.maximum(np.abs(g64).max(1), 1e-12)
res = {"nonconverged_envs": {"f64": a["bad"], "f32": b["bad"]},
       "q_abs_err": {"median": float(np.median(dq)), "p99": float(np.percentile(dq, 99)), "max": float(dq.max())},
       "tactile_rel_err_of_global_max": {"median": float(np.median(dtac)), "p99": float(np.percentile(dtac, 99)), "max": float(dtac.max())},
       "episode_grad_rel_err": {"median": float(np.median(dg)), "p90": float(np.percentile(dg, 90)), "p99": float(np.percentile(dg, 99)), "max": float(dg.max()),
                                "fraction_above_1e-4": float((dg > 1e-4).mean()), "fraction_above_1e-3": float((dg > 1e-3).mean()),
                                "fraction_above_1e-2": float((dg > 1e-2).mean())},
       "batch_gradient_rel_err": float(np.abs(g64.sum(0) - g32.sum(0)).max() / np.abs(g64.sum(0)).max()),
       "finite": bool(np.isfinite(b["q"]).all() and np.isfinite(b["du"]).all())}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fp32_vs_fp64_scale.json"), "w"), indent=1)
