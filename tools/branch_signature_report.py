"""fp32 kernels vs fp64 kernels on the whole bench batch (4096 envs x 100 env-steps), split by the branch signature
(tsim_debug_signature): environments whose 500 sub-steps went through the same smooth pieces of the contact / friction law
in both precisions, and those where one of the two crossed a kink.  (GPU box)  -> gpurun_out/branch_signature.json"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
B, T, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 100, 5
m = load_model(PUSHER_BLOB)
q0, u, _ = push_workload(B, T, seed=0)
rng = np.random.default_rng(4)
wq, wv, wt = rng.normal(size=(T, 7)), rng.normal(size=(T, 6)), rng.normal(size=(T, 390)) * 10.0
res = {}
import copy
import tactilesimulation_amd.model.blob as Bl
m_tight = copy.copy(m); m_tight.F = m.F.copy(); m_tight.F[Bl.TSIM_FH_TOL] = 1e-12
for dt in (torch.float64, torch.float32, "f64_tight"):
    mm = m
    if dt == "f64_tight":            # the fp64 kernels again with Newton tolerance 1e-12 instead of the XML's 1e-8: how much of the
        mm, dt = m_tight, torch.float64      # fp32 - fp64 difference is solver-tolerance noise that fp64 has against itself?
        key = "tight"
    else:
        key = dt
    sim = BatchSim(mm, B, dtype=dt, tape_capacity=T * S)
    sim.reset(torch.tensor(q0, device="cuda", dtype=dt), None, backward_flag=True)
    ro = sim.rollout(torch.tensor(u, device="cuda", dtype=dt).transpose(0, 1).contiguous(), S, want_qd=True)
    sig = sim.branch_signature().cpu().numpy()                     # [T*S, B, 2]
    tile = lambda w: torch.tensor(np.broadcast_to(w[:, None, :], (T, B, w.shape[1])).copy(), device="cuda", dtype=dt)
    du = sim.backward_episode(T, S, tile(wq), tile(wv), tile(wt)).double().cpu().numpy()
    res[key] = {"q": ro["q"].double().cpu().numpy(), "tac": ro["tactile"].double().cpu().numpy(), "sig": sig, "du": du,
               "bad": int((ro["status"] != 0).sum())}
    del sim
a, b = res[torch.float64], res[torch.float32]
same_step = (a["sig"] == b["sig"]).all(axis=2)                      # [T*S, B]
same = same_step.all(axis=0)                                        # [B]
first_flip = np.where(same, T * S, np.argmin(same_step, axis=0))
eg = np.abs(a["du"] - b["du"]).max(axis=(0, 2)) / np.abs(a["du"]).max(axis=(0, 2))
eq = np.abs(a["q"] - b["q"]).max(axis=(0, 2))
et = np.abs(a["tac"] - b["tac"]).max(axis=(0, 2)) / np.abs(a["tac"]).max()
def dist(x):
    return {"n": int(x.size), "median": float(np.median(x)), "p90": float(np.percentile(x, 90)), "p99": float(np.percentile(x, 99)), "max": float(x.max())} if x.size else {"n": 0}
gs = np.abs(a["du"].sum(1) - b["du"].sum(1)).max() / np.abs(a["du"].sum(1)).max()
c = res["tight"]
same_t = (a["sig"] == c["sig"]).all(axis=(0, 2))
eg_t = np.abs(a["du"] - c["du"]).max(axis=(0, 2)) / np.abs(c["du"]).max(axis=(0, 2))
out = {"B": B, "T": T,
       "fp64_tol1e-8_vs_fp64_tol1e-12": {"flipped_envs": int((~same_t).sum()), "grad_rel_err_same_signature": dist(eg_t[same_t]),
                                         "envs_over_1e-4_same_signature": int((eg_t[same_t] > 1e-4).sum()),
                                         "q_abs_err": dist(np.abs(a["q"] - c["q"]).max(axis=(0, 2)))},
       "worst_same_signature_envs_f32": [int(i) for i in np.argsort(np.where(same, eg, 0))[-5:]],
       "worst_same_signature_envs_tol": [int(i) for i in np.argsort(np.where(same_t, eg_t, 0))[-5:]], "nonconverged": {"f64": a["bad"], "f32": b["bad"]},
       "flipped_envs": int((~same).sum()), "flipped_fraction": float((~same).mean()),
       "flipped_substeps_fraction": float((~same_step).mean()),
       "first_flip_substep_of_flipped_envs": dist(first_flip[~same]),
       "grad_rel_err_same_signature": dist(eg[same]), "grad_rel_err_flipped": dist(eg[~same]),
       "envs_over_1e-4": {"same_signature": int((eg[same] > 1e-4).sum()), "flipped": int((eg[~same] > 1e-4).sum())},
       "q_abs_err_same_signature": dist(eq[same]), "q_abs_err_flipped": dist(eq[~same]),
       "tactile_rel_err_same_signature": dist(et[same]), "tactile_rel_err_flipped": dist(et[~same]),
       "batch_summed_gradient_rel_err": float(gs),
       "mean_penetrating_items_per_substep": float(a["sig"][:, :, 0].mean())}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "branch_signature.json"), "w"), indent=1)
