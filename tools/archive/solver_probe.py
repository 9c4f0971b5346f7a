"""fp32 rounding floor of ||g||: TactilePush bench batch, one launch per env-step, per-environment evaluation counts and the largest
||g|| a sub-step ended with; run with TSIM_FLOOR_FACTOR=0 / 4 / 100 (A/B of the rule in k_forward's Newton loop)."""
import os, sys, json, time, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
B, T, S = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 40, 5
m = load_model(PUSHER_BLOB)
q0, u, _ = push_workload(B, T, seed=0)
sim = BatchSim(m, B, dtype=torch.float32, tape_capacity=0)
opt = [int(x) for x in os.environ.get("TSIM_SOLVER_OPT", "1,0").split(",")]
sim.set_solver_options(bool(opt[0]), opt[1])
sim.reset(torch.tensor(q0, device="cuda", dtype=torch.float32), None, False)
U = torch.tensor(u, device="cuda", dtype=torch.float32).transpose(0, 1).contiguous()
ev, gn, bad = [], [], []
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
    o = sim.step(U[t], S, want_var=False, want_tactile=False)
    ev.append(sim.last_evals()); gn.append(sim.last_gnorm()); bad.append((o["status"] != 0).cpu().numpy())
dt = time.perf_counter() - t0
ev, gn, bad = np.array(ev), np.array(gn), np.array(bad)
print(json.dumps({"cross_kinks,budget": os.environ.get("TSIM_SOLVER_OPT", "1,0"), "s": round(dt, 3), "evals_mean": float(ev.mean()), "evals_p999": float(np.percentile(ev, 99.9)),
                  "evals_max": int(ev.max()), "env_steps_over_60_evals": int((ev > 60).sum()), "bad_env_steps": int(bad.sum()),
                  "gnorm_max": float(gn.max()), "gnorm_over_tol": int((gn > 1e-8).sum()), "gnorm_hist_over_tol": np.sort(gn[gn > 1e-8])[::-1][:12].tolist()}))
