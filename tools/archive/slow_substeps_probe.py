"""Find the TactilePush sub-steps the fp32 kernels spend > 40 evaluations on (one launch per SUB-step, B = 4096) and save their inputs
(state before the sub-step, action) for a replay by the oracle: gpurun_out/slow_substeps.npz."""
import os, sys, json, numpy as np, torch
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
rows = []
for t in range(T):
    for s in range(S):
        q, qd = sim.get_state()
        o = sim.step(U[t], 1, want_qd=True, want_var=False, want_tactile=False)
        ev, gn = sim.last_evals(), sim.last_gnorm()
        for e in np.nonzero(ev > 40)[0]:
            rows.append(dict(env=int(e), t=t, s=s, evals=int(ev[e]), gnorm=float(gn[e]), status=int(o["status"][e]),
                             q=q[e].double().cpu().numpy(), qd=qd[e].double().cpu().numpy(), u=u[e, t], q1=o["q"][e].double().cpu().numpy()))
print(len(rows), "slow sub-steps;", [(r["env"], r["t"], r["s"], r["evals"], "%.1e" % r["gnorm"], r["status"]) for r in rows][:30])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "slow_substeps.npz"), **{k: np.array([r[k] for r in rows]) for k in rows[0]})
