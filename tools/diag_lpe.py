"""Adjoint of ONE sub-step under different seeds, for the launch shape forced by TSIM_LPE (diagnostic)."""
import os, sys, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tests.workloads import push_workload
m = load_model(os.path.join(ROOT, "tests", "golden", "models", "pusher.npz"))
B, T = 8, 40
q0, u, _ = push_workload(B, T, seed=3)
res = {}
for dt in (torch.float64,):
    sim = BatchSim(m, B, dtype=dt, tape_capacity=T * 5)
    for seed in ("q", "var", "tac"):
        sim.reset(torch.tensor(q0, device="cuda", dtype=dt), None, backward_flag=True)
        for t in range(T):
            sim.step(torch.tensor(u[:, t], device="cuda", dtype=dt), 5)
        wq = torch.ones(B, sim.ndof_r, device="cuda", dtype=dt) * (seed == "q")
        wv = torch.ones(B, sim.ndof_var, device="cuda", dtype=dt) * (seed == "var")
        wt = torch.ones(B, sim.ndof_tactile, device="cuda", dtype=dt) * (seed == "tac")
        n_sub = int(os.environ.get('DIAG_N', '1'))
        for _ in range(int(os.environ.get('DIAG_CALLS', '1'))):
            du = sim.backward_steps(n_sub, wq, wv, wt)
        lq, lv = sim.get_adjoint()
        res[seed] = {"du": du.cpu().numpy().tolist(), "lq": lq.cpu().numpy().tolist(), "lv": lv.cpu().numpy().tolist()}
print(json.dumps(res))
