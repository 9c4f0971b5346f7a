"""Sub-step by sub-step: one TactileInsertion environment of the probe batch in the fp64 kernel (status, evaluations, |q|, |qd|)."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_models import _inputs
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset
m = load_model(asset("tactile_insertion"))
q0, u = _inputs("tactile_insertion", m, 4096, 14)
e = int(sys.argv[1]) if len(sys.argv) > 1 else 424
s = BatchSim(m, 1, dtype=torch.float64, tape_capacity=0); s.reset(torch.tensor(q0[e:e+1], device="cuda"), None, backward_flag=False)
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for t in range(3):
    for k in range(5):
        g = s.step(torch.tensor(u[e:e+1, t], device="cuda"), 1, want_qd=True)
        print("t %d sub %d status %d evals %3d  q %s  |qd| %.3g" % (t, k, int(g["status"][0]), int(s.last_evals()[0]), g["q"][0].cpu().numpy(), float(g["qd"].abs().max())), flush=True)
