"""One TactileInsertion environment of the B = 4096 probe batch (default: 424) through the fp64 / fp32 kernels and the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_models import _inputs
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset
from oracle.oracle import OracleSim
B, T, S = 4096, 14, 5
m = load_model(asset("tactile_insertion"))
q0, u = _inputs("tactile_insertion", m, B, T)
envs = [int(a) for a in sys.argv[1:]] or [424]
for e in envs:
    o = OracleSim(m); o.reset(q0[e])
    sims = {}
    for dt in (torch.float64, torch.float32):
        for lanes in (64, 32):
            s = BatchSim(m, 1, dtype=dt, tape_capacity=0); s.set_lanes_per_env(lanes); s.reset(torch.tensor(q0[e:e+1], device="cuda", dtype=dt), None, backward_flag=False)
            sims[(str(dt)[6:], lanes)] = s
    for t in range(4):
        rc = o.forward(u[e, t], S); q = o.state()[0]
        line = "env %d t %d oracle nonconv %d |" % (e, t, rc)
        for k, s in sims.items():
            g = s.step(torch.tensor(u[e:e+1, t], device="cuda", dtype=s.dtype), S)
            line += " %s/%d: st %d diff %.2g evals %d |" % (k[0], k[1], int(g["status"][0]), np.abs(q - g["q"][0].double().cpu().numpy()).max(), int(s.last_evals()[0]))
        print(line, flush=True)
