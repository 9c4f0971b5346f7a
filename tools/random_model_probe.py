"""Residual g and Newton matrix H of one model at random states, kernels (fp64, tsim_debug_eval) against the oracle (GPU box).
   python tools/random_model_probe.py model.xml [n]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from oracle.oracle import OracleSim
m = load_model(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nr, nu = m.ndof_r, m.ndof_u
rng = np.random.default_rng(0)
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
q0 = scale * rng.normal(size=(n, nr)); q1 = q0 + 0.2 * scale * rng.normal(size=(n, nr)); qd0 = rng.normal(size=(n, nr)); u = rng.uniform(-1, 1, size=(n, max(nu, 1)))[:, :nu]
sim = BatchSim(m, n, dtype=torch.float64, tape_capacity=4)
t = lambda a: torch.tensor(a, device="cuda:0")
g, H = sim.debug_eval(t(q1), t(q0), t(qd0), t(u).reshape(n, nu))
g, H = g.cpu().numpy(), H.cpu().numpy()
o = OracleSim(m)
for e in range(n):
    r = o.residual(q1[e], q0[e], qd0[e], u[e], which=0)
    go, Ho = (r[0], r[1]) if isinstance(r, tuple) else (r, None)
    print(e, "g kernel", np.array2string(g[e], precision=6), "oracle", np.array2string(np.asarray(go), precision=6), "max diff %.3g" % np.abs(g[e] - go).max(), ("H diff %.3g of %.3g" % (np.abs(H[e] - Ho).max(), np.abs(Ho).max())) if Ho is not None else "")
