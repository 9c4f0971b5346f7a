"""One environment of one random model: the fp32 kernels' first env-step (with and without kink crossing) against the oracle's — the states, and the
oracle's residual at each (GPU box).   TSIM_RANDOM_SEED0=200000 python tools/random_model_fp32_case.py SEED ENV [tol]"""
import os, sys, tempfile, pathlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tactilesimulation_amd.model.blob as BL
from tactilesimulation_amd.host.batch import BatchSim
from oracle.oracle import OracleSim
import test_gpu_random_models as TR
seed, e = int(sys.argv[1]), int(sys.argv[2]); tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-5
m, rng = TR._case(TR.SEED0 + seed, pathlib.Path(tempfile.mkdtemp()))
m.F[BL.TSIM_FH_TOL] = tol
nr, nu = m.ndof_r, m.ndof_u
B_, T, S = TR.B_, TR.T, TR.S
q0 = 0.02 * rng.normal(size=(B_, nr)); u = rng.uniform(-1, 1, size=(B_, T, max(nu, 1)))[:, :, :nu]
print("nr", nr, "nu", nu, "h", m.h, "integrator", int(m.I[BL.TSIM_IH_INTEGRATOR]), "max_iter", int(m.I[BL.TSIM_IH_MAX_ITER]), "max_ls", int(m.I[BL.TSIM_IH_MAX_LS]))
o = OracleSim(m)
res = {}
for name, dt, ck in (("f32 cross_kinks=1", torch.float32, 1), ("f32 cross_kinks=0", torch.float32, 0), ("f64", torch.float64, 0)):
    sim = BatchSim(m, B_, dtype=dt, tape_capacity=T * S)
    sim.set_solver_options(cross_kinks=ck)
    sim.reset(torch.tensor(q0, device="cuda:0", dtype=dt), None, backward_flag=False)
    for s_ in range(S):
        out = sim.step(torch.tensor(u[:, 0], device="cuda:0", dtype=dt).reshape(B_, nu), 1, want_qd=True)
        print(name, "sub-step", s_, "status", int(out["status"][e]), "evals", int(sim.last_evals()[e]), "q", np.array2string(out["q"][e].double().cpu().numpy(), precision=5))
o.reset(q0[e]); prev, prevd = q0[e].copy(), np.zeros(nr)
for s_ in range(S):
    bad = o.forward(u[e, 0], 1); q, qd = o.state()
    print("oracle sub-step", s_, "bad", bad, "q", np.array2string(q, precision=5), "|g| at its own root %.3g" % np.linalg.norm(o.residual(q, prev, prevd, u[e, 0])))
    prev, prevd = q, qd
