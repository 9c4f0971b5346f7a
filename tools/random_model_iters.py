"""First Newton iteration at which kernels (fp64) and oracle part ways on the first sub-step of a case of tests/test_gpu_random_models.py: the
sub-step is run with max_iter = 1, 2, 3 ... in both (a sub-step that does not converge ends on its last iterate), from the same state (GPU box).
   python tools/random_model_iters.py SEED ENV"""
import copy, os, sys, tempfile, pathlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tactilesimulation_amd.model.blob as BL
from tactilesimulation_amd.host.batch import BatchSim
from oracle.oracle import OracleSim
import test_gpu_random_models as TR
seed, e = int(sys.argv[1]), int(sys.argv[2])
m, rng = TR._case(TR.SEED0 + seed, pathlib.Path(tempfile.mkdtemp()))
nr, nu = m.ndof_r, m.ndof_u
B_, T = TR.B_, TR.T
q0 = 0.02 * rng.normal(size=(B_, nr)); u = rng.uniform(-1, 1, size=(B_, T, max(nu, 1)))[:, :, :nu]
kmax = int(m.I[BL.TSIM_IH_MAX_ITER])
print("nr", nr, "nu", nu, "max_iter", kmax, "max_ls", int(m.I[BL.TSIM_IH_MAX_LS]), "tol", m.F[BL.TSIM_FH_TOL], "integrator", int(m.I[BL.TSIM_IH_INTEGRATOR]))
for k in list(range(1, kmax + 1)):
    mk = copy.deepcopy(m); mk.I[BL.TSIM_IH_MAX_ITER] = k
    sim = BatchSim(mk, B_, dtype=torch.float64, tape_capacity=2)
    sim.reset(torch.tensor(q0, device="cuda:0"), None, backward_flag=False)
    out = sim.step(torch.tensor(u[:, 0], device="cuda:0").reshape(B_, nu), 1, want_qd=True)
    o = OracleSim(mk); o.reset(q0[e]); s0 = o.stats(); bad = o.forward(u[e, 0], 1); s1 = o.stats()
    q, _ = o.state(); qk = out["q"][e].cpu().numpy()
    g_k, g_o = o.residual(qk, q0[e], np.zeros(nr), u[e, 0]), o.residual(q, q0[e], np.zeros(nr), u[e, 0])
    print("max_iter %2d |dq| %.3g  status kernel %d oracle %d  kernel evals %d oracle ls trials %d  |g| kernel %.6g oracle %.6g" % (k, np.abs(qk - q).max(), int(out["status"][e]), bad, int(sim.last_evals()[e]), s1["evals"] - s0["evals"] - (s1["newton_iters"] - s0["newton_iters"]), np.linalg.norm(g_k), np.linalg.norm(g_o)), flush=True)
    if int(out["status"][e]) == 0 and bad == 0:
        break
