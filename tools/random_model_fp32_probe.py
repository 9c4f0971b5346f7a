"""fp32 generic kernels against the fp64 oracle on the random models of tests/test_gpu_random_models.py, with the Newton tolerance relaxed to what
single precision can reach on models of arbitrary scale (the test itself runs fp64 only): where do they differ, and by how much?  (GPU box)
   python tools/random_model_fp32_probe.py [tol] [n_models]"""
import os, sys, tempfile, pathlib, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tactilesimulation_amd.model.blob as BL
from tactilesimulation_amd.host.batch import BatchSim
from oracle.oracle import OracleSim
import test_gpu_random_models as TR
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
B_, T, S = 3, 3, 2
rows = []
for seed in range(N):
    try:
        m, rng = TR._case(TR.SEED0 + seed, pathlib.Path(tempfile.mkdtemp()))
    except BaseException:
        continue
    m.F[BL.TSIM_FH_TOL] = tol
    nr, nu = m.ndof_r, m.ndof_u
    q0 = 0.02 * rng.normal(size=(B_, nr)); u = rng.uniform(-1, 1, size=(B_, T, max(nu, 1)))[:, :, :nu]
    sim = BatchSim(m, B_, dtype=torch.float32, tape_capacity=T * S)
    sim.reset(torch.tensor(q0, device="cuda:0", dtype=torch.float32), None, backward_flag=False)
    outs = [sim.step(torch.tensor(u[:, t], device="cuda:0", dtype=torch.float32).reshape(B_, nu), S) for t in range(T)]
    o = OracleSim(m)
    for e in range(B_):
        o.reset(q0[e])
        for t in range(T):
            bad = o.forward(u[e, t], S)
            kbad = int(outs[t]["status"][e]) != 0
            q, _ = o.state()
            if bad or kbad or not np.all(np.isfinite(q)):
                rows.append({"seed": seed, "e": e, "t": t, "oracle_bad": int(bad), "kernel_bad": int(kbad)})
                break
            dq = float(np.abs(outs[t]["q"][e].double().cpu().numpy() - q).max() / (1.0 + np.abs(q).max()))
            _, tac = o.outputs()
            dtac = float(np.abs(outs[t]["tactile"][e].double().cpu().numpy() - tac).max() / (1.0 + np.abs(tac).max())) if m.ndof_tactile else 0.0
            rows.append({"seed": seed, "e": e, "t": t, "dq": dq, "dtac": dtac})
cmp_ = [r for r in rows if "dq" in r]
dq = np.array([r["dq"] for r in cmp_]); dt_ = np.array([r["dtac"] for r in cmp_])
flag = [r for r in rows if "dq" not in r]
print(json.dumps({"tol": tol, "env_steps_compared": len(cmp_), "dq_quantiles_50_90_99_max": [float(np.quantile(dq, x)) for x in (0.5, 0.9, 0.99, 1.0)],
                  "dtac_quantiles_50_90_99_max": [float(np.quantile(dt_, x)) for x in (0.5, 0.9, 0.99, 1.0)],
                  "flagged": len(flag), "only_kernel_flagged": sum(1 for r in flag if r["kernel_bad"] and not r["oracle_bad"]), "only_oracle_flagged": sum(1 for r in flag if r["oracle_bad"] and not r["kernel_bad"]),
                  "worst": sorted(cmp_, key=lambda r: -r["dq"])[:6]}))
