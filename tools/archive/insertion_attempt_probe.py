"""TactileInsertion as SURVEY.md §8d config 5 words it (workloads.insertion_attempt_workload) at B = 4096 on the GPU: evaluations per
environment and attempt, non-converged sub-steps, launch time with and without an evaluation budget, fp32 and fp64."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd import workloads as W

B = 4096
m = load_model(W.asset("tactile_insertion"))
q0, u = W.insertion_attempt_workload(B, seed=7)
mask = torch.zeros(45, dtype=torch.bool); mask[list(W.INSERTION_TACTILE_FRAMES)] = True
out = {}
for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
    sim = BatchSim(m, B, dtype=dt, tape_capacity=0)
    Q0 = torch.tensor(q0, device="cuda", dtype=dt); U = torch.tensor(u, device="cuda", dtype=dt).transpose(0, 1).contiguous()
    for budget in (0, 16, 32):
        sim.set_solver_options(cross_kinks=None, eval_budget=budget)
        ms = []
        for rep in range(4):
            sim.reset(Q0, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ro = sim.rollout(U, 1, tactile_mask=mask); e1.record()
            torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
        ev = sim.last_evals(); st = ro["status"].cpu().numpy() & 0x3FFFFFFF
        out["%s_budget%d" % (name, budget)] = {"ms_per_attempt_launch": ms, "env_steps_per_s": B * 9 / (min(ms[1:]) * 1e-3), "launch_shape": sim.launch_info(),
            "evals_per_env": {"mean": float(ev.mean()), "p50": float(np.percentile(ev, 50)), "p99": float(np.percentile(ev, 99)), "p999": float(np.percentile(ev, 99.9)), "max": int(ev.max())},
            "nonconverged_envs": int((st != 0).sum()), "nonconverged_substeps": int(st.sum()),
            "success_rate": float((ro["q"][-1, :, 8] < 0.0247).double().mean())}
    del sim
print(json.dumps(out, indent=1))
