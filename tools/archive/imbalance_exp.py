"""What sets the kernel time of an episode launch: the mean or the slowest wavefront?  (GPU box)
Batches made of copies of the heaviest / lightest environment of the bench workload, and mixtures."""
import os, sys, json, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload
B, T, S = 4096, 100, 5
model = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
q0_np, u_np, _ = push_workload(B, T, seed=0)
dt = torch.float32
sim = BatchSim(model, B, dtype=dt, tape_capacity=1)
def tot_evals(q0, u):
    sim.reset(torch.tensor(q0, device="cuda", dtype=dt), None, False)
    ud = torch.tensor(u, device="cuda", dtype=dt).transpose(0, 1).contiguous()
    ev = np.zeros(B)
    for t in range(T):
        sim.step(ud[t], S, want_var=False, want_tactile=False); ev += sim.last_evals()
    return ev
def timed(q0, u, reps=3):
    q0d = torch.tensor(q0, device="cuda", dtype=dt); ud = torch.tensor(u, device="cuda", dtype=dt).transpose(0, 1).contiguous()
    best = 1e9
    for _ in range(reps + 1):
        sim.reset(q0d, None, False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sim.rollout(ud, S); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best / T
ev = tot_evals(q0_np, u_np)
order = np.argsort(ev)
res = {"evals_per_episode": {"mean": float(ev.mean()), "min": float(ev.min()), "max": float(ev.max()), "p99": float(np.percentile(ev, 99))}}
def batch_of(idx):
    idx = np.asarray(idx); return q0_np[idx], u_np[idx]
h, l, m = order[-1], order[0], order[B // 2]
cases = {"original": np.arange(B), "all heaviest": np.full(B, h), "all lightest": np.full(B, l), "all median": np.full(B, m),
         "one heaviest, rest lightest": np.concatenate([[h], np.full(B - 1, l)]),
         "one heaviest per wavefront of 4, rest lightest": np.where(np.arange(B) % 4 == 0, h, l),
         "sorted by work": order, "heavy quarter first, interleaved": np.concatenate([order[i::4] for i in range(4)])}
for name, idx in cases.items():
    q0c, uc = batch_of(idx)
    res[name] = {"fwd_ms_per_env_step": timed(q0c, uc), "mean_evals_per_episode": float(ev[idx].mean()), "max": float(ev[idx].max())}
    print(name, res[name], flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "imbalance_exp.json"), "w"), indent=1)
