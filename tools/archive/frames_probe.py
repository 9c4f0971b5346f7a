"""Why is a 20-frame episode launch slower per env-step than a 100-frame one?  (GPU box)
Times tsim_rollout / tsim_backward_episode for several episode lengths and frame windows, and lists the per-frame
evaluation rounds (mean over environments, and per wavefront of 4 sub-step-synchronous slots)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
B, T, S = 4096, 100, 5
model = load_model(PUSHER_BLOB)
q0_np, u_np, _ = push_workload(B, T, seed=0)
dt = torch.float32
sim = BatchSim(model, B, dtype=dt, tape_capacity=T * S)
q0 = torch.tensor(q0_np, device="cuda", dtype=dt)
u = torch.tensor(u_np, device="cuda", dtype=dt).transpose(0, 1).contiguous()
Ev = lambda: torch.cuda.Event(enable_timing=True)
res = {}
def timed(n, f0=0, reps=3, record=True):
    bf, bb = 1e9, 1e9
    for _ in range(reps + 1):
        sim.reset(q0, None, backward_flag=record)
        if f0:
            sim.rollout(u[:f0], S)
        e0, e1, e2 = Ev(), Ev(), Ev()
        e0.record(); sim.rollout(u[f0:f0 + n], S); e1.record()
        if record:
            w = lambda d: torch.ones(n, B, d, device="cuda", dtype=dt)
            sim.backward_episode(n, S, w(7), w(6), w(390) * 100.0)
        e2.record(); torch.cuda.synchronize()
        bf, bb = min(bf, e0.elapsed_time(e1)), min(bb, e1.elapsed_time(e2))
    return {"fwd_ms_per_env_step": bf / n, "bwd_ms_per_env_step": bb / n}
timed(100)
for n in (1, 2, 5, 10, 20, 50, 100):
    res["frames_0_%d" % n] = timed(n); print(n, res["frames_0_%d" % n], flush=True)
for f0 in (20, 40, 60, 80):
    res["frames_%d_%d" % (f0, f0 + 20)] = timed(20, f0); print(f0, res["frames_%d_%d" % (f0, f0 + 20)], flush=True)
res["fwd_only_20"] = timed(20, record=False); res["fwd_only_100"] = timed(100, record=False)
# per-frame evaluation counts
sim2 = BatchSim(model, B, dtype=dt, tape_capacity=1)
sim2.reset(q0, None, backward_flag=False)
ev = []
for t in range(T):
    for s in range(S):
        sim2.step(u[t], 1, want_var=False, want_tactile=False); ev.append(sim2.last_evals())
ev = np.array(ev).reshape(T, S, B)
per_frame = ev.sum(1)                                   # [T, B]
g = ev.reshape(T, S, B // 4, 4).max(3).sum(1)           # rounds per wavefront and frame (sub-step-synchronous slots)
res["evals_mean_per_frame"] = per_frame.mean(1).round(2).tolist()
res["wave_rounds_mean_per_frame"] = g.mean(1).round(2).tolist()
res["wave_rounds_max_per_frame"] = g.max(1).tolist()
for n in (20, 100):
    tot = g[:n].sum(0)
    res["wave_rounds_first_%d" % n] = {"mean": float(tot.mean()), "max": int(tot.max()), "ideal_env_mean": float(per_frame[:n].sum(0).mean())}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "frames_probe.json"), "w"), indent=1)
