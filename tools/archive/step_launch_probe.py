"""(Event times around eager calls include the host's launch latency: run single sub-step counts under rocprofv3 --kernel-trace --stats
for kernel durations.)  Per-launch cost of the per-env-step path (what the closed loop pays 100 times per episode): k_forward / k_backward launch time
against the number of sub-steps per launch, B = 4096 fp32, TactilePush bench workload, state after 30 env-steps of contact.
time(n) = fixed + n * per_sub_step: the fixed part is launch + context set-up + read-out."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT)
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
B, T = 4096, 60
dt = torch.float32
q0, u, _ = push_workload(B, T, seed=0)
m = load_model(PUSHER_BLOB)
sim = BatchSim(m, B, dtype=dt, tape_capacity=1024)
U = torch.tensor(u, device="cuda", dtype=dt)
Q0 = torch.tensor(q0, device="cuda", dtype=dt)
ev = lambda: torch.cuda.Event(enable_timing=True)
res = []
NS = [int(a) for a in sys.argv[1:]] or [1, 2, 5, 10]
for n in NS:
    sim.reset(Q0, None, backward_flag=True)
    for t in range(30): sim.step(U[:, t].contiguous(), 5)
    tf, tb = [], []
    seeds = torch.randn(1, B, 7, device="cuda", dtype=dt), torch.randn(1, B, 6, device="cuda", dtype=dt), torch.randn(1, B, 390, device="cuda", dtype=dt)
    for t in range(30, 50):
        a, b = ev(), ev(); ut = U[:, t].contiguous()
        a.record(); sim.step(ut, n); b.record(); torch.cuda.synchronize(); tf.append(a.elapsed_time(b) * 1e3)
    for t in range(20):
        a, b = ev(), ev()
        a.record(); sim.backward_episode(1, n, *seeds); b.record(); torch.cuda.synchronize(); tb.append(a.elapsed_time(b) * 1e3)
    r = {"sub_steps": n, "fwd_us_median": float(np.median(tf)), "bwd_us_median": float(np.median(tb))}
    print(json.dumps(r), flush=True); res.append(r)
x = np.array([r["sub_steps"] for r in res], float)
if len(res) < 2: sys.exit(0)
for k in ("fwd_us_median", "bwd_us_median"):
    y = np.array([r[k] for r in res]); A = np.stack([np.ones_like(x), x], 1); c = np.linalg.lstsq(A, y, rcond=None)[0]
    print(k, "fixed %.1f us + %.1f us per sub-step" % (c[0], c[1]))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "step_launch_probe.json"), "w"), indent=1)
