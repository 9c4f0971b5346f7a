"""Throughput of the one-launch episode (tsim_rollout + tsim_backward_episode) vs per-step launches (GPU box)."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload

def run(B=4096, T=100, S=5, dtype=torch.float32, reps=3):
    dev = torch.device("cuda", 0)
    model = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
    q0_np, u_np, _ = push_workload(B, T, seed=0)
    sim = BatchSim(model, B, device="cuda:0", dtype=dtype, tape_capacity=T * S)
    q0 = torch.tensor(q0_np, device=dev, dtype=dtype)
    u = torch.tensor(u_np, device=dev, dtype=dtype).transpose(0, 1).contiguous()
    wq = torch.ones(T, B, sim.ndof_r, device=dev, dtype=dtype); wv = torch.ones(T, B, sim.ndof_var, device=dev, dtype=dtype)
    wt = torch.ones(T, B, sim.ndof_tactile, device=dev, dtype=dtype) * 100
    res = {}
    for mode in ("fwd", "fwd+bwd"):
        def episode():
            sim.reset(q0, None, backward_flag=(mode != "fwd"))
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(); o = sim.rollout(u, S); e1.record()
            if mode != "fwd":
                sim.backward_episode(T, S, wq, wv, wt)
            e2.record()
            return o, (e0, e1, e2)
        episode(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): o, ev = episode()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[mode] = {"env_steps_per_s": B * T * reps / dt, "fwd_ms_per_env_step": ev[0].elapsed_time(ev[1]) / T,
                     "bwd_ms_per_env_step": ev[1].elapsed_time(ev[2]) / T, "nonconverged": int((o["status"] != 0).sum())}
    return res

if __name__ == "__main__":
    out = {}
    for dt_, name in ((torch.float32, "f32"), (torch.float64, "f64")):
        out[name] = run(dtype=dt_)
        print(name, json.dumps(out[name]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rollout_probe.json"), "w"), indent=1)
