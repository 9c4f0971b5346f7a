"""Probe: does splitting the batch into G groups on G HIP streams hide the Newton stragglers? (GPU box)"""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload

def run(G, B=4096, T=100, S=5, dtype=torch.float32, reps=2, bwd=True):
    dev = torch.device("cuda", 0)
    model = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
    q0_np, u_np, _ = push_workload(B, T, seed=0)
    Bg = B // G
    sims = [BatchSim(model, Bg, device="cuda:0", dtype=dtype, tape_capacity=T * S) for _ in range(G)]
    if G > 1:
        for s_ in sims:
            s_.set_lanes_per_env(16)                     # small groups, but together they should still fill the chip
    streams = [torch.cuda.Stream(dev) for _ in range(G)]
    q0 = [torch.tensor(q0_np[g * Bg:(g + 1) * Bg], device=dev, dtype=dtype) for g in range(G)]
    u = [torch.tensor(u_np[g * Bg:(g + 1) * Bg], device=dev, dtype=dtype).transpose(0, 1).contiguous() for g in range(G)]
    s0 = sims[0]
    wq = torch.ones(Bg, s0.ndof_r, device=dev, dtype=dtype); wv = torch.ones(Bg, s0.ndof_var, device=dev, dtype=dtype)
    wt = torch.ones(Bg, s0.ndof_tactile, device=dev, dtype=dtype) * 100
    outs = [{} for _ in range(G)]
    torch.cuda.synchronize()
    def episode():
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                sims[g].reset(q0[g], None, backward_flag=bwd)
        for t in range(T):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    sims[g].step(u[g][t], S, out=outs[g])
        if bwd:
            for t in range(T):
                for g in range(G):
                    with torch.cuda.stream(streams[g]):
                        sims[g].backward_steps(S, wq, wv, wt)
    episode(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): episode()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * T * reps / dt

if __name__ == "__main__":
    res = {}
    for dt_, name in ((torch.float32, "f32"), (torch.float64, "f64")):
        for G in (1, 2, 4, 8):
            for bwd in (True, False):
                v = run(G, dtype=dt_, bwd=bwd)
                res["%s_G%d_%s" % (name, G, "fb" if bwd else "f")] = v
                print(name, "G", G, "fwd+bwd" if bwd else "fwd", "%.0f env-steps/s" % v, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "stream_groups_probe.json"), "w"), indent=1)
