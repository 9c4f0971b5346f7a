"""Probe: the bench's timed region (episodes of F frames: one forward launch, one adjoint launch) with the 4096 environments as G
groups on G HIP streams (one tsim_batch each).  A launch lasts as long as its slowest wavefront (20-frame launches: mean 297
evaluation rounds per wavefront, maximum 403), and the adjoint cannot start before the forward has finished; with G independent
chains the adjoint of one group runs under the forward tail of another.  GPU box."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB

def run(G, F, B=4096, S=5, dtype=torch.float32, reps=10, lanes=16, stagger=False):
    dev = torch.device("cuda", 0)
    model = load_model(PUSHER_BLOB)
    q0_np, u_np, _ = push_workload(B, F, seed=0)
    Bg = B // G
    sims = [BatchSim(model, Bg, device="cuda:0", dtype=dtype, tape_capacity=F * S) for _ in range(G)]
    for s_ in sims: s_.set_lanes_per_env(lanes)
    streams = [torch.cuda.Stream(dev) for _ in range(G)] if G > 1 else [torch.cuda.current_stream()]
    sl = lambda g: slice(g * Bg, (g + 1) * Bg)
    q0 = [torch.tensor(q0_np[sl(g)], device=dev, dtype=dtype) for g in range(G)]
    u = [torch.tensor(u_np[sl(g)], device=dev, dtype=dtype).transpose(0, 1).contiguous() for g in range(G)]
    s0 = sims[0]
    mk = lambda d, v: (torch.ones(F, Bg, d, device=dev, dtype=dtype) * v)
    wq, wv, wt = mk(s0.ndof_r, 1.0), mk(s0.ndof_var, 1.0), mk(s0.ndof_tactile, 100.0)
    torch.cuda.synchronize()
    def episode():
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                sims[g].reset(q0[g], None, backward_flag=True)
                sims[g].rollout(u[g], S)
                if not stagger: sims[g].backward_episode(F, S, wq, wv, wt)
        if stagger:                       # all forwards first, then the adjoints in the same order
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    sims[g].backward_episode(F, S, wq, wv, wt)
    for _ in range(3): episode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): episode()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return B * F / dt, dt / F * 1e3

if __name__ == "__main__":
    res = []
    for F in (20, 100):
        for G in (1, 2, 4, 8):
            for stagger in (False, True):
                if G == 1 and stagger: continue
                v, ms = run(G, F, reps=10 if F == 20 else 4, stagger=stagger)
                r = {"frames": F, "groups": G, "forwards_first": stagger, "env_steps_per_s": v, "ms_per_env_step": ms}
                print(json.dumps(r), flush=True); res.append(r)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "episode_groups_probe.json"), "w"), indent=1)
