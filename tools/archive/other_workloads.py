"""Forward-only throughput of the non-headline BASELINE.json configs on ONE GPU (per-GPU share of the 8-GPU batch):
DClaw (configs[3]: 16 384 envs / 8 = 2 048 per GPU, frame_skip 5) and TactileInsertion (configs[4]: 32 768 / 8 = 4 096 per
GPU, 45 single sub-steps per episode). Synthetic inputs as in tests/test_gpu_models.py. Not the bench.py headline."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model  # noqa: E402
from tactilesimulation_amd.host.batch import BatchSim  # noqa: E402
from tests.test_gpu_models import _inputs  # noqa: E402


def run(name, B, T, S, dtype=torch.float32, reps=3, variant=None):
    """variant: one of workloads.synthetic_variant's names (BASELINE.json's worded taxel sizes) on the inputs of `name`."""
    if variant:
        from tactilesimulation_amd.workloads import synthetic_variant
        m = synthetic_variant(variant)
    else:
        m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", name + ".npz"))
    if name == "pusher":
        from tactilesimulation_amd.workloads import push_workload
        q0, u, _ = push_workload(64, T, seed=0)
    else:
        q0, u = _inputs(name, m, 64, T)
    q0 = np.tile(q0, (B // 64, 1)); u = np.tile(u, (B // 64, 1, 1))
    sim = BatchSim(m, B, dtype=dtype, tape_capacity=0)
    q0d = torch.tensor(q0, device="cuda", dtype=dtype); ud = torch.tensor(u, device="cuda", dtype=dtype).transpose(0, 1).contiguous()
    out = {}
    best = None
    for r in range(reps):
        sim.reset(q0d, None, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(T):
            sim.step(ud[t], S, out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    best_ep = None                                  # the same episode as one launch (tsim_rollout)
    for r in range(reps):
        sim.reset(q0d, None, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ro = sim.rollout(ud, S)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best_ep = dt if best_ep is None else min(best_ep, dt)
    # forward + adjoint, one launch each way (seeds on q / variables / tactile of every frame)
    simr = BatchSim(m, B, dtype=dtype, tape_capacity=T * S)
    ones = lambda d: torch.ones(T, B, d, device="cuda", dtype=dtype)
    wq, wv, wt = ones(simr.ndof_r), (ones(simr.ndof_var) if simr.ndof_var else None), ones(simr.ndof_tactile)
    best_fb = None
    for r in range(reps):
        simr.reset(q0d, None, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        simr.rollout(ud, S)
        simr.backward_episode(T, S, wq, wv, wt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best_fb = dt if best_fb is None else min(best_fb, dt)
    return {"model": variant or name, "ndof_tactile": sim.ndof_tactile, "B": B, "env_steps": T, "substeps_per_env_step": S, "dtype": str(dtype),
            "episode_fwd_adjoint_env_steps_per_s": B * T / best_fb,
            "env_steps_per_s": B * T / best, "substeps_per_s": B * T * S / best, "nonconverged_last": int((out["status"] != 0).sum()),
            "episode_launch_env_steps_per_s": B * T / best_ep, "episode_nonconverged": int((ro["status"] != 0).sum()),
            "launch_shape": sim.launch_info()}


if __name__ == "__main__":
    res = [run("dclaw_position_control", 2048, 10, 5), run("tactile_insertion", 4096, 14, 5),
           run("dclaw_position_control", 2048, 10, 5, torch.float64), run("tactile_insertion", 4096, 14, 5, torch.float64),
           # BASELINE.json's worded sizes (synthetic taxel layouts on the real models), per-GPU share of the 8-GPU batches
           run("pusher", 1024, 20, 5, variant="pusher_13x13"), run("pusher", 4096, 20, 5, variant="pusher_13x13"),
           run("dclaw_position_control", 2048, 10, 5, variant="dclaw_9x9"), run("tactile_insertion", 4096, 14, 5, variant="tactile_insertion_32x32")]
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "other_workloads.json"), "w"), indent=1)
