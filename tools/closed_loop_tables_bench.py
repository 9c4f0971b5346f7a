"""The fused closed GD epoch (bench.py's closed_loop leg: B = 4096, 100 env-steps, policy inside the episode launches) on the XML's model, on an
edited model (update_* route) and with one randomised parameter table per environment (tsim_set_env_tables): env-steps/s and the kernel variant
each runs on.  GPU box:  python tools/closed_loop_tables_bench.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_legs as BL                                                             # noqa: E402
from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode, train_epoch_fused      # noqa: E402
from tactilesimulation_amd.model.compiler import load_model                        # noqa: E402
from tactilesimulation_amd import workloads as W                                   # noqa: E402
from test_gpu_closed_loop import _randomised_tables                                # noqa: E402


def run(mode, B=4096, T=100, epochs=3):
    dev, tdt = torch.device("cuda:0"), torch.float32
    model = load_model(W.asset("pusher"))
    env, q0, goal, dist_, actor, opt = BL._closed_loop_inputs(model, B, T, tdt, dev)
    if mode == "tables":
        env.sim.set_env_tables(_randomised_tables(env.sim, model, B))
    elif mode == "base_tables":
        env.sim.set_env_tables(env.sim.base_tables())
    ep = FusedPushEpisode(env, actor, T)
    train_epoch_fused(ep, opt, q0, goal, dist_, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        train_epoch_fused(ep, opt, q0, goal, dist_, B)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"mode": mode, "kernel_variant": env.sim.kernel_variant(), "value": B * T * epochs / dt, "s_per_epoch": dt / epochs, "nonconverged_envs": int((ep.status != 0).sum().item())}


if __name__ == "__main__":
    print(json.dumps([run(m) for m in ("shared", "base_tables", "tables")], indent=1))
