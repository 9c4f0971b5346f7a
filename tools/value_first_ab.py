"""Forward-only episode launches of a bench workload with the value-first trials (TSIM_OPT_VALUE_FIRST) off / on, in ONE process on ONE batch
(GPU box).  usage: python tools/value_first_ab.py dclaw|insertion|push_fwd [B] [launches]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tactilesimulation_amd.host.batch import BatchSim

name = sys.argv[1] if len(sys.argv) > 1 else "dclaw"
asset_, B0, T, fwd_only, cfg = bench.WORKLOADS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else B0
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
wl = bench.make_workload(name, B, T, 5, 0, dev, torch.float32)
S, fps = wl["S"], wl["fps"]
sim = BatchSim(wl["model"], B, dtype=torch.float32, tape_capacity=0)
sim.set_solver_options(cross_kinks=True, eval_budget=0)
mask = wl.get("tactile_mask")


def launches(first, helpers):
    sim.set_option(BatchSim.OPT_VALUE_FIRST, first); sim.set_option(BatchSim.OPT_TRIAL_HELPERS, helpers)
    ms = []
    for i in range(N + 1):
        sim.reset(wl["q0"], None, backward_flag=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        ro = sim.rollout(wl["u"], S, tactile_mask=mask)
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ev = sim.last_evals()
    env_steps = B * T / fps
    return {"value_first": first, "helpers": helpers, "launch_ms": [round(x, 3) for x in ms], "median_ms": round(float(np.median(ms[1:])), 3),
            "env_steps_per_s": round(env_steps / float(np.median(ms[1:])) * 1e3), "evals_max": int(ev.max()), "evals_mean": round(float(ev.mean()), 1),
            "nonconverged": int((ro["status"] != 0).sum()), "checksum_q": float(ro["q"].double().sum())}


out = {"workload": name, "B": B, "frames": T, "variant": sim.kernel_variant(), "lanes": sim.launch_info()["lanes_per_env"], "legs": []}
for first, helpers in ((0, 1), (1, 1), (0, 0), (1, 0), (0, 1), (1, 1)):
    out["legs"].append(launches(first, helpers))
print(json.dumps(out))
