"""Shader-clock stamps of ONE residual evaluation + solve (tsim_debug_eval with cycles) of any bench workload, at the states a roll-out of that
workload reaches (GPU box).  usage: python tools/eval_stamps.py dclaw|insertion|push [lanes] [frames]
stamps: start | link sweep | per group of 4 contact pairs: stage values, stage tangents, contacts, (fold ends the group) | projection | solve"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tactilesimulation_amd.host.batch import BatchSim

name = sys.argv[1] if len(sys.argv) > 1 else "dclaw"
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
B = 1024
wl = bench.make_workload(name, B, max(frames, 2), 5, 0, dev, torch.float32)
m, S = wl["model"], wl["S"]
sim = BatchSim(m, B, dtype=torch.float32, tape_capacity=0)
sim.reset(wl["q0"], None, backward_flag=False)
ro = sim.rollout(wl["u"][:frames], S, want_qd=True, want_tactile=False)
q, qd = ro["q"][-1], ro["qd"][-1]
q1 = q + float(m.h) * qd
uu = wl["u"][frames - 1]
if lanes:
    sim.set_lanes_per_env(lanes)
out = {"workload": name, "lanes": sim.launch_info()["lanes_per_env"], "frames": frames, "variant": sim.kernel_variant()}
for cull in (1, 0):
    sim.set_option(BatchSim.OPT_PAIR_CULL, cull)
    for _ in range(3):
        g, H, cyc = sim.debug_eval(q1, q, qd, uu, cycles=True)
    c = cyc.double().cpu().numpy()
    c = c[c[:, 0] != 0]
    n = int((c[0] != 0).sum())
    d = np.diff(c[:, :n], axis=1)
    out["cull_%d" % cull] = {"stamp_deltas_mean": [round(float(x)) for x in d.mean(0)], "total_mean": round(float(d.sum(1).mean())), "total_max": round(float(d.sum(1).max())), "waves": int(c.shape[0])}
print(json.dumps(out))
