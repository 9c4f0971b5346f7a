"""Where the difference between the HIP-event time of an episode's forward segment and rocprof's kernel duration goes (GPU box): N back-to-back
episode launches bracketed by ONE event pair against single launches bracketed each, with and without the tactile output (k_taxels_small) and the
tape.  usage: python tools/archive/launch_gap_probe.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from tactilesimulation_amd.host.batch import BatchSim

dev = torch.device("cuda:0")
B, T, S = 4096, 20, 5
wl = bench.make_workload("push", B, T, S, 0, dev, torch.float32)
out = {}
for record in (True, False):
    sim = BatchSim(wl["model"], B, dtype=torch.float32, tape_capacity=T * S if record else 0)
    for want_tac in (True, False):
        def one():
            sim.reset(wl["q0"], None, backward_flag=record)
            return sim.rollout(wl["u"], S, want_tactile=want_tac)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        singles = []
        for _ in range(6):
            sim.reset(wl["q0"], None, backward_flag=record)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sim.rollout(wl["u"], S, want_tactile=want_tac); e1.record()
            torch.cuda.synchronize()
            singles.append(e0.elapsed_time(e1))
        N = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(N):
            one()
        e1.record(); torch.cuda.synchronize()
        out["record_%d_tactile_%d" % (record, want_tac)] = {"single_launch_ms": [round(x, 3) for x in singles], "back_to_back_ms_per_episode": round(e0.elapsed_time(e1) / N, 3)}
print(json.dumps(out))
