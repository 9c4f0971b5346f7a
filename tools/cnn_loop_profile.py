"""The closed GD epoch on the per-step HIP graph with the reference's CNN policy (bench leg closed_loop_cnn_per_step_graph), for
`rocprofv3 --kernel-trace --stats`: which kernels the CNN adds to an env-step.  usage: python tools/cnn_loop_profile.py [mlp|cnn] [epochs]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_legs as BL                                                       # noqa: E402
from tactilesimulation_amd.model.compiler import load_model                  # noqa: E402
from tactilesimulation_amd import workloads as W                             # noqa: E402

if __name__ == "__main__":
    cnn = (sys.argv[1] if len(sys.argv) > 1 else "cnn") == "cnn"
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    r = BL.closed_loop_leg(load_model(W.asset("pusher")), 4096, 100, torch.float32, torch.device("cuda:0"), epochs=epochs, cnn=cnn)
    print({k: r[k] for k in ("value", "s_per_epoch")})
