"""Launch shape (lanes per environment, blocks, LDS per block) the library picks for every model at its BASELINE batch size."""
import os, sys, json, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT)
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset
for name, B in (("pusher", 4096), ("pusher", 1024), ("dclaw_position_control", 2048), ("tactile_insertion", 4096), ("stable_grasp", 4096), ("tactile_pad", 256)):
    for dt in (torch.float32, torch.float64):
        s = BatchSim(load_model(asset(name)), B, dtype=dt, tape_capacity=8)
        print(name, B, str(dt)[6:], json.dumps(s.launch_info()), flush=True)
        del s
