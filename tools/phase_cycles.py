"""Shader-clock stamps of one residual evaluation + solve of a lone wavefront, per lanes-per-environment shape (GPU box).
usage: TSIM_LPE=16 python tools/phase_cycles.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from accuracy_report import phase_cycles
r = phase_cycles(torch.float32, 256)
r["lpe"] = os.environ.get("TSIM_LPE", "64")
r["stamps"] = "start | phase1 | stage value | stage tangent | contacts | fold | phase3 | solve"
print(json.dumps(r))
