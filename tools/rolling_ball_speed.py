"""The reference's only simulator-speed measurement, examples/RollingBallExp/test_sim_speed.py:72-104 (350 steps of
tactile_pad.xml with a 200x200 tactile read-out every 5 steps, FPS = 350 / elapsed), replayed (i) through the redmax_py
shim on the GPU (one environment, like the reference) and (ii) with the fp64 CPU oracle."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tactilesimulation_amd", "compat"))
import redmax_py as redmax  # noqa: E402
from tactilesimulation_amd.model.compiler import load_model  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

acts = np.asarray([[0, 0, .2]] * 100 + [[.1, 0, .2]] * 50 + [[-.2, 0, .2]] * 50 + [[0, .1, .2]] * 50 + [[0, -.2, .2]] * 100, dtype=np.float64)
m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "tactile_pad.npz"))
res = {}
for name, dt in (("gpu_shim_f32", torch.float32), ("gpu_shim_f64", torch.float64)):
    sim = redmax.Simulation(m, dtype=dt)
    for rep in range(2):
        sim.reset(backward_flag=False)
        torch.cuda.synchronize(); t0 = time.time()
        for i in range(len(acts)):
            sim.set_u(acts[i]); sim.forward(1, verbose=False, test_derivatives=False)
            if i % 5 == 0:
                tac = sim.get_tactile_force_vector().copy().reshape(200, 200, 3)
        torch.cuda.synchronize(); res[name] = 350 / (time.time() - t0)
o = OracleSim(m)
o.reset(np.zeros(9)); t0 = time.time()
for i in range(len(acts)):
    o.forward(acts[i], 1)
    if i % 5 == 0:
        o.outputs()
res["cpu_oracle_f64_1thread"] = 350 / (time.time() - t0)
print(json.dumps({"metric": "RollingBall test_sim_speed FPS (350 steps, 40 000-taxel read-out every 5), 1 env", **res}))
