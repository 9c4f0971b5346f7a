"""TactilePush fwd + adjoint (B = 4096, 20 env-steps x 5 sub-steps per launch, the driver's timed region) on each kernel instantiation a batch
can be on: fully static (the XML's model), structure-static after an update_* edit, structure-static with per-environment tables, generic with
the same tables (GPU box).  usage: python tools/param_tables_probe.py [B] [T]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset, push_workload

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S, DEV = 5, "cuda:0"
m = load_model(asset("pusher"))
q0, u, _ = push_workload(B, 100, seed=0); u = u[:, :T]      # bench.py's inputs: the first T frames of the 100-frame table
q0 = torch.tensor(q0, device=DEV, dtype=torch.float32)
u = torch.tensor(u, device=DEV, dtype=torch.float32).transpose(0, 1).contiguous()
g = torch.Generator().manual_seed(1)
wq, wv, wt = (torch.randn(T, B, n, generator=g).to(DEV) for n in (7, 6, 390))


def timed(sim, reps=7):
    ts = []
    for i in range(reps + 2):
        sim.reset(q0, None, backward_flag=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        ro = sim.rollout(u, S)
        sim.backward_episode(T, S, wq, wv, wt)
        e1.record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    return {"variant": sim.kernel_variant(), "lanes": sim.launch_info()["lanes_per_env"], "ms_per_env_step": round(ms / T, 4), "env_steps_per_s": round(B * T / ms * 1e3),
            "min_max_ms": [round(ts[0] / T, 4), round(ts[-1] / T, 4)], "nonconverged": int((ro["status"] != 0).sum())}


out = {}
sim = BatchSim(m, B, dtype=torch.float32, tape_capacity=T * S)
out["static"] = timed(sim)
sim.set_env_tables(sim.base_tables())
out["param_tables"] = timed(sim)
sim.set_static(False)
out["generic_tables"] = timed(sim)
sim.set_env_tables(None)
out["generic"] = timed(sim)
sim.set_static(True)
import copy
import tactilesimulation_amd.model.blob as BL
m2 = copy.copy(m); m2.F = m.F.copy(); m2.F[m.I[BL.TSIM_IH_FOFF_PAIR] + BL.TSIM_PF_SIZE + BL.TSIM_PF_KN] *= 1.0000001
sim.update_model(m2)
out["param_shared"] = timed(sim)
print(json.dumps({"B": B, "T": T, "legs": out}))
