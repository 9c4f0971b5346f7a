"""Shader-clock stamps of the adjoint kernel's first sub-steps (wavefront 0), bench workload (GPU box)."""
import os, sys, json, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.host import capi
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
B, T, S = 4096, 20, 5
m = load_model(PUSHER_BLOB)
q0, u, _ = push_workload(B, T, seed=0)
dt = torch.float32
sim = BatchSim(m, B, dtype=dt, tape_capacity=T * S)
cyc = torch.zeros(32, dtype=torch.int64, device="cuda")
for rep in range(2):
    sim.reset(torch.tensor(q0, device="cuda", dtype=dt), None, backward_flag=True)
    sim.rollout(torch.tensor(u, device="cuda", dtype=dt).transpose(0, 1).contiguous(), S)
    cyc.zero_()
    capi.lib().tsim_debug_stamps(sim._h, C.c_void_p(cyc.data_ptr()))
    w = lambda d: torch.ones(T, B, d, device="cuda", dtype=dt)
    sim.backward_episode(T, S, w(7), w(6), w(390))
    torch.cuda.synchronize()
    capi.lib().tsim_debug_stamps(sim._h, None)
c = cyc.cpu().numpy()
n = int((c != 0).sum())
print(json.dumps({"n": n, "deltas": np.diff(c[:n]).tolist(), "names": "top | record->LDS | phase1 | (seeded: output_vjp) | solve | [stage value, stage tangent, contacts] | fold | phase3 | M z + lam update"}))
