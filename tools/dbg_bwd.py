import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
m = load_model(PUSHER_BLOB)
dt = torch.float64 if sys.argv[1] == "f64" else torch.float32
lanes = int(sys.argv[2])
B, T, S = 8, 3, 5
q0, u, _ = push_workload(B, T, seed=1)
sim = BatchSim(m, B, dtype=dt, tape_capacity=T * S)
sim.set_lanes_per_env(lanes)
print(sim.launch_info(), flush=True)
sim.reset(torch.tensor(q0), None, backward_flag=True)
for t in range(T):
    o = sim.step(torch.tensor(u[:, t]), S)
torch.cuda.synchronize(); print("fwd ok", o["q"][0, :3].tolist(), flush=True)
du = sim.backward_steps(S, torch.ones(B, 7, dtype=torch.float64), torch.ones(B, 6, dtype=torch.float64), torch.ones(B, 390, dtype=torch.float64))
torch.cuda.synchronize(); print("bwd ok", du[0, 0].tolist(), flush=True)
