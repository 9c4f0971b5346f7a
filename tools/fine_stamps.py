"""Fine-grained shader-clock stamps of one evaluation + solve (A/B build with -DTS_FINE_STAMPS; GPU box):
   TSIM_HIP_LIB=tactilesimulation_amd/csrc/ab/libtsim_fine.so TSIM_LPE=16 python tools/fine_stamps.py"""
import os, sys, json, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
from oracle.oracle import OracleSim
m = load_model(PUSHER_BLOB)
o = OracleSim(m)
q0s, us, _ = push_workload(64, 10, seed=4)
st = []
for e in range(64):
    o.reset(q0s[e])
    for t in range(4 + e % 6):
        o.forward(us[e, t], 5)
    q, qd = o.state()
    st.append((q + m.h * qd, q, qd, us[e, 9]))
B = 256
q1, q0, qd0, u = (torch.tensor(np.tile(np.stack([s[i] for s in st]), (B // 64, 1))) for i in range(4))
sim = BatchSim(m, B, dtype=torch.float32, tape_capacity=4)
for _ in range(3):
    g, H, cyc = sim.debug_eval(q1, q0, qd0, u, cycles=True)
c = cyc.double().cpu().numpy()
c = c[c[:, 0] != 0]
counts = collections.Counter(int((r != 0).sum()) for r in c)
n = counts.most_common(1)[0][0]
sel = c[[int((r != 0).sum()) == n for r in c]]
d = np.diff(sel[:, :n], axis=1)
print(json.dumps({"stamp_counts": dict(counts), "n": n, "waves": len(sel), "deltas_mean": d.mean(0).round(0).tolist(), "total": float(d.sum(1).mean()),
                  "lpe": os.environ.get("TSIM_LPE", "auto")}))
