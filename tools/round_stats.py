"""Rounds (residual evaluations of the wavefront) and shader clocks per wavefront of one forward episode launch (A/B build with -DTS_ROUND_STATS; GPU box):
   python tools/build_ab.py rounds -DTS_ROUND_STATS
   TSIM_HIP_LIB=tactilesimulation_amd/csrc/ab/libtsim_rounds.so python tools/round_stats.py            (TSIM_NO_FREE_RUN=1 TSIM_INKERNEL_READOUT=1: the lock-step loop)"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
B, T, S = int(os.environ.get("BATCH", "4096")), int(os.environ.get("STEPS", "20")), 5
from tactilesimulation_amd import workloads as W
m = W.synthetic_variant("pusher_13x13") if os.environ.get("MODEL") == "13x13" else load_model(PUSHER_BLOB)
q0, u, _ = push_workload(B, 100, seed=0)       # the bench's table (episodes of 100 env-steps), its first T frames
u = u[:, :T]
REC = os.environ.get("RECORD", "1") == "1"
sim = BatchSim(m, B, dtype=torch.float32, tape_capacity=T * S if REC else 0)
ut = torch.tensor(u, device="cuda:0", dtype=torch.float32).transpose(0, 1).contiguous()
out = {}
for rep in range(3):
    sim.reset(torch.tensor(q0, device="cuda:0", dtype=torch.float32), None, backward_flag=REC)
    ro = sim.rollout(ut, S)
    torch.cuda.synchronize()
    rounds = ro["status"].cpu().numpy().astype(np.int64)
    cyc = sim.last_gnorm().astype(np.float64)
    ev = sim.last_evals().astype(np.int64)
    out = {"rep": rep, "rounds_max": int(rounds.max()), "rounds_mean": float(rounds.mean()), "cycles_max": float(cyc.max()), "cycles_mean": float(cyc.mean()),
           "cycles_per_round_mean": float((cyc / rounds).mean()), "cycles_per_round_of_the_slowest": float(cyc[np.argmax(cyc)] / rounds[np.argmax(cyc)]),
           "rounds_of_the_slowest": int(rounds[np.argmax(cyc)]), "evals_mean": float(ev.mean()), "evals_max": int(ev.max())}
    print(json.dumps(out))
