"""How full the SIMDs are during one forward episode launch of a bench workload (A/B build with -DTS_ROUND_STATS: rounds and shader clocks of every
wavefront; GPU box): SIMD-idle fraction = 1 - sum of the wavefronts' busy clocks / (resident wavefront slots x launch duration), the cost of a
round, the launch against its slowest wavefront.  What a per-slot work queue could recover is bounded by the idle fraction that is NOT the
slowest environment's own chain.
   python tools/build_ab.py rounds -DTS_ROUND_STATS
   TSIM_HIP_LIB=tactilesimulation_amd/csrc/ab/libtsim_rounds.so python tools/queue_stats.py dclaw|insertion|push [B]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from tactilesimulation_amd.host.batch import BatchSim

name = sys.argv[1] if len(sys.argv) > 1 else "dclaw"
asset_, B0, T, fwd_only, cfg = bench.WORKLOADS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else B0
if name == "push":
    T = 20
dev = torch.device("cuda:0")
wl = bench.make_workload(name, B, T, 5, 0, dev, torch.float32)
sim = BatchSim(wl["model"], B, dtype=torch.float32, tape_capacity=0)
sim.set_solver_options(cross_kinks=True, eval_budget=0)
info = sim.launch_info()
ns = 64 // info["lanes_per_env"]
res = []
for helpers in (0, 1):
    sim.set_option(BatchSim.OPT_TRIAL_HELPERS, helpers)
    for rep in range(3):
        sim.reset(wl["q0"], None, backward_flag=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        ro = sim.rollout(wl["u"], wl["S"], tactile_mask=wl.get("tactile_mask"), want_tactile=False)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    rounds = ro["status"].cpu().numpy().astype(np.int64)[::ns]          # per wavefront (every slot of a wavefront reports the wavefront's numbers)
    cyc = sim.last_gnorm().astype(np.float64)[::ns]
    ev = sim.last_evals().astype(np.int64)
    waves, slots = len(rounds), 1024
    # shader clock: the slowest wavefront of a launch whose wavefronts are all resident at once spans (nearly) the whole launch
    res.append({"helpers": helpers, "launch_ms": round(ms, 3), "wavefronts": waves, "wavefronts_per_simd": waves / slots, "rounds_mean": float(rounds.mean()), "rounds_max": int(rounds.max()),
                "clocks_per_round_mean": float((cyc / rounds).mean()), "busy_clocks_sum": float(cyc.sum()), "busy_clocks_max": float(cyc.max()),
                "evals_mean": float(ev.mean()), "evals_max": int(ev.max())})
# clock rate from the all-resident case if this is one (else report clocks only)
out = {"workload": name, "B": B, "frames": T, "lanes": info["lanes_per_env"], "legs": res}
for r in res:
    if r["wavefronts"] <= 1024:
        r["clock_ghz_if_slowest_wave_spans_the_launch"] = r["busy_clocks_max"] / (r["launch_ms"] * 1e6)
print(json.dumps(out))
