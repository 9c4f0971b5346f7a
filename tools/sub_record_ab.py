"""The bench's sub-record legs (other BASELINE configs, generic kernels) under the environment's A/B switches (GPU box):
   TSIM_NO_FREE_RUN=1 / TSIM_INKERNEL_READOUT=1 python tools/sub_record_ab.py push_fwd dclaw insertion"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    r = bench.sub_record(name, "f32", dev)
    print(json.dumps({"leg": name, "value": round(r["value"]), "ms_per_step": round(r["ms_per_step"], 4), "kernel_ms": {k: v["ms"] for k, v in r["roofline"]["per_kernel"].items()}, "idle_share": r["idle_share"], "evals": r["residual_evals_per_substep_last_launch"],
                      "lanes": r["launch_shape"].get("lanes_per_env"), "nonconverged": r["nonconverged_envs"], "switches": {k: v for k, v in os.environ.items() if k.startswith("TSIM_")}}))
