"""The generic-kernel collection legs (D'Claw, TactileInsertion) against the batch size of ONE GPU (GPU box): a launch lasts its slowest
environment's chain of evaluations whatever the batch is, so a larger batch amortises it over more wavefronts taken in turn by each SIMD.
   python tools/sub_record_batch.py insertion 4096 8192 16384 32768"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench_legs as BL
dev = torch.device("cuda:0")
name = sys.argv[1]
for B in map(int, sys.argv[2:]):
    w = BL.WORKLOADS[name]
    BL.WORKLOADS[name] = (w[0], B) + w[2:]
    r = BL.sub_record(name, "f32", dev)
    print(json.dumps({"leg": name, "B": B, "value_M": round(r["value"] / 1e6, 3), "ms_per_step": round(r["ms_per_step"], 4), "kernel_ms": {k: round(v["ms"], 3) for k, v in r["roofline"]["per_kernel"].items()},
                      "idle_share": round(r["idle_share"], 3), "max_env_evals": r["residual_evals_per_substep_last_launch"]["max_env_total"], "mean_env_evals": round(r["residual_evals_per_substep_last_launch"]["mean_env_total"], 1),
                      "lanes": r["launch_shape"].get("lanes_per_env"), "blocks": r["launch_shape"].get("blocks"), "nonconverged_envs": r["nonconverged_envs"],
                      "switches": {k: v for k, v in os.environ.items() if k.startswith("TSIM_")}}), flush=True)
