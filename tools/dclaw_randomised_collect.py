"""Roll-out collection on BatchedDClawRotateEnv with the reference's per-reset domain randomisation: continuous draws per environment on the
device (randomize=True) against the round-3/4 pool of 16 compiled variants, eager and from one HIP graph per step (GPU box).
usage: python tools/dclaw_randomised_collect.py [B] [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tactilesimulation_amd.envs.dclaw_rotate import BatchedDClawRotateEnv, GraphedCollector

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = int(sys.argv[3]) if len(sys.argv) > 3 else 128
out = {"B": B, "steps": T, "eval_budget": budget, "legs": []}
for name, kw in (("none", {}), ("pool_of_16_variants", {"variants": 16}), ("continuous_per_environment", {"randomize": True})):
    env = BatchedDClawRotateEnv(B, dtype=torch.float32, seed=0, **kw)
    env.sim.set_solver_options(cross_kinks=True, eval_budget=budget)      # roll-out collection: a creeping sub-step must not stall the batch (flagged in status)
    torch.manual_seed(1)
    W = torch.randn(env.obs_dim, env.act_dim, device="cuda") * 0.02
    policy = lambda obs: torch.tanh(obs @ W) + 0.3 * torch.randn(B, 9, device="cuda")
    col = GraphedCollector(env, policy)
    for _ in range(10):
        col.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flagged = torch.zeros((), device="cuda", dtype=torch.long)
    resets = torch.zeros((), device="cuda", dtype=torch.long)
    for _ in range(T):
        col.step()
        flagged += (col.status != 0).sum(); resets += col.done.sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["legs"].append({"randomisation": name, "env_steps_per_s": round(B * T / dt), "ms_per_env_step": dt / T * 1e3, "flagged_env_steps": int(flagged), "resets": int(resets),
                        "kernel": env.sim.kernel_variant(), "lanes": env.sim.launch_info()["lanes_per_env"]})
    del col, env
    torch.cuda.empty_cache()
print(json.dumps(out))
