"""tsim_readout on RollingBall's 200 x 200 taxels, 256 environments, 30 calls: run under `rocprofv3 --kernel-trace --stats` to get the
per-kernel durations of k_readout and k_taxels (A/B of library builds via TSIM_HIP_LIB; profiles/r03_readout_ab.md)."""
import os, sys, json, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT)
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for dt in (torch.float32, torch.float64):
    r = bench.readout_leg(dt, torch.device("cuda", 0), B=B, reps=30)
    print(json.dumps({"dtype": str(dt)[6:], "B": B, "us_call_min": round(r["ms"] * 1e3, 1), "GB_per_s": round(r["achieved"], 1)}), flush=True)
