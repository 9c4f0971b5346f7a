"""k_readout on RollingBall's 200 x 200 taxels at B = 1, 16, 256, 1024: time per read-out and GB/s written (bench.py readout_leg at other batch sizes)."""
import os, sys, json, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT)
import bench
for dt in (torch.float32, torch.float64):
    for B in (1, 16, 256, 1024):
        r = bench.readout_leg(dt, torch.device("cuda", 0), B=B, reps=10)
        print(json.dumps({"dtype": str(dt)[6:], "B": B, "us": round(r["ms"] * 1e3, 1), "GB_per_s": round(r["achieved"], 1), "frac_of_hbm_peak": round(r["frac"], 4)}), flush=True)
