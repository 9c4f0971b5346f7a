"""bench.py end to end on the GPU (small batch, short windows): ONE stdout line under the limit with the contract's fields, the per-kernel roofline from the
library's own HIP events — and a leg that crashes its process (as the segmentation fault inside a graph capture did in round 6) costs that leg only:
the optional legs run in child processes, the headline and the other legs still arrive.  The reference prints its timing the same way, one short line:
examples/RollingBallExp/test_sim_speed.py:102-104."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench(extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--batch", "512", "--episode", "6", "--repeats", "2"] + extra,
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) < 6144, (len(lines), [len(l) for l in lines])
    return json.loads(lines[0])


def test_line_with_legs_in_child_processes_and_a_crashing_leg():
    j = _bench(["--legs", "step_mode,push_fwd,readout"], env={"TSIM_BENCH_CRASH_LEG": "push_fwd"})
    assert j["value"] > 0 and j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2 and j["dtype"] == "f32" and j["unit"] == "env-steps/s"
    rl = j["roofline"]
    assert rl["bound"] == "hbm" and rl["kernel"] == "k_forward" and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-6
    assert set(rl["per_kernel"]) == {"k_forward", "k_taxels", "k_backward"} and all(v["ms"] > 0 for v in rl["per_kernel"].values())
    assert abs(rl["per_kernel"]["k_forward"]["algorithmic_bytes"] - 356 * 512 * 6) < 1e-4 * 356 * 512 * 6      # its OWN bytes: u in, q / variables out, tape q / qd (the line rounds to 5 digits)
    assert "error" in j["sub"]["push_fwd"] and "push_fwd (error)" in j["legs"]["errors"]     # the leg that aborted ...
    assert j["sub"]["readout"]["GBps"] > 0 and j["launch"]["other_mode_value"] > 0          # ... cost nothing else
    assert os.path.exists(os.path.join(ROOT, "bench_detail.json"))


def test_legs_in_process_give_the_same_records():
    j = _bench(["--legs", "env_tables", "--legs-in-process"])
    k = _bench(["--legs", "env_tables"])
    for x in (j, k):
        assert x["sub"]["env_tables"]["value"] > 0 and x["sub"]["env_tables"]["kernel"].startswith("k_forward<float,8,false,") and "TsParam" in x["sub"]["env_tables"]["kernel"]
