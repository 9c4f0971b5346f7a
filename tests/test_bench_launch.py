"""bench.py's rank handling: `python bench.py --gpus N` without a launcher starts its own N ranks
(torch.distributed.run, one process per GPU — the reference's only parallel launch is SubprocVecEnv's process per worker,
externals/pytorch-a2c-ppo-acktr-gail/a2c_ppo_acktr/envs.py:100-108), and any mismatch between --gpus, the launcher's world size and the
visible devices is an error with a non-zero exit code, never a silent single-rank run.  CPU part: `--plumbing-only` (process group + the
118 296-B policy-gradient all-reduce on gloo, no simulator); the same checks with the simulator on a GPU: test_gpu_sharded.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--backend", "gloo", "--plumbing-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_in_allreduce"] == 2 and j["allreduce_bytes"] == 118296


def test_single_rank_plumbing():
    j = _json_line(_run(["--plumbing-only"]).stdout)
    assert j["n_gpus"] == 1 and j["ranks_in_allreduce"] == 1


def test_mismatches_fail_loudly():
    r = _run(["--gpus", "2", "--plumbing-only"], env={"WORLD_SIZE": "4", "RANK": "0"})          # launcher and --gpus disagree
    assert r.returncode != 0 and "WORLD_SIZE 4" in r.stderr
    r = _run(["--gpus", "1", "--plumbing-only"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "disagree" in r.stderr
    import torch
    if torch.cuda.device_count() < 2:                                                           # RCCL ranks need a device each
        r = _run(["--gpus", "2"])
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
