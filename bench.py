#!/usr/bin/env python
"""bench.py — env-steps/s (forward + adjoint) of the batched TactilePush step on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: launched under torch.distributed.run)
A "step" = one env-step (5 implicit BDF1 sub-steps, tactile read-out) forward AND its adjoint for one batch of B = 4096 TactilePush
environments per GPU (BASELINE.json configs[2]: gd_tactile fwd+adjoint, batch 4096).  A timed window is exactly K env-steps, run as episodes of
min(K, 100) env-steps: forward all, then backward all (the order autograd imposes in algorithms/gd.py:239-259), one launch per episode each way
(tsim_rollout / tsim_backward_episode — the open-loop episode of EpisodicSimFunction, envs/redmax_torch_functions.py:46-57,77-92, with the
synthetic actions resident in HBM); `--launch step` times one launch per env-step (StepSimFunction granularity).  Every frame's q / variables /
tactile outputs are written in both modes.  Inputs are resident in HBM before the timed region.  Environments shard across ranks with no
data-path exchange (weak scaling); the only collective is the GD outer loop's policy-gradient all-reduce (118 296 B), once per
episode.

Output: ONE compact JSON line on stdout (rank 0; < 6 KB, tests/test_bench_line.py) — the contract's fields, `roofline` (the dominant kernel
charged ITS OWN algorithmic bytes, timed by HIP events the library records on the launching stream around that kernel alone; per-kernel table
for k_forward / k_taxels / k_backward; `traffic` from in-run rocprofv3 --pmc passes), `cpu_baseline`, and a one-line summary of each optional
leg.  Everything else (counter dumps, per-window lists, launch shapes, full sub-records) goes to bench_detail.json next to this file and to
stderr.  The headline is measured FIRST; the optional legs (bench_legs.py, chosen with --legs) run after it inside a wall-clock budget, each in a
child process of its own (the contract's other fields first: the PMC passes, the CPU baseline), and a watchdog prints the line with whatever is finished
if one overruns — a leg that crashes or hangs costs that leg only.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_legs as BL      # noqa: E402  (optional legs + shared helpers; imports nothing of the simulator at module level)
from bench_legs import WORKLOADS, make_workload, sub_record, readout_leg, Leg, kernel_record      # noqa: E402,F401  (tools/*.py use them through this module)

ALL_LEGS = ["step_mode", "cpu", "pmc", "env_tables", "f64", "push_fwd", "dclaw", "insertion", "closed_loop", "readout"]      # (the contract's fields first: roofline.traffic, cpu_baseline)
LINE_LIMIT = 6144            # bytes of the final stdout line (the driver keeps an 8.5 KB tail)

_T0 = time.perf_counter()


def progress(msg):
    """leg timings on stderr (the JSON line on stdout stays alone)"""
    print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def fatal(msg, rc=2):
    print("bench.py: " + msg, file=sys.stderr, flush=True)
    sys.exit(rc)


# ---------------------------------------------------------------------------------------------------- the compact line
def _r(x, sig=5):
    """numbers rounded to `sig` significant digits (the line is for reading and parsing, bench_detail.json keeps full precision)"""
    if isinstance(x, str):
        return x if len(x) <= 230 else x[:227] + "..."
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return _r(float(x), sig)


def _short_inst(s):
    """k_forward<float, NRM=8, EXPJ=false, LPE=16, POLICY=false, TsStaticPusher> -> k_forward<float,8,false,16,false,TsStaticPusher>"""
    if not s:
        return s
    for k in ("NRM=", "EXPJ=", "LPE=", "POLICY="):
        s = s.replace(k, "")
    return s.replace(", ", ",")


def _sub_summary(rec):
    """<= ~300 B per optional leg: value, its dominant kernel's roofline fraction (own bytes) and VALU fraction, instantiation, non-converged"""
    if not isinstance(rec, dict):
        return None
    if "error" in rec:
        return {"error": str(rec["error"])[:120]}
    if "skipped" in rec:
        return {"skipped": rec["skipped"]}
    out = {"value": rec.get("value")}
    rl = rec.get("roofline") or {}
    if rl:
        out["kernel"] = _short_inst(rl.get("instantiation") or rl.get("kernel"))
        out["kernel_ms"] = rl.get("kernel_ms")
        out["frac"] = rl.get("frac")
        if rl.get("traffic_over_algorithmic") is not None:
            out["traffic_x"] = rl["traffic_over_algorithmic"]
        if rl.get("valu"):
            out["valu_frac"] = rl["valu"].get("frac")
            out["lanes"] = rl["valu"].get("active_lane_frac")
    for k_in, k_out in (("nonconverged_envs", "nonconv_envs"), ("batch", "B"), ("dtype", "dtype"), ("idle_share", "idle_share"), ("achieved", "GBps"),
                        ("s_per_epoch", "s_per_epoch"), ("value_budgeted", "value_budgeted"), ("flagged_frac_budgeted", "flagged_frac_budgeted"),
                        ("value_whole_config", "value_whole_config")):
        if rec.get(k_in) is not None:
            out[k_out] = rec[k_in]
    return out


def compact_line(res):
    """The ONE stdout line, from the full result dict (which goes to bench_detail.json).  Pure function: tests/test_bench_line.py feeds it a
    canned result and asserts strict JSON and len < LINE_LIMIT."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "host_gap_ms")
    line = {k: res.get(k) for k in keep}
    cfg = res.get("config") or {}
    line["config"] = {"workload": cfg.get("workload"), "global_batch": cfg.get("global_batch"), "parallelism": cfg.get("parallelism")}
    k = res.get("kernel") or {}
    line["kernel"] = {"variant": k.get("variant"), "lanes_per_env": k.get("lanes_per_env")}
    for kn in ("k_forward", "k_backward"):
        if kn in k:
            line["kernel"][kn] = {"inst": _short_inst(k[kn].get("instantiation")), "vgpr": k[kn].get("vgpr_count"), "agpr": k[kn].get("agpr_count"),
                                  "sgpr_spill": k[kn].get("sgpr_spill_count"), "vgpr_spill": k[kn].get("vgpr_spill_count"), "code_bytes": k[kn].get("code_bytes")}
    rp = res.get("repeats") or {}
    line["repeats"] = {"windows": rp.get("windows"), "value_is": "median window", "values": rp.get("values")}
    rl = res.get("roofline") or {}
    rlo = {kk: rl.get(kk) for kk in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "traffic_over_algorithmic",
                                      "env_steps_per_launch", "kernel_ms", "timed_by", "traffic_source")}
    rlo["instantiation"] = _short_inst(rl.get("instantiation"))
    if rl.get("per_kernel"):
        rlo["per_kernel"] = {kn: {kk: v.get(kk) for kk in ("ms", "launches_per_window", "algorithmic_bytes", "traffic", "traffic_over_algorithmic", "achieved_gbs", "frac")}
                             for kn, v in rl["per_kernel"].items()}
    if rl.get("forward_side"):
        rlo["forward_side"] = rl["forward_side"]
    v = rl.get("valu")
    if v:
        rlo["valu"] = {kk: v.get(kk) for kk in ("frac", "achieved_tflops", "peak_tflops", "active_lane_frac", "wave_valu_frac", "wave_waiting_frac", "wavefronts_per_simd",
                                                  "valu_wave_insts_per_env_step")}
    rlo["bound_note"] = "HBM fraction as the contract asks; the kernel is bound by dependent-instruction latency of one wavefront per SIMD: see valu"
    line["roofline"] = rlo
    cb = res.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {kk: cb.get(kk) for kk in ("value", "unit", "cores", "kind", "sample", "single_thread_value")} if "error" not in cb else cb
    nt = res.get("nonconverged_timed")
    if nt:
        line["nonconverged_timed"] = nt
    if res.get("launch"):
        line["launch"] = {"mode": res["launch"].get("mode"), "other_mode": res["launch"].get("other_mode"), "other_mode_value": res["launch"].get("other_mode_value")}
    if res.get("per_rank"):
        line["per_rank"] = [{kk: p.get(kk) for kk in ("rank", "kernel_ms_total", "allreduce_ms_total", "host_gap_ms", "wait_for_slowest_rank_ms")} for p in res["per_rank"]][:8]
    if res.get("ranks"):
        line["ranks"] = res["ranks"]
    subs = {}
    for kk in ("env_tables", "f32_bare_xml_loop", "f64", "f64_library_default", "push_fwd", "dclaw", "insertion", "closed_loop", "closed_loop_per_step_graph", "closed_loop_cnn_per_step_graph", "readout"):
        if kk in res:
            subs[kk] = _sub_summary(res[kk])
    if subs:
        line["sub"] = subs
    lg = res.get("legs") or {}
    line["legs"] = {kk: lg[kk] for kk in ("skipped", "watchdog") if lg.get(kk)} or None
    if any("(error)" in d for d in lg.get("done", [])):
        line["legs"] = dict(line["legs"] or {}, errors=[d for d in lg["done"] if "(error)" in d])
    line["detail"] = res.get("detail_file")
    line = _r(line)
    s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:      # never exceed the limit: drop the optional parts, largest first, until it fits
        for drop in ("sub", "per_rank", "launch", "repeats", "kernel"):
            if drop in line:
                line[drop] = {"dropped": "line would exceed %d bytes; see %s" % (LINE_LIMIT, line.get("detail"))}
                s = json.dumps(line, allow_nan=False, separators=(",", ":"))
                if len(s) < LINE_LIMIT:
                    break
    return s


class Emitter:
    """Exactly one line on stdout, whoever gets there first: the normal end of main() or the watchdog."""

    def __init__(self):
        self.lock = threading.Lock()
        self.done = False
        self.res = None
        self.detail_path = os.path.join(ROOT, "bench_detail.json")

    def emit(self, why=None):
        import copy
        with self.lock:
            if self.done or self.res is None:
                return
            self.done = True
            # The watchdog calls this from its own thread while the main thread may be inside a leg that is adding records to the same dict: work on a
            # snapshot (a few attempts: a copy can meet a dict that changes size under it), and whatever happens, print A line
            res = None
            for _ in range(5):
                try:
                    res = copy.deepcopy(self.res)
                    break
                except RuntimeError:
                    time.sleep(0.05)
            if res is None:
                res = {k: self.res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                                                    "data", "config", "roofline", "cpu_baseline", "repeats", "kernel")}
            if why:
                res.setdefault("legs", {})["watchdog"] = why
            res["detail_file"] = os.path.basename(self.detail_path)
            try:
                with open(self.detail_path, "w") as fh:
                    json.dump(res, fh, indent=1, default=lambda o: float(o) if hasattr(o, "__float__") else str(o))
            except Exception as e:      # a read-only checkout must not cost the line
                res["detail_file"] = "not written: %r" % (e,)
            try:
                line = compact_line(res)
            except Exception as e:      # ... nor may a malformed optional record
                line = compact_line({k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                                               "dtype", "data", "config", "roofline", "cpu_baseline", "repeats", "kernel")} | {"legs": {"watchdog": "line rebuilt without the optional records: %r" % (e,)}})
            print(line, flush=True)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves — one process per GPU under
    torch.distributed.run, as the driver does (the reference's only parallel launch is SubprocVecEnv's process-per-worker,
    externals/pytorch-a2c-ppo-acktr-gail/a2c_ppo_acktr/envs.py:100-108)."""
    import socket
    if args.backend == "nccl" and os.environ.get("TSIM_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() < args.gpus:
        fatal("--gpus %d but only %d GPU(s) visible (one process per GPU; TSIM_BENCH_SHARE_GPU=1 with --backend gloo is the 1-GPU plumbing test)"
              % (args.gpus, torch.cuda.device_count()))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def plumbing_only(args, world, rank):
    """No simulator, no GPU needed: the process group of `--gpus N`, one barrier, one all-reduce of the policy-gradient payload."""
    import torch.distributed as dist
    ranks = 1
    if world > 1:
        t = torch.ones(BL.POLICY_GRAD_FLOATS, dtype=torch.float32)
        dist.all_reduce(t)
        ranks = int(t[0].item())
        dist.barrier()
    if rank == 0:
        print(json.dumps({"plumbing_only": True, "n_gpus": world, "ranks_in_allreduce": ranks, "backend": args.backend if world > 1 else None,
                          "allreduce_bytes": 4 * BL.POLICY_GRAD_FLOATS, "value": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


IN_PROCESS_LEGS = ("step_mode", "pmc")      # step_mode uses the headline's own batch; pmc fills the headline's roofline (and is child processes itself)


def leg_child(args):
    """`bench.py --leg-child NAME` (started by the parent's run_leg_in_child): ONE optional leg in a process of its own; prints the records it produced."""
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd import workloads as W
    name = args.leg_child
    asset_, B_def, T_def, fwd_only_def, _ = BL.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    forward_only = args.forward_only or fwd_only_def
    S = 1 if args.workload == "insertion" else args.frame_skip
    ctx = {"args": args, "B": args.batch or B_def, "T": args.episode or T_def, "S": S, "fps": BL.FRAMES_PER_ENV_STEP[args.workload], "dev": dev, "tdt": tdt,
           "esz": 4 if args.dtype == "f32" else 8, "forward_only": forward_only, "model": load_model(W.asset(asset_)), "leg": None, "progress": progress}
    if os.environ.get("TSIM_BENCH_CRASH_LEG") == name:      # tests/test_gpu_bench_line.py: a leg that dies the hard way (as a segmentation fault would)
        os.abort()
    out = {}
    BL.run_leg(name, out, ctx)
    print(json.dumps(out, default=lambda o: float(o) if hasattr(o, "__float__") else str(o)), flush=True)


def run_leg_in_child(name, res, args, timeout_s):
    """One optional leg in a child process (same command line + --leg-child NAME): its records are merged into `res`; a crash, a hang or garbage on its
    stdout becomes an {"error": ...} record of that leg and nothing else."""
    cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + ["--leg-child", name]
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        res[name if name != "cpu" else "cpu_baseline"] = {"error": "leg ran into its %d s limit (child process killed)" % timeout_s}
        return False
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        tail = (r.stderr or "").strip().splitlines()[-3:]
        res[name if name != "cpu" else "cpu_baseline"] = {"error": "child process exited with %d: %s" % (r.returncode, " | ".join(tail)[-300:])}
        return False
    res.update(json.loads(lines[-1]))
    return True


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="push", choices=["push", "dclaw", "insertion"], help="push = the headline (BASELINE configs[2]); dclaw / insertion = "
                    "configs[3] / [4] at their per-GPU share, forward-only (reported as sub-records of the default N = 1 line too)")
    ap.add_argument("--batch", type=int, default=None, help="environments per GPU (default: the workload's per-GPU share)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--frame-skip", type=int, default=5)
    ap.add_argument("--episode", type=int, default=None, help="env-steps per episode (tape length / frame_skip; default: the workload's)")
    ap.add_argument("--repeats", type=int, default=5, help="timed windows of --steps steps each; `value` is the median window (all are listed in `repeats`)")
    ap.add_argument("--graph", action="store_true", help="replay each episode of the timed windows from ONE HIP graph (host/graphed.GraphedEpisode) after two eager windows "
                    "that carry the kernels' HIP events (measured equal to eager at the headline's shape: profiles/r05_graphed_episode.md)")
    ap.add_argument("--legs", default="all", help="comma-separated optional legs after the headline, of: " + ",".join(ALL_LEGS) + "; 'all' (default), 'none'")
    ap.add_argument("--budget-s", type=float, default=240.0, help="optional legs are not STARTED after this many seconds of wall clock; the watchdog prints the line "
                    "with what is finished 90 s later")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="= drop 'cpu' from --legs")
    ap.add_argument("--no-pmc", action="store_true", help="= drop 'pmc' from --legs: no in-run rocprofv3 --pmc passes (roofline.traffic / valu stay null)")
    ap.add_argument("--no-closed-loop", action="store_true", help="= drop 'closed_loop' from --legs")
    ap.add_argument("--no-sub-records", action="store_true", help="= drop the env_tables / f64 / push_fwd / dclaw / insertion legs")
    ap.add_argument("--pmc-dump", default=None, help="write the counters of the in-run --pmc passes to this JSON file (profiles/)")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE.json configs[1] style run (not the headline)")
    ap.add_argument("--launch", default="episode", choices=["episode", "step"],
                    help="episode: tsim_rollout + tsim_backward_episode, one launch each way per episode; step: one tsim_step / tsim_backward_steps launch "
                         "per env-step (StepSimFunction granularity, what a closed-loop torch policy needs)")
    ap.add_argument("--readout-only", action="store_true", help="only the RollingBall read-out leg at --batch environments (the --pmc passes of the readout leg run this)")
    ap.add_argument("--timed-only", action="store_true", help="skip every leg after the timed region: profiler runs (and the in-run --pmc passes)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, the real multi-GPU path) or gloo (plumbing test: with TSIM_BENCH_SHARE_GPU=1 all ranks share cuda:0)")
    ap.add_argument("--leg-child", default=None, help=argparse.SUPPRESS)      # internal: run ONE optional leg in this (child) process and print its records as JSON
    ap.add_argument("--legs-in-process", action="store_true", help="run the optional legs in this process (default: each in a child process, so that a leg that "
                    "crashes or hangs — a segmentation fault inside a graph capture, say — cannot take the headline with it)")
    ap.add_argument("--plumbing-only", action="store_true", help="set up the ranks of --gpus N, run one all-reduce of the policy-gradient payload, print n_gpus and exit")
    args = ap.parse_args(argv)
    legs = [] if args.legs == "none" else (list(ALL_LEGS) if args.legs == "all" else [l for l in args.legs.split(",") if l])
    for l in legs:
        if l not in ALL_LEGS:
            ap.error("unknown leg %r (of %s)" % (l, ",".join(ALL_LEGS)))
    drop = set()
    if args.no_cpu_baseline:
        drop.add("cpu")
    if args.no_pmc:
        drop.add("pmc")
    if args.no_closed_loop:
        drop.add("closed_loop")
    if args.no_sub_records:
        drop.update(("env_tables", "f64", "push_fwd", "dclaw", "insertion"))
    args.leg_list = [l for l in legs if l not in drop]
    return args


def main():
    args = parse_args()
    if args.gpus < 1:
        fatal("--gpus must be >= 1")
    if args.readout_only:
        dev = torch.device("cuda", 0)
        print(json.dumps(BL.readout_leg(torch.float32 if args.dtype == "f32" else torch.float64, dev, B=args.batch or 256)), flush=True)
        return
    if args.forward_only and args.workload != "push":
        fatal("--forward-only is a TactilePush option; dclaw / insertion are forward-only already")
    if args.leg_child:
        return leg_child(args)

    # ---- ranks: under a launcher WORLD_SIZE is set; a bare `python bench.py --gpus N` starts its own N ranks
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        fatal("--gpus %d but WORLD_SIZE %d: the launcher's rank count and --gpus disagree" % (args.gpus, world))
    share = os.environ.get("TSIM_BENCH_SHARE_GPU") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what this host driver supports (task statement)
        if share:
            local_rank = 0
        if args.backend == "nccl":
            if share:
                fatal("TSIM_BENCH_SHARE_GPU=1 needs --backend gloo (RCCL wants one device per rank)")
            if torch.cuda.device_count() < world:
                fatal("--gpus %d but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            fatal("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    if args.plumbing_only:
        return plumbing_only(args, world, rank)
    if not torch.cuda.is_available():
        fatal("no GPU visible: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        t = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t)                                            # the first collective: every rank must be in it
        ranks_seen = int(t.item())
        if ranks_seen != args.gpus:
            fatal("all-reduce saw %d ranks, --gpus %d" % (ranks_seen, args.gpus))

    asset_, B_def, T_def, fwd_only_def, cfg_text = BL.WORKLOADS[args.workload]
    B = args.batch or B_def
    S, T = args.frame_skip, (args.episode or T_def)
    forward_only = args.forward_only or fwd_only_def
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    wl = BL.make_workload(args.workload, B, T, S, rank, dev, tdt)
    S, fps = wl["S"], wl["fps"]               # TactileInsertion: frames of one sub-step, 5 frames per env-step
    model = wl["model"]
    leg = BL.Leg(wl, dev, tdt, forward_only, world, args.backend)
    sim = leg.sim
    nr, nu, nvar, ntac = leg.nr, leg.nu, leg.nvar, leg.ntac
    run_steps = leg.run

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    # clocks / caches: one untimed episode before the W warm-up steps of the contract (W = 10 is ~7 ms of GPU work, which
    # is shorter than the clock ramp: the same binary measured 4.9 M with only that and 5.9 M after a full episode)
    if not args.timed_only:
        run_steps(T, False, args.launch)
    bad_warm = run_steps(args.warmup * fps, False, args.launch) if args.warmup > 0 else 0
    sim.kernel_times()                                # forget the warm-up's kernel events: the per-kernel times below are those of the timed windows

    # The timed region of the contract — exactly K steps between barrier + synchronize on both sides, max over ranks — REPEATED (--repeats, default
    # 5): at K = 20 the region is one forward and one backward launch, 4 ms, and run-to-run spread is +-5 %.  `value` is the MEDIAN window's;
    # every window's value is listed next to it (`repeats`).  The HIP-event kernel times are those of all windows.
    def timed_window(graphed):
        sync_all()
        t0 = time.perf_counter()
        run_steps(args.steps * fps, True, args.launch, graphed=graphed)
        torch.cuda.synchronize()
        dt_own_ = time.perf_counter() - t0          # this rank's own work, before it waits for the others
        sync_all()
        dt_ = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt_], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_, dt_own_
    graph_note = None
    n_ep = min(T, args.steps * fps)
    if args.launch == "episode" and args.graph:
        try:
            leg.capture(n_ep)
            graph_note = {"used": True, "frames_per_replay": n_ep, "what": "reset + tsim_rollout + tsim_backward_episode + reduction of dL/du, one HIP graph per episode"}
        except Exception as e:          # the eager path is always there
            leg.graph = None
            graph_note = {"used": False, "error": repr(e)}
    eager_windows = [timed_window(False) for _ in range(2 if leg.graph is not None else max(1, args.repeats))]
    ktimes = sim.kernel_times()                       # per-kernel HIP events of the eager windows (library-side, on the launching stream)
    windows = [timed_window(True) for _ in range(max(1, args.repeats))] if leg.graph is not None else eager_windows
    order_ = sorted(range(len(windows)), key=lambda i: windows[i][0])
    dt, dt_own = windows[order_[len(order_) // 2]]
    n_win = len(eager_windows)                       # the windows the HIP events of the individual kernels cover
    per_rank = None
    if world > 1:
        import torch.distributed as dist
        # What a SCALE record needs to decompose its efficiency without another round: per rank, the kernel time of its launches (HIP
        # events), the all-reduce as its launching stream saw it (includes waiting for the slowest rank), and what is left of its own wall
        # clock — host launch gaps and idle time
        ar = [a.elapsed_time(b) * 1e3 for a, b in leg.ar_ev]
        k_ms = sum(ms for ms, _ in ktimes.values()) / n_win                                                                           # per window
        ar = ar[len(ar) - len(ar) // n_win:] if n_win > 1 and len(ar) >= n_win else ar                                                # the last window's all-reduces
        mine = {"rank": rank, "device": torch.cuda.get_device_name(dev),
                "k_forward_ms_per_launch": ktimes["k_forward"][0] / max(ktimes["k_forward"][1], 1), "k_backward_ms_per_launch": ktimes["k_backward"][0] / max(ktimes["k_backward"][1], 1),
                "launches": ktimes["k_forward"][1] // n_win, "allreduce_us_mean": float(np.mean(ar)) if ar else None, "allreduce_us_max": float(np.max(ar)) if ar else None,
                "own_wall_ms": dt_own * 1e3, "kernel_ms_total": k_ms, "allreduce_ms_total": sum(ar) / 1e3,
                "host_gap_ms": dt_own * 1e3 - k_ms - sum(ar) / 1e3, "wait_for_slowest_rank_ms": (dt - dt_own) * 1e3}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if args.timed_only:
        if rank == 0:
            print(json.dumps({"timed_only": True, "value": B * world * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                              "values": [B * world * args.steps / w[0] for w in windows], "kernel_variant": sim.kernel_variant(),
                              "kernel_ms": {k: ms / max(n, 1) for k, (ms, n) in ktimes.items()}, "per_rank": per_rank}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return

    em = Emitter()
    if rank == 0:
        frames_per_launch = n_ep if args.launch == "episode" else 1
        rl = BL.per_kernel_roofline(leg, esz, ktimes, n_win, frames_per_launch)
        value = B * world * args.steps / dt
        what = "fwd only" if forward_only else "fwd+bwd"
        label = {"push": "TactilePush", "dclaw": "DClaw rotate", "insertion": "TactileInsertion"}[args.workload]
        bad_sub_timed, bad_env_timed = leg.timed_nonconverged()
        res = {
            "metric": "env-steps/sec (%s) %s batch=%d" % (what, label, B),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("TactilePush (pusher.xml, 13x10 taxels, ndof_r 7) gd_tactile %s, frame_skip %d, batch %d envs/GPU; a timed window = %d env-steps as episodes of "
                                    "%d (one forward + one adjoint launch each)" % ("fwd+adjoint" if not forward_only else "forward only", S, B, args.steps, n_ep // fps))
                                   if args.workload == "push" else
                                   "%s; %s.xml, ndof_r %d, %d tactile values, frame_skip %d, batch %d envs/GPU, episodes of %d frames" % (cfg_text, asset_, nr, ntac, S, B, n_ep),
                       "global_batch": B * world, "parallelism": "env-sharded x%d, policy-grad all-reduce %d B/episode" % (world, 4 * BL.POLICY_GRAD_FLOATS)},
            "solver": leg.solver,
            "repeats": {"windows": len(windows), "steps_per_window": args.steps, "value_is": "median window" + (" (episodes replayed from one HIP graph each)" if leg.graph is not None else ""),
                        "graph": graph_note, "eager_values": [B * world * args.steps / w[0] for w in eager_windows],
                        "values": [B * world * args.steps / w[0] for w in windows], "timed_region_s_each": [w[0] for w in windows]},
            "kernel": BL.kernel_record(sim, args.dtype, forward_only),
            "ranks": {"world_size": world, "ranks_in_first_allreduce": ranks_seen, "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if world > 1 else None,
                      "shared_gpu": share} if world > 1 else None,
            "roofline": rl,
            # what a window spends outside its simulation kernels (host launch gaps, the reset / reduction launches, the all-reduce): wall clock of the
            # median window minus the HIP-event kernel time of an average window — SURVEY §8e names this as the expected limiter of the multi-GPU curve
            "host_gap_ms": dt * 1e3 - sum(ms for ms, _ in ktimes.values()) / max(n_win, 1),
            "nonconverged_warmup": bad_warm,
            "nonconverged_timed": {"substeps": bad_sub_timed, "envs": bad_env_timed, "of_substeps": B * args.steps * fps * S * len(eager_windows)},
            "per_rank": per_rank,
            "launch_shape": sim.launch_info(),      # LDS bytes / block, blocks, lanes per environment
            "legs": {"asked": args.leg_list if world == 1 else [], "done": [], "skipped": []},
        }
        res["roofline"]["instantiation"] = res["kernel"].get(rl["kernel"], res["kernel"].get("k_forward", {})).get("instantiation")
        em.res = res
        progress("headline: %.3f M env-steps/s (median of %d windows), k_forward %.3f ms / k_taxels %.3f ms / k_backward %.3f ms per launch"
                 % (value / 1e6, len(windows), *(ktimes[k][0] / max(ktimes[k][1], 1) for k in ("k_forward", "k_taxels", "k_backward"))))

    if rank == 0 and world == 1:
        # ---- optional legs: after the headline, inside the budget; the watchdog prints the line if one of them hangs
        wd = threading.Timer(max(30.0, args.budget_s + 90.0 - (time.perf_counter() - _T0)), lambda: (em.emit("deadline reached inside an optional leg"), os._exit(0)))
        wd.daemon = True
        wd.start()
        ctx = {"args": args, "B": B, "T": T, "S": S, "fps": fps, "dev": dev, "tdt": tdt, "esz": esz, "forward_only": forward_only, "model": model, "leg": leg,
               "n_ep": n_ep, "frames_per_launch": frames_per_launch, "progress": progress}
        for name in args.leg_list:
            if time.perf_counter() - _T0 > args.budget_s:
                res["legs"]["skipped"].append(name)
                continue
            if name != "step_mode" and ctx.get("leg") is not None:      # free the headline's batch (tape: 0.6 GB) before the other legs
                ctx["leg"] = None
                leg = sim = run_steps = None
                torch.cuda.empty_cache()
            try:
                if name in IN_PROCESS_LEGS or args.legs_in_process:
                    BL.run_leg(name, res, ctx)
                    ok = True
                else:                       # a process of its own: a crash or a hang there costs that leg only
                    ok = run_leg_in_child(name, res, args, timeout_s=max(30.0, min(180.0, args.budget_s + 60.0 - (time.perf_counter() - _T0))))
                res["legs"]["done"].append(name if ok else name + " (error)")
            except Exception as e:      # the headline must not die with an optional leg
                res[name] = {"error": repr(e)}
                res["legs"]["done"].append(name + " (error)")
            progress("leg %s done" % name)
        wd.cancel()
        em.emit()
    elif rank == 0:
        em.emit()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
