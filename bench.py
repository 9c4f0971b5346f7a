#!/usr/bin/env python
"""bench.py — env-steps/s (forward + adjoint) of the batched TactilePush step on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: launched under torch.distributed.run)
A "step" = one env-step (5 implicit BDF1 sub-steps, tactile read-out) forward AND its adjoint for one batch of
B = 4096 TactilePush environments per GPU (BASELINE.json configs[2]: gd_tactile fwd+adjoint, batch 4096).  The K steps
are run as episodes of <= 100 env-steps (forward all, then backward all — the order autograd imposes in
algorithms/gd.py:239-259).  Default launch granularity: one launch per episode each way (tsim_rollout /
tsim_backward_episode — the open-loop episode of EpisodicSimFunction, envs/redmax_torch_functions.py:46-57,77-92, with the
synthetic actions resident in HBM); `--launch step` times one launch per env-step (StepSimFunction granularity) and
its rate is reported in the same JSON line either way (`launch.other_mode_value`).  Every frame's q / variables /
tactile outputs are written in both modes.  Inputs are resident in HBM before the timed region.  Environments shard across ranks with
no data-path exchange (weak scaling); the only collective is the GD outer loop's policy-gradient all-reduce
(29 574 fp32 = 118 296 B), issued once per episode.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline`
(the fp64 CPU oracle — this build's restatement, NOT DiffRedMax — on a bounded sample, 1 thread and all host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

POLICY_GRAD_FLOATS = 29574      # DiagGaussianActor(393 -> 64 -> 64 -> 3), SURVEY.md §2.2
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3


def algorithmic_bytes(nr, nu, nvar, ntac, S, esz):
    """per env-step, state on chip across the S sub-steps, model constants batch-shared.  The taped state is
    (q as double, qd) per sub-step: the pose chain is double also in the fp32 kernels."""
    tape = S * nr * (8 + esz)
    fwd = esz * (nu + nr + nvar + ntac) + tape
    bwd = tape + esz * (nr + nvar + ntac + nu * S)
    return fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--frame-skip", type=int, default=5)
    ap.add_argument("--episode", type=int, default=100, help="env-steps per episode (tape length / frame_skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE.json configs[1] style run (not the headline)")
    ap.add_argument("--launch", default="episode", choices=["episode", "step"],
                    help="episode: tsim_rollout + tsim_backward_episode, one launch each way per episode (the open-loop "
                         "episode of EpisodicSimFunction); step: one tsim_step / tsim_backward_steps launch per env-step "
                         "(StepSimFunction granularity, what a closed-loop policy needs)")
    ap.add_argument("--timed-only", action="store_true",
                    help="skip the legs after the timed region (other launch mode, evaluation statistics): profiler runs")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, the real multi-GPU path) or gloo (plumbing test: with TSIM_BENCH_SHARE_GPU=1 "
                         "all ranks share cuda:0 and the collectives go through host copies)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("TSIM_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.host.batch import BatchSim
    from tests.workloads import push_workload

    model = load_model(os.path.join(ROOT, "tests", "golden", "models", "pusher.npz"))
    B, S, T = args.batch, args.frame_skip, args.episode
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    sim = BatchSim(model, B, device=str(dev), dtype=tdt, tape_capacity=T * S)
    nr, nu, nvar, ntac = sim.ndof_r, sim.ndof_u, sim.ndof_var, sim.ndof_tactile

    # synthetic inputs, resident in HBM (seed differs per rank so that ranks do different work)
    q0_np, u_np, _ = push_workload(B, T, seed=rank)
    q0 = torch.tensor(q0_np, device=dev, dtype=tdt)
    u = torch.tensor(u_np, device=dev, dtype=tdt).transpose(0, 1).contiguous()      # [T, B, 6]
    wq = torch.ones(B, nr, device=dev, dtype=tdt)
    wv = torch.ones(B, nvar, device=dev, dtype=tdt)
    wt = torch.ones(B, ntac, device=dev, dtype=tdt) * 100.0
    wqT, wvT, wtT = (w.unsqueeze(0).expand(T, -1, -1).contiguous() for w in (wq, wv, wt))    # per-frame seeds [T, B, dim]
    grad_buf = torch.zeros(POLICY_GRAD_FLOATS, device=dev, dtype=torch.float32)
    out = {}
    ev = {"fwd": [], "bwd": []}

    def run_steps(k_total, timed, launch):
        """k_total env-steps as episodes of <= T: forward all, backward all."""
        done = 0
        bad = 0
        Ev = lambda: torch.cuda.Event(enable_timing=True)
        while done < k_total:
            n = min(T, k_total - done)
            sim.reset(q0, None, backward_flag=not args.forward_only)
            if launch == "episode":
                e0, e1, e2 = Ev(), Ev(), Ev()
                e0.record()
                ro = sim.rollout(u[:n], S)
                e1.record()
                status = ro["status"]
                if not args.forward_only:
                    du = sim.backward_episode(n, S, wqT[:n], wvT[:n], wtT[:n])
                    e2.record()
                    grad_buf[:6] = du.sum((0, 1)).float()[:6] if nu >= 6 else 0.0
                if timed:
                    ev["fwd"].append((e0, e1, n))
                    if not args.forward_only:
                        ev["bwd"].append((e1, e2, n))
            else:
                for t in range(n):
                    if timed:
                        e0, e1 = Ev(), Ev()
                        e0.record()
                    sim.step(u[t], S, out=out)
                    if timed:
                        e1.record()
                        ev["fwd"].append((e0, e1, 1))
                status = out["status"]
                if not args.forward_only:
                    for t in reversed(range(n)):
                        if timed:
                            e0, e1 = Ev(), Ev()
                            e0.record()
                        du = sim.backward_steps(S, wq, wv, wt)
                        if timed:
                            e1.record()
                            ev["bwd"].append((e0, e1, 1))
                    grad_buf[:6] = du[0].sum(0).float()[:6] if nu >= 6 else 0.0
            bad += int((status != 0).sum().item()) if not timed else 0
            if world > 1:
                import torch.distributed as dist
                if args.backend == "nccl":
                    dist.all_reduce(grad_buf)       # GD outer loop: policy-gradient all-reduce over xGMI (RCCL)
                else:
                    g = grad_buf.cpu(); dist.all_reduce(g); grad_buf.copy_(g)
            done += n
        return bad

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
    bad_warm = run_steps(args.warmup, False, args.launch) if args.warmup > 0 else 0
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps, True, args.launch)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # kernel durations of the timed region (HIP events on the launching stream), per launch and per env-step
    def ev_stats(lst):
        if not lst:
            return 0.0, 0.0, 0
        ms = [a.elapsed_time(b) for a, b, _ in lst]
        fr = [n for _, _, n in lst]
        return float(np.mean(ms)), float(sum(ms) / sum(fr)), int(round(np.mean(fr)))
    fwd_ms, fwd_ms_step, fwd_frames = ev_stats(ev["fwd"])
    bwd_ms, bwd_ms_step, bwd_frames = ev_stats(ev["bwd"])

    # second leg, reported next to the headline: the other launch granularity on the same workload (short, after the
    # timed region)
    other = "step" if args.launch == "episode" else "episode"
    ev_main, ev = ev, {"fwd": [], "bwd": []}
    k_other = min(args.steps, 40)
    other_value = None
    if not args.timed_only:
        run_steps(min(k_other, 5), False, other)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(k_other, False, other)
        torch.cuda.synchronize()
        other_value = B * k_other / (time.perf_counter() - t1)
    ev = ev_main

    # untimed: Newton work statistics (residual evaluations per env-step) of the same workload
    sim.reset(q0, None, backward_flag=False)
    evs = []
    for t in range(min(T, 30) if not args.timed_only else 0):
        sim.step(u[t], S, out=out)
        evs.append(sim.last_evals())
    evs = np.array(evs) if evs else np.zeros((1, 1))
    if args.timed_only:
        out["status"] = torch.zeros(1, dtype=torch.int32)
    status_bad = int((out["status"] != 0).sum().item())

    if rank == 0:
        fb, bb = algorithmic_bytes(nr, nu, nvar, ntac, S, esz)
        dom, dom_ms, dom_bytes, dom_frames = ("k_forward", fwd_ms, fb, fwd_frames) if fwd_ms >= bwd_ms else ("k_backward", bwd_ms, bb, bwd_frames)
        achieved = dom_bytes * B * dom_frames / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        value = B * world * args.steps / dt
        traffic, traffic_src, issue = None, None, None
        pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_%s.json" % args.dtype)
        if os.path.exists(pmc_file) and B == 4096:       # separate rocprofv3 --pmc run of this same command (tools/gpu_pmc.sh)
            try:
                pj = json.load(open(pmc_file))
                pk = [v for k, v in pj["per_kernel"].items() if dom in k][0]
                # gfx950: FETCH_SIZE counts 1/2 (MI355X_MICROARCH.md §HBM); scaled from the profiled launch (pmc frames) to this one
                traffic = (2.0 * pk["FETCH_SIZE"] + pk["WRITE_SIZE"]) * 1024.0 * dom_frames / pj.get("frames_per_launch", 1)
                traffic_src = ("profiles/" + os.path.basename(pmc_file) + " (2*FETCH_SIZE + WRITE_SIZE KiB per launch of %d env-steps, "
                               "scaled to %d)" % (pj.get("frames_per_launch", 1), dom_frames))
                # what actually bounds the kernel: instruction issue of the one wavefront each SIMD holds (same PMC file)
                issue = {"valu_insts_per_env_step": pk["SQ_INSTS_VALU"] / (4096.0 * pj.get("frames_per_launch", 1)),
                         "wave_issuing_frac": pk["SQ_ACTIVE_INST_ANY"] / pk["SQ_WAVE_CYCLES"],
                         "wave_valu_frac": pk["SQ_ACTIVE_INST_VALU"] / pk["SQ_WAVE_CYCLES"],
                         "wavefronts": pk["SQ_WAVES"], "source": "profiles/" + os.path.basename(pmc_file)}
            except Exception:
                pass
        res = {
            "metric": ("env-steps/sec (fwd+bwd) TactilePush batch=%d" % B) if not args.forward_only else ("env-steps/sec (fwd only) TactilePush batch=%d" % B),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "TactilePush (pusher.xml, 13x10 taxels, ndof_r 7) gd_tactile fwd+adjoint, frame_skip %d, "
                                   "batch %d envs/GPU, episodes of %d env-steps" % (S, B, T),
                       "global_batch": B * world, "parallelism": "env-sharded x%d, policy-grad all-reduce %d B/episode" % (world, 4 * POLICY_GRAD_FLOATS)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes * B * dom_frames, "env_steps_per_launch": dom_frames,
                         "algorithmic_bytes_per_env_step": {"forward": fb, "backward": bb},
                         "kernel_ms": {"k_forward": fwd_ms, "k_backward": bwd_ms},
                         "kernel_ms_per_env_step": {"k_forward": fwd_ms_step, "k_backward": bwd_ms_step},
                         "issue": issue,
                         "note": "state stays in LDS across sub-steps, so this path is latency/VALU-bound, not HBM-bound (SURVEY.md §0.6)"},
            "launch": {"mode": args.launch,
                       "episode": "tsim_rollout + tsim_backward_episode: one launch each way per episode (EpisodicSimFunction's open-loop episode)",
                       "step": "tsim_step + tsim_backward_steps: one launch per env-step each way (StepSimFunction granularity)",
                       "other_mode": other, "other_mode_value": other_value, "other_mode_env_steps": k_other},
            "nonconverged_envs_last_step": status_bad, "nonconverged_warmup": bad_warm,
            "launch_shape": sim.launch_info(),      # LDS bytes / block, blocks, lanes per environment, wavefronts per SIMD
            "residual_evals_per_env_step": {"mean": float(evs.mean()), "p99": float(np.percentile(evs, 99)),
                                            "mean_of_per_step_max": float(evs.max(axis=1).mean()), "max": int(evs.max())},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(model, S, not args.forward_only)
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(model, S, with_backward):
    """fp64 CPU oracle (oracle/tsim_oracle.cpp — the build's own restatement) on a bounded sample of the same workload:
    one thread, and all host cores (environments are independent: one oracle instance per thread, ctypes releases the GIL)."""
    import threading
    from oracle.oracle import OracleSim
    from tests.workloads import push_workload
    nenv, nstep = 8, 100
    q0, u, _ = push_workload(nenv, nstep, seed=0)
    o = OracleSim(model)
    o.bench_rollout(q0[:1], u[:1, :5], S, with_backward)       # warm
    t0 = time.perf_counter()
    n, _ = o.bench_rollout(q0, u, S, with_backward)
    dt1 = time.perf_counter() - t0
    st = o.stats()
    single = n / dt1
    # all cores: 2 environments x 100 env-steps per thread
    nthr = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    per = 2
    q0m, um, _ = push_workload(per * nthr, nstep, seed=1)
    sims = [OracleSim(model) for _ in range(nthr)]
    done = [0] * nthr

    def work(i):
        done[i], _ = sims[i].bench_rollout(q0m[i * per:(i + 1) * per], um[i * per:(i + 1) * per], S, with_backward)
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    c0 = os.times()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dtm = time.perf_counter() - t0
    c1 = os.times()
    busy = ((c1.user - c0.user) + (c1.system - c0.system)) / dtm      # cores actually kept busy (cgroup limits show here)
    what = "fwd+adjoint" if with_backward else "fwd only"
    return {"value": sum(done) / dtm, "unit": "env-steps/s", "cores": nthr, "kind": "port",
            "sample": "%d threads x %d envs x %d env-steps of the same TactilePush workload, %s, fp64, g++ -O3 (one oracle instance "
                      "per thread); single thread: %d envs x %d env-steps; mean Newton iterations/sub-step %.2f"
                      % (nthr, per, nstep, what, nenv, nstep, st["newton_iters"] / max(st["substeps"], 1)),
            "single_thread_value": single, "host_cpus": os.cpu_count(), "cores_busy": busy}


if __name__ == "__main__":
    main()
