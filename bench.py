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
tactile outputs are written in both modes.  Inputs are resident in HBM before the timed region.  Environments shard
across ranks with no data-path exchange (weak scaling); the only collective is the GD outer loop's policy-gradient
all-reduce (29 574 fp32 = 118 296 B), issued once per episode.

Prints ONE JSON line (rank 0).  Besides the contract's fields:
  roofline      dominant kernel, HIP-event timed in this run; HBM fraction from SURVEY.md §8d's algorithmic bytes, plus
                `valu`: what actually bounds the path (instruction issue of one wavefront per SIMD), from hardware counters
                collected IN THIS RUN by short `rocprofv3 --pmc` passes of this same script (separate passes, counters only);
                `traffic` = 2 * FETCH_SIZE + WRITE_SIZE of those passes (gfx950 correction of MI355X_MICROARCH.md §HBM)
  closed_loop   BASELINE config 3 as cfg/gd_tactile.yaml runs it: policy between env-steps, BPTT, all-reduce, clip, Adam — the
                policy evaluated inside the simulator's episode launches (envs/push_closed_loop.FusedPushEpisode), timed here, not by
                a side script; closed_loop_per_step_graph: the same epoch as one launch per env-step replayed from one HIP graph
                (algorithms/batched_gd.GraphedRollout: any torch policy)
  readout_hbm   the one HBM-relevant kernel of this path: 200 x 200-taxel read-out of RollingBall, GB/s
  cpu_baseline  the fp64 CPU oracle (this build's restatement, NOT DiffRedMax), built -O3 -march=native on this host,
                one instance per usable core
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_EVAL_BUDGET = 256           # evaluations per sub-step in the f64 legs
POLICY_GRAD_FLOATS = 29574      # DiagGaussianActor(393 -> 64 -> 64 -> 3), SURVEY.md §2.2
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3   # 256 CU x 4 SIMD x 64 lanes x 2 flop per 2 cycles at 2.4 GHz (packed / two wavefronts per SIMD)
N_SIMD, CLOCK_GHZ = 1024, 2.4
PMC_PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"],
    ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
     "SQ_INSTS_VALU_FMA_F64", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]


def algorithmic_bytes(nr, nu, nvar, ntac, S, esz, tape=True):
    """per env-step, state on chip across the S sub-steps, model constants batch-shared:
    fwd = esz (nu + nr + nvar + ntac + 2 nr S), bwd = esz (2 nr S + nr + nvar + ntac + nu S)   (1 916 / 2 012 B for fp32
    TactilePush).  As built, the tape holds q as double and the Newton matrix: that is implementation traffic
    and shows in `traffic`, not here."""
    fwd = esz * (nu + nr + nvar + ntac + (2 * nr * S if tape else 0))      # forward-only: no tape (1 636 B for fp32 TactilePush)
    bwd = esz * (2 * nr * S + nr + nvar + ntac + nu * S)
    return fwd, bwd


def usable_cores():
    """Cores this process may actually keep busy: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


WORKLOADS = {
    # name: (model asset, environments per GPU, env-steps per episode, forward-only, BASELINE.json config)
    "push": ("pusher", 4096, 100, False, "configs[2]: TactilePush gd_tactile fwd+adjoint, batch 4096 on one MI355X"),
    "push_fwd": ("pusher_13x13", 1024, 20, True, "configs[1]: TactilePush with a 13 x 13 taxel pad (the XML's 13 x 10 pad re-gridded: workloads.synthetic_variant), batch 1024, forward-only on one MI355X"),
    "dclaw": ("dclaw_position_control", 2048, 50, True, "configs[3]: D'Claw rotate, 16 384 environments over 8 GPUs = 2048 per GPU, forward-only (PPO roll-out): q_init + 0.05 N(0, 1), "
                                                        "random policy under relative position control (SURVEY.md §8d config 4; envs/dclaw_rotate_env.py:74-77,163,201-204), 50 of the "
                                                        "episode's 200 env-steps per launch"),
    "insertion": ("tactile_insertion", 4096, 45, True, "configs[4]: TactileInsertion, 32 768 environments over 8 GPUs = 4096 per GPU, one 45-sub-step insertion attempt per "
                                                       "episode from the settled grasp moved by U(+-6 mm, +-6 mm, +-10 deg) (SURVEY.md §8d config 5; envs/tactile_insertion_env.py:"
                                                       "200-216,344-357), six captured tactile frames, forward-only + the 118 296-B policy-gradient all-reduce per episode"),
}
# TactileInsertion's episode is 45 frames of ONE sub-step (envs/tactile_insertion_env.py:53,359: frame_skip 1, a new joint target every
# sub-step); to keep the unit of the metric (one env-step = 5 sub-steps) 5 of its frames count as one env-step
FRAMES_PER_ENV_STEP = {"push": 1, "push_fwd": 1, "dclaw": 1, "insertion": 5}


_T0 = time.perf_counter()


def progress(msg):
    """leg timings on stderr (the JSON line on stdout stays alone)"""
    print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def fatal(msg, rc=2):
    print("bench.py: " + msg, file=sys.stderr, flush=True)
    sys.exit(rc)


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


def make_workload(name, B, T, S, rank, dev, tdt):
    """Synthetic inputs of one BASELINE config, resident in HBM (seed differs per rank so that ranks do different work)."""
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd import workloads as W
    model = W.synthetic_variant(WORKLOADS[name][0]) if name == "push_fwd" else load_model(W.asset(WORKLOADS[name][0]))
    if name in ("push", "push_fwd"):
        q0, u, _ = W.push_workload(B, T, seed=rank)
    elif name == "dclaw":
        q0, u = W.dclaw_random_workload(B, T, seed=7 + rank)
    else:
        q0, u = W.insertion_attempt_workload(B, seed=7 + rank)
        u, S = u[:, :T], 1
    wl = {"name": name, "model": model, "S": S, "T": T, "B": B, "fps": FRAMES_PER_ENV_STEP[name],
          "q0": torch.tensor(q0, device=dev, dtype=tdt), "u": torch.tensor(u, device=dev, dtype=tdt).transpose(0, 1).contiguous()}
    if name == "insertion":
        mask = torch.zeros(T, dtype=torch.bool)
        mask[[f for f in W.INSERTION_TACTILE_FRAMES if f < T]] = True
        wl["tactile_mask"] = mask
    return wl


def kernel_record(sim, dtype, forward_only=False, policy=False):
    """Which instantiation of the simulation kernels the batch's next launches run (include/tsim.h tsim_kernel_variant + the launch shape) and
    what it uses: registers, spills, LDS, code bytes from the built code object's metadata (host/buildhash.py write_kernel_table)."""
    from tactilesimulation_amd.host import buildhash
    try:
        table = json.load(open(buildhash.KERNELS_JSON))
    except OSError:
        table = {}
    info = sim.launch_info()
    variant = sim.kernel_variant()
    I_ = np.asarray(sim.model.I)
    has_exp = any(int(I_[int(I_[14]) + i * 8 + 1]) == 7 for i in range(int(I_[2])))      # a rotation-vector joint (include/tsim_blob.h TSIM_J_SPHERICAL_EXP): the EXPJ kernels
    rec = {"variant": variant, "lanes_per_env": info["lanes_per_env"], "blocks": info["blocks"], "dynamic_lds_bytes": info["lds_bytes"],
           "options": {"pair_cull": sim.get_option(sim.OPT_PAIR_CULL), "value_trials": sim.get_option(sim.OPT_VALUE_TRIALS),
                       "trial_helpers": sim.get_option(sim.OPT_TRIAL_HELPERS), "value_first": sim.get_option(sim.OPT_VALUE_FIRST)}}
    for k in ("k_forward",) + (() if forward_only else ("k_backward",)):
        mangled, readable = buildhash.kernel_name(k, dtype, sim.ndof_r, has_exp, info["lanes_per_env"], variant, policy)
        rec[k] = dict({"instantiation": readable, "symbol": mangled}, **(table.get(mangled) or {"metadata": "not found in %s" % os.path.basename(buildhash.KERNELS_JSON)}))
    return rec


class Leg:
    """One workload on one BatchSim: runs env-steps as episodes of <= T (forward all, then backward all) and keeps the HIP-event times
    of the launches of its timed part."""

    def __init__(self, wl, dev, tdt, forward_only, world=1, backend="nccl", solver="bench"):
        from tactilesimulation_amd.host.batch import BatchSim
        self.wl, self.dev, self.tdt, self.forward_only, self.world, self.backend = wl, dev, tdt, forward_only, world, backend
        B, T, S = wl["B"], wl["T"], wl["S"]
        self.sim = sim = BatchSim(wl["model"], B, device=str(dev), dtype=tdt, tape_capacity=0 if forward_only else T * S)
        # Solver options (include/tsim.h tsim_set_solver_options).  Every leg of this bench runs the XML's Newton loop with kink
        # crossing near convergence — the library's default for fp32 batches; f64 legs are given the same option so that they differ
        # from the headline in arithmetic only (the library's fp64 default is the bare loop: what the parity tests pin).
        # TactileInsertion (round 4: the reference's episode — an attempt from the SETTLED grasp) converges everywhere
        # under the library's default loop: no evaluation budget.  (Rounds 1-3 timed a stand-in that closed the grasp inside the episode;
        # its two finger-meets-box sub-steps are where plain backtracking creeps, and it needed a budget of 128.)
        # f64 legs: 2 of the 4096 TactilePush environments cycle between the two sides of a kink (the loop then runs ~1000 evaluations to
        # max_iter, non-converged either way): bounded as well.  The fp32 headline has no budget (its largest sub-step: 43 evaluations).
        self.eval_budget = F64_EVAL_BUDGET if (tdt == torch.float64 and solver != "library") else 0
        if solver != "library":
            sim.set_solver_options(cross_kinks=True, eval_budget=self.eval_budget)
            self.solver = "XML Newton loop (tol / max_iter / max_ls of the model) + kink crossing near convergence" + (
                "" if not self.eval_budget else ", at most %d evaluations per sub-step (flagged in status beyond)" % self.eval_budget)
        else:
            self.solver = "library default for this dtype: " + ("XML Newton loop + kink crossing near convergence" if tdt == torch.float32
                                                                else "the bare XML Newton loop (what the fp64 parity tests pin), no evaluation budget")
        self.status_log, self.ar_ev = [], []
        self.nr, self.nu, self.nvar, self.ntac = sim.ndof_r, sim.ndof_u, sim.ndof_var, sim.ndof_tactile
        one = lambda d, s=1.0: torch.ones(B, d, device=dev, dtype=tdt) * s
        self.wq, self.wv, self.wt = one(self.nr), (one(self.nvar) if self.nvar else None), one(self.ntac, 100.0)
        if not forward_only:
            self.wqT, self.wvT, self.wtT = ((w.unsqueeze(0).expand(T, -1, -1).contiguous() if w is not None else None) for w in (self.wq, self.wv, self.wt))
        self.grad_buf = torch.zeros(POLICY_GRAD_FLOATS, device=dev, dtype=torch.float32)
        self.out = {}
        self.ev = {"fwd": [], "bwd": [], "episode": []}
        self.graph = None

    def capture(self, n):
        """An episode of n frames — reset, episode launch forward, episode launch backward, reduction of dL/du into the gradient buffer — as ONE HIP
        graph (host/graphed.GraphedEpisode).  Replayed by run(..., graphed=True) for episodes of exactly that length; BDF1 models only."""
        from tactilesimulation_amd.host.graphed import GraphedEpisode
        wl, S = self.wl, self.wl["S"]
        ng = min(6, self.nu)

        def post(ro, du):
            if du is not None:
                self.grad_buf[:ng] = du.sum((0, 1)).float()[:ng]
            return None
        seeds = None if self.forward_only else (self.wqT[:n], self.wvT[:n] if self.wvT is not None else None, self.wtT[:n])
        mask = wl["tactile_mask"][:n] if "tactile_mask" in wl else None
        self.graph = GraphedEpisode(self.sim, wl["q0"], wl["u"][:n], S, seeds=seeds, tactile_mask=mask, post=post)
        self.graph_n = n

    def run(self, k_total, timed, launch, graphed=False):
        sim, wl, T, S, u = self.sim, self.wl, self.wl["T"], self.wl["S"], self.wl["u"]
        done = bad = 0
        Ev = lambda: torch.cuda.Event(enable_timing=True)
        ng = min(6, self.nu)
        while done < k_total:
            n = min(T, k_total - done)
            if graphed and launch == "episode" and self.graph is not None and n == self.graph_n:
                # one replay = one whole episode (reset, forward launch, backward launch, gradient reduction): no events INSIDE a graph, the pair brackets it
                e0, e1 = Ev(), Ev()
                e0.record()
                ro, _, _ = self.graph.replay()
                e1.record()
                status = ro["status"]
                if timed:
                    self.status_log.append(status.clone())
                    self.ev["episode"].append((e0, e1, n))
            elif launch == "episode":
                sim.reset(wl["q0"], None, backward_flag=not self.forward_only)
                e0, e1, e2 = Ev(), Ev(), Ev()
                e0.record()
                ro = sim.rollout(u[:n], S, tactile_mask=wl["tactile_mask"][:n]) if "tactile_mask" in wl else sim.rollout(u[:n], S)
                e1.record()
                status = ro["status"]
                if timed:
                    self.status_log.append(status)
                if not self.forward_only:
                    du = sim.backward_episode(n, S, self.wqT[:n], self.wvT[:n] if self.wvT is not None else None, self.wtT[:n])
                    e2.record()
                    self.grad_buf[:ng] = du.sum((0, 1)).float()[:ng]
                if timed:
                    self.ev["fwd"].append((e0, e1, n))
                    if not self.forward_only:
                        self.ev["bwd"].append((e1, e2, n))
            else:
                sim.reset(wl["q0"], None, backward_flag=not self.forward_only)
                for t in range(n):
                    if timed:
                        e0, e1 = Ev(), Ev()
                        e0.record()
                    sim.step(u[t], S, out=self.out)
                    if timed:
                        e1.record()
                        self.ev["fwd"].append((e0, e1, 1))
                status = self.out["status"]
                if not self.forward_only:
                    for t in reversed(range(n)):
                        if timed:
                            e0, e1 = Ev(), Ev()
                            e0.record()
                        du = sim.backward_steps(S, self.wq, self.wv, self.wt)
                        if timed:
                            e1.record()
                            self.ev["bwd"].append((e0, e1, 1))
                    self.grad_buf[:ng] = du[0].sum(0).float()[:ng]
            bad += int((status != 0).sum().item()) if not timed else 0
            if self.world > 1:
                import torch.distributed as dist
                a0, a1 = Ev(), Ev()
                a0.record()
                if self.backend == "nccl":
                    dist.all_reduce(self.grad_buf)       # GD outer loop: policy-gradient all-reduce over xGMI (RCCL), 118 296 B
                else:
                    g = self.grad_buf.cpu(); dist.all_reduce(g); self.grad_buf.copy_(g)
                a1.record()                              # the launching stream waits for the collective: the pair brackets it
                if timed:
                    self.ar_ev.append((a0, a1))
            done += n
        return bad

    def timed_nonconverged(self):
        """(sub-steps that ended above the Newton tolerance — or were cut by the evaluation budget —, environments with at least one) over
        the launches of the TIMED part; the status tensors are only looked at after the timed region."""
        if not self.status_log:
            return 0, 0
        st = torch.stack(self.status_log) & 0x3FFFFFFF
        return int(st.sum().item()), int((st != 0).any(0).sum().item())

    def ev_stats(self, key):
        lst = self.ev[key]
        if not lst:
            return 0.0, 0.0, 0
        ms = [a.elapsed_time(b) for a, b, _ in lst]
        fr = [n for _, _, n in lst]
        return float(np.mean(ms)), float(sum(ms) / sum(fr)), int(round(np.mean(fr)))

    def roofline(self, esz):
        """HBM side of the roofline for the dominant kernel of this leg: SURVEY.md §8d's algorithmic bytes over the HIP-event time."""
        S, fps = self.wl["S"], self.wl["fps"]
        fb, bb = algorithmic_bytes(self.nr, self.nu, self.nvar, self.ntac, S, esz, tape=not self.forward_only)
        if "tactile_mask" in self.wl:           # per frame: u in, q out; the tactile frame only where the mask says so (6 of 45)
            T = self.wl["T"]
            fb = esz * (self.nu + self.nr + self.nvar) + esz * self.ntac * int(self.wl["tactile_mask"].sum()) / T
        fwd_ms, fwd_ms_step, fwd_frames = self.ev_stats("fwd")
        bwd_ms, bwd_ms_step, bwd_frames = self.ev_stats("bwd")
        dom, dom_ms, dom_bytes, dom_frames = ("k_forward", fwd_ms, fb, fwd_frames) if fwd_ms >= bwd_ms else ("k_backward", bwd_ms, bb, bwd_frames)
        achieved = dom_bytes * self.wl["B"] * dom_frames / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        return {"bound": "hbm",          # the roofline the contract asks the figures against (achieved / peak / frac / traffic are HBM bytes)
                "bound_note": "what actually bounds the kernel is vector-instruction issue of one wavefront per SIMD, not HBM (SURVEY.md §0.6): see `valu`",
                "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                "algorithmic_bytes_per_launch": dom_bytes * self.wl["B"] * dom_frames, "env_steps_per_launch": dom_frames,
                "algorithmic_bytes_per_env_step": {"forward": fb * fps, "backward": bb * fps, "source": "SURVEY.md §8d (general formula)" + (
                    "; %d frames of one sub-step per env-step, tactile frames as masked" % fps if fps > 1 else "")},
                "kernel_ms": {"k_forward": fwd_ms, "k_backward": bwd_ms},
                "kernel_ms_per_env_step": {"k_forward": fwd_ms_step * fps, "k_backward": bwd_ms_step * fps},
                "valu": None}, (dom, dom_ms, dom_frames)


def sub_record(name, dtype, dev, steps=None, warm=None, solver="bench", pmc=False, env_tables=False):
    """A short N = 1 leg of another BASELINE config (or of the headline workload in another dtype), reported inside the headline's JSON
    line: value, kernel times by HIP events, HBM roofline from the general formula of SURVEY.md §8d.  `steps` in env-steps (5 sub-steps)."""
    asset_, B, T, fwd_only, cfg = WORKLOADS[name]
    tdt = torch.float32 if dtype == "f32" else torch.float64
    esz = 4 if dtype == "f32" else 8
    if name == "push" and env_tables:      # the HEADLINE's inputs (the first 20 env-steps of its 100-step table): this record is read against the headline
        wl = make_workload(name, B, T, 5, 0, dev, tdt)
        T = min(T, 20)
        wl["u"], wl["T"] = wl["u"][:T].contiguous(), T
    else:                                  # (the f64 records keep their own 20-step table, as in every round: two of its environments do not converge in fp64)
        T = min(T, 20) if name == "push" else T
        wl = make_workload(name, B, T, 5, 0, dev, tdt)
    fps = wl["fps"]
    leg = Leg(wl, dev, tdt, fwd_only, solver=solver)
    if env_tables:      # one parameter table per environment (domain randomisation, include/tsim.h tsim_set_env_tables): here every row the model's own
        leg.sim.set_env_tables(leg.sim.base_tables())
    steps = steps or 2 * T // fps                       # two episodes
    leg.run(warm * fps if warm else T, False, "episode")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    leg.run(steps * fps, True, "episode")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad_sub, bad_env = leg.timed_nonconverged()
    rl, (dom, dom_ms, dom_frames) = leg.roofline(esz)
    info = leg.sim.launch_info()
    evals = leg.sim.last_evals()
    helped = leg.sim.last_helper_trials()
    krec = kernel_record(leg.sim, dtype, fwd_only)
    leg_solver = leg.solver
    del leg
    torch.cuda.empty_cache()
    S = wl["S"]
    rl["kernel_instantiation"] = krec[dom]["instantiation"]
    if pmc:      # the same counters as the headline's, from `bench.py --workload <name> --timed-only` under rocprofv3 --pmc (separate passes)
        ns = argparse.Namespace(steps=T // fps, warmup=T // fps, dtype=dtype, workload=name, frame_skip=5, launch="episode", forward_only=False)
        c = pmc_passes(ns, B, T, kernels=(dom,))
        if c and dom in c:
            fill_roofline_counters(rl, c[dom], "measured in this run: rocprofv3 --pmc passes of `bench.py --workload %s --timed-only`" % name, B, dom_frames, dom_ms)
            rl["counters_per_launch"] = c
    return {"workload": cfg, "model": asset_, "batch": B, "dtype": dtype, "value": B * steps / dt, "unit": "env-steps/s",
            "solver": leg_solver,
            "what": ("forward only" if fwd_only else "forward + adjoint") + ", %s, episodes of %d frames, one launch per episode each way" % (
                "5 sub-steps per env-step" if fps == 1 else "one env-step = %d frames of %d sub-step (a new joint target every sub-step)" % (fps, S), T),
            "steps": steps, "ms_per_step": dt / steps * 1e3,
            # counted over the launches of the TIMED region itself (status is read after it)
            "nonconverged_envs": bad_env, "nonconverged_substeps": bad_sub, "substeps_timed": B * steps * fps * S,
            "residual_evals_per_substep_last_launch": {"mean": float(evals.mean()) / (T * S), "max_env_total": int(evals.max()), "mean_env_total": float(evals.mean()),
                                                       # a launch lasts its slowest environment's chain: the share of the SIMD time of a launch that is idle by that alone
                                                       "idle_share_if_launch_lasts_slowest_env": 1.0 - float(evals.mean()) / max(float(evals.max()), 1.0),
                                                       "trials_evaluated_by_helper_slots": int(helped.sum()), "of_them_for_the_slowest_env": int(helped[int(evals.argmax())])},
            "launch_shape": info, "kernel": krec, "roofline": rl}


def plumbing_only(args, world, rank):
    """No simulator, no GPU needed: the process group of `--gpus N`, one barrier, one all-reduce of the policy-gradient payload."""
    import torch.distributed as dist
    ranks = 1
    if world > 1:
        t = torch.ones(POLICY_GRAD_FLOATS, dtype=torch.float32)
        dist.all_reduce(t)
        ranks = int(t[0].item())
        dist.barrier()
    if rank == 0:
        print(json.dumps({"plumbing_only": True, "n_gpus": world, "ranks_in_allreduce": ranks, "backend": args.backend if world > 1 else None,
                          "allreduce_bytes": 4 * POLICY_GRAD_FLOATS, "value": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
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
    ap.add_argument("--graph", action="store_true", help="replay each episode of the timed windows from ONE HIP graph (host/graphed.GraphedEpisode) after two eager windows that carry the kernels' HIP events; "
                    "measured at the headline's shape: 20.95 M against 20.97 M eager — the window is kernel time, not host time (profiles/r05_graphed_episode.md) — so off by default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes (roofline.traffic / roofline.valu then "
                    "come from the committed profile of this command, with the source stated)")
    ap.add_argument("--no-closed-loop", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true", help="skip the f64 / dclaw / insertion legs of the N = 1 line")
    ap.add_argument("--pmc-dump", default=None, help="write the counters of the in-run --pmc passes to this JSON file (profiles/)")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE.json configs[1] style run (not the headline)")
    ap.add_argument("--launch", default="episode", choices=["episode", "step"],
                    help="episode: tsim_rollout + tsim_backward_episode, one launch each way per episode (the open-loop "
                         "episode of EpisodicSimFunction); step: one tsim_step / tsim_backward_steps launch per env-step "
                         "(StepSimFunction granularity, what a closed-loop policy needs)")
    ap.add_argument("--readout-only", action="store_true", help="only the RollingBall read-out leg at --batch environments (the --pmc passes of readout_hbm run this)")
    ap.add_argument("--timed-only", action="store_true",
                    help="skip every leg after the timed region: profiler runs (and the in-run --pmc passes)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, the real multi-GPU path) or gloo (plumbing test: with TSIM_BENCH_SHARE_GPU=1 "
                         "all ranks share cuda:0 and the collectives go through host copies)")
    ap.add_argument("--plumbing-only", action="store_true", help="set up the ranks of --gpus N, run one all-reduce of the policy-gradient "
                    "payload, print n_gpus and exit (no simulator; works without a GPU on gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        fatal("--gpus must be >= 1")
    if args.readout_only:
        dev = torch.device("cuda", 0)
        print(json.dumps(readout_leg(torch.float32 if args.dtype == "f32" else torch.float64, dev, B=args.batch or 256)), flush=True)
        return
    if args.forward_only and args.workload != "push":
        fatal("--forward-only is a TactilePush option; dclaw / insertion are forward-only already")

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

    asset_, B_def, T_def, fwd_only_def, cfg_text = WORKLOADS[args.workload]
    B = args.batch or B_def
    S, T = args.frame_skip, (args.episode or T_def)
    forward_only = args.forward_only or fwd_only_def
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    wl = make_workload(args.workload, B, T, S, rank, dev, tdt)
    S, fps = wl["S"], wl["fps"]               # TactileInsertion: frames of one sub-step, 5 frames per env-step
    model = wl["model"]
    leg = Leg(wl, dev, tdt, forward_only, world, args.backend)
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
    # --graph: the episode as ONE HIP graph (reset, forward launch, backward launch, gradient reduction; host/graphed.GraphedEpisode).  The value then comes
    # from graph-replayed windows; two EAGER windows are timed the same way before them — they carry the HIP events of the individual kernels (no events
    # inside a graph) and are listed next to the others (`repeats.eager_values`).  Default: eager windows only (at the headline's shape a window IS its
    # kernels: 3.08 + 0.83 ms of 3.91).
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
        fm, _, _ = leg.ev_stats("fwd"); bm, _, _ = leg.ev_stats("bwd")
        ar = [a.elapsed_time(b) * 1e3 for a, b in leg.ar_ev]
        k_ms = (sum(a.elapsed_time(b) for a, b, _ in leg.ev["fwd"]) + sum(a.elapsed_time(b) for a, b, _ in leg.ev["bwd"])) / n_win      # per window (the events cover all windows)
        ar = ar[len(ar) - len(ar) // n_win:] if n_win > 1 and len(ar) >= n_win else ar                                                # the last window's all-reduces
        mine = {"rank": rank, "device": torch.cuda.get_device_name(dev), "k_forward_ms_per_launch": fm, "k_backward_ms_per_launch": bm,
                "launches": len(leg.ev["fwd"]) // n_win, "allreduce_us_mean": float(np.mean(ar)) if ar else None, "allreduce_us_max": float(np.max(ar)) if ar else None,
                "own_wall_ms": dt_own * 1e3, "kernel_ms_total": k_ms, "allreduce_ms_total": sum(ar) / 1e3,
                "host_gap_ms": dt_own * 1e3 - k_ms - sum(ar) / 1e3, "wait_for_slowest_rank_ms": (dt - dt_own) * 1e3}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # kernel durations of the timed region (HIP events on the launching stream), per launch and per env-step
    fwd_ms, fwd_ms_step, fwd_frames = leg.ev_stats("fwd")
    bwd_ms, bwd_ms_step, bwd_frames = leg.ev_stats("bwd")
    if args.timed_only:
        if rank == 0:
            print(json.dumps({"timed_only": True, "value": B * world * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                              "values": [B * world * args.steps / w[0] for w in windows], "kernel_variant": sim.kernel_variant(),
                              "kernel_ms": {"k_forward": fwd_ms, "k_backward": bwd_ms}, "per_rank": per_rank}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return

    # second leg, reported next to the headline: the other launch granularity on the same workload (short, after the
    # timed region)
    other = "step" if args.launch == "episode" else "episode"
    k_other = min(args.steps, 40)
    run_steps(min(k_other, 5) * fps, False, other)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run_steps(k_other * fps, False, other)
    torch.cuda.synchronize()
    other_value = B * k_other / (time.perf_counter() - t1)
    bad_sub_timed, bad_env_timed = leg.timed_nonconverged()

    # untimed: Newton work statistics (residual evaluations per env-step) of the same workload
    sim.reset(wl["q0"], None, backward_flag=False)
    evs = []
    out = {}
    for t in range(min(T, 30)):
        sim.step(wl["u"][t], S, out=out)
        evs.append(sim.last_evals())
    evs = np.array(evs)
    status_bad = int((out["status"] != 0).sum().item())
    launch_shape = sim.launch_info()

    if rank == 0:
        rl, (dom, dom_ms, dom_frames) = leg.roofline(esz)
        value = B * world * args.steps / dt
        what = "fwd only" if forward_only else "fwd+bwd"
        label = {"push": "TactilePush", "dclaw": "DClaw rotate", "insertion": "TactileInsertion"}[args.workload]
        res = {
            "metric": "env-steps/sec (%s) %s batch=%d" % (what, label, B),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("TactilePush (pusher.xml, 13x10 taxels, ndof_r 7) gd_tactile %s, frame_skip %d, batch %d envs/GPU, episodes of %d env-steps"
                                    % ("fwd+adjoint" if not forward_only else "forward only", S, B, T)) if args.workload == "push" else
                                   "%s; %s.xml, ndof_r %d, %d tactile values, frame_skip %d, batch %d envs/GPU, episodes of %d frames" % (cfg_text, asset_, nr, ntac, S, B, T),
                       "global_batch": B * world, "parallelism": "env-sharded x%d, policy-grad all-reduce %d B/episode" % (world, 4 * POLICY_GRAD_FLOATS)},
            "solver": leg.solver,
            "repeats": {"windows": len(windows), "steps_per_window": args.steps, "value_is": "median window" + (" (episodes replayed from one HIP graph each)" if leg.graph is not None else ""),
                        "graph": graph_note, "eager_values": [B * world * args.steps / w[0] for w in eager_windows],
                        "values": [B * world * args.steps / w[0] for w in windows],
                        "min": B * world * args.steps / max(w[0] for w in windows), "max": B * world * args.steps / min(w[0] for w in windows),
                        "first": B * world * args.steps / windows[0][0], "timed_region_s_each": [w[0] for w in windows]},
            "kernel": kernel_record(sim, args.dtype, forward_only),
            "ranks": {"world_size": world, "ranks_in_first_allreduce": ranks_seen, "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if world > 1 else None,
                      "shared_gpu": share},
            "roofline": rl,
            "launch": {"mode": args.launch,
                       "episode": "tsim_rollout + tsim_backward_episode: one launch each way per episode (EpisodicSimFunction's open-loop episode)",
                       "step": "tsim_step + tsim_backward_steps: one launch per env-step each way (StepSimFunction granularity)",
                       "other_mode": other, "other_mode_value": other_value, "other_mode_env_steps": k_other},
            "nonconverged_envs_last_step": status_bad, "nonconverged_warmup": bad_warm,
            "nonconverged_timed": {"substeps": bad_sub_timed, "envs": bad_env_timed, "of_substeps": B * args.steps * fps * S},
            "per_rank": per_rank,
            "launch_shape": launch_shape,      # LDS bytes / block, blocks, lanes per environment
            "residual_evals_per_env_step": {"mean": float(evs.mean()), "p99": float(np.percentile(evs, 99)),
                                            "mean_of_per_step_max": float(evs.max(axis=1).mean()), "max": int(evs.max())},
        }
        res["roofline"]["kernel_instantiation"] = res["kernel"][dom]["instantiation"]
        # free the batch before the other legs (tape: 0.6 GB) — and so that the profiled child runs see an idle GPU
        del sim, leg
        torch.cuda.empty_cache()
        progress("timed region + launch-mode leg + evaluation statistics done")
        if world == 1:
            pmc = None
            if not args.no_pmc:
                pmc = pmc_passes(args, B, T)
                progress("rocprofv3 --pmc passes done")
            src = "measured in this run: rocprofv3 --pmc passes of `bench.py --timed-only` with this run's --steps / --batch / --dtype / --workload"
            if pmc is None or dom not in pmc:
                pmc, src = pmc_from_profile(args, B), "profiles/r03_pmc_%s.json (committed rocprofv3 --pmc run of this command; the in-run passes were skipped or failed)" % args.dtype
            if pmc is not None and dom in pmc:
                fill_roofline_counters(res["roofline"], pmc[dom], src, B, dom_frames, dom_ms)
                res["roofline"]["counters_per_launch"] = pmc
                if args.pmc_dump:
                    json.dump({"note": "rocprofv3 --pmc, separate passes " + " | ".join(" ".join(p_) for p_ in PMC_PASSES) + "; mean per dispatch of `python bench.py "
                               "--steps %d --warmup %d --timed-only` %s B=%d; FETCH_SIZE / WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE under-reports "
                               "wide reads by 2x); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles" % (args.steps, args.steps, args.dtype, B),
                               "frames_per_launch": dom_frames, "per_kernel": pmc}, open(args.pmc_dump, "w"), indent=1)
            if args.workload == "push" and not args.no_sub_records:
                # the reference's arithmetic type (envs/tactile_push_env.py:29 torch.double) and the other two multi-GPU configs, each a
                # short leg with its own roofline
                for key, (nm, dty) in {"f64": ("push", "f64"), "push_forward_only_b1024": ("push_fwd", args.dtype), "dclaw": ("dclaw", args.dtype), "insertion": ("insertion", args.dtype)}.items():
                    if key == "f64" and (args.dtype == "f64" or forward_only):
                        continue
                    try:
                        res[key] = sub_record(nm, dty, dev, pmc=(not args.no_pmc) and key in ("dclaw", "insertion"))
                    except Exception as e:      # the headline must not die with an optional leg
                        res[key] = {"error": repr(e)}
                    progress("sub-record %s done" % key)
                try:      # the headline workload with one parameter table per environment: must stay on compiled-in kernels (round 4: fell to the generic ones)
                    res["push_env_tables"] = sub_record("push", args.dtype, dev, steps=20, warm=20, env_tables=True)
                except Exception as e:
                    res["push_env_tables"] = {"error": repr(e)}
                progress("sub-record push_env_tables done")
                if "f64" in res and "value" in res["f64"]:
                    res["f64_value"] = res["f64"]["value"]      # NB: the bench's solver options (kink crossing + a budget of 256), see f64.solver
                    try:                                          # ... and the same leg under the library's fp64 default: the loop the parity tests pin
                        lib_ = sub_record("push", "f64", dev, steps=20, warm=5, solver="library")
                        res["f64_library_default"] = {k: lib_[k] for k in ("value", "ms_per_step", "solver", "nonconverged_envs", "nonconverged_substeps", "substeps_timed",
                                                                           "residual_evals_per_substep_last_launch")}
                    except Exception as e:
                        res["f64_library_default"] = {"error": repr(e)}
                    progress("sub-record f64 (library default solver) done")
            if args.workload == "push" and not args.no_closed_loop and not forward_only:
                try:                            # the path examples/train_tactile_push_gd_batched.py runs by default
                    res["closed_loop"] = closed_loop_fused_leg(model, B, T, tdt, dev)
                except Exception as e:      # the headline must not die with an optional leg
                    res["closed_loop"] = {"error": repr(e)}
                try:
                    res["closed_loop_per_step_graph"] = closed_loop_leg(model, B, T, tdt, dev)
                except Exception as e:
                    res["closed_loop_per_step_graph"] = {"error": repr(e)}
                progress("closed loop done")
            try:
                res["readout_hbm"] = readout_legs(tdt, dev, args.dtype, pmc=not args.no_pmc)
            except Exception as e:
                res["readout_hbm"] = {"error": repr(e)}
            progress("read-out leg done")
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.workload, model, S, not forward_only)
                progress("cpu baseline done")
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------- hardware counters
def pmc_passes(args, B, T, kernels=("k_forward", "k_backward")):
    """Counters of the bench kernels, collected by re-running this script's timed region under `rocprofv3 --pmc` (counters
    only, one pass per counter group: FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md §rocprofv3 PMC slots).
    Returns {kernel: {counter: mean per dispatch}} or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="tsim_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {k: {} for k in kernels}
    try:
        for i, counters in enumerate(PMC_PASSES):
            d = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", str(args.steps), "--warmup", str(args.steps), "--batch", str(B), "--dtype", args.dtype, "--workload", args.workload,
                   "--episode", str(T), "--frame-skip", str(args.frame_skip), "--launch", args.launch, "--timed-only", "--no-pmc", "--repeats", "1",
                   "--no-cpu-baseline"] + (["--forward-only"] if args.forward_only else [])
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150, check=True)
            per = {k: {} for k in kernels}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    for k in kernels:
                        if k in r.get("Kernel_Name", ""):
                            per[k].setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                            per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for k in kernels:
                for c in counters:
                    if c in per[k]:
                        vals[k][c] = float(np.mean(list(per[k][c].values())))
        need = [c for p_ in PMC_PASSES for c in p_]
        return {k: v for k, v in vals.items() if all(c in v for c in need)} or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_from_profile(args, B):
    f = os.path.join(ROOT, "profiles", "r03_pmc_%s.json" % args.dtype)
    if not (os.path.exists(f) and B == 4096 and args.workload == "push"):
        return None
    try:
        pj = json.load(open(f))
        if pj.get("frames_per_launch") != min(args.steps, args.episode or 100):
            return None
        return pj["per_kernel"]
    except Exception:
        return None


def fill_roofline_counters(rl, c, src, B, frames, kernel_ms):
    """HBM traffic and the VALU side of the roofline from the counters of one launch of the dominant kernel."""
    # gfx950: FETCH_SIZE reports half of the bytes of wide reads, WRITE_SIZE as is; both in KiB (MI355X_MICROARCH.md §HBM)
    rl["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    rl["traffic_source"] = src + "; 2 * FETCH_SIZE + WRITE_SIZE (KiB) per launch of %d env-steps" % frames
    t = kernel_ms * 1e-3
    valu = c["SQ_INSTS_VALU"]                                   # wave-level VALU instructions of the launch
    f32 = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + 2.0 * c["SQ_INSTS_VALU_FMA_F32"]
    f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]
    # mean fraction of the 64 lanes active in a VALU instruction: SQ_THREAD_CYCLES_VALU counts active lanes per instruction (a
    # full-lane elementwise kernel reads exactly 64 per SQ_INSTS_VALU: profiles/r02_pmc_calibration.json)
    lanes = c["SQ_THREAD_CYCLES_VALU"] / max(64.0 * c["SQ_INSTS_VALU"], 1.0)
    flops = 64.0 * lanes * (f32 + f64)                          # lane-level flops (FMA = 2), idle lanes not counted
    rl["valu"] = {
        "source": src,
        "valu_wave_insts_per_env_step": valu / (B * frames),
        # issue roofline: a SIMD issues at most one VALU instruction per 4 cycles from ONE wavefront (2 cycles with >= 2)
        "achieved_wave_insts_per_s": valu / t, "peak_wave_insts_per_s_one_wave_per_simd": N_SIMD * CLOCK_GHZ * 1e9 / 4.0,
        "frac_of_one_wave_issue_rate": (valu / t) / (N_SIMD * CLOCK_GHZ * 1e9 / 4.0),
        "frac_of_chip_issue_rate": (valu / t) / (N_SIMD * CLOCK_GHZ * 1e9 / 2.0),
        "wavefronts": c["SQ_WAVES"], "wavefronts_per_simd": c["SQ_WAVES"] / N_SIMD,
        "wave_issuing_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "wave_valu_frac": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
        "wave_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "active_lane_frac": lanes,
        "fp_flops_per_env_step": flops / (B * frames), "fp64_share_of_flops": f64 / max(f32 + f64, 1.0),
        "achieved_tflops": flops / t / 1e12, "peak_tflops": FP32_VALU_PEAK_TFLOPS, "frac": flops / t / 1e12 / FP32_VALU_PEAK_TFLOPS,
    }


# ---------------------------------------------------------------------------------------------------- closed GD loop
def closed_loop_leg(model, B, T, tdt, dev, epochs=3):
    """BASELINE config 3 as algorithms/gd.py:224-259 runs it — observation -> policy -> env-step, 100 env-steps, BPTT, one
    gradient all-reduce + clip + Adam per epoch — with every environment of the batch as one episode and the episode + its
    backward replayed from one HIP graph (algorithms/batched_gd.GraphedRollout)."""
    from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
    from tactilesimulation_amd.algorithms.batched_gd import Actor, GraphedRollout, train_epoch_graphed
    env = BatchedTactilePushEnv(model, B, device=str(dev), dtype=tdt, gradient=True, seed=0, tape_steps=T)
    env.reset()
    q0, goal = env.q0.clone(), env.goal.clone()
    rng = np.random.default_rng(1)
    dist_ = torch.tensor(rng.uniform(-1, 1, size=(T, B, 2)) * (rng.uniform(size=(T, B, 1)) < 0.5), device=dev, dtype=tdt)
    torch.manual_seed(0)
    actor = Actor(dtype=tdt).to(dev)
    opt = torch.optim.Adam(actor.parameters(), lr=5e-3, betas=(0.7, 0.95))          # cfg/gd_tactile.yaml
    gr = GraphedRollout(env, actor, T, q0, goal, dist_, warmup=1)
    train_epoch_graphed(gr, opt, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [train_epoch_graphed(gr, opt, B).detach().clone() for _ in range(epochs)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"value": B * T * epochs / dt, "unit": "env-steps/s", "s_per_epoch": dt / epochs, "epochs": epochs, "horizon": T, "batch": B,
           "what": "closed GD epoch: policy MLP (29 574 parameters) between env-steps, per-env-step launches of the simulator, BPTT, "
                   "gradient normalise + clip + Adam; one HIP graph replay per episode",
           "loss_per_episode": [float(l) / B for l in losses]}
    del gr, env
    torch.cuda.empty_cache()
    return res


def closed_loop_fused_leg(model, B, T, tdt, dev, epochs=3):
    """The same GD epoch with the policy INSIDE the simulator's episode launches (envs/push_closed_loop.FusedPushEpisode,
    include/tsim_env.h tsim_push_closed_rollout / _backward): one launch each way per episode, so no env-step waits for the batch's
    slowest environment; reward, its partials and the weight-gradient GEMMs stay in torch.  Same episode data, same optimiser, same
    cold start as closed_loop_leg; the two legs' gradients agree (tests/test_gpu_closed_loop.py)."""
    from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
    from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode, train_epoch_fused
    from tactilesimulation_amd.algorithms.batched_gd import Actor
    env = BatchedTactilePushEnv(model, B, device=str(dev), dtype=tdt, gradient=True, seed=0, tape_steps=T)
    env.reset()
    q0, goal = env.q0.clone(), env.goal.clone()
    rng = np.random.default_rng(1)
    dist_ = torch.tensor(rng.uniform(-1, 1, size=(T, B, 2)) * (rng.uniform(size=(T, B, 1)) < 0.5), device=dev, dtype=tdt)
    torch.manual_seed(0)
    actor = Actor(dtype=tdt).to(dev)
    opt = torch.optim.Adam(actor.parameters(), lr=5e-3, betas=(0.7, 0.95))          # cfg/gd_tactile.yaml
    ep = FusedPushEpisode(env, actor, T)
    train_epoch_fused(ep, opt, q0, goal, dist_, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [train_epoch_fused(ep, opt, q0, goal, dist_, B).detach().clone() for _ in range(epochs)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = int((ep.status != 0).sum().item())
    res = {"value": B * T * epochs / dt, "unit": "env-steps/s", "s_per_epoch": dt / epochs, "epochs": epochs, "horizon": T, "batch": B,
           "what": "closed GD epoch with the policy (393-64-64-3 ELU MLP, observation, action mapping) evaluated inside the simulator's episode "
                   "launches: one launch each way per episode; reward partials and weight-gradient GEMMs in torch; normalise + clip + Adam",
           "loss_per_episode": [float(l) / B for l in losses], "nonconverged_envs_last_epoch": bad}
    del ep, env
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------- HBM-relevant read-out
L3_BYTES = 256 * 1024 * 1024      # Infinity Cache (MI355X_MICROARCH.md): a write stream smaller than this is absorbed on-die


def readout_pmc(B, dtype):
    """WRITE_SIZE / FETCH_SIZE of k_taxels from two counters-only rocprofv3 passes of `bench.py --readout-only` (bytes per launch)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="tsim_pmc_ro_", dir="/tmp")
    out = {}
    try:
        for c in ("WRITE_SIZE", "FETCH_SIZE"):
            d = os.path.join(tmp, c)
            cmd = [exe, "--pmc", c, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--readout-only", "--batch", str(B), "--dtype", dtype]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_taxels" in r.get("Kernel_Name", "") and r["Counter_Name"] == c:
                        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            if not per:
                return None
            out[c] = float(np.median(list(per.values()))) * 1024.0      # KiB as reported
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def readout_legs(tdt, dev, dtype, pmc=True):
    """The read-out against HBM, not against the Infinity Cache: 256 environments write 123 MB (inside the 256 MiB L3: that figure is an
    on-die rate), 1024 write 0.49 GB, 4096 write 1.97 GB = 7.7 x L3.  The record's headline numbers are the LARGEST batch's; WRITE_SIZE of
    a counters-only rocprofv3 pass confirms that the bytes went out."""
    legs = []
    for B in (256, 1024, 4096):
        legs.append(readout_leg(tdt, dev, B=B))
        torch.cuda.empty_cache()
    big = dict(legs[-1])
    big["by_batch"] = [{k: l[k] for k in ("environments", "bytes_written", "x_l3", "ms", "achieved", "frac", "ms_cold", "achieved_cold")} for l in legs]
    if pmc:
        c = readout_pmc(4096, dtype)
        if c:
            big["pmc"] = {"WRITE_SIZE_bytes": c["WRITE_SIZE"], "FETCH_SIZE_bytes_x2": 2.0 * c["FETCH_SIZE"], "written_over_algorithmic": c["WRITE_SIZE"] / big["bytes_written"],
                          "source": "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate counters-only passes) of `bench.py --readout-only --batch 4096`, median k_taxels dispatch"}
    return big


def readout_leg(tdt, dev, B=256, reps=5):
    """k_readout on RollingBall's 200 x 200 taxels (assets/tactile_pad/tactile_pad.xml:29): 480 KB written per
    environment and read-out, taxel constants (12 planes) re-read per environment from L2 — the one kernel of this path
    whose time is set by memory traffic."""
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.workloads import asset
    m = load_model(asset("tactile_pad"))
    sim = BatchSim(m, B, device=str(dev), dtype=tdt, tape_capacity=0)
    sim.reset(torch.zeros(B, sim.ndof_r, device=dev, dtype=tdt), None, backward_flag=False)
    u = torch.zeros(B, sim.ndof_u, device=dev, dtype=tdt)
    u[:, 2] = 0.2                                               # the first 100 steps of examples/RollingBallExp/test_sim_speed.py:43-48:
    for _ in range(100):                                        # the pad comes down on the ball
        sim.step(u, 1, want_var=False, want_tactile=False)
    sim.readout()
    torch.cuda.synchronize()
    # as the reference's loop runs it (test_sim_speed.py:50-56): forward(1), then the read-out — only the read-out is timed.  The forward
    # launch leaves the pose records of its final state, so the read-out is k_taxels alone; "cold" is the read-out of a state that no
    # forward launch produced (after reset(q, qd)): kinematics kernel + k_taxels.
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        sim.step(u, 1, want_var=False, want_tactile=False)
        e[i][0].record()
        _, tac = sim.readout(want_var=False)
        e[i][1].record()
    torch.cuda.synchronize()
    ms = min(a.elapsed_time(b) for a, b in e)
    q, qd = sim.get_state()
    for i in range(reps):
        sim.reset(q, qd, backward_flag=False)
        e[i][0].record()
        sim.readout(want_var=False)
        e[i][1].record()
    torch.cuda.synchronize()
    ms_cold = min(a.elapsed_time(b) for a, b in e)
    esz = 4 if tdt == torch.float32 else 8
    written = B * sim.ndof_tactile * esz
    del sim
    return {"kernel": "k_taxels (tsim_readout after a forward launch; cold: k_readout + k_taxels)", "workload": "RollingBall tactile_pad.xml, 200 x 200 taxels, %d environments" % B, "ms": ms,
            "environments": B, "x_l3": written / L3_BYTES,
            "ms_cold": ms_cold, "achieved_cold": written / (ms_cold * 1e-3) / 1e9,
            "bytes_written": written, "achieved": written / (ms * 1e-3) / 1e9, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": written / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "taxels_in_contact_max": int((tac.reshape(B, -1, 3)[:, :, 2] != 0).sum(1).max().item())}


# ---------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(workload, model, S, with_backward):
    """fp64 CPU oracle (oracle/tsim_oracle.cpp — the build's own restatement, kind "port") on a bounded sample of the same
    workload, rebuilt here with -O3 -march=native: one thread, and one oracle instance per USABLE core (affinity mask capped
    by the cgroup CPU quota; environments are independent, ctypes releases the GIL)."""
    import threading
    from oracle.oracle import OracleSim
    from tactilesimulation_amd import workloads as W
    nstep = {"push": 100, "dclaw": 50, "insertion": 45}[workload]
    gen = {"push": lambda n, seed: W.push_workload(n, nstep, seed=seed)[:2], "dclaw": lambda n, seed: W.dclaw_random_workload(n, nstep, seed=7 + seed),
           "insertion": lambda n, seed: W.insertion_attempt_workload(n, seed=7 + seed)}[workload]
    unit = 1.0 / FRAMES_PER_ENV_STEP[workload]                 # TactileInsertion: frames of one sub-step, 5 of them = one env-step
    nenv = 8
    q0, u = gen(nenv, 0)
    try:
        o = OracleSim(model, native=True)
        flags = "g++ -O3 -march=native (built on this host)"
        native = True
    except Exception:
        o = OracleSim(model)
        flags = "g++ -O3 (portable build; the native rebuild failed)"
        native = False
    o.bench_rollout(q0[:1], u[:1, :5], S, with_backward)       # warm
    t0 = time.perf_counter()
    n, _ = o.bench_rollout(q0, u, S, with_backward)
    dt1 = time.perf_counter() - t0
    st = o.stats()
    single = n / dt1
    nthr = usable_cores()
    per = max(2, int(round(12.0 * single / nstep)))            # ~12 s of CPU work per thread
    q0m, um = gen(per * nthr, 1)
    sims = [OracleSim(model, native=native) for _ in range(nthr)]
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
    busy = ((c1.user - c0.user) + (c1.system - c0.system)) / dtm      # cores actually kept busy
    what = "fwd+adjoint" if with_backward else "fwd only"
    single *= unit
    return {"value": sum(done) / dtm * unit, "unit": "env-steps/s", "cores": nthr, "kind": "port",
            "sample": "%d threads (usable cores) x %d envs x %d env-steps of the same workload, %s, fp64, %s, one oracle "
                      "instance per thread; single thread: %d envs x %d env-steps; mean Newton iterations/sub-step %.2f"
                      % (nthr, per, nstep, what, flags, nenv, nstep, st["newton_iters"] / max(st["substeps"], 1)),
            "single_thread_value": single, "host_cpus": os.cpu_count(), "cores_busy": busy}


if __name__ == "__main__":
    main()
