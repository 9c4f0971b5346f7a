"""bench_legs.py — what bench.py measures with: the workloads, the timed leg, the per-kernel roofline, and the optional legs that follow the headline.

Nothing here imports the simulator at module level (bench.py --plumbing-only runs without a GPU).  Only `cpu_baseline` touches oracle/ (the checker,
timed as the CPU baseline: task statement ④)."""
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
BENCH = os.path.join(ROOT, "bench.py")

F64_EVAL_BUDGET = 256           # evaluations per sub-step in the f64 legs
DCLAW_EVAL_BUDGET = 64          # ... in the budgeted D'Claw collection figure (reported NEXT to the unbudgeted one)
POLICY_GRAD_FLOATS = 29574      # DiagGaussianActor(393 -> 64 -> 64 -> 3), SURVEY.md §2.2
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3   # 256 CU x 4 SIMD x 64 lanes x 2 flop per 2 cycles at 2.4 GHz (packed / two wavefronts per SIMD)
N_SIMD, CLOCK_GHZ = 1024, 2.4
KERNELS = ("k_forward", "k_taxels", "k_backward")      # host/batch.BatchSim.KERNEL_KINDS; rocprof names contain them (k_taxels_small contains k_taxels)
PMC_PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"],
    ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
     "SQ_INSTS_VALU_FMA_F64", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]

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
# BASELINE.json configs[3] / configs[4] are 8-GPU jobs (16 384 / 32 768 environments): the same jobs on ONE GPU, next to their per-GPU shares
WHOLE_CONFIG_BATCH = {"dclaw": 16384, "insertion": 32768}


def kernel_bytes(nr, nu, nvar, ntac, S, esz, tape=True, tac_frac=1.0, inkernel_readout=False):
    """ALGORITHMIC bytes per frame (one env-step of S sub-steps) of SURVEY.md §8d, charged to the kernel that moves each term:
        k_forward   reads u, writes q and variables, writes the tape (q, qd per sub-step) when recording     esz (nu + nr + nvar + 2 nr S)   =   356 B
        k_taxels    writes the tactile frame (from the pose records k_forward leaves: implementation traffic)  esz ntac                      = 1 560 B
        k_backward  reads the tape and the three seeds, writes dL/du per sub-step                             esz (2 nr S + nr + nvar + ntac + nu S) = 2 012 B
    (fp32 TactilePush figures; forward side 1 916 B, forward + adjoint 3 928 B as in the survey's table).  tac_frac: share of the frames whose
    tactile frame is captured (tactile_masks); inkernel_readout: the launch had no separate read-out kernel, k_forward wrote the tactile frame itself."""
    tac = esz * ntac * tac_frac
    kf = esz * (nu + nr + nvar + (2 * nr * S if tape else 0)) + (tac if inkernel_readout else 0.0)
    kt = 0.0 if inkernel_readout else tac
    kb = esz * (2 * nr * S + nr + nvar + nu * S) + tac
    return {"k_forward": kf, "k_taxels": kt, "k_backward": kb}


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
        mangled, readable = buildhash.kernel_name(k, dtype, sim.ndof_r, has_exp, info["lanes_per_env"], variant, policy, default_opts=sim.get_option(sim.OPT_ALL_DEFAULT) == 1)
        rec[k] = dict({"instantiation": readable, "symbol": mangled}, **(table.get(mangled) or {"metadata": "not found in %s" % os.path.basename(buildhash.KERNELS_JSON)}))
    return rec


class Leg:
    """One workload on one BatchSim: runs env-steps as episodes of <= T (forward all, then backward all).  The kernels of its launches are timed by
    the library's own HIP events (include/tsim.h tsim_kernel_timing: one pair per kernel launch, on the launching stream)."""

    def __init__(self, wl, dev, tdt, forward_only, world=1, backend="nccl", solver="bench", eval_budget=None):
        from tactilesimulation_amd.host.batch import BatchSim
        self.wl, self.dev, self.tdt, self.forward_only, self.world, self.backend = wl, dev, tdt, forward_only, world, backend
        B, T, S = wl["B"], wl["T"], wl["S"]
        self.sim = sim = BatchSim(wl["model"], B, device=str(dev), dtype=tdt, tape_capacity=0 if forward_only else T * S)
        sim.kernel_timing(True)
        # Solver options (include/tsim.h tsim_set_solver_options).  Every leg of this bench runs the XML's Newton loop with kink
        # crossing near convergence — the library's default for fp32 batches; f64 legs are given the same option so that they differ
        # from the headline in arithmetic only (the library's fp64 default is the bare loop: what the parity tests pin).
        # f64 legs: 2 of the 4096 TactilePush environments cycle between the two sides of a kink (the loop then runs ~1000 evaluations to
        # max_iter, non-converged either way): bounded.  The fp32 headline has no budget (its largest sub-step: 43 evaluations).
        self.eval_budget = eval_budget if eval_budget is not None else (F64_EVAL_BUDGET if (tdt == torch.float64 and solver != "library") else 0)
        if solver == "bare":       # the XML-stated loop and nothing else, whatever the dtype (the reference has no other; fp32 batches run kink crossing by default)
            sim.set_solver_options(cross_kinks=False, eval_budget=0)
            self.solver = "the bare XML Newton loop (tol / max_iter / max_ls of the model), no kink crossing, no evaluation budget"
        elif solver != "library":
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
                ro, _, _ = self.graph.replay()      # one replay = one whole episode (reset, forward launch, backward launch, gradient reduction)
                status = ro["status"]
                if timed:
                    self.status_log.append(status.clone())
            elif launch == "episode":
                sim.reset(wl["q0"], None, backward_flag=not self.forward_only)
                ro = sim.rollout(u[:n], S, tactile_mask=wl["tactile_mask"][:n]) if "tactile_mask" in wl else sim.rollout(u[:n], S)
                status = ro["status"]
                if timed:
                    self.status_log.append(status)
                if not self.forward_only:
                    du = sim.backward_episode(n, S, self.wqT[:n], self.wvT[:n] if self.wvT is not None else None, self.wtT[:n])
                    self.grad_buf[:ng] = du.sum((0, 1)).float()[:ng]
            else:
                sim.reset(wl["q0"], None, backward_flag=not self.forward_only)
                for t in range(n):
                    sim.step(u[t], S, out=self.out)
                status = self.out["status"]
                if not self.forward_only:
                    for t in reversed(range(n)):
                        du = sim.backward_steps(S, self.wq, self.wv, self.wt)
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


def per_kernel_roofline(leg, esz, ktimes, n_win, frames_per_launch):
    """HBM side of the roofline, one row per kernel, each charged ITS OWN algorithmic bytes (kernel_bytes) over ITS OWN average launch duration
    (HIP events recorded by the library around that kernel alone).  The headline `roofline` fields are the dominant kernel's (longest launch)."""
    wl = leg.wl
    S, B = wl["S"], wl["B"]
    tac_frac = float(wl["tactile_mask"].sum()) / wl["T"] if "tactile_mask" in wl else 1.0
    inkernel = ktimes["k_taxels"][1] == 0 and leg.ntac > 0
    per_frame = kernel_bytes(leg.nr, leg.nu, leg.nvar, leg.ntac, S, esz, tape=not leg.forward_only, tac_frac=tac_frac, inkernel_readout=inkernel)
    rows = {}
    for k in KERNELS:
        ms_sum, n = ktimes[k]
        if n == 0:
            continue
        ms = ms_sum / n
        alg = per_frame[k] * B * frames_per_launch
        gbs = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        rows[k] = {"ms": ms, "launches_per_window": n / max(n_win, 1), "algorithmic_bytes": alg, "algorithmic_bytes_per_env_frame": per_frame[k],
                   "achieved_gbs": gbs, "frac": gbs / HBM_PEAK_GBS, "traffic": None, "traffic_over_algorithmic": None}
    dom = max(rows, key=lambda k: rows[k]["ms"]) if rows else "k_forward"
    d = rows.get(dom, {"ms": 0.0, "algorithmic_bytes": 0.0, "achieved_gbs": 0.0, "frac": 0.0})
    rl = {"bound": "hbm", "kernel": dom, "achieved": d["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
          "traffic": None, "traffic_source": None, "traffic_over_algorithmic": None,
          "algorithmic_bytes_per_launch": d["algorithmic_bytes"], "env_steps_per_launch": frames_per_launch / wl["fps"], "frames_per_launch": frames_per_launch,
          "kernel_ms": d["ms"], "timed_by": "HIP events around this kernel alone on its launching stream (tsim_kernel_timing), mean of the timed windows",
          "algorithmic_bytes_source": "SURVEY.md §8d, split by kernel (bench_legs.kernel_bytes)" + ("; tactile frames as masked (%.3f of the frames)" % tac_frac if tac_frac != 1.0 else "")
                                      + ("; read-out inside k_forward" if inkernel else ""),
          "per_kernel": rows, "valu": None}
    if "k_forward" in rows and "k_taxels" in rows:
        rl["forward_side"] = {"ms": rows["k_forward"]["ms"] + rows["k_taxels"]["ms"], "algorithmic_bytes": rows["k_forward"]["algorithmic_bytes"] + rows["k_taxels"]["algorithmic_bytes"],
                              "traffic": None, "traffic_over_algorithmic": None}
    return rl


# ---------------------------------------------------------------------------------------------------- hardware counters
def pmc_passes(args, B, T, kernels=KERNELS):
    """Counters of the bench kernels, collected by re-running bench.py's timed region under `rocprofv3 --pmc` (counters only, one pass per
    counter group: FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md §rocprofv3 PMC slots).
    Returns {kernel: {counter: mean per dispatch}} (kernels with every counter only) or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="tsim_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {k: {} for k in kernels}
    try:
        for i, counters in enumerate(PMC_PASSES):
            d = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, BENCH,
                   "--steps", str(args.steps), "--warmup", str(args.steps), "--batch", str(B), "--dtype", args.dtype, "--workload", args.workload,
                   "--episode", str(T), "--frame-skip", str(args.frame_skip), "--launch", args.launch, "--timed-only", "--repeats", "1"] + (["--forward-only"] if args.forward_only else [])
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60, check=True)      # (a pass takes ~3 s)
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


def valu_record(c, src, B, frames, kernel_ms):
    """The VALU side of the roofline from the counters of one launch of a kernel."""
    t = kernel_ms * 1e-3
    valu = c["SQ_INSTS_VALU"]                                   # wave-level VALU instructions of the launch
    f32 = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + 2.0 * c["SQ_INSTS_VALU_FMA_F32"]
    f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]
    # mean fraction of the 64 lanes active in a VALU instruction: SQ_THREAD_CYCLES_VALU counts active lanes per instruction (a
    # full-lane elementwise kernel reads exactly 64 per SQ_INSTS_VALU: profiles/r02_pmc_calibration.json)
    lanes = c["SQ_THREAD_CYCLES_VALU"] / max(64.0 * c["SQ_INSTS_VALU"], 1.0)
    flops = 64.0 * lanes * (f32 + f64)                          # lane-level flops (FMA = 2), idle lanes not counted
    return {
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


def fill_roofline_counters(rl, pmc, src, B):
    """HBM traffic of every kernel row (2 * FETCH_SIZE + WRITE_SIZE: gfx950 reports half of the bytes of wide reads, both in KiB — MI355X_MICROARCH.md
    §HBM) and the VALU record of the dominant kernel, from the counters of one launch each."""
    frames = rl["frames_per_launch"]
    for k, row in rl["per_kernel"].items():
        c = pmc.get(k)
        if not c:
            continue
        row["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        row["traffic_over_algorithmic"] = row["traffic"] / row["algorithmic_bytes"] if row["algorithmic_bytes"] else None
        row["valu_frac"] = valu_record(c, src, B, frames, row["ms"])["frac"]
    dom = rl["kernel"]
    if dom in pmc and dom in rl["per_kernel"]:
        rl["traffic"] = rl["per_kernel"][dom]["traffic"]
        rl["traffic_over_algorithmic"] = rl["per_kernel"][dom]["traffic_over_algorithmic"]
        rl["traffic_source"] = src + "; 2*FETCH_SIZE + WRITE_SIZE per launch"
        rl["valu"] = valu_record(pmc[dom], src, B, frames, rl["per_kernel"][dom]["ms"])
    fs = rl.get("forward_side")
    if fs and all(rl["per_kernel"][k].get("traffic") is not None for k in ("k_forward", "k_taxels")):
        fs["traffic"] = rl["per_kernel"]["k_forward"]["traffic"] + rl["per_kernel"]["k_taxels"]["traffic"]
        fs["traffic_over_algorithmic"] = fs["traffic"] / fs["algorithmic_bytes"]


# ---------------------------------------------------------------------------------------------------- sub-records
def sub_record(name, dtype, dev, steps=None, warm=None, solver="bench", pmc=False, env_tables=False, eval_budget=None, batch=None):
    """A short N = 1 leg of another BASELINE config (or of the headline workload in another dtype): value, per-kernel roofline.  `steps` in
    env-steps (5 sub-steps).  batch: another number of environments on this GPU than the config's per-GPU share."""
    asset_, B, T, fwd_only, cfg = WORKLOADS[name]
    B = batch or B
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
    leg = Leg(wl, dev, tdt, fwd_only, solver=solver, eval_budget=eval_budget)
    if env_tables:      # one parameter table per environment (domain randomisation, include/tsim.h tsim_set_env_tables): here every row the model's own
        leg.sim.set_env_tables(leg.sim.base_tables())
    steps = steps or 2 * T // fps                       # two episodes
    leg.run(warm * fps if warm else T, False, "episode")
    torch.cuda.synchronize()
    leg.sim.kernel_times()
    t0 = time.perf_counter()
    leg.run(steps * fps, True, "episode")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ktimes = leg.sim.kernel_times()
    bad_sub, bad_env = leg.timed_nonconverged()
    rl = per_kernel_roofline(leg, esz, ktimes, 1, min(T, steps * fps))
    dom = rl["kernel"]
    info = leg.sim.launch_info()
    evals = leg.sim.last_evals()
    helped = leg.sim.last_helper_trials()
    krec = kernel_record(leg.sim, dtype, fwd_only)
    leg_solver = leg.solver
    del leg
    torch.cuda.empty_cache()
    S = wl["S"]
    rl["instantiation"] = krec.get(dom, krec["k_forward"])["instantiation"] if dom != "k_taxels" else "k_taxels"
    if pmc:      # the same counters as the headline's, from `bench.py --workload <name> --timed-only` under rocprofv3 --pmc (separate passes)
        ns = argparse.Namespace(steps=T // fps, warmup=T // fps, dtype=dtype, workload=name, frame_skip=5, launch="episode", forward_only=False)
        c = pmc_passes(ns, B, T)
        if c:
            fill_roofline_counters(rl, c, "measured in this run: rocprofv3 --pmc passes of `bench.py --workload %s --timed-only`" % name, B)
            rl["counters_per_launch"] = c
    return {"workload": cfg, "model": asset_, "batch": B, "dtype": dtype, "value": B * steps / dt, "unit": "env-steps/s",
            "solver": leg_solver,
            "what": ("forward only" if fwd_only else "forward + adjoint") + ", %s, episodes of %d frames, one launch per episode each way" % (
                "5 sub-steps per env-step" if fps == 1 else "one env-step = %d frames of %d sub-step (a new joint target every sub-step)" % (fps, S), T),
            "steps": steps, "ms_per_step": dt / steps * 1e3,
            # counted over the launches of the TIMED region itself (status is read after it)
            "nonconverged_envs": bad_env, "nonconverged_substeps": bad_sub, "substeps_timed": B * steps * fps * S,
            # a launch lasts its slowest environment's chain: the share of the SIMD time of a launch that is idle by that alone
            "idle_share": 1.0 - float(evals.mean()) / max(float(evals.max()), 1.0),
            "residual_evals_per_substep_last_launch": {"mean": float(evals.mean()) / (T * S), "max_env_total": int(evals.max()), "mean_env_total": float(evals.mean()),
                                                       "trials_evaluated_by_helper_slots": int(helped.sum()), "of_them_for_the_slowest_env": int(helped[int(evals.argmax())])},
            "launch_shape": info, "kernel": krec, "roofline": rl}


# ---------------------------------------------------------------------------------------------------- closed GD loop
CNN_CFG = {"actor_cnn": {"kernel_sizes": [3, 3], "layer_sizes": [8, 16], "stride_sizes": [1, 1], "hidden_size": 32, "activation": "elu"}, "actor_logstd_init": -1.0}


def _closed_loop_inputs(model, B, T, tdt, dev, cnn=False):
    from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
    from tactilesimulation_amd.algorithms.batched_gd import Actor, CNNActor
    env = BatchedTactilePushEnv(model, B, device=str(dev), dtype=tdt, gradient=True, seed=0, tape_steps=T, observation_type="tactile_map" if cnn else "tactile_flatten")
    env.reset()
    q0, goal = env.q0.clone(), env.goal.clone()
    rng = np.random.default_rng(1)
    dist_ = torch.tensor(rng.uniform(-1, 1, size=(T, B, 2)) * (rng.uniform(size=(T, B, 1)) < 0.5), device=dev, dtype=tdt)
    torch.manual_seed(0)
    actor = (CNNActor((3, 13, 10), 3, CNN_CFG, state_dim=3, dtype=tdt) if cnn else Actor(dtype=tdt)).to(dev)
    opt = torch.optim.Adam(actor.parameters(), lr=5e-3, betas=(0.7, 0.95))          # cfg/gd_tactile.yaml
    return env, q0, goal, dist_, actor, opt


def closed_loop_leg(model, B, T, tdt, dev, epochs=3, cnn=False):
    """BASELINE config 3 as algorithms/gd.py:224-259 runs it — observation -> policy -> env-step, 100 env-steps, BPTT, one
    gradient all-reduce + clip + Adam per epoch — with every environment of the batch as one episode and the episode + its
    backward replayed from one HIP graph (algorithms/batched_gd.GraphedRollout: any torch policy)."""
    from tactilesimulation_amd.algorithms.batched_gd import GraphedRollout, train_epoch_graphed
    env, q0, goal, dist_, actor, opt = _closed_loop_inputs(model, B, T, tdt, dev, cnn=cnn)
    gr = GraphedRollout(env, actor, T, q0, goal, dist_, warmup=1)
    train_epoch_graphed(gr, opt, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [train_epoch_graphed(gr, opt, B).detach().clone() for _ in range(epochs)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"value": B * T * epochs / dt, "unit": "env-steps/s", "s_per_epoch": dt / epochs, "epochs": epochs, "horizon": T, "batch": B,
           "what": "closed GD epoch: %s between env-steps, per-env-step launches of the simulator, BPTT, gradient normalise + clip + Adam; one HIP graph replay per episode"
                   % ("the reference's CNN policy on its default tactile_map observation (utils/model.py:37-98; %d parameters, MIOpen convolutions)" % sum(p.numel() for p in actor.parameters())
                      if cnn else "policy MLP (29 574 parameters)"),
           "loss_per_episode": [float(l) / B for l in losses]}
    del gr, env
    torch.cuda.empty_cache()
    return res


def closed_loop_fused_leg(model, B, T, tdt, dev, epochs=3):
    """The same GD epoch with the policy INSIDE the simulator's episode launches (envs/push_closed_loop.FusedPushEpisode,
    include/tsim_env.h tsim_push_closed_rollout / _backward): one launch each way per episode, so no env-step waits for the batch's
    slowest environment; reward, its partials and the weight-gradient GEMMs stay in torch.  Same episode data, same optimiser, same
    cold start as closed_loop_leg; the two legs' gradients agree (tests/test_gpu_closed_loop.py)."""
    from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode, train_epoch_fused
    env, q0, goal, dist_, actor, opt = _closed_loop_inputs(model, B, T, tdt, dev)
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
           "loss_per_episode": [float(l) / B for l in losses], "nonconverged_envs": bad}
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
            cmd = [exe, "--pmc", c, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, BENCH, "--readout-only", "--batch", str(B), "--dtype", dtype]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
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
    big["value"] = None
    big["by_batch"] = [{k: l[k] for k in ("environments", "bytes_written", "x_l3", "ms", "achieved", "frac", "ms_cold", "achieved_cold")} for l in legs]
    big["roofline"] = {"kernel": "k_taxels", "instantiation": "k_taxels<%s>" % ("float" if dtype == "f32" else "double"), "kernel_ms": big["ms"], "frac": big["frac"],
                       "traffic_over_algorithmic": None}
    if pmc:
        c = readout_pmc(4096, dtype)
        if c:
            big["pmc"] = {"WRITE_SIZE_bytes": c["WRITE_SIZE"], "FETCH_SIZE_bytes_x2": 2.0 * c["FETCH_SIZE"], "written_over_algorithmic": c["WRITE_SIZE"] / big["bytes_written"],
                          "source": "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate counters-only passes) of `bench.py --readout-only --batch 4096`, median k_taxels dispatch"}
            big["roofline"]["traffic_over_algorithmic"] = (c["WRITE_SIZE"] + 2.0 * c["FETCH_SIZE"]) / big["bytes_written"]
    return big


def readout_leg(tdt, dev, B=256, reps=5):
    """k_taxels on RollingBall's 200 x 200 taxels (assets/tactile_pad/tactile_pad.xml:29): 480 KB written per
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
def cpu_baseline(workload, model, S, with_backward, seconds=10.0):
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
        flags = "g++ -O3 -march=native on this host"
        native = True
    except Exception:
        o = OracleSim(model)
        flags = "g++ -O3 (portable; native rebuild failed)"
        native = False
    o.bench_rollout(q0[:1], u[:1, :5], S, with_backward)       # warm
    t0 = time.perf_counter()
    n, _ = o.bench_rollout(q0, u, S, with_backward)
    dt1 = time.perf_counter() - t0
    st = o.stats()
    single = n / dt1
    nthr = usable_cores()
    per = max(2, int(round(seconds * single / nstep)))         # ~`seconds` of CPU work per thread
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
            "sample": "%d threads x %d envs x %d env-steps of the same workload, %s, fp64 oracle (%s), one instance per thread; %.2f Newton iterations/sub-step"
                      % (nthr, per, nstep, what, flags, st["newton_iters"] / max(st["substeps"], 1)),
            "single_thread_value": single, "single_thread_sample": "%d envs x %d env-steps" % (nenv, nstep), "host_cpus": os.cpu_count(), "cores_busy": busy}


# ---------------------------------------------------------------------------------------------------- the optional legs, by name
def run_leg(name, res, ctx):
    """One optional leg of the N = 1 line: fills res[...] (full record; bench.compact_line summarises it)."""
    args, B, T, dev, tdt, model = ctx["args"], ctx["B"], ctx["T"], ctx["dev"], ctx["tdt"], ctx["model"]
    push = args.workload == "push"
    if name == "pmc":
        pmc = pmc_passes(args, B, T)
        if pmc is None:
            res["roofline"]["traffic_source"] = "in-run rocprofv3 --pmc passes failed or rocprofv3 is missing: traffic / valu not measured"
            return
        src = "in-run rocprofv3 --pmc passes of this command's timed region"
        fill_roofline_counters(res["roofline"], pmc, src, B)
        res["roofline"]["counters_per_launch"] = pmc
        if args.pmc_dump:
            json.dump({"note": "rocprofv3 --pmc, separate passes " + " | ".join(" ".join(p_) for p_ in PMC_PASSES) + "; mean per dispatch of `python bench.py "
                       "--steps %d --warmup %d --timed-only` %s B=%d; FETCH_SIZE / WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE under-reports "
                       "wide reads by 2x); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles" % (args.steps, args.steps, args.dtype, B),
                       "frames_per_launch": res["roofline"]["frames_per_launch"], "per_kernel": pmc}, open(args.pmc_dump, "w"), indent=1)
    elif name == "step_mode":
        leg = ctx.get("leg")
        if leg is None:
            res["launch"] = {"skipped": "headline batch already freed"}
            return
        other = "step" if args.launch == "episode" else "episode"
        k_other = min(args.steps, 40)
        fps = ctx["fps"]
        leg.run(min(k_other, 5) * fps, False, other)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        leg.run(k_other * fps, False, other)
        torch.cuda.synchronize()
        res["launch"] = {"mode": args.launch, "other_mode": other, "other_mode_value": B * k_other / (time.perf_counter() - t1), "other_mode_env_steps": k_other,
                         "episode": "tsim_rollout + tsim_backward_episode: one launch each way per episode (EpisodicSimFunction's open-loop episode)",
                         "step": "tsim_step + tsim_backward_steps: one launch per env-step each way (StepSimFunction granularity)"}
        # Newton work statistics (residual evaluations per env-step) of the same workload
        sim, wl = leg.sim, leg.wl
        sim.reset(wl["q0"], None, backward_flag=False)
        evs, out = [], {}
        for t in range(min(T, 30)):
            sim.step(wl["u"][t], ctx["S"], out=out)
            evs.append(sim.last_evals())
        evs = np.array(evs)
        res["residual_evals_per_env_step"] = {"mean": float(evs.mean()), "p99": float(np.percentile(evs, 99)), "mean_of_per_step_max": float(evs.max(axis=1).mean()), "max": int(evs.max())}
    elif name in ("f64", "push_fwd", "dclaw", "insertion", "env_tables"):
        if not push:
            res[name] = {"skipped": "sub-records belong to the push headline"}
            return
        pmc_on = "pmc" in args.leg_list
        if name == "f64":
            if args.dtype == "f64" or ctx["forward_only"]:
                res[name] = {"skipped": "headline is f64 / forward-only already"}
                return
            res["f64"] = sub_record("push", "f64", dev)
            # ... and the same leg under the library's fp64 default: the loop the parity tests pin
            lib_ = sub_record("push", "f64", dev, steps=20, warm=5, solver="library")
            res["f64_library_default"] = {k: lib_[k] for k in ("value", "ms_per_step", "solver", "batch", "dtype", "nonconverged_envs", "nonconverged_substeps", "substeps_timed",
                                                               "idle_share", "residual_evals_per_substep_last_launch", "roofline")}
            if args.dtype == "f32":      # ... and the fp32 headline WITHOUT the one solver option it uses that the reference does not have
                bare_ = sub_record("push", "f32", dev, steps=20, warm=5, solver="bare", env_tables=False)
                res["f32_bare_xml_loop"] = {k: bare_[k] for k in ("value", "ms_per_step", "solver", "batch", "dtype", "nonconverged_envs", "nonconverged_substeps", "substeps_timed",
                                                                  "idle_share", "residual_evals_per_substep_last_launch", "roofline")}
        elif name == "env_tables":      # the headline workload with one parameter table per environment: must stay on compiled-in kernels (round 4: fell to the generic ones)
            res[name] = sub_record("push", args.dtype, dev, steps=20, warm=20, env_tables=True)
        elif name == "dclaw":
            res[name] = sub_record("dclaw", args.dtype, dev, pmc=pmc_on)
            # the collection figure BOTH ways: unbudgeted above; here with an evaluation budget per sub-step, and the fraction of sub-steps it cut (flagged in status)
            b_ = sub_record("dclaw", args.dtype, dev, eval_budget=DCLAW_EVAL_BUDGET)
            res[name]["value_budgeted"] = b_["value"]
            res[name]["flagged_frac_budgeted"] = b_["nonconverged_substeps"] / max(b_["substeps_timed"], 1)
            res[name]["budgeted"] = {"eval_budget": DCLAW_EVAL_BUDGET, "value": b_["value"], "nonconverged_envs": b_["nonconverged_envs"], "nonconverged_substeps": b_["nonconverged_substeps"],
                                     "substeps_timed": b_["substeps_timed"], "idle_share": b_["idle_share"], "kernel_ms": b_["roofline"]["kernel_ms"]}
        else:
            res[name] = sub_record(name, args.dtype, dev, pmc=pmc_on and name == "insertion")
        if name in WHOLE_CONFIG_BATCH and "value" in res.get(name, {}):
            # the BASELINE config's WHOLE batch on this one GPU (its 8-GPU job in 288 GB): a launch lasts its slowest environment's chain whatever
            # the batch is, so the larger batch amortises that chain over more wavefronts taken in turn by each SIMD
            w_ = sub_record(name, args.dtype, dev, batch=WHOLE_CONFIG_BATCH[name])
            res[name]["value_whole_config"] = w_["value"]
            res[name]["whole_config_on_one_gpu"] = {"batch": w_["batch"], "value": w_["value"], "ms_per_step": w_["ms_per_step"], "nonconverged_envs": w_["nonconverged_envs"],
                                                    "nonconverged_substeps": w_["nonconverged_substeps"], "substeps_timed": w_["substeps_timed"], "idle_share": w_["idle_share"],
                                                    "kernel_ms": w_["roofline"]["kernel_ms"], "launch_shape": w_["launch_shape"]}
    elif name == "closed_loop":
        if not push or ctx["forward_only"]:
            res[name] = {"skipped": "closed loop belongs to the push fwd+adjoint headline"}
            return
        res["closed_loop"] = closed_loop_fused_leg(model, B, T, tdt, dev)
        res["closed_loop_per_step_graph"] = closed_loop_leg(model, B, T, tdt, dev)
        res["closed_loop_cnn_per_step_graph"] = closed_loop_leg(model, B, T, tdt, dev, epochs=2, cnn=True)
    elif name == "readout":
        res["readout"] = readout_legs(tdt, dev, args.dtype, pmc="pmc" in args.leg_list)
    elif name == "cpu":
        res["cpu_baseline"] = cpu_baseline(args.workload, model, ctx["S"], not ctx["forward_only"])
    else:
        raise ValueError("unknown leg " + name)
