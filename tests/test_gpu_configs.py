"""BASELINE.json configs at their stated sizes, through the C ABI, against the fp64 oracle.

* config 3 — TactilePush gd_tactile fwd + adjoint, B = 4096 fp32: the launch shape the bench times (4 environments per
  wavefront, `k_forward<float, 8, false, 16>`), a 64-environment subset OF THAT BATCH against the oracle (q, qd, variables,
  tactile, 100-step episode gradients), and properties on all 4096 rows (converged, identical environments give identical
  rows, a row does not depend on its position in the batch).
* config 2 — B = 1024 forward-only, 100 env-steps.
* the "1e-4" gradient statement at scale (SURVEY.md §7 "Non-smoothness"): fp32 kernels against fp64 kernels on 1024
  environments x 100 env-steps, compared where the branch signatures (contact / stick-slip / face pattern of every
  sub-step, exported by kernel AND oracle) agree; the fraction of environments that crossed a kink is asserted too.

The inputs are the bench's own (`push_workload(B, 100, seed=0)`); the oracle runs one instance per host thread.
"""
import os
import threading

import numpy as np
import pytest
import torch

from tactilesimulation_amd.workloads import push_workload

pytestmark = pytest.mark.gpu
T, S = 100, 5
DEV = "cuda:0"


def _weights():
    rng = np.random.default_rng(4)
    return rng.normal(size=(T, 7)), rng.normal(size=(T, 6)), rng.normal(size=(T, 390)) * 10.0


def _oracle_subset(model, q0, u, idx, weights=None, want_sig=False, solver="literal"):
    """Oracle trajectories (and episode gradients / branch signatures) of the environments idx, on all host threads.
    solver: the oracle's Newton driver ("literal": the XML's own loop, which the kernels run; "r02": the legacy globalisation)."""
    from oracle.oracle import OracleSim
    n = len(idx)
    out = {"q": np.zeros((T, n, 7)), "qd": np.zeros((T, n, 7)), "var": np.zeros((T, n, 6)), "tac": np.zeros((T, n, 390)),
           "du": np.zeros((T, n, 6)), "sig": np.zeros((T * S, n, 2), dtype=np.int64)}
    nthr = max(1, min(len(os.sched_getaffinity(0)), 32, n))
    err = []

    def work(i):
        try:
            o = OracleSim(model, solver=solver)
            for j in range(i, n, nthr):
                e = idx[j]
                o.reset(q0[e], record=weights is not None)
                for t in range(T):
                    if want_sig:
                        bad, sg = o.forward_sig(u[e, t], S)
                        out["sig"][t * S:(t + 1) * S, j] = sg
                    else:
                        bad = o.forward(u[e, t], S)
                    assert bad == 0
                    out["q"][t, j], out["qd"][t, j] = o.state()
                    out["var"][t, j], out["tac"][t, j] = o.outputs()
                if weights is not None:
                    wq, wv, wt = weights
                    for t in reversed(range(T)):
                        dq = np.zeros((S, 7)); dq[-1] = wq[t]
                        dv = np.zeros((S, 6)); dv[-1] = wv[t]
                        dt_ = np.zeros((S, 390)); dt_[-1] = wt[t]
                        out["du"][t, j] = o.backward_steps(S, dq, dv, dt_).sum(0)
        except Exception as ex:      # surface worker failures in the test thread
            err.append(ex)
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    [t.start() for t in th]
    [t.join() for t in th]
    if err:
        raise err[0]
    return out


def _tile(w, B, dtype):
    return torch.tensor(np.broadcast_to(w[:, None, :], (T, B, w.shape[1])).copy(), device=DEV, dtype=dtype)


def test_config3_push_b4096_fwd_adjoint_fp32(pusher_model):
    from tactilesimulation_amd.host.batch import BatchSim
    B = 4096
    q0, u, _ = push_workload(B, T, seed=0)                     # exactly what bench.py feeds rank 0
    dup = {4095: 0, 2049: 1, 1027: 2, 517: 3}                  # identical environments in other wavefronts / other slots
    for d, s in dup.items():
        q0[d], u[d] = q0[s], u[s]
    weights = _weights()
    dt = torch.float32
    sim = BatchSim(pusher_model, B, dtype=dt, tape_capacity=T * S)
    info = sim.launch_info()
    assert os.environ.get("TSIM_LPE") or (info["lanes_per_env"] == 16 and info["blocks"] == 1024), info   # the instantiation bench.py times (TSIM_LPE: the whole suite under a forced shape)
    assert os.environ.get("TSIM_NO_STATIC") or (sim.static_model() == 1 and sim.kernel_variant() == "static:pusher")      # ... by name: the compiled-in TactilePush kernels, not the generic ones
    sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None, backward_flag=True)
    ud = torch.tensor(u, device=DEV, dtype=dt).transpose(0, 1).contiguous()
    ro = sim.rollout(ud, S, want_qd=True)
    sig = sim.branch_signature().cpu().numpy()
    du = sim.backward_episode(T, S, *(_tile(w, B, dt) for w in weights))
    assert sim.tape_len() == 0

    # ---- properties on all 4096 rows
    assert int((ro["status"] != 0).sum()) == 0
    for k in ("q", "qd", "var", "tactile"):
        assert bool(torch.isfinite(ro[k]).all()), k
    assert bool(torch.isfinite(du).all())
    for d, s in dup.items():
        for k in ("q", "qd", "var", "tactile"):
            assert torch.equal(ro[k][:, d], ro[k][:, s]), (k, d, s)
        assert torch.equal(du[:, d], du[:, s]), (d, s)
    perm = torch.arange(B - 1, -1, -1, device=DEV)             # the same environments in reverse batch order
    sim2 = BatchSim(pusher_model, B, dtype=dt, tape_capacity=T * S)
    sim2.reset(torch.tensor(q0, device=DEV, dtype=dt)[perm], None, backward_flag=True)
    ro2 = sim2.rollout(ud[:, perm].contiguous(), S, want_qd=True)
    du2 = sim2.backward_episode(T, S, *(_tile(w, B, dt) for w in weights))
    for k in ("q", "qd", "var", "tactile"):
        assert torch.equal(ro2[k][:, perm], ro[k]), k
    assert torch.equal(du2[:, perm], du)
    del sim2, ro2, du2

    # ---- 64 environments of this batch against the oracle
    idx = np.arange(0, B, 64)
    o = _oracle_subset(pusher_model, q0, u, idx, weights, want_sig=True)
    g = {k: ro[k][:, idx].double().cpu().numpy() for k in ("q", "qd", "var", "tactile")}
    assert np.abs(g["q"] - o["q"]).max() < 5e-6
    assert np.abs(g["qd"] - o["qd"]).max() < 2e-4 * max(np.abs(o["qd"]).max(), 1.0)
    assert np.abs(g["var"] - o["var"]).max() < 5e-6
    assert np.abs(g["tactile"] - o["tac"]).max() < 2e-4 * np.abs(o["tac"]).max()
    same = (sig[:, idx] == o["sig"]).all(axis=(0, 2))            # same contact / friction branches in all 500 sub-steps
    dg = du[:, idx].double().cpu().numpy()
    eg = np.abs(dg - o["du"]).max(axis=(0, 2)) / np.abs(o["du"]).max(axis=(0, 2))
    print("config 3, 64 environments of the 4096 batch vs the oracle: %d on the same branches throughout; gradient error median %.2e, max %.2e (on those: %.2e)"
          % (same.sum(), np.median(eg), eg.max(), eg[same].max()))
    assert same.sum() >= 63, "more than 1 of 64 environments crossed a kink: %d" % (64 - same.sum())     # measured: 64 of 64, max error 1.1e-5
    assert eg[same].max() < 1e-4, eg[same].max()                 # BASELINE.json: gradients within 1e-4 rel of the CPU path


@pytest.mark.parametrize("lanes", [0, 16])
def test_config2_push_b1024_forward_only_fp32(pusher_model, lanes):
    from tactilesimulation_amd.host.batch import BatchSim
    B = 1024
    q0, u, _ = push_workload(B, T, seed=0)
    dt = torch.float32
    sim = BatchSim(pusher_model, B, dtype=dt, tape_capacity=0)       # forward-only: no tape
    if lanes:
        sim.set_lanes_per_env(lanes)
        assert sim.launch_info()["lanes_per_env"] == lanes
    sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None, backward_flag=False)
    ro = sim.rollout(torch.tensor(u, device=DEV, dtype=dt).transpose(0, 1).contiguous(), S, want_qd=True)
    assert int((ro["status"] != 0).sum()) == 0
    idx = np.arange(0, B, 32)
    o = _oracle_subset(pusher_model, q0, u, idx)
    g = {k: ro[k][:, idx].double().cpu().numpy() for k in ("q", "qd", "var", "tactile")}
    assert np.abs(g["q"] - o["q"]).max() < 5e-6
    assert np.abs(g["qd"] - o["qd"]).max() < 2e-4 * max(np.abs(o["qd"]).max(), 1.0)
    assert np.abs(g["var"] - o["var"]).max() < 5e-6
    assert np.abs(g["tactile"] - o["tac"]).max() < 2e-4 * np.abs(o["tac"]).max()
    q, qd = sim.get_state()                                      # get_q / get_qdot after the episode
    assert torch.equal(q, ro["q"][-1]) and torch.equal(qd, ro["qd"][-1])


def test_branch_signature_kernel_equals_oracle(pusher_model):
    """The exported signature itself: fp64 kernels and the oracle report the same (count, hash) for every sub-step."""
    from tactilesimulation_amd.host.batch import BatchSim
    B = 48
    q0, u, _ = push_workload(B, T, seed=17)
    sim = BatchSim(pusher_model, B, dtype=torch.float64, tape_capacity=T * S)
    sim.reset(torch.tensor(q0, device=DEV), None, backward_flag=True)
    sim.rollout(torch.tensor(u, device=DEV).transpose(0, 1).contiguous(), S)
    sig = sim.branch_signature().cpu().numpy()
    o = _oracle_subset(pusher_model, q0, u, np.arange(B), want_sig=True)
    assert sig[:, :, 0].max() > 20                               # contacts and taxels do penetrate in this workload
    assert (sig == o["sig"]).mean() > 0.9995, (sig != o["sig"]).sum()     # equal but for knife-edge states (|d| ~ 1e-16)
    part = sim.branch_signature(t_first=10, n=5).cpu().numpy()
    assert (part == sig[10:15]).all()


def test_fp32_gradients_within_1e4_where_branches_agree_b1024(pusher_model):
    """1024 environments x 100 env-steps, fp32 kernels against fp64 kernels (which equal the oracle to round-off,
    test_full_episodes_against_the_oracle).  Asserted:
      * at most 0.5 % of the environments take another contact / friction branch somewhere in their 500 sub-steps;
      * on the environments whose branch signatures agree the 100-step episode gradients agree to 1e-4 (99.5 % of them;
        every one within 1e-3 — measured on the whole 4096 batch: p99 1.1e-5, 3 environments between 1e-4 and 2.6e-4,
        profiles/r02_branch_signature.json);
      * that residue is far below what the XML's own Newton tolerance (1e-8 on ||g||) does to the fp64 path itself: the
        fp64 kernels with tol 1e-12 differ from the fp64 kernels with tol 1e-8 by more (median) than fp32 from fp64 (p99)."""
    import copy
    import tactilesimulation_amd.model.blob as Bl
    from tactilesimulation_amd.host.batch import BatchSim
    B = 1024
    q0, u, _ = push_workload(B, T, seed=0)
    weights = _weights()
    tight = copy.copy(pusher_model)
    tight.F = pusher_model.F.copy()
    tight.F[Bl.TSIM_FH_TOL] = 1e-12
    res = {}
    for key, model, dt in (("f64", pusher_model, torch.float64), ("f32", pusher_model, torch.float32), ("f64_tight", tight, torch.float64)):
        sim = BatchSim(model, B, dtype=dt, tape_capacity=T * S)
        sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None, backward_flag=True)
        ro = sim.rollout(torch.tensor(u, device=DEV, dtype=dt).transpose(0, 1).contiguous(), S)
        assert int((ro["status"] != 0).sum()) == 0
        sig = sim.branch_signature().cpu().numpy()
        du = sim.backward_episode(T, S, *(_tile(w, B, dt) for w in weights)).double().cpu().numpy()
        res[key] = (sig, du, ro["q"].double().cpu().numpy())
        del sim
    (s64, g64, q64), (s32, g32, q32), (s_t, g_t, _) = res["f64"], res["f32"], res["f64_tight"]
    same = (s64 == s32).all(axis=(0, 2))
    flipped = float((~same).mean())
    eg = np.abs(g64 - g32).max(axis=(0, 2)) / np.abs(g64).max(axis=(0, 2))
    same_t = (s64 == s_t).all(axis=(0, 2))
    eg_t = np.abs(g64 - g_t).max(axis=(0, 2)) / np.abs(g_t).max(axis=(0, 2))
    print("fp32 vs fp64: flipped fraction %.4f; gradient error on agreeing environments: median %.2e p99 %.2e max %.2e; on flipped: "
          "max %.2e | fp64 tol 1e-8 vs 1e-12: flipped %.3f, median %.2e"
          % (flipped, np.median(eg[same]), np.percentile(eg[same], 99), eg[same].max(), eg[~same].max() if (~same).any() else 0.0,
             float((~same_t).mean()), np.median(eg_t[same_t])))
    # what the optimiser sees: the gradient summed over the batch (the GD loop's all-reduced policy gradient is a linear map of it)
    gsum64, gsum32 = g64.sum(axis=1), g32.sum(axis=1)
    batch_err = float(np.abs(gsum32 - gsum64).max() / np.abs(gsum64).max())
    print("batch-summed gradient, fp32 vs fp64 kernels, %d environments: %.2e" % (B, batch_err))
    assert batch_err <= 2e-5, batch_err                           # measured 1.9e-6 (round 2, with its own globalisation: 1.1e-4)
    assert flipped <= 0.002, flipped                              # measured 0 of 1024 (round 2: 0.07 %)
    assert np.percentile(eg[same], 99.5) < 1e-4, np.percentile(eg[same], 99.5)
    assert eg[same].max() < 1e-3, eg[same].max()
    assert np.abs(q64 - q32)[:, same].max() < 2e-5
    assert np.percentile(eg[same], 99) < np.median(eg_t[same_t])          # fp32 error sits below the solver-tolerance noise floor


# ------------------------------------------------------------------------------------------------ configs 4 and 5 at their per-GPU sizes
@pytest.mark.parametrize("name,B,T,n_oracle,tq,tt", [
    # BASELINE.json configs[3]: D'Claw, 16 384 environments over 8 GPUs = 2048 per GPU, forward-only (PPO collection); here 12 of the
    # config's 200 env-steps at the real batch size and launch shape
    ("dclaw_position_control", 2048, 12, 6, 2e-5, 2e-3),
    # configs[4]: TactileInsertion, 32 768 over 8 GPUs = 4096 per GPU, forward-only: the reference's episode, ONE insertion attempt of 45 single
    # sub-steps from the settled grasp moved by U(+-6 mm, +-6 mm, +-10 deg) (envs/tactile_insertion_env.py:200-216,344-359; workloads.py) — what
    # bench.py's `insertion` record times.  (Rounds 1 - 4 also kept a stand-in here that closed the grasp inside the episode and tolerated 1 % of
    # non-converged environments; it was the wrong episode and is gone.)
    ("tactile_insertion", 4096, 45, 4, 2e-5, 2e-3),
])
def test_config4_config5_per_gpu_share_forward_only_fp32(name, B, T, n_oracle, tq, tt):
    """The other two multi-GPU configurations at the batch one GPU gets: all environments converge, rows that start from
    identical inputs are bit-identical wherever they sit in the batch (duplicates planted at both ends and in the middle), the
    episode launch equals per-step launches bit for bit, and a subset of the batch matches the fp64 oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_models import _inputs
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.workloads import asset
    from oracle.oracle import OracleSim
    from tactilesimulation_amd.workloads import insertion_attempt_workload
    m = load_model(asset(name))
    S = 5
    if name == "tactile_insertion":
        q0, u = insertion_attempt_workload(B, seed=0)            # [B, 45, 6], one sub-step per frame
        S = 1
    else:
        q0, u = _inputs(name, m, B, T)
    dup = [(0, B - 1), (1, B // 2), (2, B - 7)]                    # (source, copy) environments
    for a, b in dup:
        q0[b], u[b] = q0[a], u[a]
    dt = torch.float32
    sim = BatchSim(m, B, dtype=dt, tape_capacity=0)
    Q0, U = torch.tensor(q0, device=DEV, dtype=dt), torch.tensor(u, device=DEV, dtype=dt).transpose(0, 1).contiguous()
    sim.reset(Q0, None, backward_flag=False)
    ro = sim.rollout(U, S, want_qd=True)
    # every environment of every config converges under the XML's Newton loop (+ kink crossing, the fp32 default)
    bad32 = (ro["status"] != 0)
    assert int(bad32.sum()) == 0, "%d environments did not converge" % int(bad32.sum())
    assert float(ro["q"].abs().max()) < 4.0 and float(ro["qd"].abs().max()) < 200.0
    assert bool(torch.isfinite(ro["q"]).all()) and bool(torch.isfinite(ro["tactile"]).all())
    for a, b in dup:
        for k in ("q", "qd", "tactile"):
            assert torch.equal(ro[k][:, a], ro[k][:, b]), (k, a, b)
    # per-step launches of the same batch: the same bits
    sim.reset(Q0, None, backward_flag=False)
    for t in range(min(T, 6)):
        o_ = sim.step(U[t], S, want_qd=True)
        assert torch.equal(o_["q"], ro["q"][t]) and torch.equal(o_["tactile"], ro["tactile"][t]), t
    # oracle on a subset of THIS batch
    idx = np.linspace(3, B - 11, n_oracle).astype(int)
    for e in idx:
        o = OracleSim(m); o.reset(q0[e])
        for t in range(T):
            assert o.forward(u[e, t], S) == 0
            q, _ = o.state()
            _, tac = o.outputs()
            gq, gt = ro["q"][t, e].double().cpu().numpy(), ro["tactile"][t, e].double().cpu().numpy()
            assert np.abs(gq - q).max() < tq * max(1.0, np.abs(q).max()), (e, t, np.abs(gq - q).max())
            assert np.abs(gt - tac).max() < tt * max(np.abs(tac).max(), 1e-3), (e, t)


def test_config3_reward_loss_gradient_as_survey_words_it(pusher_model):
    """SURVEY.md §8d row 3 literally: L = sum_t [reward_pos + reward_rot + reward_touch] (envs/tactile_push_env.py:206-208) with the goals
    of the workload, gradient w.r.t. u[B, 100, 6]; fp32 kernels at B = 4096 against the fp64 oracle on a 64-environment subset of the
    batch, by the survey's metric max_i |g_gpu - g_cpu| / max(|g_cpu|, 1e-12 |g_cpu|_inf) (per environment, scaled by the environment's
    largest gradient entry) and by the reference's own pair, relative error and cosine (algorithms/gd.py:459-465)."""
    import math
    from tactilesimulation_amd.host.batch import BatchSim
    from oracle.oracle import OracleSim
    B = 4096
    q0, u, goal = push_workload(B, T, seed=0)
    dt = torch.float32
    sim = BatchSim(pusher_model, B, dtype=dt, tape_capacity=T * S)
    sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None, backward_flag=True)
    ro = sim.rollout(torch.tensor(u, device=DEV, dtype=dt).transpose(0, 1).contiguous(), S)
    assert int((ro["status"] != 0).sum()) == 0

    def seeds(q, var, g):                                       # dL/dq [.., 7], dL/dvar [.., 6] of the three reward terms
        dq = np.zeros(q.shape); dv = np.zeros(var.shape)
        dq[..., 3:5] = -200.0 * (q[..., 3:5] - g[..., 0:2])     # -0.01 * 2 (p - g) / 0.01^2
        dq[..., 6] = -0.2 * (q[..., 6] - g[..., 2]) / (math.pi / 36.0) ** 2
        d = var[..., 0:3] - var[..., 3:6]
        dv[..., 0:3], dv[..., 3:6] = -2.0 * d / 0.02 ** 2, 2.0 * d / 0.02 ** 2
        return dq, dv
    q_g, v_g = ro["q"].double().cpu().numpy(), ro["var"].double().cpu().numpy()
    dq, dv = seeds(q_g, v_g, goal[None])
    du = sim.backward_episode(T, S, torch.tensor(dq, device=DEV, dtype=dt), torch.tensor(dv, device=DEV, dtype=dt), None).double().cpu().numpy()
    idx = np.arange(5, B, 64)
    G = np.zeros((T, len(idx), 6)); L_o = np.zeros(len(idx)); sig_o = np.zeros((T * S, len(idx), 2), dtype=np.int64)
    nthr = max(1, min(len(os.sched_getaffinity(0)), 32, len(idx)))

    def work(i):
        o = OracleSim(pusher_model)
        for j in range(i, len(idx), nthr):
            e = idx[j]
            o.reset(q0[e], record=True)
            qs, vs = np.zeros((T, 7)), np.zeros((T, 6))
            for t in range(T):
                bad, sg = o.forward_sig(u[e, t], S)
                assert bad == 0
                sig_o[t * S:(t + 1) * S, j] = sg
                qs[t], vs[t] = o.state()[0], o.outputs()[0]
            sq, sv = seeds(qs, vs, goal[e][None])
            for t in reversed(range(T)):
                a = np.zeros((S, 7)); a[-1] = sq[t]
                b = np.zeros((S, 6)); b[-1] = sv[t]
                G[t, j] = o.backward_steps(S, a, b, np.zeros((S, 390))).sum(0)
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    [t.start() for t in th]
    [t.join() for t in th]
    g = du[:, idx]
    scale = np.abs(G).max(axis=(0, 2), keepdims=True)
    survey = (np.abs(g - G) / np.maximum(scale, 1e-12 * scale.max())).max(axis=(0, 2))       # per environment
    rel = np.linalg.norm((g - G).reshape(T, len(idx), -1).transpose(1, 0, 2).reshape(len(idx), -1), axis=1) / np.linalg.norm(G.transpose(1, 0, 2).reshape(len(idx), -1), axis=1)
    cos = np.array([float(g[:, j].reshape(-1) @ G[:, j].reshape(-1) / (np.linalg.norm(g[:, j]) * np.linalg.norm(G[:, j]))) for j in range(len(idx))])
    ok = survey < 1e-4
    print("config 3 reward loss: %d of %d environments within 1e-4 by the survey's metric; median %.2e, max %.2e; gd.py relative error median %.2e"
          % (ok.sum(), len(idx), np.median(survey), survey.max(), np.median(rel)))
    assert ok.sum() >= len(idx) - 1, (np.sort(survey)[-6:], "environments off: %d" % (~ok).sum())     # the few that crossed a kink (§5)
    assert np.median(survey) < 2e-5 and np.median(rel) < 2e-5 and cos[ok].min() > 1.0 - 1e-8, (np.median(survey), np.median(rel), cos[ok].min())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
def test_adjoint_properties_at_full_size(pusher_model, dtype, tol):
    """Size-independent properties of the adjoint launch on the bench's own batch (B = 4096, 100 env-steps): it is LINEAR in the seeds,
    CAUSAL (an action after the last seeded frame gets exactly zero), DETERMINISTIC (same tape, same seeds: same bits), and independent of
    what else is in the batch (the first 1024 environments alone give the same rows)."""
    from tactilesimulation_amd.host.batch import BatchSim
    B = 4096
    q0, u, _ = push_workload(B, T, seed=0)
    q0t = torch.tensor(q0, device=DEV, dtype=dtype)
    ut = torch.tensor(u, device=DEV, dtype=dtype).transpose(0, 1).contiguous()
    g = torch.Generator(device="cpu").manual_seed(9)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(DEV, dtype)
    s1 = (rnd(T, B, 7), rnd(T, B, 6), rnd(T, B, 390) * 10.0)
    s2 = (rnd(T, B, 7), rnd(T, B, 6), rnd(T, B, 390) * 10.0)
    K = 60                                                          # s2 seeds frames 0..K-1 only
    for w in s2:
        w[K:] = 0.0
    a, b = 0.7, -1.3
    sim = BatchSim(pusher_model, B, dtype=dtype, tape_capacity=T * S)

    def grad(seeds, sim_=sim, n=B):
        sim_.reset(q0t[:n], None, backward_flag=True)
        ro = sim_.rollout(ut[:, :n], S)
        assert int((ro["status"] != 0).sum()) == 0
        return sim_.backward_episode(T, S, *(w[:, :n].contiguous() for w in seeds))
    d1, d2 = grad(s1), grad(s2)
    d12 = grad(tuple(a * x + b * y for x, y in zip(s1, s2)))
    scale = float((a * d1 + b * d2).abs().max())
    assert float((d12 - (a * d1 + b * d2)).abs().max()) <= tol * scale                      # linear
    assert float(d2[K:].abs().max()) == 0.0 and float(d2[:K].abs().max()) > 0.0             # causal
    assert torch.equal(grad(s1), d1)                                                        # deterministic
    small = BatchSim(pusher_model, 1024, dtype=dtype, tape_capacity=T * S)
    small.set_lanes_per_env(sim.launch_info()["lanes_per_env"])                             # the same launch shape: the same summation orders
    if dtype == torch.float64 and os.environ.get("TSIM_LPE") == "16":
        # the suite under a forced 16-lane shape: `sim` (forced by the environment) runs the generic fp64 kernels, `small` (forced to the fall-back
        # shape by the line above) the compiled-in ones — two kernels, equal to round-off, not to the bit
        assert torch.allclose(grad(s1, small, 1024), d1[:, :1024], rtol=0, atol=1e-7 * float(d1.abs().max()))
        return
    assert torch.equal(grad(s1, small, 1024), d1[:, :1024])                                 # batch-composition independent
