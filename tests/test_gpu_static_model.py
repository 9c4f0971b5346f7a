"""The statically specialised TactilePush kernels (csrc/tsim_static.h: the model's tree, joint types, joint frames and axes as compile-time
constants, the link sweep and the contact loops folded to what that structure leaves) against the generic kernels on the same inputs: folding a
zero out of a dot product is exact but lets the compiler contract the remaining products the other way round, so the two are fp32 roundings of the
same arithmetic — outputs to 1e-5, the same Newton work in all but a handful of environments, per-environment gradients to 1e-4 — and the
specialisation switches itself off for any batch whose blob is not the compiled-in one."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

import tactilesimulation_amd.model.blob as BL
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.workloads import push_workload

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


from test_gpu_param_model import _run, _check_outliers, _oracle_gradient      # noqa: E402  (the run + the outlier-vs-oracle check are shared)


def test_static_pusher_kernels_agree_with_the_generic_ones_to_fp32_rounding(pusher_model):
    B, T, S = 4096, 12, 5
    q0, u, _ = push_workload(B, T, seed=5)
    g = torch.Generator().manual_seed(2)
    wq, wv, wt = (torch.randn(T, B, n, generator=g).to(DEV) for n in (7, 6, 390))
    a = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=T * S)
    b = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=T * S)
    b.set_static(False)
    assert a.static_model() == 1 and b.static_model() == 0
    assert os.environ.get("TSIM_LPE") or a.launch_info()["lanes_per_env"] == 16      # the shape bench.py times (TSIM_LPE: the whole suite under a forced shape)
    ra, rb = _run(a, q0, u, T, S, wq, wv, wt), _run(b, q0, u, T, S, wq, wv, wt)
    # Folding a structural zero out of  a0 b0 + a1 b1 + a2 b2  is exact, but the compiler is then free to contract the two products that remain
    # the other way round (fma(a0, b0, a1 b1) or fma(a1, b1, a0 b0)): the two kernels are fp32 roundings of the same arithmetic, not the same
    # bits.  Measured at B = 4096 over 60 sub-steps: q 1e-6 - 2.4e-6, tactile 9e-6 of its maximum, the same Newton work in 99.9 % of the environments,
    # episode gradients per environment: median 1e-6 (profiles/r04_static_model.md).
    rel = lambda x, y: float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
    assert torch.equal(ra[0]["status"], rb[0]["status"]) and int(ra[0]["status"].abs().max()) == 0
    assert float((ra[0]["q"] - rb[0]["q"]).abs().max()) < 5e-6 and float((ra[0]["qd"] - rb[0]["qd"]).abs().max()) < 2e-4
    assert rel(ra[0]["var"], rb[0]["var"]) < 1e-5
    # tactile forces: both are fp32 roundings of the fp64 kernels' values, which are the third party here — per environment, relative to the
    # batch's largest taxel force: 99.9 % of the environments within 4e-5 of fp64 on either path, every environment within 2e-4 (measured at 16 lanes per environment: p999 1.4e-5,
    # max static 6.3e-5, generic 1.2e-4 in ONE environment of 4096 whose Newton iteration takes one evaluation less than in fp64; profiles/r05_static_vs_generic.json)
    d = BatchSim(pusher_model, B, dtype=torch.float64, tape_capacity=0)
    d.set_static(False)                                                  # the third party: the generic fp64 kernels
    d.reset(torch.tensor(q0, device=DEV, dtype=torch.float64), None, backward_flag=False)
    rd = d.rollout(torch.tensor(u, device=DEV, dtype=torch.float64).transpose(0, 1).contiguous(), S)
    tmax = float(rd["tactile"].abs().max())
    for name, r in (("static", ra), ("generic", rb)):
        e = (r[0]["tactile"].double() - rd["tactile"]).abs().amax(dim=(0, 2)) / tmax
        assert float(e.max()) < 2e-4 and float(torch.quantile(e, 0.999)) < 4e-5, (name, float(e.max()), float(torch.quantile(e, 0.999)))
    assert rel(ra[0]["tactile"], rb[0]["tactile"]) < 3e-4
    assert float(ra[0]["tactile"].abs().max()) > 0
    assert (ra[1] == rb[1]).mean() > 0.99 and abs(int(ra[1].sum()) - int(rb[1].sum())) < 1e-3 * rb[1].sum()      # Newton work
    # gradients, environment by environment (du [T, B, nu], adjoints [B, nr]): two fp32 roundings of one trajectory agree to ~1e-6 unless it
    # crosses a contact / friction kink on different sides — the gradient is discontinuous there (fp32 vs fp64 kernels, 3 of 4096)
    def per_env(x, y):
        x, y = (t.transpose(0, 1).reshape(B, -1) if t.dim() == 3 else t for t in (x, y))
        return ((x - y).abs().max(1).values / y.abs().max(1).values.clamp_min(1e-30)).cpu().numpy()
    errs = {name: per_env(x, y) for x, y, name in ((ra[2], rb[2], "du"), (ra[3], rb[3], "lamq"), (ra[4], rb[4], "lamv"))}
    for name, e in errs.items():
        assert np.median(e) < 1e-5 and (e < 1e-4).mean() > 0.995 and (e > 1e-2).sum() <= 2 and e.max() < 0.2, (name, float(np.median(e)), float((e < 1e-4).mean()), int((e > 1e-2).sum()), float(e.max()))
    # every outlier environment: different smooth pieces in the two runs, and whichever run is on the ORACLE's pieces has the oracle's gradient (1e-4)
    outl = sorted(set(int(i) for e in errs.values() for i in np.nonzero(e > 1e-2)[0]))
    _check_outliers(outl, ra, rb, "static vs generic", lambda e: _oracle_gradient(pusher_model, q0, u, T, S, wq, wv, wt, e))
    from _report import rep
    rep("static_vs_generic", q=float((ra[0]["q"] - rb[0]["q"]).abs().max()), qd=float((ra[0]["qd"] - rb[0]["qd"]).abs().max()), tactile=rel(ra[0]["tactile"], rb[0]["tactile"]),
        du_median=float(np.median(errs["du"])), du_p999=float(np.quantile(errs["du"], 0.999)), du_max=float(errs["du"].max()), du_within_1e4=float((errs["du"] < 1e-4).mean()),
        lamq_median=float(np.median(errs["lamq"])), lamq_max=float(errs["lamq"].max()), same_evals=float((ra[1] == rb[1]).mean()))


def test_one_static_evaluation_against_the_generic_and_the_fp64_one(pusher_model):
    """g and H of ONE residual evaluation (tsim_debug_eval) at 512 states of the workload, a third of them with the pad on the box: the fused
    static evaluation (tsim_static_eval.h) and the generic three-phase one are both fp32 roundings of what the fp64 kernel computes, equally close."""
    B, T, S = 512, 8, 5
    q0, u, _ = push_workload(B, T, seed=11)
    d = BatchSim(pusher_model, B, dtype=torch.float64, tape_capacity=0)
    d.set_static(False)                                                  # the generic fp64 kernels (debug_eval has no static fp64 instantiation anyway)
    d.reset(torch.tensor(q0, device=DEV, dtype=torch.float64), None, backward_flag=False)
    ro = d.rollout(torch.tensor(u, device=DEV, dtype=torch.float64).transpose(0, 1).contiguous(), S, want_qd=True)
    assert float((ro["tactile"][-1].abs().sum(1) > 0).float().mean()) > 0.05          # some environments are in contact at the probe state
    q, qd = ro["q"][-1], ro["qd"][-1]
    h = float(pusher_model.h)
    q1 = q + h * qd + 1e-4 * torch.randn(B, 7, generator=torch.Generator().manual_seed(3), dtype=torch.float64).to(DEV)
    uu = torch.tensor(u[:, -1], device=DEV, dtype=torch.float64)
    gd, Hd = d.debug_eval(q1, q, qd, uu)
    a = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=0)
    b = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=0)
    b.set_static(False); a.set_lanes_per_env(16); b.set_lanes_per_env(16)
    assert a.static_model() == 1 and b.static_model() == 0
    f = lambda x: x.float()
    (ga, Ha), (gb, Hb) = (s_.debug_eval(f(q1), f(q), f(qd), f(uu)) for s_ in (a, b))
    def err(x, y):      # per environment, relative to the environment's largest entry
        x, y = x.double().reshape(B, -1), y.reshape(B, -1)
        return ((x - y).abs().max(1).values / y.abs().max(1).values.clamp_min(1e-30)).cpu().numpy()
    eHa, eHb, ega, egb = err(Ha, Hd), err(Hb, Hd), err(ga, gd), err(gb, gd)
    # H entries: stiffness x h^2 of penalty contacts (relative rounding of fp32 ~1e-7 per operation, a few hundred operations deep)
    assert np.median(eHa) < 2e-6 and np.median(eHb) < 2e-6 and eHa.max() < 1e-3 and eHb.max() < 1e-3, (np.median(eHa), np.median(eHb), eHa.max(), eHb.max())
    assert np.median(ega) < 1e-5 and np.median(egb) < 1e-5 and ega.max() < 1e-2 and egb.max() < 1e-2, (np.median(ega), np.median(egb), ega.max(), egb.max())
    assert np.median(eHa) < 3 * np.median(eHb) + 1e-7 and np.median(ega) < 3 * np.median(egb) + 1e-7
    from _report import rep
    rep("static_eval_vs_f64", H_static_median=float(np.median(eHa)), H_generic_median=float(np.median(eHb)), H_static_max=float(eHa.max()), H_generic_max=float(eHb.max()),
        g_static_median=float(np.median(ega)), g_generic_median=float(np.median(egb)), g_static_max=float(ega.max()), g_generic_max=float(egb.max()))


def test_the_taxel_layout_is_not_part_of_the_static_model():
    """BASELINE configs[1] words the TactilePush pad as 13 x 13 taxels (the XML has 13 x 10: workloads.synthetic_variant).  Taxels take no part
    in the dynamics and the static kernels read none of the layout, so that model runs on the same instantiation — and agrees with the generic
    kernels like the XML's own model does (forward, tactile frames of the re-gridded pad, episode gradient with tactile seeds)."""
    from tactilesimulation_amd.workloads import synthetic_variant
    m = synthetic_variant("pusher_13x13")
    assert m.ndof_tactile == 3 * 169
    B, T, S = 1024, 8, 5
    q0, u, _ = push_workload(B, T, seed=6)
    u[:, :, 0] = np.abs(u[:, :, 0])
    g = torch.Generator().manual_seed(5)
    wq, wv, wt = (torch.randn(T, B, n, generator=g).to(DEV) for n in (7, 6, 507))
    a = BatchSim(m, B, dtype=torch.float32, tape_capacity=T * S)
    b = BatchSim(m, B, dtype=torch.float32, tape_capacity=T * S)
    b.set_static(False)
    assert a.static_model() == 1 and b.static_model() == 0
    ra, rb = _run(a, q0, u, T, S, wq, wv, wt), _run(b, q0, u, T, S, wq, wv, wt)
    rel = lambda x, y: float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
    assert int(ra[0]["status"].abs().max()) == 0 and float(ra[0]["tactile"].abs().max()) > 0 and tuple(ra[0]["tactile"].shape) == (T, B, 507)
    # (tactile forces are penalty stiffness x a penetration depth of ~1e-4 m: a position difference of 1e-6 between the two fp32 roundings shows
    # as 1e-4 ... 2e-4 of the frame's largest force on a pad that is pressed on the box the whole episode)
    assert float((ra[0]["q"] - rb[0]["q"]).abs().max()) < 5e-6 and rel(ra[0]["tactile"], rb[0]["tactile"]) < 5e-4
    e = ((ra[2] - rb[2]).abs().amax((0, 2)) / rb[2].abs().amax((0, 2)).clamp_min(1e-30)).cpu().numpy()
    assert np.median(e) < 1e-5 and (e < 1e-4).mean() > 0.99, (float(np.median(e)), float((e < 1e-4).mean()))


def test_static_kernels_switch_off_for_other_blobs_shapes_and_tables(pusher_model):
    """The FULLY static instantiation needs the compiled-in blob bit for bit; a batch that keeps the model's structure moves to the
    structure-static one (round 5, tests/test_gpu_param_model.py), everything else to the generic kernels — and kernel_variant() says which."""
    B = 4096
    sim = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=0)
    assert sim.static_model() == 1 and sim.kernel_variant() == "static:pusher"
    sim.set_lanes_per_env(32)
    assert sim.kernel_variant() == "static:pusher"                      # every launch shape has its static instantiation
    sim.set_lanes_per_env(0)
    sim.set_env_tables(sim.base_tables())
    assert sim.static_model() == 1 and sim.kernel_variant() == "param:pusher"      # per-environment tables: parameters at run time
    sim.set_env_tables(None)
    assert sim.kernel_variant() == "static:pusher"
    m = copy.copy(pusher_model); m.F = pusher_model.F.copy()
    m.F[m.I[BL.TSIM_IH_FOFF_PAIR] + BL.TSIM_PF_KN] *= 1.5                # one float record edited: no longer the compiled-in model, still its structure
    sim.update_model(m)
    assert sim.kernel_variant() == "param:pusher"
    m2 = copy.copy(pusher_model); m2.F = pusher_model.F.copy()
    m2.F[m2.I[BL.TSIM_IH_FOFF_LINK] + BL.TSIM_LF_R + 1] = 0.01            # a joint frame that is no identity any more: another structure
    sim.update_model(m2)
    assert sim.static_model() == 0 and sim.kernel_variant() == "generic"
    sim.update_model(pusher_model)
    assert sim.static_model() == 1 and sim.kernel_variant() == "static:pusher"
    sim.set_static(False)
    assert sim.static_model() == 0 and sim.kernel_variant() == "generic"
    sim.update_model(pusher_model)
    assert sim.kernel_variant() == "generic"                            # tsim_set_static(0) survives a model update
    d = BatchSim(pusher_model, B, dtype=torch.float64, tape_capacity=0)      # fp64 (round 5): the same two instantiations, two or one environments per wavefront
    if os.environ.get("TSIM_LPE") == "16":      # (the suite under a forced 16-lane shape: fp64 has no compiled-in instantiation there, the generic kernels run)
        assert d.static_model() == 0 and d.kernel_variant() == "generic"
        return
    assert d.static_model() == 1 and d.kernel_variant() == "static:pusher" and d.launch_info()["lanes_per_env"] in (32, 64)
    d.update_model(m)
    assert d.kernel_variant() == "param:pusher"
    d.set_env_tables(d.base_tables())
    assert d.kernel_variant() == "generic"                               # (the per-environment table check is fp32 only)
    assert BatchSim(pusher_model, 64, dtype=torch.float32, tape_capacity=0).static_model() == 1      # small batch: one environment per wavefront, static too


@pytest.mark.parametrize("edited", [False, True])
def test_fp64_static_kernels_against_the_generic_fp64_ones(pusher_model, edited):
    """The fp64 instantiations of the compiled-in TactilePush kernels (fully static; structure-static on an edited model) against the generic fp64
    kernels — the ones tests/test_gpu_literal.py walks the oracle's iterates with: the same arithmetic up to the order of a few sums, so states to
    1e-11, tactile forces to 1e-9 of their maximum, the same Newton work in all but a handful of environments, episode gradients to 1e-7."""
    if os.environ.get("TSIM_LPE") == "16":
        pytest.skip("fp64 batches forced to 16 lanes per environment run the generic kernels: nothing to compare")
    B, T, S = 1024, 12, 5
    m = pusher_model
    if edited:
        m = copy.copy(pusher_model); m.F = pusher_model.F.copy()
        m.F[m.I[BL.TSIM_IH_FOFF_PAIR] + BL.TSIM_PF_KN] *= 1.5
        m.F[m.I[BL.TSIM_IH_FOFF_DOF] + BL.TSIM_DF_DAMPING] = 0.7
    q0, u, _ = push_workload(B, T, seed=5)
    g = torch.Generator().manual_seed(2)
    wq, wv, wt = (torch.randn(T, B, n, generator=g, dtype=torch.float64).to(DEV) for n in (7, 6, 390))

    def run(static):
        sim = BatchSim(m, B, dtype=torch.float64, tape_capacity=T * S)
        sim.set_static(static)
        assert sim.kernel_variant() == (("param:pusher" if edited else "static:pusher") if static else "generic")
        sim.reset(torch.tensor(q0, device=DEV, dtype=torch.float64), None, backward_flag=True)
        ro = sim.rollout(torch.tensor(u, device=DEV, dtype=torch.float64).transpose(0, 1).contiguous(), S, want_qd=True)
        ev = sim.last_evals().copy()
        du = sim.backward_episode(T, S, wq, wv, wt)
        lq, lv = sim.get_adjoint()
        return ro, ev, du, lq, lv
    a, b = run(True), run(False)
    assert torch.equal(a[0]["status"], b[0]["status"])
    assert float((a[0]["q"] - b[0]["q"]).abs().max()) < 1e-11 and float((a[0]["qd"] - b[0]["qd"]).abs().max()) < 1e-8
    tmax = float(b[0]["tactile"].abs().max())
    assert tmax > 0 and float((a[0]["tactile"] - b[0]["tactile"]).abs().max()) < 1e-9 * tmax
    assert (a[1] == b[1]).mean() > 0.99
    for x, y, name in ((a[2], b[2], "du"), (a[3], b[3], "lamq"), (a[4], b[4], "lamv")):
        x, y = (t.transpose(0, 1).reshape(B, -1) if t.dim() == 3 else t for t in (x, y))
        e = ((x - y).abs().max(1).values / y.abs().max(1).values.clamp_min(1e-30)).cpu().numpy()
        assert np.median(e) < 1e-10 and (e < 1e-7).mean() > 0.995, (name, float(np.median(e)), float((e < 1e-7).mean()), float(e.max()))


def test_a_blob_that_differs_below_float_resolution_is_the_compiled_in_model_to_an_fp32_batch(pusher_model):
    """Last-bit differences of the doubles (another host's BLAS in the Python compiler; the native loader, include/tsim_model.h, on the mesh-derived
    mass properties of pusher.xml) do not reach an fp32 kernel: the batch stays on the fully static instantiation and not one output bit moves.
    An fp64 batch sees them and takes the structure-static kernels."""
    import copy
    m = copy.deepcopy(pusher_model)
    fl = int(m.I[BL.TSIM_IH_FOFF_LINK])
    for k in (BL.TSIM_LF_MASS, BL.TSIM_LF_COM, BL.TSIM_LF_INERTIA, BL.TSIM_LF_INERTIA + 3):
        i = fl + BL.TSIM_LF_SIZE + k
        m.F[i] = np.nextafter(m.F[i], np.inf)
    assert (m.F != pusher_model.F).sum() == 4 and np.array_equal(m.F.astype(np.float32), pusher_model.F.astype(np.float32))
    B, T, S = 256, 6, 5
    q0, u, _ = push_workload(B, T, seed=3)
    outs = []
    for mod in (pusher_model, m):
        sim = BatchSim(mod, B, dtype=torch.float32, tape_capacity=T * S)
        assert sim.kernel_variant() == "static:pusher"
        sim.reset(torch.tensor(q0, device=DEV, dtype=torch.float32), None, backward_flag=True)
        o = sim.rollout(torch.tensor(u, device=DEV, dtype=torch.float32).transpose(0, 1).contiguous(), S)
        g = sim.backward_episode(T, S, df_dq=torch.ones(T, B, 7, device=DEV))
        outs.append((o["q"], o["tactile"], g[0] if isinstance(g, (tuple, list)) else g))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    f16 = os.environ.get("TSIM_LPE") == "16"      # (forced 16 lanes: fp64 runs the generic kernels)
    assert BatchSim(m, 8, dtype=torch.float64, tape_capacity=8).kernel_variant() == ("generic" if f16 else "param:pusher")
    assert BatchSim(pusher_model, 8, dtype=torch.float64, tape_capacity=8).kernel_variant() == ("generic" if f16 else "static:pusher")
