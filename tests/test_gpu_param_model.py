"""Structure-static kernels (csrc/tsim_static.h ts_F, csrc/tsim_param_pusher.hip): the TactilePush instantiation with only the model's STRUCTURE
compiled in — tree, joint types, contact pairs, and the exact 0 / 1 / -1 entries of the float records — and every other parameter read from the
batch's records.  It must (a) be what a batch runs after an `update_*` edit and with per-environment tables (the reference's domain
randomisation: envs/tactile_insertion_env.py:238-281, envs/dclaw_rotate_env.py:169-178 use exactly these two routes), (b) agree with the generic
kernels on the edited / randomised models as the fully static instantiation does on the XML's (two fp32 roundings of one arithmetic), and (c)
step aside for a model whose structure differs."""
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


def _run(sim, q0, u, T, S, wq, wv, wt):
    sim.reset(torch.tensor(q0, device=DEV, dtype=torch.float32), None, backward_flag=True)
    ro = sim.rollout(torch.tensor(u, device=DEV, dtype=torch.float32).transpose(0, 1).contiguous(), S, want_qd=True)
    ev = sim.last_evals().copy()
    sig = sim.branch_signature().cpu().numpy()          # [T S, B, 2] smooth pieces of the contact / friction law each sub-step was on (read before the tape is popped)
    du = sim.backward_episode(T, S, wq, wv, wt)
    lq, lv = sim.get_adjoint()
    return ro, ev, du, lq, lv, sig


MAX_OUTLIERS = 2        # environments whose gradients differ by more than 1e-2 between the two kernels (tests/test_gpu_static_model.py caps them the same way)


def _oracle_gradient(model, q0, u, T, S, wq, wv, wt, e):
    """dL/du [T, nu] and the branch signature [T S, 2] of environment e from the fp64 CPU oracle (the checker: oracle/), same seeds."""
    from oracle.oracle import OracleSim
    o = OracleSim(model)
    o.reset(q0[e], record=True)
    sig = np.zeros((T * S, 2), dtype=np.int64)
    for t in range(T):
        _, sg = o.forward_sig(u[e, t], S)
        sig[t * S:(t + 1) * S] = sg
    G = np.zeros((T, u.shape[2]))
    for t in reversed(range(T)):
        dq = np.zeros((S, q0.shape[1])); dq[-1] = wq[t, e].double().cpu().numpy()
        dv = np.zeros((S, wv.shape[2])); dv[-1] = wv[t, e].double().cpu().numpy()
        dt = np.zeros((S, wt.shape[2])); dt[-1] = wt[t, e].double().cpu().numpy()
        G[t] = o.backward_steps(S, dq, dv, dt).sum(0)
    return G, sig


def _check_outliers(outliers, ra, rb, tag, oracle_of):
    """Two fp32 roundings of one trajectory may cross a contact / friction kink on different sides; the gradient is discontinuous there, so a large
    per-environment difference is legitimate ONLY if the two runs went through different smooth pieces.  VERDICT r05 #7: a wrong adjoint term in a
    rare branch must not hide under `e.max() < 0.2` — so (1) at most MAX_OUTLIERS environments, (2) an outlier whose two runs report the SAME
    branch signature is an error outright, (3) every outlier is taken to the fp64 oracle (the reference's own criterion is analytic-vs-third-party:
    algorithms/gd.py:459-465): whichever run shares the oracle's signature must have the oracle's gradient to 1e-4."""
    assert len(outliers) <= MAX_OUTLIERS, (tag, "environments with gradient differences > 1e-2", list(outliers))
    for e in outliers:
        sa, sb = ra[5][:, e], rb[5][:, e]
        assert not (sa == sb).all(), (tag, "environment %d: same smooth pieces in every sub-step, yet the gradients of the two kernels differ by > 1e-2" % e)
        G, so = oracle_of(int(e))
        n_on_oracle_branch = 0
        for name, r in (("a", ra), ("b", rb)):
            if (r[5][:, e] == so).all():
                n_on_oracle_branch += 1
                g = r[2][:, e].double().cpu().numpy()
                err = np.abs(g - G).max() / max(np.abs(G).max(), 1e-30)
                assert err < 1e-4, (tag, "environment %d, run %s: on the oracle's branches but dL/du differs from the oracle's by %.2e" % (e, name, err))
        print("%s: outlier environment %d explained — the runs differ in %d of %d sub-step signatures; %d of the two on the oracle's branches (checked to 1e-4)"
              % (tag, e, int((sa != sb).any(axis=1).sum()), sa.shape[0], n_on_oracle_branch))


def _agree(ra, rb, B, tag, allow_bad=0, oracle_of=None):
    rel = lambda x, y: float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
    assert torch.equal(ra[0]["status"], rb[0]["status"]) and int((ra[0]["status"] != 0).sum()) <= allow_bad, tag
    assert float((ra[0]["q"] - rb[0]["q"]).abs().max()) < 5e-6 and float((ra[0]["qd"] - rb[0]["qd"]).abs().max()) < 5e-4, tag      # (2e-4 at the automatic shape; 4.6e-4 measured with four environments per wavefront forced)
    assert rel(ra[0]["var"], rb[0]["var"]) < 1e-5 and rel(ra[0]["tactile"], rb[0]["tactile"]) < 2e-4, tag
    assert float(ra[0]["tactile"].abs().max()) > 0, tag
    assert (ra[1] == rb[1]).mean() > 0.99, tag

    def per_env(x, y):
        x, y = (t.transpose(0, 1).reshape(B, -1) if t.dim() == 3 else t for t in (x, y))
        return ((x - y).abs().max(1).values / y.abs().max(1).values.clamp_min(1e-30)).cpu().numpy()
    outliers = set()
    for x, y, name in ((ra[2], rb[2], "du"), (ra[3], rb[3], "lamq"), (ra[4], rb[4], "lamv")):
        e = per_env(x, y)
        assert np.median(e) < 1e-5 and (e < 1e-4).mean() > 0.995 and e.max() < 0.2, (tag, name, float(np.median(e)), float((e < 1e-4).mean()), float(e.max()))
        outliers.update(int(i) for i in np.nonzero(e > 1e-2)[0])
    _check_outliers(sorted(outliers), ra, rb, tag, oracle_of)
    return {"q": float((ra[0]["q"] - rb[0]["q"]).abs().max()), "tactile": rel(ra[0]["tactile"], rb[0]["tactile"]),
            "du_median": float(np.median(per_env(ra[2], rb[2]))), "same_evals": float((ra[1] == rb[1]).mean())}


def _edited(pusher_model):
    """What the reference's randomisers do to a model, on the TactilePush blob: contact and tactile parameters (update_contact_parameters /
    update_tactile_parameters), a joint damping (update_joint_damping), a body's mass and inertia (update_body_density)."""
    m = copy.copy(pusher_model); m.F = pusher_model.F.copy(); I = m.I
    fp, fs, fd, fl = I[BL.TSIM_IH_FOFF_PAIR], I[BL.TSIM_IH_FOFF_SENSOR], I[BL.TSIM_IH_FOFF_DOF], I[BL.TSIM_IH_FOFF_LINK]
    m.F[fp + 1 * BL.TSIM_PF_SIZE + BL.TSIM_PF_KN] *= 1.4; m.F[fp + 1 * BL.TSIM_PF_SIZE + BL.TSIM_PF_MU] *= 0.7; m.F[fp + BL.TSIM_PF_KT] *= 1.2
    m.F[fs + BL.TSIM_SF_KN] *= 0.8; m.F[fs + BL.TSIM_SF_KD] *= 1.3
    m.F[fd + 6 * BL.TSIM_DF_SIZE + BL.TSIM_DF_DAMPING] = 0.05
    m.F[fl + 3 * BL.TSIM_LF_SIZE + BL.TSIM_LF_MASS] *= 1.25
    for e in range(3):
        m.F[fl + 3 * BL.TSIM_LF_SIZE + BL.TSIM_LF_INERTIA + e] *= 1.25
    return m


def test_an_edited_model_stays_on_the_compiled_in_structure(pusher_model):
    B, T, S = 2048, 10, 5
    q0, u, _ = push_workload(B, T, seed=5)
    g = torch.Generator().manual_seed(2)
    wq, wv, wt = (torch.randn(T, B, n, generator=g).to(DEV) for n in (7, 6, 390))
    m = _edited(pusher_model)
    a = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=T * S)
    assert a.kernel_variant() == "static:pusher" and a.static_model() == 1
    a.update_model(m)
    assert a.kernel_variant() == "param:pusher" and a.static_model() == 1
    b = BatchSim(m, B, dtype=torch.float32, tape_capacity=T * S)
    b.set_static(False)
    assert b.kernel_variant() == "generic" and b.static_model() == 0
    ra, rb = _run(a, q0, u, T, S, wq, wv, wt), _run(b, q0, u, T, S, wq, wv, wt)
    out = _agree(ra, rb, B, "edited model", oracle_of=lambda e: _oracle_gradient(m, q0, u, T, S, wq, wv, wt, e))
    # ... and the edit matters: against the XML's model the trajectories differ visibly
    c = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=T * S)
    rc = _run(c, q0, u, T, S, wq, wv, wt)
    assert float((rc[0]["q"] - ra[0]["q"]).abs().max()) > 1e-4
    a.update_model(pusher_model)
    assert a.kernel_variant() == "static:pusher"
    from _report import rep
    rep("param_vs_generic_edited", **out)


def test_per_environment_tables_stay_on_the_compiled_in_structure(pusher_model):
    B, T, S = 2048, 10, 5
    q0, u, _ = push_workload(B, T, seed=6)
    g = torch.Generator().manual_seed(3)
    wq, wv, wt = (torch.randn(T, B, n, generator=g).to(DEV) for n in (7, 6, 390))
    a = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=T * S)
    b = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=T * S)
    b.set_static(False)
    # (1) every environment the XML's own table: the structure-static kernels against the fully static ones
    a.set_env_tables(a.base_tables())
    assert a.kernel_variant() == "param:pusher"
    s = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=T * S)
    assert s.kernel_variant() == "static:pusher"
    ra, rs = _run(a, q0, u, T, S, wq, wv, wt), _run(s, q0, u, T, S, wq, wv, wt)
    o1 = _agree(ra, rs, B, "base tables vs static", oracle_of=lambda e: _oracle_gradient(pusher_model, q0, u, T, S, wq, wv, wt, e))
    # (2) every environment its own draw of contact / tactile parameters, box mass, yaw damping — as the reference's reset-time randomisers draw them
    I = pusher_model.I
    fp, fs, fd, fl = I[BL.TSIM_IH_FOFF_PAIR], I[BL.TSIM_IH_FOFF_SENSOR], I[BL.TSIM_IH_FOFF_DOF], I[BL.TSIM_IH_FOFF_LINK]
    tab = a.base_tables()
    r = torch.rand(B, 8, generator=g).to(DEV)
    tab[:, fp + BL.TSIM_PF_SIZE + BL.TSIM_PF_KN] *= 0.7 + 0.6 * r[:, 0]
    tab[:, fp + BL.TSIM_PF_SIZE + BL.TSIM_PF_MU] *= 0.5 + r[:, 1]
    tab[:, fp + BL.TSIM_PF_SIZE + BL.TSIM_PF_KD] *= 0.7 + 0.6 * r[:, 2]
    tab[:, fs + BL.TSIM_SF_KN] *= 0.7 + 0.6 * r[:, 3]
    tab[:, fd + 6 * BL.TSIM_DF_SIZE + BL.TSIM_DF_DAMPING] = 0.01 + 0.1 * r[:, 4]
    scale = 0.8 + 0.4 * r[:, 5]
    for e in (BL.TSIM_LF_MASS, BL.TSIM_LF_INERTIA, BL.TSIM_LF_INERTIA + 1, BL.TSIM_LF_INERTIA + 2):
        tab[:, fl + 3 * BL.TSIM_LF_SIZE + e] *= scale
    a.set_env_tables(tab); b.set_env_tables(tab)
    assert a.kernel_variant() == "param:pusher" and b.kernel_variant() == "generic"
    ra, rb = _run(a, q0, u, T, S, wq, wv, wt), _run(b, q0, u, T, S, wq, wv, wt)
    def model_of_env(e):      # the oracle takes a model: the XML's with environment e's row of the table in place of the shared records
        m_ = copy.copy(pusher_model); m_.F = pusher_model.F.copy()
        row = tab[e].double().cpu().numpy()
        m_.F[:row.size] = row
        return m_
    o2 = _agree(ra, rb, B, "randomised tables vs generic", allow_bad=4, oracle_of=lambda e: _oracle_gradient(model_of_env(e), q0, u, T, S, wq, wv, wt, e))      # (a random draw may leave an environment at max_iter: both kernels must flag the same ones)
    assert float((ra[0]["q"] - rs[0]["q"]).abs().max()) > 1e-4       # the randomisation matters
    # (3) a table that breaks the structure (a joint axis that is no unit vector of the joint frame any more) takes the batch to the generic kernels
    bad = a.base_tables()
    bad[B // 2, fl + 0 * BL.TSIM_LF_SIZE + BL.TSIM_LF_AXES + 0] = 0.1
    a.set_env_tables(bad)
    assert a.kernel_variant() == "generic"
    a.set_env_tables(None)
    assert a.kernel_variant() == "static:pusher"
    from _report import rep
    rep("param_tables", base_vs_static=o1, randomised_vs_generic=o2)


def test_one_structure_static_evaluation_against_the_fp64_one(pusher_model):
    """g and H of ONE residual evaluation (tsim_debug_eval) of an edited model: structure-static, generic and fp64 kernels."""
    B, T, S = 512, 8, 5
    m = _edited(pusher_model)
    q0, u, _ = push_workload(B, T, seed=11)
    d = BatchSim(m, B, dtype=torch.float64, tape_capacity=0)
    d.set_static(False)                                                  # the generic fp64 kernels
    d.reset(torch.tensor(q0, device=DEV, dtype=torch.float64), None, backward_flag=False)
    ro = d.rollout(torch.tensor(u, device=DEV, dtype=torch.float64).transpose(0, 1).contiguous(), S, want_qd=True)
    q, qd = ro["q"][-1], ro["qd"][-1]
    q1 = q + float(m.h) * qd + 1e-4 * torch.randn(B, 7, generator=torch.Generator().manual_seed(3), dtype=torch.float64).to(DEV)
    uu = torch.tensor(u[:, -1], device=DEV, dtype=torch.float64)
    gd, Hd = d.debug_eval(q1, q, qd, uu)
    a = BatchSim(m, B, dtype=torch.float32, tape_capacity=0)
    b = BatchSim(m, B, dtype=torch.float32, tape_capacity=0)
    b.set_static(False); a.set_lanes_per_env(16); b.set_lanes_per_env(16)
    assert a.kernel_variant() == "param:pusher" and b.kernel_variant() == "generic"
    f = lambda x: x.float()
    (ga, Ha), (gb, Hb) = (s_.debug_eval(f(q1), f(q), f(qd), f(uu)) for s_ in (a, b))

    def err(x, y):
        x, y = x.double().reshape(B, -1), y.reshape(B, -1)
        return ((x - y).abs().max(1).values / y.abs().max(1).values.clamp_min(1e-30)).cpu().numpy()
    eHa, eHb, ega, egb = err(Ha, Hd), err(Hb, Hd), err(ga, gd), err(gb, gd)
    assert np.median(eHa) < 2e-6 and eHa.max() < 1e-3 and np.median(ega) < 1e-5 and ega.max() < 1e-2, (np.median(eHa), eHa.max(), np.median(ega), ega.max())
    assert np.median(eHa) < 3 * np.median(eHb) + 1e-7 and np.median(ega) < 3 * np.median(egb) + 1e-7


@pytest.mark.parametrize("seed", range(16))
def test_structure_static_kernels_on_randomly_scaled_parameters_follow_the_oracle(pusher_model, seed):
    """Every float record of pusher.xml's blob that is a PARAMETER or a non-trivial geometric value — masses, centres of mass, inertias, joint and
    primitive positions, end-effector offsets, dampings, limits, motor records, primitive sizes, contact and tactile parameters, contact points and
    taxel positions — scaled by its own random factor (frames' rotations and axes are left alone: scaled they would not be rotations): the
    batch keeps the compiled-in STRUCTURE (`param:pusher`), and the fp64 structure-static kernels land on the oracle's trajectory and gradient."""
    import copy
    from oracle.oracle import OracleSim
    rng = np.random.default_rng(400 + seed)
    m = copy.deepcopy(pusher_model)
    I, F = m.I, m.F
    nl, nr_, nu_ = int(I[BL.TSIM_IH_NL]), int(I[BL.TSIM_IH_NR]), int(I[BL.TSIM_IH_NU])
    fac = lambda n=1: rng.uniform(0.8, 1.25, size=n)
    for l_ in range(nl):
        o_ = int(I[BL.TSIM_IH_FOFF_LINK]) + l_ * BL.TSIM_LF_SIZE
        F[o_ + BL.TSIM_LF_P:o_ + BL.TSIM_LF_P + 3] *= fac(3)
        k = fac()[0]
        F[o_ + BL.TSIM_LF_MASS] *= k
        F[o_ + BL.TSIM_LF_INERTIA:o_ + BL.TSIM_LF_INERTIA + 6] *= k
        F[o_ + BL.TSIM_LF_COM:o_ + BL.TSIM_LF_COM + 3] *= fac(3)
    for d in range(nr_):
        o_ = int(I[BL.TSIM_IH_FOFF_DOF]) + d * BL.TSIM_DF_SIZE
        F[o_:o_ + BL.TSIM_DF_SIZE] *= fac(BL.TSIM_DF_SIZE)
    for k_ in range(nu_):
        o_ = int(I[BL.TSIM_IH_FOFF_MOTOR]) + k_ * BL.TSIM_MF_SIZE
        F[o_:o_ + BL.TSIM_MF_SIZE] *= fac(BL.TSIM_MF_SIZE)
    for v_ in range(int(I[BL.TSIM_IH_NVAR])):
        o_ = int(I[BL.TSIM_IH_FOFF_VAR]) + v_ * BL.TSIM_VF_SIZE
        F[o_:o_ + 3] *= fac(3)
    for p_ in range(int(I[BL.TSIM_IH_NPAIR])):
        o_ = int(I[BL.TSIM_IH_FOFF_PAIR]) + p_ * BL.TSIM_PF_SIZE
        F[o_ + BL.TSIM_PF_P:o_ + BL.TSIM_PF_P + 3] *= fac(3)
        F[o_ + BL.TSIM_PF_SHAPE:o_ + BL.TSIM_PF_SIZE] *= fac(BL.TSIM_PF_SIZE - BL.TSIM_PF_SHAPE)
    for s_ in range(int(I[BL.TSIM_IH_NSENSOR])):
        o_ = int(I[BL.TSIM_IH_FOFF_SENSOR]) + s_ * BL.TSIM_SF_SIZE
        F[o_:o_ + BL.TSIM_SF_SIZE] *= fac(BL.TSIM_SF_SIZE)
    ncpt, ntax = int(I[BL.TSIM_IH_NCPT]), int(I[BL.TSIM_IH_NTAXEL])
    F[int(I[BL.TSIM_IH_FOFF_CPT]):int(I[BL.TSIM_IH_FOFF_CPT]) + 3 * ncpt] *= fac(3 * ncpt)
    F[int(I[BL.TSIM_IH_FOFF_TAXEL]):int(I[BL.TSIM_IH_FOFF_TAXEL]) + 3 * ntax] *= rng.uniform(0.95, 1.05, size=3 * ntax)      # (taxel positions only)
    F[BL.TSIM_FH_TOL] = 1e-12
    B, T, S = 16, 6, 5
    q0, u, _ = push_workload(B, T, seed=60 + seed)
    dt = torch.float64
    sim = BatchSim(m, B, dtype=dt, tape_capacity=T * S)
    # (an fp64 batch forced to 16 lanes per environment — TSIM_LPE=16 — has no compiled-in instantiation and runs the generic kernels: include/tsim.h)
    assert sim.kernel_variant() == ("generic" if os.environ.get("TSIM_LPE") == "16" else "param:pusher")
    assert BatchSim(m, 4, dtype=torch.float32, tape_capacity=4).kernel_variant() == "param:pusher"
    sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None, backward_flag=True)
    outs = [sim.step(torch.tensor(u[:, t], device=DEV, dtype=dt), S) for t in range(T)]
    wq = rng.normal(size=(B, 7))
    wt = rng.normal(size=(B, 390))
    du = sim.backward_steps(T * S, df_dq=torch.tensor(wq, device=DEV, dtype=dt), df_dtactile=torch.tensor(wt, device=DEV, dtype=dt)).cpu().numpy()
    o = OracleSim(m)
    n = T * S
    for e in range(0, B, 4):
        o.reset(q0[e], record=True)
        for t in range(T):
            bad = o.forward(u[e, t], S)
            assert bad == 0 and int(outs[t]["status"][e]) == 0, (seed, e, t)
            q, _ = o.state()
            var, tac = o.outputs()
            assert np.abs(outs[t]["q"][e].cpu().numpy() - q).max() < 1e-9, (seed, e, t)
            assert np.abs(outs[t]["var"][e].cpu().numpy() - var).max() < 1e-9
            assert np.abs(outs[t]["tactile"][e].cpu().numpy() - tac).max() < 1e-8 * (1.0 + np.abs(tac).max())
        g = o.backward_steps(n, df_dq=np.concatenate([np.zeros((n - 1) * 7), wq[e]]), df_dtac=np.concatenate([np.zeros((n - 1) * 390), wt[e]]))
        assert np.abs(du[e].reshape(n, 6) - g).max() < 1e-7 * (1.0 + np.abs(g).max()), (seed, e, np.abs(du[e].reshape(n, 6) - g).max(), np.abs(g).max())
