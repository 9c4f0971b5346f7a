"""tsim_rollout / tsim_backward_episode (the open-loop episode of EpisodicSimFunction, envs/redmax_torch_functions.py:
46-57, 77-92, as one launch each): bit-identical to the per-step entry points, and checked against the oracle."""
import os

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _report import rep as _rep
import numpy as np
import pytest
import torch

from workloads import push_workload
from tactilesimulation_amd.workloads import asset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_rollout_equals_steps_and_episode_adjoint_equals_step_adjoints(pusher_model, dtype):
    from tactilesimulation_amd.host.batch import BatchSim
    B, T, S = 16, 12, 5
    q0_np, u_np, _ = push_workload(B, T, seed=5)
    dev = "cuda:0"
    q0 = torch.tensor(q0_np, device=dev, dtype=dtype)
    u = torch.tensor(u_np, device=dev, dtype=dtype).transpose(0, 1).contiguous()        # [T, B, nu]
    a = BatchSim(pusher_model, B, dtype=dtype, tape_capacity=T * S)
    b = BatchSim(pusher_model, B, dtype=dtype, tape_capacity=T * S)
    nr, nu, nv, nt = a.ndof_r, a.ndof_u, a.ndof_var, a.ndof_tactile
    g = torch.Generator(device="cpu").manual_seed(1)
    wq = torch.randn(T, B, nr, generator=g).to(dev, dtype)
    wv = torch.randn(T, B, nv, generator=g).to(dev, dtype)
    wt = (torch.randn(T, B, nt, generator=g) * 10).to(dev, dtype)

    a.reset(q0, None, backward_flag=True)
    ro = a.rollout(u, S, want_qd=True)
    b.reset(q0, None, backward_flag=True)
    for t in range(T):
        so = b.step(u[t], S, want_qd=True)
        for k in ("q", "qd", "var", "tactile"):
            assert torch.equal(ro[k][t], so[k]), (k, t)
    assert int(ro["status"].sum()) == 0

    du_ep = a.backward_episode(T, S, wq, wv, wt)                                          # [T, B, nu]
    du_st = torch.zeros_like(du_ep)
    for t in reversed(range(T)):
        du_st[t] = b.backward_steps(S, wq[t], wv[t], wt[t]).sum(1)
    la, lb = a.get_adjoint(), b.get_adjoint()
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    sc = float(du_st.abs().max())
    assert float((du_ep - du_st).abs().max()) <= tol * sc
    for x, y in zip(la, lb):
        assert float((x - y).abs().max()) <= tol * max(float(y.abs().max()), 1.0)
    assert a.tape_len() == 0


def test_episode_adjoint_matches_oracle(pusher_model):
    from tactilesimulation_amd.host.batch import BatchSim
    from oracle.oracle import OracleSim
    B, T, S = 3, 6, 5
    q0_np, u_np, _ = push_workload(B, T, seed=9)
    m = pusher_model
    dev = "cuda:0"
    sim = BatchSim(m, B, dtype=torch.float64, tape_capacity=T * S)
    sim.reset(torch.tensor(q0_np, device=dev), None, backward_flag=True)
    u = torch.tensor(u_np, device=dev).transpose(0, 1).contiguous()
    out = sim.rollout(u, S)
    rng = np.random.default_rng(2)
    nr, nu, nv, nt = sim.ndof_r, sim.ndof_u, sim.ndof_var, sim.ndof_tactile
    wq, wv, wt = rng.normal(size=(T, nr)), rng.normal(size=(T, nv)), rng.normal(size=(T, nt))
    tile = lambda w: torch.tensor(np.broadcast_to(w[:, None, :], (T, B, w.shape[1])).copy(), device=dev)
    du = sim.backward_episode(T, S, tile(wq), tile(wv), tile(wt)).cpu().numpy()
    o = OracleSim(m)
    for e in range(B):
        o.reset(q0_np[e], record=True)
        for t in range(T):
            o.forward(u_np[e, t], S)
            q, _ = o.state(); var, tac = o.outputs()
            assert np.abs(out["q"][t, e].cpu().numpy() - q).max() < 1e-6
            assert np.abs(out["tactile"][t, e].cpu().numpy() - tac).max() < 1e-5 * max(np.abs(tac).max(), 1e-3)
        n = T * S
        sq, sv, st = np.zeros((n, nr)), np.zeros((n, nv)), np.zeros((n, nt))
        sq[S - 1::S], sv[S - 1::S], st[S - 1::S] = wq, wv, wt
        g = o.backward_steps(n, sq, sv, st).reshape(T, S, nu).sum(1)
        assert np.abs(du[:, e] - g).max() < 1e-4 * max(np.abs(g).max(), 1e-12)


def test_batched_episodic_function_with_tactile_masks(pusher_model):
    """BatchedEpisodicSimFunction (rows a3/a4 batched): masked tactile frames, dL/dq0, dL/dqdot0, dL/dactions against the
    per-step entry points with zero seeds at the unmasked frames."""
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.functions import BatchedEpisodicSimFunction
    B, T, S = 6, 9, 1                                   # forward(1) per action, as the reference's episodic function
    q0_np, u_np, _ = push_workload(B, T, seed=12)
    dev, dt = "cuda:0", torch.float64
    mask = torch.tensor([0, 1, 1, 0, 0, 1, 0, 1, 0], dtype=torch.bool)
    q0 = torch.tensor(q0_np, device=dev, dtype=dt, requires_grad=True)
    qd0 = (0.05 * torch.randn(B, q0.shape[1], generator=torch.Generator().manual_seed(3))).to(dev, dt).requires_grad_(True)
    act = torch.tensor(u_np, device=dev, dtype=dt).transpose(0, 1).contiguous().requires_grad_(True)
    sim = BatchSim(pusher_model, B, dtype=dt, tape_capacity=T * S)
    qs, vs, ts = BatchedEpisodicSimFunction.apply(q0, qd0, act, mask, sim, True, S)
    assert qs.shape == (T, B, sim.ndof_r) and vs.shape == (T, B, sim.ndof_var) and ts.shape == (int(mask.sum()), B, sim.ndof_tactile)
    g = torch.Generator().manual_seed(4)
    wq, wv = torch.randn(qs.shape, generator=g).to(dev, dt), torch.randn(vs.shape, generator=g).to(dev, dt)
    wt = (10 * torch.randn(ts.shape, generator=g)).to(dev, dt)
    ((qs * wq).sum() + (vs * wv).sum() + (ts * wt).sum()).backward()

    ref = BatchSim(pusher_model, B, dtype=dt, tape_capacity=T * S)
    ref.reset(q0.detach(), qd0.detach(), backward_flag=True)
    k = 0
    for t in range(T):
        o = ref.step(act.detach()[t], S)
        assert torch.equal(o["q"], qs[t].detach()) and torch.equal(o["var"], vs[t].detach())
        if mask[t]:
            assert torch.equal(o["tactile"], ts[k].detach())
            k += 1
    du = torch.zeros_like(act)
    slot = (torch.cumsum(mask.int(), 0) - 1).tolist()
    for t in reversed(range(T)):
        wtt = wt[slot[t]] if mask[t] else torch.zeros(B, sim.ndof_tactile, device=dev, dtype=dt)
        du[t] = ref.backward_steps(S, wq[t], wv[t], wtt).sum(1)
    lq, lv = ref.get_adjoint()
    for got, want in ((act.grad, du), (q0.grad, lq), (qd0.grad, lv)):
        assert float((got - want).abs().max()) <= 1e-11 * max(float(want.abs().max()), 1.0)


def _grasp_actions(gp, q0):
    """The five-stage grasp of envs/stable_grasp_env.py:197-229 (move, close, lift + capture, put down, open) with
    shorter stages: linearly interpolated joint-position targets [x, y, z, yaw, finger, finger]."""
    lift, gh, fp = 0.2029862 + 0.03, 0.2029862, -0.008
    tq = [q0[:6].copy()] + [np.array([0, gp, h, 0, f, f]) for h, f in ((gh, fp), (gh, fp), (lift, fp), (lift, fp), (gh, fp), (gh, fp), (gh, q0[4]))]
    ns = [6, 3, 10, 4, 10, 3, 5]
    return np.array([(tq[s + 1] - tq[s]) / ns[s] * (i + 1) + tq[s] for s in range(len(ns)) for i in range(ns[s])])


@pytest.mark.parametrize("dtype,tq,tt", [(torch.float64, 1e-8, 1e-6), (torch.float32, 1e-6, 1e-4)])      # measured fp32: q 1.2e-8, tactile 2.0e-5 (same taxels in contact)
def test_stable_grasp_episode_matches_oracle(dtype, tq, tt):
    """StableGrasp (stable_grasp.xml: 12 dofs, position-controlled gripper, 11 rigidly joined boxes, 2 pads): the env's
    grasp episode through BatchedEpisodicSimFunction with a tactile capture mask, against the oracle frame by frame."""
    import os
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.model import blob as Bl
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.functions import BatchedEpisodicSimFunction
    from oracle.oracle import OracleSim
    m = load_model(asset("stable_grasp"))
    m.F[Bl.TSIM_FH_TOL] = 1e-12 if dtype == torch.float64 else 1e-8
    gps = [0.0, 0.02, -0.035]
    Bn = len(gps)
    q0 = np.zeros((Bn, 12)); q0[:, 2] = 0.2; q0[:, 4] = q0[:, 5] = -0.03
    acts = np.stack([_grasp_actions(g, q0[e]) for e, g in enumerate(gps)], axis=1)        # [T, B, 6]
    T = acts.shape[0]
    mask = torch.zeros(T, dtype=torch.bool); mask[14] = True; mask[21] = True
    dev = "cuda:0"
    sim = BatchSim(m, Bn, dtype=dtype, tape_capacity=1)
    qs, _, tacs = BatchedEpisodicSimFunction.apply(torch.tensor(q0, device=dev, dtype=dtype), torch.zeros(Bn, 12, device=dev, dtype=dtype),
                                                   torch.tensor(acts, device=dev, dtype=dtype), mask, sim, False, 1)
    assert tacs.shape == (2, Bn, sim.ndof_tactile)
    qs, tacs = qs.double().cpu().numpy(), tacs.double().cpu().numpy()
    o = OracleSim(m)
    lifted = 0.0
    for e in range(Bn):
        o.reset(q0[e])
        k = 0
        for t in range(T):
            assert o.forward(acts[t, e], 1) == 0
            q, _ = o.state()
            assert np.abs(qs[t, e] - q).max() < tq, (e, t)
            if mask[t]:
                _, tac = o.outputs()
                assert np.abs(tac).max() > 1e-3                                       # the pads do press on the object
                _rep("site4_stable_grasp", dtype=str(dtype), env=e, t=t, tac=np.abs(tacs[k, e] - tac).max() / np.abs(tac).max(), q=np.abs(qs[t, e] - q).max(),
                     n_contact_oracle=int((tac.reshape(-1, 3)[:, 2] != 0).sum()), n_contact_hip=int((tacs[k, e].reshape(-1, 3)[:, 2] != 0).sum()))
                assert np.abs(tacs[k, e] - tac).max() < tt * np.abs(tac).max(), (e, t)
                k += 1
            lifted = max(lifted, q[8])
    assert lifted > 5e-3                                                               # the object left the table


def test_masked_reset_restarts_only_the_masked_environments(pusher_model):
    """tsim_reset_masked: a roll-out collector restarts single environments; the others must not notice."""
    from tactilesimulation_amd.host.batch import BatchSim
    B, T, S = 8, 10, 5
    q0_np, u_np, _ = push_workload(B, T, seed=21)
    dev, dt = "cuda:0", torch.float64
    q0 = torch.tensor(q0_np, device=dev, dtype=dt)
    u = torch.tensor(u_np, device=dev, dtype=dt).transpose(0, 1).contiguous()
    a = BatchSim(pusher_model, B, dtype=dt, tape_capacity=0)
    ref = BatchSim(pusher_model, B, dtype=dt, tape_capacity=0)
    a.reset(q0, None, False); ref.reset(q0, None, False)
    mask = torch.tensor([0, 1, 0, 0, 1, 0, 0, 1], device=dev)
    sel, keep = mask.bool(), ~mask.bool()
    t_reset = 4
    early, late = [], []
    for t in range(T):
        if t == t_reset:
            a.reset_masked(q0, mask)                          # envs 1, 4, 7 start over; their actions restart too
        ua = u[t].clone()
        if t >= t_reset:
            ua[sel] = u[t - t_reset][sel]
        oa = a.step(ua, S)
        orf = ref.step(u[t], S)
        assert torch.equal(oa["q"][keep], orf["q"][keep]) and torch.equal(oa["tactile"][keep], orf["tactile"][keep])
        early.append(orf["q"][sel].clone())
        if t >= t_reset:
            late.append(oa["q"][sel].clone())
    for x, y in zip(late, early):                               # the restarted envs replay their first steps exactly
        assert torch.equal(x, y)
    with pytest.raises(RuntimeError):
        rec = BatchSim(pusher_model, B, dtype=dt, tape_capacity=4)
        rec.reset(q0, None, True)
        rec.reset_masked(q0, mask)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 5e-4)])
def test_launch_shapes_agree(pusher_model, dtype, tol, monkeypatch):
    """1, 2 and 4 environments per wavefront (TSIM_LPE = 64 / 32 / 16, read at tsim_batch_create) run the same arithmetic
    up to the summation order: forward outputs and the whole-episode adjoint of the three shapes agree to round-off."""
    monkeypatch.delenv("TSIM_LPE", raising=False)
    from tactilesimulation_amd.host.batch import BatchSim
    B, T, S = 11, 10, 5                                   # 11: the last wavefront of the packed shapes has idle slots
    q0_np, u_np, _ = push_workload(B, T, seed=31)
    dev = "cuda:0"
    q0 = torch.tensor(q0_np, device=dev, dtype=dtype)
    u = torch.tensor(u_np, device=dev, dtype=dtype).transpose(0, 1).contiguous()
    g = torch.Generator().manual_seed(6)
    res = {}
    for lpe in (64, 32, 16):
        if lpe == 32:
            monkeypatch.setenv("TSIM_LPE", "32")              # either way of forcing the shape
            sim = BatchSim(pusher_model, B, dtype=dtype, tape_capacity=T * S)
            monkeypatch.delenv("TSIM_LPE")
        else:
            sim = BatchSim(pusher_model, B, dtype=dtype, tape_capacity=T * S)
            sim.set_lanes_per_env(lpe)
        if dtype == torch.float64 and lpe == 16:
            assert sim.launch_info()["lanes_per_env"] in (16, 32)      # 4 fp64 environments exceed a block's 64 KB of LDS
        else:
            assert sim.launch_info()["lanes_per_env"] == lpe
        sim.reset(q0, None, backward_flag=True)
        out = sim.rollout(u, S)
        if lpe == 64:
            wq = torch.randn(out["q"].shape, generator=g).to(dev, dtype)
            wv = torch.randn(out["var"].shape, generator=g).to(dev, dtype)
            wt = (10 * torch.randn(out["tactile"].shape, generator=g)).to(dev, dtype)
        du = sim.backward_episode(T, S, wq, wv, wt)
        lq, lv = sim.get_adjoint()
        assert int(out["status"].sum()) == 0
        res[lpe] = {"q": out["q"], "var": out["var"], "tactile": out["tactile"], "du": du, "lq": lq, "lv": lv}
    for lpe in (32, 16):
        for k, ref in res[64].items():
            err = float((res[lpe][k] - ref).abs().max())
            assert err <= tol * max(float(ref.abs().max()), 1e-3), (lpe, k, err)


_EPISODE_ORACLE = {}


@pytest.mark.parametrize("lanes", [64, 32, 16])
@pytest.mark.parametrize("dtype,tq,tt,tg", [(torch.float64, 1e-10, 1e-9, 1e-9), (torch.float32, 5e-6, 2e-4, 1e-4)])
def test_full_episodes_against_the_oracle(pusher_model, dtype, tq, tt, tg, lanes):
    """64 environments x 100 env-steps (the bench's episode length) through tsim_rollout / tsim_backward_episode against the
    oracle, with the XML's own Newton tolerance: trajectories, tactile fields and the 100-step episode gradients.  The fp32
    bound on the gradients is the BASELINE.md target (1e-4)."""
    import threading
    from tactilesimulation_amd.host.batch import BatchSim
    from oracle.oracle import OracleSim
    B, T, S = 64, 100, 5
    q0, u, _ = push_workload(B, T, seed=23)
    rng = np.random.default_rng(4)
    wq, wv, wt = rng.normal(size=(T, 7)), rng.normal(size=(T, 6)), rng.normal(size=(T, 390)) * 10.0
    Q, QD, TAC, G = np.zeros((T, B, 7)), np.zeros((T, B, 7)), np.zeros((T, B, 390)), np.zeros((T, B, 6))
    nthr = min(len(os.sched_getaffinity(0)), 16)

    def work(i):                                              # one oracle instance per thread (ctypes releases the GIL)
        o = OracleSim(pusher_model)
        for e in range(i, B, nthr):
            o.reset(q0[e], record=True)
            for t in range(T):
                assert o.forward(u[e, t], S) == 0
                Q[t, e], QD[t, e] = o.state()
                _, TAC[t, e] = o.outputs()
            for t in reversed(range(T)):
                dq = np.zeros((S, 7)); dq[-1] = wq[t]
                dv = np.zeros((S, 6)); dv[-1] = wv[t]
                dt_ = np.zeros((S, 390)); dt_[-1] = wt[t]
                G[t, e] = o.backward_steps(S, dq, dv, dt_).sum(0)
    if "ref" not in _EPISODE_ORACLE:                          # the oracle run is the same for every shape / precision
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
        [t.start() for t in th]
        [t.join() for t in th]
        _EPISODE_ORACLE["ref"] = (Q, QD, TAC, G)
    Q, QD, TAC, G = _EPISODE_ORACLE["ref"]
    dev = "cuda:0"
    sim = BatchSim(pusher_model, B, dtype=dtype, tape_capacity=T * S)
    sim.set_lanes_per_env(lanes)
    got = sim.launch_info()["lanes_per_env"]
    assert got == lanes or (dtype == torch.float64 and lanes == 16 and got == 32)     # 4 fp64 environments exceed 64 KB of LDS
    sim.reset(torch.tensor(q0, device=dev, dtype=dtype), None, backward_flag=True)
    ro = sim.rollout(torch.tensor(u, device=dev, dtype=dtype).transpose(0, 1).contiguous(), S, want_qd=True)
    assert int((ro["status"] != 0).sum()) == 0
    tile = lambda w: torch.tensor(np.broadcast_to(w[:, None, :], (T, B, w.shape[1])).copy(), device=dev, dtype=dtype)
    du = sim.backward_episode(T, S, tile(wq), tile(wv), tile(wt)).double().cpu().numpy()
    assert np.abs(ro["q"].double().cpu().numpy() - Q).max() < tq
    assert np.abs(ro["qd"].double().cpu().numpy() - QD).max() < 200 * tq * max(1.0, np.abs(QD).max())     # qd = dq / h, h = 5e-3
    assert np.abs(ro["tactile"].double().cpu().numpy() - TAC).max() < tt * np.abs(TAC).max()
    eg = np.abs(du - G).max(axis=(0, 2)) / np.abs(G).max(axis=(0, 2))
    assert eg.max() < tg, eg.max()


@pytest.mark.parametrize("name,adjoint", [("pusher", True), ("pusher", False), ("dclaw_position_control", False)])
def test_graphed_open_loop_episode_equals_the_eager_calls(name, adjoint):
    """host/graphed.GraphedEpisode: reset + episode launch forward + episode launch backward replayed from ONE HIP graph give the bits of the eager
    calls — also after new episode data was written into the static inputs, and replay after replay."""
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.host.graphed import GraphedEpisode
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.workloads import asset, dclaw_random_workload, push_workload
    B, T, S, dt = 512, 8, 5, torch.float32
    m = load_model(asset(name))
    data = [push_workload(B, T, seed=s_)[:2] if name == "pusher" else dclaw_random_workload(B, T, seed=s_) for s_ in (1, 2)]
    g = torch.Generator().manual_seed(4)
    seeds = None
    if adjoint:
        seeds = tuple(torch.randn(T, B, n, generator=g).to("cuda") for n in (m.ndof_r, m.ndof_var, m.ndof_tactile))

    def eager(q0, u):
        sim = BatchSim(m, B, dtype=dt, tape_capacity=T * S if adjoint else 0)
        for _ in range(2):                                      # twice: the second episode launch runs in the LPT order of the first, as a replay does
            sim.reset(torch.tensor(q0, device="cuda", dtype=dt), None, backward_flag=adjoint)
            ro = sim.rollout(torch.tensor(u, device="cuda", dtype=dt).transpose(0, 1).contiguous(), S)
            du = sim.backward_episode(T, S, *seeds) if adjoint else None
        return ro, du
    sim = BatchSim(m, B, dtype=dt, tape_capacity=T * S if adjoint else 0)
    q0s = torch.tensor(data[0][0], device="cuda", dtype=dt)
    us = torch.tensor(data[0][1], device="cuda", dtype=dt).transpose(0, 1).contiguous()
    ge = GraphedEpisode(sim, q0s, us, S, seeds=seeds)
    for k in (0, 1, 0):
        q0s.copy_(torch.tensor(data[k][0], device="cuda", dtype=dt)); us.copy_(torch.tensor(data[k][1], device="cuda", dtype=dt).transpose(0, 1))
        ro, du, _ = ge.replay()
        torch.cuda.synchronize()
        ro_e, du_e = eager(*data[k])
        for key in ("q", "var", "tactile", "status"):
            if key in ro_e:
                assert torch.equal(ro[key], ro_e[key]), (name, k, key)
        if adjoint:
            assert torch.equal(du, du_e), (name, k)
