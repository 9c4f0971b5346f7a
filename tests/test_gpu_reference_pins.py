"""tests/test_reference_pins.py on the HIP kernels: the hand-derived kinematics, the magnitudes the reference's renderers are scaled for, the
envs' own success criteria, the settled grasp of generate_initial_pose() and BASELINE configs[4] as SURVEY.md §8d words it — through the C ABI."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_reference_pins import PUSHER_VARS_Q0, push_magnitudes, insertion_relative_shear   # noqa: E402  (hand-derived expectations live there)

from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd import workloads as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_pusher_kinematics_on_the_kernels(pusher_model, dt):
    """k_readout's forward kinematics: end-effector positions at q = 0 and under each dof, against the numbers derived by hand from
    pusher.xml:17-66 in tests/test_reference_pins.py (revolute about z at (0.02, 0, 0.18), planar x / y, half turn about x, quarter turn about y)."""
    B = 8
    sim = BatchSim(pusher_model, B, dtype=dt, tape_capacity=0)
    d = 1e-3
    q = np.zeros((B, 7))
    for k in range(7):
        q[k + 1, k] = d
    sim.reset(torch.tensor(q, device=DEV, dtype=dt), None)
    var = sim.readout(want_tactile=False)[0].double().cpu().numpy()
    tol = 3e-5 if dt == torch.float64 else 4e-5
    assert np.allclose(var[0], PUSHER_VARS_Q0, atol=tol)
    dv = var[1:] - var[0]
    t = 1e-9 if dt == torch.float64 else 3e-8
    assert np.allclose(dv[0], (0.004 * (math.cos(d) - 1), 0.004 * math.sin(d), 0, 0, 0, 0), atol=t)
    assert np.allclose(dv[1], (d, 0, 0, 0, 0, 0), atol=t) and np.allclose(dv[2], (0, d, 0, 0, 0, 0), atol=t)
    for k, ax in ((3, 0), (4, 1), (5, 2)):
        e = np.zeros(6); e[3 + ax] = d
        assert np.allclose(dv[k], e, atol=t)
    assert np.allclose(dv[6], (0, 0, 0, -0.025 * (math.cos(d) - 1), -0.025 * math.sin(d), 0), atol=t)


@pytest.mark.parametrize("static", [True, False])
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_push_taxel_magnitudes_on_the_kernels(pusher_model, dt, static):
    """envs/tactile_push_env.py:285-286 draws shear / 3e-6 and normal / 3e-3: a steady straight push puts the kernels' taxel outputs within
    a decade of both (and the two three decades apart)."""
    forces = (0.2, 0.3)
    B = len(forces)
    q0, _, _ = W.push_workload(B, 1, seed=0)
    sim = BatchSim(pusher_model, B, dtype=dt, tape_capacity=0)
    sim.set_static(static)                                      # the compiled-in TactilePush kernels (both precisions) and the generic ones
    sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None)
    u = torch.zeros(100, B, 6, device=DEV, dtype=dt)
    for e, f in enumerate(forces):
        u[:, e, 0] = f
    ro = sim.rollout(u, 5, want_var=False)
    assert int(ro["status"].abs().max()) == 0
    tac = ro["tactile"].double().cpu().numpy()
    for e in range(B):
        sh, nm = push_magnitudes(lambda t: tac[t, e])
        assert 3e-4 < np.median(nm) < 3e-2, np.median(nm)
        assert 3e-7 < np.percentile(sh, 90) < 3e-5, np.percentile(sh, 90)
        assert np.median(sh) < 0.01 * np.median(nm)


def _attempt_states(cases):
    q0 = np.tile(np.asarray(W.INSERTION_Q_REF), (len(cases), 1))
    for e, (dx, dy, rot) in enumerate(cases):
        q0[e, 0] += dx; q0[e, 6] += dx; q0[e, 1] += dy; q0[e, 7] += dy; q0[e, 3] += rot
    q0[:, 9:12] = W._rotvec_mul_z(q0[:, 9:12], np.array([c[2] for c in cases]))
    return q0, W.insertion_attempt_table(q0)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_insertion_success_criterion_on_the_kernels(dt):
    """envs/tactile_insertion_env.py:387-391 on the HIP path: an attempt inside the hole's 2.25 mm clearance ends with the box 0.7 mm into the
    hole (z < 0.0247, |x|, |y| <= 2.2 mm), one that is 3 - 6 mm or 10 degrees off ends on the rim."""
    cases = [(0.0, 0.0, 0.0), (0.001, 0.001, 0.0), (0.002, 0.0, 0.0), (0.0015, -0.001, 0.02), (0.006, 0.0, 0.0), (0.0, 0.006, 0.0), (0.0, 0.0, math.pi / 18), (0.003, 0.0, 0.0)]
    inserted = [True, True, True, True, False, False, False, False]
    q0, u = _attempt_states(cases)
    m = load_model(W.asset("tactile_insertion"))
    sim = BatchSim(m, len(cases), dtype=dt, tape_capacity=0)
    sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None)
    mask = torch.zeros(45, dtype=torch.bool); mask[list(W.INSERTION_TACTILE_FRAMES)] = True
    ro = sim.rollout(torch.tensor(u, device=DEV, dtype=dt).transpose(0, 1).contiguous(), 1, tactile_mask=mask)
    assert int(ro["status"].abs().max()) == 0
    q = ro["q"][-1].double().cpu().numpy()
    for e, ins in enumerate(inserted):
        assert (q[e, 8] < 0.0247) == ins, (cases[e], q[e, 8])
        if ins:
            assert abs(q[e, 6]) <= 0.0022 and abs(q[e, 7]) <= 0.0022 and 0.0240 < q[e, 8] < 0.0245
        else:
            assert 0.02495 < q[e, 8] < 0.02505


def test_settled_grasp_reproduced_by_the_kernels():
    """generate_initial_pose() (envs/tactile_insertion_env.py:126-170) run by the fp64 kernels — 500 scripted + 500 settling sub-steps — lands
    on workloads.INSERTION_Q_REF, the oracle's numbers, and every sub-step converges."""
    m = load_model(W.asset("tactile_insertion"))
    sim = BatchSim(m, 1, dtype=torch.float64, tape_capacity=0)
    q = np.zeros(12); q[2], q[4], q[5] = 0.2, -0.03, -0.03
    sim.reset(torch.tensor(q[None], device=DEV), None)
    tq = [np.array([0, 0, 0.2, -0.03, 0.0, 0.0]), np.array([0.0, 0.0, 0.2, 0.0, 0.0, 0.0]), np.array([0.0, 0.0, 0.2, 0.0, 1.0, 1.0]), np.array([0.0, 0.0, 0.2, 0.0, 1.0, 1.0])]
    rows = []
    for stage, n in enumerate((100, 100, 300)):
        rows += [(tq[stage + 1] - tq[stage]) / n * (i + 1) + tq[stage] for i in range(n)]
    ro = sim.rollout(torch.tensor(np.array(rows)[:, None, :], device=DEV), 1, want_tactile=False)
    assert int(ro["status"][0]) == 0
    qs = sim.get_state()[0][0].clone()
    qs[2] += 0.029; qs[8] += 0.029
    sim.reset(qs[None], None)
    u = qs[:6].clone(); u[4:6] = 1.0
    o = sim.step(u[None], 500, want_tactile=False)
    assert int(o["status"][0]) == 0
    # 1000 sub-steps at the XML's Newton tolerance of 1e-8: kernel and oracle may stop an iterate apart (measured 3.1e-9)
    assert np.abs(sim.get_state()[0][0].cpu().numpy() - np.asarray(W.INSERTION_Q_REF)).max() < 2e-8


@pytest.mark.parametrize("dt,tq,tt", [(torch.float32, 2e-5, 2e-3), (torch.float64, 1e-8, 1e-6)])
def test_config5_insertion_attempts_b4096_all_converge_and_match_the_oracle(dt, tq, tt):
    """BASELINE configs[4] as SURVEY.md §8d words it — 4096 environments per GPU, the settled grasp moved by U(+-6 mm, +-6 mm, +-10 deg), ONE
    45-sub-step attempt with the six captured tactile frames — under the library's default solver (no evaluation budget): EVERY environment
    converges in every sub-step, and a subset of the batch (every row checked, none skipped) matches the fp64 oracle."""
    from oracle.oracle import OracleSim
    B = 4096
    m = load_model(W.asset("tactile_insertion"))
    q0, u = W.insertion_attempt_workload(B, seed=7)
    sim = BatchSim(m, B, dtype=dt, tape_capacity=0)
    sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None)
    mask = torch.zeros(45, dtype=torch.bool); mask[list(W.INSERTION_TACTILE_FRAMES)] = True
    ro = sim.rollout(torch.tensor(u, device=DEV, dtype=dt).transpose(0, 1).contiguous(), 1, want_qd=True, tactile_mask=mask)
    assert int((ro["status"] != 0).sum()) == 0, "%d environments did not converge" % int((ro["status"] != 0).sum())
    assert bool(torch.isfinite(ro["q"]).all()) and bool(torch.isfinite(ro["tactile"]).all())
    ev = sim.last_evals()
    assert ev.mean() / 45 < 2.6                                       # 2.1 evaluations per sub-step; a few environments have one long line search
    succ = (ro["q"][-1, :, 8] < 0.0247).double().mean().item()
    assert 0.01 < succ < 0.12                                         # oracle, 2048 environments: 3.5 % of the random pre-grasp poses go in
    # envs/tactile_insertion_env.py:508,435-441: the largest relative shear of an attempt where the reference's renderer looks at it (30 px x 2e-6)
    tac = ro["tactile"][:, :64].double().cpu().numpy()
    mx = [insertion_relative_shear(tac[:, e]) for e in range(64)]
    assert 2e-5 < np.median(mx) < 2e-4, np.median(mx)
    idx = np.linspace(3, B - 11, 6).astype(int)
    for e in idx:
        o = OracleSim(m); o.reset(q0[e])
        k = 0
        for t in range(45):
            assert o.forward(u[e, t], 1) == 0
            q, _ = o.state()
            gq = ro["q"][t, e].double().cpu().numpy()
            assert np.abs(gq - q).max() < tq * max(1.0, np.abs(q).max()), (e, t, np.abs(gq - q).max())
            if t in W.INSERTION_TACTILE_FRAMES:
                tac = o.outputs()[1]
                gt = ro["tactile"][k, e].double().cpu().numpy(); k += 1
                assert np.abs(gt - tac).max() < tt * max(np.abs(tac).max(), 1e-3), (e, t)


@pytest.mark.parametrize("dt,tol", [(torch.float64, 6e-8), (torch.float32, 3e-7)])
def test_stable_grasp_settles_at_the_reference_constant_on_the_kernels(dt, tol):
    """envs/stable_grasp_env.py:198-199 `grasp_height = 0.2029862` — the seven digits DiffRedMax left the gripper at after
    generate_initial_state() — reproduced by the HIP kernels (tests/test_reference_pins.py derives what the number measures: the gripper's
    total mass incl. the default-density bodies, gravity, the position motor)."""
    from test_reference_pins import stable_grasp_settled_state
    m = load_model(W.asset("stable_grasp"))
    sim = BatchSim(m, 1, dtype=dt, tape_capacity=0)

    def step500(q, u):
        sim.reset(torch.tensor(q[None], device=DEV, dtype=dt), None)
        o = sim.step(torch.tensor(u[None], device=DEV, dtype=dt), 500, want_tactile=False)
        assert int(o["status"][0]) == 0
        return sim.get_state()[0][0].double().cpu().numpy()
    q = stable_grasp_settled_state(step500)
    assert abs(q[2] - 0.2029862) < tol, q[2]


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_stable_grasp_success_criterion_on_the_kernels(dt):
    """envs/stable_grasp_env.py:262-266 at capture frame 60: the uniform bar lifts level when gripped at its centre, hangs when gripped 1 - 5 cm
    off it."""
    from test_reference_pins import stable_grasp_settled_state, stable_grasp_episode
    m = load_model(W.asset("stable_grasp"))
    one = BatchSim(m, 1, dtype=torch.float64, tape_capacity=0)

    def step500(q, u):
        one.reset(torch.tensor(q[None], device=DEV), None)
        one.step(torch.tensor(u[None], device=DEV), 500, want_tactile=False)
        return one.get_state()[0][0].cpu().numpy()
    q_ref = stable_grasp_settled_state(step500)
    gps = [0.0, 0.01, 0.04, -0.05]
    eps = [stable_grasp_episode(None, None, q_ref, gp) for gp in gps]
    U = torch.tensor(np.stack([np.array(r[:61]) for r, _ in eps], axis=1), device=DEV, dtype=dt)       # [61, B, 6]
    sim = BatchSim(m, len(gps), dtype=dt, tape_capacity=0)
    sim.reset(torch.tensor(np.stack([qi for _, qi in eps]), device=DEV, dtype=dt), None)
    ro = sim.rollout(U, 1, want_tactile=False)
    assert int(ro["status"].abs().max()) == 0
    q60 = ro["q"][-1].double().cpu().numpy()
    ang = np.linalg.norm(q60[:, 9:12], axis=1)
    assert (q60[:, 8] > 0.005).all()
    assert ang[0] < 1e-3 and (ang[1:] > 0.05).all(), ang


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_rolling_ball_peaks_on_the_kernels(dt):
    """utils/tactile_utils.py:4,28 (depth image white at -normal = 1.2e-3, force image red at 8e-4, one cell of arrow at shear 1.5e-4): the
    kernels' read-outs over the test_sim_speed.py sequence peak at those scales, compression negative in the taxel frame."""
    from test_reference_pins import ROLLING_BALL_ACTIONS, rolling_ball_peaks, check_rolling_ball_peaks
    m = load_model(W.asset("tactile_pad"))
    sim = BatchSim(m, 1, dtype=dt, tape_capacity=0)
    sim.reset(torch.zeros(1, 9, device=DEV, dtype=dt), None)
    frames = []
    for i, a in enumerate(ROLLING_BALL_ACTIONS):
        sim.step(torch.tensor([a], device=DEV, dtype=dt), 1, want_var=False, want_tactile=False)
        if i % 5 == 0:
            frames.append(sim.readout(want_var=False)[1][0].double().cpu().numpy())
    check_rolling_ball_peaks(*rolling_ball_peaks(frames))


def test_config4_dclaw_random_policy_b2048_converges_touches_and_matches_the_oracle():
    """BASELINE configs[3] as SURVEY.md §8d words it, at one GPU's share (2048 environments, fp32, forward only): q_init + 0.05 N(0, 1), random
    relative position control for 50 env-steps.  The kernels flag the environments the oracle flags (3 of 2048), fingers do meet the cap by the env's own criterion (summed taxel
    force >= 1.0, envs/dclaw_rotate_env.py:131-133), and a subset of the batch matches the fp64 oracle."""
    from oracle.oracle import OracleSim
    B, T, S = 2048, 50, 5
    m = load_model(W.asset("dclaw_position_control"))
    q0, u = W.dclaw_random_workload(B, T, seed=7)
    dt = torch.float32
    sim = BatchSim(m, B, dtype=dt, tape_capacity=0)
    sim.reset(torch.tensor(q0, device=DEV, dtype=dt), None)
    ro = sim.rollout(torch.tensor(u, device=DEV, dtype=dt).transpose(0, 1).contiguous(), S)
    # The XML's Newton loop itself gives up on one sub-step of 3 of these 2048 random walks (a fingertip jammed against the cap: max_iter
    # after ~1 200 evaluations) — the fp64 oracle flags exactly environments 529, 700, 1855 on the same inputs (165 s of CPU for the whole
    # batch; profiles/r04_dclaw_config4.md), and a quarter of the environments have a sub-step of >= 100 evaluations.  The kernels must flag
    # the same three and nothing else.
    flagged = torch.nonzero(ro["status"] != 0).flatten().tolist()
    assert flagged == [529, 700, 1855], flagged
    for e in flagged:                                                                 # ... and the oracle does give up on each of them
        o = OracleSim(m); o.reset(q0[e])
        assert sum(o.forward(u[e, t], S) != 0 for t in range(T)) == 1, e
    tot = ro["tactile"].reshape(T, B, 3, 302, 3).norm(dim=-1).sum(-1)                 # [T, B, finger]
    assert float((tot >= 1.0).double().mean()) > 0.01                                # measured 0.06 over the first 50 env-steps
    for e in np.linspace(5, B - 9, 4).astype(int):
        o = OracleSim(m); o.reset(q0[e])
        for t in range(T):
            assert o.forward(u[e, t], S) == 0
            q = o.state()[0]; tac = o.outputs()[1]
            assert np.abs(ro["q"][t, e].double().cpu().numpy() - q).max() < 2e-5 * max(1.0, np.abs(q).max()), (e, t)
            assert np.abs(ro["tactile"][t, e].double().cpu().numpy() - tac).max() < 2e-3 * max(np.abs(tac).max(), 1e-3), (e, t)
