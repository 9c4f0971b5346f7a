"""Known-answer and self-consistency tests that pin the CPU oracle (and with it the formulation in DESIGN.md).

The reference offers no golden numbers for this path (its simulator source is absent — PARITY UNPINNED); what it
offers is a METHOD: analytic-vs-finite-difference gradient checks with rel-error + cosine (algorithms/gd.py:407-468).
That method is applied here, together with closed-form mechanics answers.
"""
import os

import numpy as np
import pytest

import tactilesimulation_amd.model.blob as B
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset
from oracle.oracle import OracleSim

HERE = os.path.dirname(os.path.abspath(__file__))


def _model(name, tol=None):
    m = load_model(os.path.join(HERE, "models", name + ".xml"))
    if tol is not None:
        m.F[B.TSIM_FH_TOL] = tol
    return m


def test_discrete_free_fall_and_force_motor():
    """BDF1 closed form: v_n = v_{n-1} + a h, z_n = z_{n-1} + h v_n  =>  z_n = a h^2 n(n+1)/2."""
    m = _model("point_fall")
    o = OracleSim(m)
    o.reset(np.zeros(3))
    n, h = 50, m.h
    u = np.array([0.5, -1.0, 0.0])               # ctrl_range [-2, 2] N: +1 N, -2 N, 0 N on a 1 kg body
    assert o.forward(u, n) == 0
    q, qd = o.state()
    acc = np.array([1.0, -2.0, -9.8])
    assert np.allclose(qd, acc * n * h, rtol=0, atol=1e-10)
    assert np.allclose(q, acc * h * h * n * (n + 1) / 2, rtol=0, atol=1e-10)
    var, _ = o.outputs(tactile=False)
    assert np.allclose(var, q + np.array([0.05, 0, 1.0]), atol=1e-10)     # end-effector = joint pos + offset


def test_static_rest_penetration():
    """Cube on 4 corner points: d = m g / (4 kn)."""
    m = _model("box_rest")
    o = OracleSim(m)
    o.reset(np.zeros(3))
    assert o.forward(np.zeros(0), 4000) == 0
    q, qd = o.state()
    assert abs(q[2] + 0.5 * 9.8 / (4 * 2e3)) < 1e-9 and np.abs(qd).max() < 1e-9 and np.abs(q[:2]).max() < 1e-12


def test_mass_matrix_and_gravity_torque_of_double_pendulum():
    m = _model("pendulum")
    o = OracleSim(m)
    mu_, ml, ms = 0.02 * 0.02 * 0.4 * 1000, 0.02 * 0.02 * 0.3 * 1000, 2000 * 4 / 3 * np.pi * 0.03 ** 3
    Iu = mu_ * (0.02 ** 2 + 0.4 ** 2) / 12 + mu_ * 0.2 ** 2
    Il = ml * (0.02 ** 2 + 0.3 ** 2) / 12 + ml * 0.15 ** 2 + 0.4 * ms * 0.03 ** 2 + ms * 0.3 ** 2
    z = np.zeros(2)
    u0 = np.array([0.0, 0.0])
    base = o.inverse_dynamics(z, z, z, u0)                     # hanging straight down: no gravity torque, motors at rest
    assert np.abs(base).max() < 1e-12
    M = np.stack([o.inverse_dynamics(z, z, e, u0) for e in np.eye(2)], axis=1)
    assert abs(M[1, 1] - Il) < 1e-12
    # lower link about the shoulder when straight: parallel axis with d + 0.4
    Il_sh = ml * (0.02 ** 2 + 0.3 ** 2) / 12 + ml * 0.55 ** 2 + 0.4 * ms * 0.03 ** 2 + ms * 0.7 ** 2
    assert abs(M[0, 0] - (Iu + Il_sh)) < 1e-12 and abs(M[0, 1] - M[1, 0]) < 1e-14
    # gravity: holding torque at shoulder angle th is  g * sum(m_i * lever_i) * sin(th)  (restoring, so r = +...)
    th = 0.3
    r = o.inverse_dynamics(np.array([th, 0.0]), z, z, np.array([0.0, 0.0]))
    lever = (mu_ * 0.2 + ml * 0.55 + ms * 0.7) * np.sin(th)
    assert abs(r[0] - 9.8 * lever) < 1e-12
    # position motor on the elbow: tau = P (u - q) - D qd enters the residual with a minus sign
    r2 = o.inverse_dynamics(z, z, z, np.array([0.0, 0.5]))
    assert abs(r2[1] + 2.0 * 0.5) < 1e-12
    r3 = o.inverse_dynamics(z, np.array([0.0, 1.0]), z, np.array([0.0, 0.0]))
    # elbow velocity 1 rad/s: D term + the centripetal terms vanish in the straight configuration for dof 1
    assert abs(r3[1] - 0.05 * 1.0) < 1e-12


def test_energy_is_dissipated_not_created():
    """Free double pendulum under BDF1 (implicit Euler): T + V must never increase (numerical dissipation only)."""
    m = _model("pendulum")
    m.F[m.I[B.TSIM_IH_FOFF_MOTOR] + B.TSIM_MF_SIZE + B.TSIM_MF_P] = 0.0      # switch the elbow PD motor off
    m.F[m.I[B.TSIM_IH_FOFF_MOTOR] + B.TSIM_MF_SIZE + B.TSIM_MF_D] = 0.0
    o = OracleSim(m)
    o.reset(np.array([0.8, -0.5]))
    mu_, ml, ms = 0.16, 0.12, 2000 * 4 / 3 * np.pi * 0.03 ** 3

    def energy():
        q, qd = o.state()
        z = np.zeros(2)
        g0 = o.inverse_dynamics(q, z, z, z)
        Mq = np.stack([o.inverse_dynamics(q, z, e, z) - g0 for e in np.eye(2)], 1)
        c1, c12 = np.cos(q[0]), np.cos(q[0] + q[1])
        V = 9.8 * (mu_ * (-0.2 * c1) + ml * (-0.4 * c1 - 0.15 * c12) + ms * (-0.4 * c1 - 0.3 * c12))
        return 0.5 * qd @ Mq @ qd + V
    E = [energy()]
    for _ in range(400):
        o.forward([0.0, 0.0], 1)
        E.append(energy())
    E = np.array(E)
    assert np.all(np.diff(E) < 1e-12), "energy increased"
    assert E[0] - E[-1] < 0.1 * abs(E[0] - E.min() + 1.0)        # ... and only slowly at h = 1e-3


@pytest.mark.parametrize("name,q0,nu", [("slider_push", [0, 0, 0, 0], 1), ("pendulum", [0.4, -0.2], 2)])
def test_adjoint_matches_finite_differences(name, q0, nu):
    """gd.py:407-468 style check: dL/du, dL/dq0, dL/dqdot0 vs central differences; rel-err and cosine."""
    m = _model(name, tol=1e-13)
    o = OracleSim(m)
    rng = np.random.default_rng(0)
    T, S = 8, 3
    U = rng.uniform(-0.8, 0.8, size=(T, nu))
    if name == "slider_push":
        U = np.abs(U) + 0.1
    q0 = np.asarray(q0, dtype=np.float64)
    qd0 = rng.normal(size=q0.size) * 0.05
    wq = rng.normal(size=(T, o.nr)); wv = rng.normal(size=(T, o.nvar)); wt = rng.normal(size=(T, o.ntac)) * 5

    def loss(U, q0, qd0, grad=False):
        o.reset(q0, qd0, record=grad)
        L = 0.0
        for t in range(T):
            o.forward(U[t], S)
            q, _ = o.state()
            var, tac = o.outputs()
            L += wq[t] @ q + wv[t] @ var + (wt[t] @ tac if o.ntac else 0.0)
        if not grad:
            return L
        G = np.zeros_like(U)
        for t in reversed(range(T)):
            dq = np.zeros((S, o.nr)); dq[-1] = wq[t]
            dv = np.zeros((S, o.nvar)); dv[-1] = wv[t]
            dt = np.zeros((S, o.ntac)); dt[-1] = wt[t] if o.ntac else 0
            G[t] = o.backward_steps(S, dq, dv, dt).sum(0)
        lq, lv = o.adjoint()
        return L, G, lq, lv

    L, G, lq, lv = loss(U, q0, qd0, True)
    eps = 1e-6
    Gfd = np.zeros_like(U)
    for t in range(T):
        for j in range(nu):
            Up, Um = U.copy(), U.copy(); Up[t, j] += eps; Um[t, j] -= eps
            Gfd[t, j] = (loss(Up, q0, qd0) - loss(Um, q0, qd0)) / (2 * eps)
    lqfd, lvfd = np.zeros_like(q0), np.zeros_like(q0)
    for k in range(q0.size):
        d = np.zeros_like(q0); d[k] = eps
        lqfd[k] = (loss(U, q0 + d, qd0) - loss(U, q0 - d, qd0)) / (2 * eps)
        lvfd[k] = (loss(U, q0, qd0 + d) - loss(U, q0, qd0 - d)) / (2 * eps)
    for a, b in ((G, Gfd), (lq, lqfd), (lv, lvfd)):
        a, b = a.ravel(), b.ravel()
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-6, (a, b)
        assert 1 - a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) < 1e-10


def test_newton_matrix_is_exact(pusher_model):
    """Dual-number H = dg/dq1 vs central differences at contact states of the TactilePush model."""
    from tactilesimulation_amd.workloads import push_workload, asset
    o = OracleSim(pusher_model)
    q0s, us, _ = push_workload(4, 12, seed=3)
    for e in range(4):
        o.reset(q0s[e])
        for t in range(8 + e):
            o.forward(us[e, t], 5)
        q, qd = o.state()
        q1 = q + pusher_model.h * qd
        g, H = o.residual(q1, q, qd, us[e, 11], which=0)
        Hfd = np.zeros_like(H)
        for k in range(7):
            d = np.zeros(7); d[k] = 1e-7
            Hfd[:, k] = (o.residual(q1 + d, q, qd, us[e, 11]) - o.residual(q1 - d, q, qd, us[e, 11])) / 2e-7
        assert np.abs(H - Hfd).max() < 1e-8 * np.abs(H).max()


def test_pusher_static_sag_and_contact(pusher_model):
    """TactilePush box (0.075 kg) on 4 corners with kn = 1e3 (pusher.xml:50): sag m g / (4 kn) = 1.8375e-4 m."""
    o = OracleSim(pusher_model)
    q0 = np.zeros(7); q0[1] = -0.001
    o.reset(q0)
    o.forward(np.zeros(6), 400)
    q, _ = o.state()
    assert abs(q[5] + 0.075 * 9.8 / 4e3) < 1e-7


def _rolling_ball_actions():
    """examples/RollingBallExp/test_sim_speed.py:43-48"""
    acts = [[0, 0, .2]] * 100 + [[.1, 0, .2]] * 50 + [[-.2, 0, .2]] * 50 + [[0, .1, .2]] * 50 + [[0, -.2, .2]] * 100
    return np.asarray(acts, dtype=np.float64)


def test_rolling_ball_kinematic_known_answer():
    """RollingBall (tactile_pad.xml: BDF2, free3d-exp sphere between ground and pad): a ball rolling without slipping
    between a fixed plane and a moving plate travels HALF the plate's displacement and turns by x / r."""
    m = load_model(asset("tactile_pad"))
    assert (m.ndof_r, m.ndof_u, m.ndof_tactile) == (9, 3, 120000)         # test_sim_speed.py:53-54, 200 x 200 taxels
    o = OracleSim(m)
    o.reset(np.zeros(9))
    A = _rolling_ball_actions()
    for i in range(150):
        assert o.forward(A[i], 1) == 0
    q, _ = o.state()
    assert abs(q[3] - 0.5 * q[0]) < 0.03 * abs(q[0]) and abs(q[7] - q[3] / 0.02) < 0.02 * abs(q[7])
    _, tac = o.outputs()
    nz = np.count_nonzero(tac) // 3
    assert 1000 < nz < 4000 and tac[2::3].min() < -5e-4 and tac[2::3].max() <= 0.0      # normal negative under compression


def test_exponential_joint_newton_matrix_is_exact():
    m = load_model(asset("tactile_pad"))
    o = OracleSim(m)
    o.reset(np.zeros(9))
    A = _rolling_ball_actions()
    for i in range(230):
        o.forward(A[i], 1)
    q, qd = o.state()
    assert np.linalg.norm(q[6:9]) > 0.5            # a genuinely 3-D rotation vector
    q1 = q + m.h * qd
    g, H = o.residual(q1, q, qd, A[230], which=0)
    Hfd = np.zeros_like(H)
    for k in range(9):
        d = np.zeros(9); d[k] = 1e-7
        Hfd[:, k] = (o.residual(q1 + d, q, qd, A[230]) - o.residual(q1 - d, q, qd, A[230])) / 2e-7
    # ~1900 pad points are in stick/slip contact with the ball: a central difference straddles friction kinks in the
    # translational columns, so those are compared loosely; the rotation-vector columns (the new joint) must be exact
    assert np.abs(H[:, 6:9] - Hfd[:, 6:9]).max() < 1e-6 * np.abs(H).max()
    assert np.abs(H - Hfd).max() < 2e-2 * np.abs(H).max()


@pytest.mark.parametrize("name,nsub0,T", [("pusher", 60, 8), ("ball_push", 12, 8)])
def test_bdf2_adjoint_matches_finite_differences(name, nsub0, T):
    """The oracle's adjoint of BDF2 sub-steps (round 3; the reference's only BDF2 model, tactile_pad.xml, is never differentiated —
    examples/RollingBallExp/test_sim_speed.py:51 — so the integrator is forced on two models that are): first recorded sub-step BDF1
    (start-up), the rest BDF2; dL/du of every sub-step and dL/dqd0 against central differences of the roll-out itself."""
    import os
    import tactilesimulation_amd.model.blob as Bl
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
    from oracle.oracle import OracleSim
    here = os.path.dirname(os.path.abspath(__file__))
    m = load_model(PUSHER_BLOB if name == "pusher" else os.path.join(here, "models", name + ".xml"))
    m.F[Bl.TSIM_FH_TOL] = 1e-13
    m.I[Bl.TSIM_IH_INTEGRATOR] = 2
    rng = np.random.default_rng(0)
    if name == "pusher":
        q0, u, _ = push_workload(1, 40, seed=5)
        u = np.repeat(u[0], 5, axis=0)                                    # one action per sub-step
    else:
        q0 = np.array([[0, 0, 0, 0, 0, 0, 0.3, -0.2, 0.5]], dtype=np.float64)
        u = np.stack([[0.3 * np.sin(t), 0.25 * np.cos(t), -0.4] for t in range(40)])
    o = OracleSim(m)
    o.reset(q0[0])
    for t in range(nsub0):                                                # into the contact phase
        assert o.forward(u[t], 1) == 0
    qs, qds = o.state()
    U, W = u[nsub0:nsub0 + T].copy(), rng.normal(size=(T, m.ndof_r))

    def loss(qd0_, U_):
        o.reset(qs, qd0_, record=False)
        L = 0.0
        for t in range(T):
            assert o.forward(U_[t], 1) == 0
            L += W[t] @ o.state()[0]
        return L
    o.reset(qs, qds, record=True)
    for t in range(T):
        o.forward(U[t], 1)
    G = np.zeros((T, m.ndof_u))
    for t in reversed(range(T)):
        G[t] = o.backward_steps(1, W[t][None])[0]
    _, lv = o.adjoint()
    eps = 1e-6
    Gfd, lvf = np.zeros_like(G), np.zeros(m.ndof_r)
    for t in range(T):
        for k in range(m.ndof_u):
            Up, Um = U.copy(), U.copy()
            Up[t, k] += eps; Um[t, k] -= eps
            Gfd[t, k] = (loss(qds, Up) - loss(qds, Um)) / (2 * eps)
    for k in range(m.ndof_r):
        e = np.zeros(m.ndof_r); e[k] = eps
        lvf[k] = (loss(qds + e, U) - loss(qds - e, U)) / (2 * eps)
    assert np.abs(Gfd).max() > 1e-3
    assert np.abs(G - Gfd).max() < 1e-6 * np.abs(Gfd).max(), np.abs(G - Gfd).max() / np.abs(Gfd).max()
    assert np.abs(lv - lvf).max() < 1e-6 * np.abs(lvf).max(), np.abs(lv - lvf).max() / np.abs(lvf).max()


def test_friction_creep_and_sliding_closed_forms():
    """The tangential penalty law ft = -min(kt |vt|, mu |fn|) vt/|vt| on a cube pushed along the ground (4 corner points, N = m g):
    below the Coulomb limit the box creeps at the velocity where 4 kt v balances the push, above it the discrete (BDF1) velocity grows by
    h (F - mu m g) / m per sub-step — exactly, the force being constant — and the vertical equilibrium m g = 4 kn d is not disturbed."""
    m = _model("box_slide")
    o = OracleSim(m)
    o.reset(np.zeros(3))
    mass, g, kn, kt, mu, h = 0.5, 9.8, 2e3, 5.0, 0.8, m.h
    assert o.forward(np.zeros(3), 4000) == 0                       # settle
    force = lambda F: np.array([F / 10.0, 0.0, 0.0])               # ctrl_range [-10, 10] N
    assert o.forward(force(1.0), 4000) == 0                        # 1 N < mu m g = 3.92 N
    q, qd = o.state()
    assert abs(qd[0] - 1.0 / (4 * kt)) < 1e-8 and abs(qd[1]) < 1e-14 and abs(qd[2]) < 1e-12
    assert abs(q[2] + mass * g / (4 * kn)) < 1e-9
    assert o.forward(force(6.0), 40) == 0                          # 6 N: through the creep regime into sliding (v > mu N / (4 kt) = 0.196)
    _, v0 = o.state()
    assert v0[0] > mu * mass * g / (4 * kt)
    n = 40
    assert o.forward(force(6.0), n) == 0
    q, v1 = o.state()
    assert abs((v1[0] - v0[0]) - n * h * (6.0 - mu * mass * g) / mass) < 1e-10
    assert abs(q[2] + mass * g / (4 * kn)) < 1e-9 and abs(v1[2]) < 1e-12
    # pushed along the diagonal the friction force opposes the velocity direction: the same growth of |v|
    o.reset(np.zeros(3)); assert o.forward(np.zeros(3), 4000) == 0
    d = np.array([0.6, 0.6, 0.0]) / np.sqrt(2.0)
    assert o.forward(d, 80) == 0
    _, va = o.state()
    assert o.forward(d, n) == 0
    _, vb = o.state()
    assert abs(va[0] - va[1]) < 1e-12 and abs((np.hypot(vb[0], vb[1]) - np.hypot(va[0], va[1])) - n * h * (6.0 - mu * mass * g) / mass) < 1e-10


def test_tactile_law_closed_forms():
    """A 3 x 3-taxel pad pressed flat onto a block by d = 1 mm, no gravity (tests/models/pad_press.xml; tactile kn 1e2, kt 8, mu 1,
    damping 1e1 as in pusher.xml:11): every taxel reads, in its own frame (axis0, axis1, normal),
      normal  -(kn - kd ddot) d          negative under compression (utils/tactile_utils.py:18,28),
      shear   min(kt |vt|, mu fn)        against the taxel's motion relative to the block."""
    m = _model("pad_press")
    assert (m.ndof_r, m.ndof_tactile) == (6, 27)
    o = OracleSim(m)
    kn, kt, mu, kd, d = 1e2, 8.0, 1.0, 1e1, 1e-3
    q = np.zeros(6); q[2] = -d                                     # the taxel plane is the block's top face at q = 0

    def read(qd):
        o.reset(q, np.asarray(qd, dtype=float))
        return o.outputs()[1].reshape(9, 3)
    t = read(np.zeros(6))
    assert np.allclose(t, np.tile([0.0, 0.0, -kn * d], (9, 1)), rtol=0, atol=1e-15)
    t = read([0.002, 0, 0, 0, 0, 0])                               # creeping along +x: axis0 = -x, the force opposes the motion
    assert np.allclose(t, np.tile([kt * 0.002, 0.0, -kn * d], (9, 1)), rtol=0, atol=1e-15)
    t = read([0, 0, 0, 0.002, 0, 0])                               # the block moves instead: the opposite shear
    assert np.allclose(t, np.tile([-kt * 0.002, 0.0, -kn * d], (9, 1)), rtol=0, atol=1e-15)
    t = read([0.5, 0, 0, 0, 0, 0])                                 # sliding: the Coulomb limit
    assert np.allclose(t, np.tile([mu * kn * d, 0.0, -kn * d], (9, 1)), rtol=0, atol=1e-15)
    t = read([0, 0.3, -0.01, 0, 0, 0])                             # pressing in at 1 cm/s while sliding along +y (axis1 = -y)
    fn = (kn + kd * 0.01) * d
    assert np.allclose(t, np.tile([0.0, mu * fn, -fn], (9, 1)), rtol=0, atol=1e-15)
    q[2] = 1e-4                                                    # lifted off: nothing
    assert not read(np.zeros(6)).any()


def test_joint_space_laws_closed_forms():
    """tests/models/joint_laws.xml, no gravity, no contact: a prismatic joint pushed past its limits rests where the limit spring
    balances the push (q = hi + F / k, lo - F / k: `lim`, `lim_stiffness`), a damped joint reaches the terminal velocity F / d, a PD
    position motor (ctrl="position": P (u - q) - D qd, tactile_insertion.xml:91-92) settles on its target."""
    m = _model("joint_laws")
    o = OracleSim(m)
    o.reset(np.zeros(3))
    assert o.forward(np.array([0.5, 0.5, 0.7]), 6000) == 0           # force motors: ctrl_range [-2, 2] N -> +1 N, +1 N; servo target 0.7 rad
    q, qd = o.state()
    assert abs(q[0] - (0.03 + 1.0 / 50.0)) < 1e-9 and abs(qd[0]) < 1e-8
    assert abs(qd[1] - 1.0 / 2.0) < 1e-10
    assert abs(q[2] - 0.7) < 1e-10 and abs(qd[2]) < 1e-10
    assert o.forward(np.array([-0.25, 0.5, -0.4]), 6000) == 0         # -0.5 N: the lower limit
    q, qd = o.state()
    assert abs(q[0] - (-0.02 - 0.5 / 50.0)) < 1e-9 and abs(q[2] + 0.4) < 1e-10 and abs(qd[1] - 0.5) < 1e-10


def test_sphere_on_ground_closed_forms():
    """A sphere's ground contact is ONE moving point, the lowest point of the sphere (tactile_pad.xml's ball): rest penetration m g / kn,
    creep at F / kt, sliding with h (F - mu m g) / m per sub-step (tests/models/sphere_rest.xml, translational joint: no rolling)."""
    m = _model("sphere_rest")
    o = OracleSim(m)
    o.reset(np.zeros(3))
    mass, g, kn, kt, mu, h = 2000.0 * 4.0 / 3.0 * np.pi * 0.02 ** 3, 9.8, 5e3, 2.0, 0.8, m.h
    assert o.forward(np.zeros(3), 4000) == 0
    q, qd = o.state()
    assert abs(q[2] + mass * g / kn) < 1e-9
    assert o.forward(np.array([0.1, 0, 0]), 4000) == 0                # 0.1 N < mu m g = 0.525 N
    assert abs(o.state()[1][0] - 0.1 / kt) < 1e-7
    assert o.forward(np.array([0.9, 0, 0]), 400) == 0
    va = o.state()[1][0]
    assert o.forward(np.array([0.9, 0, 0]), 40) == 0
    assert abs((o.state()[1][0] - va) - 40 * h * (0.9 - mu * mass * g) / mass) < 1e-9


def _cyl_cases():
    kn, kt, mu, kd, d = 1e2, 8.0, 1.0, 1e1, 1e-3
    q = np.zeros((6, 6)); q[:, 0] = -d; q[5, 2] = 0.035             # case 5: the pad slid up the post, its top taxel beyond the cap's edge
    qd = np.zeros((6, 6)); qd[1, 2] = 0.002; qd[2, 1] = 0.002; qd[3, 2] = 0.9; qd[4, 0] = -0.01
    fn4 = (kn + kd * 0.01) * d
    row = lambda a, b, c: np.tile([a, b, c], (3, 1))
    want = np.stack([row(0, 0, -kn * d), row(kt * 0.002, 0, -kn * d), row(0, -kt * 0.002, -kn * d), row(mu * kn * d, 0, -kn * d), row(0, 0, -fn4),
                     np.array([[0, 0, 0], [0, 0, -kn * d], [0, 0, -kn * d]])])
    return q, qd, want


def test_cylinder_primitive_tactile_closed_forms():
    """tests/models/cyl_press.xml: a 3-taxel strip pressed radially onto the curved side of a cylinder primitive (the D'Claw cap's type,
    dclaw_position_control.xml:114): radial depth d -> normal -kn d, shear kt v along the axis / around it against the motion, the Coulomb
    limit, the damping term, and no force on a taxel past the end cap."""
    m = _model("cyl_press")
    o = OracleSim(m)
    q, qd, want = _cyl_cases()
    for k in range(len(q)):
        o.reset(q[k], qd[k])
        assert np.allclose(o.outputs()[1].reshape(3, 3), want[k], rtol=0, atol=1e-14), k


def test_cylinder_medial_surface_is_a_jump_of_the_penalty_force():
    """[CHOICE] pinned as a known answer (profiles/r06_dclaw_nonconv.md): a penetrating point is pushed out along the normal of the NEAREST face
    of the primitive — inside a cylinder that is the side wall where rho - r > |z| - l/2 and the cap face otherwise.  Across the surface where the
    two distances are equal the force keeps its magnitude kn d and turns by 90 degrees: the residual of a sub-step is DISCONTINUOUS there, which
    is why 3 of the 2048 D'Claw environments of BASELINE configs[3] (a fingertip point 0.1 - 0.3 mm inside the cap's rim) run the XML's Newton
    loop to max_iter in the oracle and in the kernels alike.  tests/models/cyl_press.xml: taxel 0 of the pad, 1 mm inside the wall and
    1 mm -/+ 1e-9 below the cap face."""
    m = _model("cyl_press")
    o = OracleSim(m)
    out = []
    for eps in (-1e-9, +1e-9):                       # taxel 0 sits at rho = 0.02 + qx, z = 0.01 + qz;  cylinder r = 0.02, l / 2 = 0.04
        q = np.array([-0.001, 0.0, 0.029 + eps, 0.0, 0.0, 0.0])
        o.reset(q, np.zeros(6))
        out.append(o.outputs()[1].reshape(3, 3)[0].copy())
    side, cap = out
    # side wall nearest: the force is radial = along the pad's normal (third component; negative under compression), 100 N/m x 1 mm
    assert abs(side[2] + 0.1) < 1e-9 and abs(side[0]) < 1e-12 and abs(side[1]) < 1e-12, side
    # cap face nearest (1 nm further up): the same magnitude along the cylinder's axis = the taxel's first shear axis (0, 0, -1)
    assert abs(abs(cap[0]) - 0.1) < 1e-7 and abs(cap[2]) < 1e-12 and abs(cap[1]) < 1e-12, cap
    assert np.linalg.norm(side - cap) > 0.14            # a jump of sqrt(2) kn d over 2 nm: no root of the residual lies on that surface
