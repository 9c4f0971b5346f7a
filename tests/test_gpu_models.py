"""HIP path vs fp64 oracle on further models: the build's own small XMLs (prismatic joint + limits, planar, position
motor, double pendulum without contacts, cylinder sampling) and the reference's DClaw model (10 revolute dofs, 10 links,
432 contact points from files, 3 x 302 abstract taxels, cylinder primitive, position motors, joint limits)."""
import os

import numpy as np
import pytest
import torch

import tactilesimulation_amd.model.blob as B
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, tol=1e-13):
    p = os.path.join(HERE, "models", name + ".xml")
    m = load_model(p if os.path.exists(p) else asset(name))
    m.F[B.TSIM_FH_TOL] = tol
    return m


CASES = {
    # name: (q0, u sampler, T, S)
    "point_fall": (np.zeros(3), lambda r: r.uniform(-1, 1, 3), 6, 3),
    "box_rest": (np.zeros(3), lambda r: np.zeros(0), 6, 5),
    # both regimes of the tangential penalty law (creep below the Coulomb limit, sliding above) and lift-off, with closed forms in
    # tests/test_oracle_physics.py::test_friction_creep_and_sliding_closed_forms
    "box_slide": (np.array([0.0, 0.0, -6e-4]), lambda r: np.array([r.uniform(-0.1, 0.9), r.uniform(-0.5, 0.5), r.uniform(-0.3, 0.55)]), 16, 4),
    # 3 x 3 taxels pressed flat onto a block, no gravity: the tactile law's closed forms are in tests/test_oracle_physics.py
    "pad_press": (np.array([0.0, 0.0, -1e-3, 0.0, 0.0, 0.0]), lambda r: np.zeros(0), 8, 4),
    # limit spring, joint damping, PD position motor — closed forms in tests/test_oracle_physics.py::test_joint_space_laws_closed_forms
    "joint_laws": (np.array([0.02, 0.0, 0.3]), lambda r: np.array([r.uniform(-1, 1), r.uniform(-1, 1), r.uniform(-1.5, 1.5)]), 10, 4),
    # sphere primitive on the ground: one moving contact point (closed forms in tests/test_oracle_physics.py)
    "sphere_rest": (np.array([0.0, 0.0, -1.3e-4]), lambda r: np.array([r.uniform(-0.2, 0.9), r.uniform(-0.4, 0.4), r.uniform(-0.3, 0.5)]), 12, 4),
    "pendulum": (np.array([0.7, -0.4]), lambda r: r.uniform(-1, 1, 2), 8, 4),
    "slider_push": (np.zeros(4), lambda r: np.array([r.uniform(0.2, 1.0)]), 16, 4),
    "dclaw_position_control": (None, None, 10, 5),
    # 2 pads x 13x10 taxels, prismatic fingers with limits, free3d-euler box (3 translations + 3 revolutes after the
    # compiler's decomposition), 11 contact pairs incl. world-fixed general bodies (tactile_insertion.xml)
    "tactile_insertion": (None, None, 14, 5),
    # rotation-vector joint under BDF1, forward AND adjoint (the reference's only free3d-exp model is forward-only)
    "ball_push": (np.array([0, 0, 0, 0, 0, 0, 0.3, -0.2, 0.5]), None, 12, 3),
}


def _inputs(name, m, B_, T):
    rng = np.random.default_rng(7)
    q0c, us, _, _ = CASES[name]
    if name == "dclaw_position_control":
        from tactilesimulation_amd.workloads import dclaw_workload        # shared with bench.py --workload dclaw
        return dclaw_workload(B_, T, seed=7)
    if name == "ball_push":
        q0 = np.tile(q0c, (B_, 1)) + 0.02 * rng.normal(size=(B_, 9)) * np.array([0, 0, 0, 0.05, 0.05, 0, 1, 1, 1])
        u = np.stack([[[0.3 * np.sin(t + e), 0.25 * np.cos(t - e), -0.4 + 0.05 * rng.uniform(-1, 1)] for t in range(T)] for e in range(B_)])
        return q0, u
    if name == "tactile_insertion":
        from tactilesimulation_amd.workloads import insertion_workload    # shared with bench.py --workload insertion
        return insertion_workload(B_, T, seed=7)
    q0 = np.tile(q0c, (B_, 1)) + 1e-3 * rng.normal(size=(B_, q0c.size)) * (name != "box_rest")
    u = np.stack([[us(rng) for _ in range(T)] for _ in range(B_)]).reshape(B_, T, m.ndof_u)
    return q0, u


_ORACLE_CACHE = {}


def _oracle_case(name, m, q0, u, wq, wv, wt, T, S):
    """Oracle trajectory + adjoint of one case (the same for every launch shape and kernel precision)."""
    from oracle.oracle import OracleSim
    key = (name, float(m.F[B.TSIM_FH_TOL]))
    if key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    nr, nu, nv, nt = m.ndof_r, m.ndof_u, m.ndof_var, m.ndof_tactile
    o = OracleSim(m)
    res = []
    for e in range(q0.shape[0]):
        o.reset(q0[e], record=True)
        tr = []
        for t in range(T):
            assert o.forward(u[e, t], S) == 0
            tr.append(o.state() + o.outputs())
        Go = np.zeros((T, max(nu, 1)))
        for t in reversed(range(T)):
            dq = np.zeros((S, nr)); dq[-1] = wq[t]
            dv = np.zeros((S, nv)); dv[-1] = wv[t] if nv else 0
            dt = np.zeros((S, nt)); dt[-1] = wt[t] if nt else 0
            du = o.backward_steps(S, dq, dv, dt)
            if nu:
                Go[t] = du.sum(0)
        res.append((tr, Go) + o.adjoint())
    _ORACLE_CACHE[key] = res
    return res


@pytest.mark.parametrize("lanes", [64, 32, 16])
@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("dtype,tq,tt,tg", [(torch.float64, 1e-12, 1e-9, 1e-10), (torch.float32, 2e-6, 2e-3, 1e-4)])
def test_model_forward_and_adjoint(name, dtype, tq, tt, tg, lanes):
    """All seven models, every launch shape, against the oracle.  The kernels run the oracle's Newton loop (round 3), so the bounds are
    those of the arithmetic: fp64 round-off (measured: q <= 7e-15, gradients <= 1e-12), fp32 q <= 9e-8 and gradients <= 5.4e-5 — the
    gradient bound asserted for fp32 is BASELINE.json's 1e-4 on EVERY model (round 2 needed 2e-2 on the stiff ones)."""
    from tactilesimulation_amd.host.batch import BatchSim
    m = _load(name, 1e-13 if dtype == torch.float64 else 1e-8)
    _, _, T, S = CASES[name]
    B_ = 4
    q0, u = _inputs(name, m, B_, T)
    nr, nu, nv, nt = m.ndof_r, m.ndof_u, m.ndof_var, m.ndof_tactile
    rng = np.random.default_rng(11)
    wq, wv, wt = rng.normal(size=(T, nr)), rng.normal(size=(T, nv)), rng.normal(size=(T, nt))
    sim = BatchSim(m, B_, dtype=dtype, tape_capacity=T * S)
    sim.set_lanes_per_env(lanes)
    got = sim.launch_info()["lanes_per_env"]
    if got != lanes:
        # the rotation-vector joint runs one environment per wavefront; a block holds at most 64 KB of LDS
        pytest.skip("%s %s runs %d lanes per environment (LDS / joint constraints), covered by that case" % (name, dtype, got))
    sim.reset(torch.tensor(q0), None, backward_flag=True)
    outs = []
    for t in range(T):
        r = sim.step(torch.tensor(u[:, t]) if nu else torch.zeros(B_, 0), S, want_qd=True)
        outs.append({k: v.double().cpu().numpy() for k, v in r.items()})
    G = np.zeros((B_, T, max(nu, 1)))
    for t in reversed(range(T)):
        du = sim.backward_steps(S, torch.tensor(np.tile(wq[t], (B_, 1))), torch.tensor(np.tile(wv[t], (B_, 1))) if nv else None,
                                torch.tensor(np.tile(wt[t], (B_, 1))) if nt else None)
        if nu:
            G[:, t] = du.double().cpu().numpy().sum(1)
    lq, lv = (x.double().cpu().numpy() for x in sim.get_adjoint())
    ref = _oracle_case(name, m, q0, u, wq, wv, wt, T, S)
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-9))
    print("MEASURED %s %s lanes %d: q %.1e | dL/du %.1e | lam_q %.1e | lam_v %.1e" % (
        name, str(dtype)[6:], lanes, max(np.abs(outs[t]["q"][e] - ref[e][0][t][0]).max() for e in range(B_) for t in range(T)),
        max(rel(G[e], ref[e][1]) for e in range(B_)) if nu else 0.0, max(rel(lq[e], ref[e][2]) for e in range(B_)), max(rel(lv[e], ref[e][3]) for e in range(B_))))
    for e in range(B_):
        tr, Go, alq, alv = ref[e]
        for t in range(T):
            q, qd, v, tc = tr[t]
            assert np.abs(outs[t]["q"][e] - q).max() <= tq * max(1.0, np.abs(q).max()), (name, e, t)
            # get_qdot (envs/dclaw_rotate_env.py:94): (q1 - q0) / h of the last sub-step
            assert np.abs(outs[t]["qd"][e] - qd).max() <= tq / m.h * max(1.0, np.abs(q).max()) + tq * np.abs(qd).max(), (name, e, t, "qd")
            if nv:
                assert np.abs(outs[t]["var"][e] - v).max() <= tq * 10
            if nt:
                assert np.abs(outs[t]["tactile"][e] - tc).max() <= tt * max(np.abs(tc).max(), 1e-3), (name, e, t)
        if nu:
            assert np.abs(G[e] - Go).max() <= tg * max(np.abs(Go).max(), 1e-9), (name, e, np.abs(G[e] - Go).max() / np.abs(Go).max())
        assert np.abs(lq[e] - alq).max() <= tg * max(np.abs(alq).max(), 1e-9), (name, "lam_q")
        assert np.abs(lv[e] - alv).max() <= tg * max(np.abs(alv).max(), 1e-9), (name, "lam_v")


def test_per_environment_tables_domain_randomisation(pusher_model):
    """Each environment gets its own contact / tactile parameters (batched form of update_contact_parameters /
    update_tactile_parameters, envs/tactile_insertion_env.py:238-281); every row must equal an oracle run on a model
    compiled with those parameters."""
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.model import compiler as mc
    from oracle.oracle import OracleSim
    from tactilesimulation_amd.workloads import push_workload, asset
    m = _load("pusher")
    B_, T = 4, 8
    q0, u, _ = push_workload(B_, T, seed=31)
    rng = np.random.default_rng(2)
    kn, mu, tkn = rng.uniform(50, 400, B_), rng.uniform(0.3, 2.0, B_), rng.uniform(50, 200, B_)
    sim = BatchSim(m, B_, dtype=torch.float64, tape_capacity=4)
    tab = sim.base_tables()
    tab[:, m.table_offset("pair", ("tactile_pad_left", "box"), "kn")] = torch.tensor(kn, device="cuda")
    tab[:, m.table_offset("pair", ("tactile_pad_left", "box"), "mu")] = torch.tensor(mu, device="cuda")
    tab[:, m.table_offset("sensor", "tactile_pad_left", "kn")] = torch.tensor(tkn, device="cuda")
    sim.set_env_tables(tab)
    sim.reset(torch.tensor(q0), None, False)
    for t in range(T):
        out = sim.step(torch.tensor(u[:, t]), 5)
    for e in range(B_):
        spec = mc.compile_spec(m.spec).spec
        spec["options"]["tol"] = 1e-13
        mc.edit_spec(spec, "contact_parameters", ("tactile_pad_left", "box"), kn=kn[e], mu=mu[e])
        mc.edit_spec(spec, "tactile_parameters", "tactile_pad_left", kn=tkn[e])
        o = OracleSim(mc.compile_spec(spec)); o.reset(q0[e])
        for t in range(T):
            o.forward(u[e, t], 5)
        tac = o.outputs()[1]
        assert np.abs(out["q"][e].cpu().numpy() - o.state()[0]).max() < 1e-9
        assert np.abs(out["tactile"][e].cpu().numpy() - tac).max() < 1e-7 * max(np.abs(tac).max(), 1e-4)
    sim.set_env_tables(None)        # back to the shared model
    sim.reset(torch.tensor(q0), None, False)
    sim.step(torch.tensor(u[:, 0]), 5)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-14), (torch.float32, 2e-7)])
def test_tactile_read_out_closed_forms_on_the_kernels(dtype, tol):
    """The tactile law's closed forms (tests/test_oracle_physics.py::test_tactile_law_closed_forms) asserted on the HIP read-out itself,
    without the oracle in between: five environments = the five (velocity) cases, plus one lifted off."""
    from tactilesimulation_amd.host.batch import BatchSim
    m = _load("pad_press")
    kn, kt, mu, kd, d = 1e2, 8.0, 1.0, 1e1, 1e-3
    q = np.zeros((6, 6)); q[:, 2] = -d; q[5, 2] = 1e-4
    qd = np.zeros((6, 6))
    qd[1, 0] = 0.002; qd[2, 3] = 0.002; qd[3, 0] = 0.5; qd[4, 1], qd[4, 2] = 0.3, -0.01
    fn4 = (kn + kd * 0.01) * d
    want = np.array([[0, 0, -kn * d], [kt * 0.002, 0, -kn * d], [-kt * 0.002, 0, -kn * d], [mu * kn * d, 0, -kn * d], [0, mu * fn4, -fn4], [0, 0, 0]])
    sim = BatchSim(m, 6, dtype=dtype, tape_capacity=0)
    sim.reset(torch.tensor(q, device="cuda", dtype=dtype), torch.tensor(qd, device="cuda", dtype=dtype), backward_flag=False)
    _, tac = sim.readout(want_var=False)
    tac = tac.double().cpu().numpy().reshape(6, 9, 3)
    assert np.abs(tac - want[:, None, :]).max() <= tol, np.abs(tac - want[:, None, :]).max()


def test_dynamics_closed_forms_on_the_kernels():
    """Closed-form mechanics asserted on the HIP path itself (fp64 kernels), not through the oracle: discrete free fall under a force
    motor (tests/test_oracle_physics.py::test_discrete_free_fall_and_force_motor), the rest penetration m g / (4 kn) of a cube on its
    corners, and the two regimes of the friction law (creep at F / (4 kt), sliding with h (F - mu m g) / m per sub-step)."""
    from tactilesimulation_amd.host.batch import BatchSim
    dt = torch.float64
    t = lambda a: torch.tensor(np.asarray(a, dtype=float), device="cuda", dtype=dt)
    # free fall: v_n = a h n, z_n = a h^2 n (n + 1) / 2
    m = _load("point_fall", tol=1e-12)
    sim = BatchSim(m, 2, dtype=dt, tape_capacity=0)
    sim.reset(t(np.zeros((2, 3))), None, backward_flag=False)
    n, h = 50, m.h
    out = sim.step(t([[0.5, -1.0, 0.0], [0.0, 0.0, 1.0]]), n, want_qd=True, want_var=False, want_tactile=False)
    acc = np.array([[1.0, -2.0, -9.8], [0.0, 0.0, -9.8 + 2.0]])
    assert np.abs(out["qd"].cpu().numpy() - acc * n * h).max() < 1e-10
    assert np.abs(out["q"].cpu().numpy() - acc * h * h * n * (n + 1) / 2).max() < 1e-10
    # friction: environment 0 creeps under 1 N, environment 1 slides under 6 N, environment 2 is pushed along the diagonal
    m = _load("box_slide", tol=1e-12)
    mass, g, kn, kt, mu, h = 0.5, 9.8, 2e3, 5.0, 0.8, m.h
    sim = BatchSim(m, 3, dtype=dt, tape_capacity=0)
    sim.reset(t(np.zeros((3, 3))), None, backward_flag=False)
    out = sim.step(t(np.zeros((3, 3))), 4000, want_qd=True, want_var=False, want_tactile=False)
    assert int(out["status"].sum()) == 0
    assert np.abs(out["q"].cpu().numpy()[:, 2] + mass * g / (4 * kn)).max() < 1e-9               # rest penetration
    d = 0.6 / np.sqrt(2.0)
    u = t([[0.1, 0, 0], [0.6, 0, 0], [d, d, 0]])
    o1 = sim.step(u, 4000, want_qd=True, want_var=False, want_tactile=False)
    v1 = o1["qd"].cpu().numpy()
    assert abs(v1[0, 0] - 1.0 / (4 * kt)) < 1e-8                                                   # creep
    nn = 40
    o2 = sim.step(u, nn, want_qd=True, want_var=False, want_tactile=False)
    v2 = o2["qd"].cpu().numpy()
    growth = nn * h * (6.0 - mu * mass * g) / mass
    assert abs((v2[1, 0] - v1[1, 0]) - growth) < 1e-9                                              # sliding
    assert abs((np.hypot(*v2[2, :2]) - np.hypot(*v1[2, :2])) - growth) < 1e-9 and abs(v2[2, 0] - v2[2, 1]) < 1e-12
    assert np.abs(o2["q"].cpu().numpy()[:, 2] + mass * g / (4 * kn)).max() < 1e-9
    # joint-space laws: limit spring (q = hi + F / k, lo - F / k), terminal velocity F / d, PD servo on its target
    m = _load("joint_laws", tol=1e-12)
    sim = BatchSim(m, 2, dtype=dt, tape_capacity=0)
    sim.reset(t(np.zeros((2, 3))), None, backward_flag=False)
    o3 = sim.step(t([[0.5, 0.5, 0.7], [-0.25, 0.5, -0.4]]), 6000, want_qd=True, want_var=False, want_tactile=False)
    q3, v3 = o3["q"].cpu().numpy(), o3["qd"].cpu().numpy()
    assert int(o3["status"].sum()) == 0
    assert abs(q3[0, 0] - (0.03 + 1.0 / 50.0)) < 1e-9 and abs(q3[1, 0] - (-0.02 - 0.5 / 50.0)) < 1e-9
    assert np.abs(v3[:, 1] - 0.5).max() < 1e-10 and abs(q3[0, 2] - 0.7) < 1e-10 and abs(q3[1, 2] + 0.4) < 1e-10


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-13), (torch.float32, 2e-7)])
def test_cylinder_primitive_read_out_closed_forms_on_the_kernels(dtype, tol):
    """tests/test_oracle_physics.py::test_cylinder_primitive_tactile_closed_forms on the HIP read-out itself."""
    import sys
    sys.path.insert(0, HERE)
    from test_oracle_physics import _cyl_cases
    from tactilesimulation_amd.host.batch import BatchSim
    q, qd, want = _cyl_cases()
    sim = BatchSim(_load("cyl_press"), len(q), dtype=dtype, tape_capacity=0)
    sim.reset(torch.tensor(q, device="cuda", dtype=dtype), torch.tensor(qd, device="cuda", dtype=dtype), backward_flag=False)
    tac = sim.readout(want_var=False)[1].double().cpu().numpy().reshape(len(q), 3, 3)
    assert np.abs(tac - want).max() <= tol, np.abs(tac - want).max()


def test_cylinder_medial_surface_jump_on_the_kernels():
    """tests/test_oracle_physics.py::test_cylinder_medial_surface_is_a_jump_of_the_penalty_force on the HIP read-out (fp64 kernels: the two states
    differ by 2 nm): nearest-face normal inside the cylinder, so the taxel force turns by 90 degrees across rho - r = |z| - l/2 — what leaves 3 of
    2048 D'Claw environments of BASELINE configs[3] at max_iter (profiles/r06_dclaw_nonconv.md); the kernels implement the oracle's law there."""
    from tactilesimulation_amd.host.batch import BatchSim
    q = np.array([[-0.001, 0.0, 0.029 - 1e-9, 0, 0, 0], [-0.001, 0.0, 0.029 + 1e-9, 0, 0, 0]])
    sim = BatchSim(_load("cyl_press"), 2, dtype=torch.float64, tape_capacity=0)
    sim.reset(torch.tensor(q, device="cuda"), None, backward_flag=False)
    tac = sim.readout(want_var=False)[1].cpu().numpy().reshape(2, 3, 3)[:, 0]
    from oracle.oracle import OracleSim
    o = OracleSim(_load("cyl_press"))
    for k in range(2):
        o.reset(q[k], np.zeros(6))
        assert np.abs(tac[k] - o.outputs()[1].reshape(3, 3)[0]).max() < 1e-13
    assert abs(tac[0, 2] + 0.1) < 1e-9 and abs(abs(tac[1, 0]) - 0.1) < 1e-7 and np.linalg.norm(tac[0] - tac[1]) > 0.14
