"""RollingBall (BASELINE.json configs[0]: examples/RollingBallExp/test_sim_speed.py — tactile_pad.xml, BDF2, free3d-exp
sphere, 200 x 200 taxels, 350 steps with a tactile read-out every 5): HIP path vs the fp64 oracle, driven through the
redmax_py shim exactly like the reference script drives its simulator."""
import os
import sys

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _report import rep as _rep
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tactilesimulation_amd", "compat"))


def _model(tol=None):
    import tactilesimulation_amd.model.blob as B
    from tactilesimulation_amd.model.compiler import load_model
    m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "tactile_pad.npz"))
    if tol is not None:
        m.F[B.TSIM_FH_TOL] = tol
        m.spec["options"]["tol"] = tol
    return m


def _actions():
    acts = [[0, 0, .2]] * 100 + [[.1, 0, .2]] * 50 + [[-.2, 0, .2]] * 50 + [[0, .1, .2]] * 50 + [[0, -.2, .2]] * 100
    return np.asarray(acts, dtype=np.float64)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-8), (torch.float32, 5e-3)])
def test_exponential_joint_residual_and_newton_matrix(dtype, tol):
    from tactilesimulation_amd.host.batch import BatchSim
    from oracle.oracle import OracleSim
    m = _model()
    o = OracleSim(m); o.reset(np.zeros(9))
    A = _actions()
    states = []
    for i in range(300):
        o.forward(A[i], 1)
        if i in (120, 180, 230, 260, 299):
            q, qd = o.state()
            states.append((q + m.h * qd, q.copy(), qd.copy(), A[i]))
    sim = BatchSim(m, len(states), dtype=dtype, tape_capacity=0)
    g, H = sim.debug_eval(*(torch.tensor(np.stack([s[i] for s in states])) for i in range(4)))
    g, H = g.double().cpu().numpy(), H.double().cpu().numpy()
    for e, s in enumerate(states):
        go, Ho = o.residual(s[0], s[1], s[2], s[3], which=0)
        assert np.abs(g[e] - go).max() <= tol * max(np.abs(go).max(), 1e-7), (e, g[e], go)
        assert np.abs(H[e] - Ho).max() <= tol * np.abs(Ho).max(), (e, np.abs(H[e] - Ho).max() / np.abs(Ho).max())


@pytest.mark.parametrize("dtype,newton_tol,tq,tt", [(torch.float64, 1e-13, 1e-8, 1e-5), (torch.float32, None, 2e-6, 1e-4)])      # measured fp32: q 2.9e-7, tactile 2.2e-5 of the frame's maximum, the same taxels in contact in all 70 read-outs
def test_sim_speed_script_sequence(dtype, newton_tol, tq, tt):
    import redmax_py as redmax
    from oracle.oracle import OracleSim
    m = _model(newton_tol)
    sim = redmax.Simulation(m, dtype=dtype)
    sim.reset(backward_flag=False)
    assert (sim.ndof_u, sim.ndof_r) == (3, 9)
    pos = sim.get_tactile_image_pos("pad")
    assert len(pos) == 40000 and max(p[0] for p in pos) == 199 and max(p[1] for p in pos) == 199
    o = OracleSim(m); o.reset(np.zeros(9))
    A = _actions()
    for i in range(len(A)):
        sim.set_u(A[i]); sim.forward(1, verbose=False, test_derivatives=False)
        assert o.forward(A[i], 1) == 0
        if i % 5 == 0:
            tac = sim.get_tactile_force_vector().copy()
            assert tac.shape[0] // 3 == 200 * 200
            _, to = o.outputs()
            q, _ = o.state()
            _rep("site3_rollingball", dtype=str(dtype), step=i, q=np.abs(sim.get_q() - q).max() / max(1.0, np.abs(q).max()), tac=np.abs(tac - to).max() / max(np.abs(to).max(), 1e-4),
                 tac_scale=np.abs(to).max(), n_contact_oracle=int((to.reshape(-1, 3)[:, 2] != 0).sum()), n_contact_hip=int((tac.reshape(-1, 3)[:, 2] != 0).sum()))
            assert np.abs(sim.get_q() - q).max() <= tq * max(1.0, np.abs(q).max()), i
            assert np.abs(tac - to).max() <= tt * max(np.abs(to).max(), 1e-4), i
