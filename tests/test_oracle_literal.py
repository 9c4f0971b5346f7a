"""The oracle's literal solver — the loop the HIP kernels run — against the globalisation rounds 1-2 used (CPU only; the HIP path against
the literal solver: test_gpu_literal.py).

VERDICT r02 "what's weak" 1: kernel and oracle shared one hand-tuned Newton globalisation (backtracking cut short after 4 halvings, then
the full Newton step taken anyway; restart; trust region), so a sub-step on which the non-monotone step lands on a different root than
plain backtracking would was invisible to every parity test.  Round 3 added `solver="literal"` — Newton + monotone backtracking exactly
as `<solver_option tol max_iter max_ls>` states it (pusher.xml:4) — and found exactly that on the stiff TactileInsertion grasp: on 11 of
4096 environments the r02 solver "converges" (|g| < tol) to a state 0.15 rad / 2 cm away from the root plain backtracking reaches in
four iterations.  The kernels now run the literal loop; the r02 solver survives in the oracle as `solver="r02"` so that this file can
keep the finding as a regression:

* TactilePush bench inputs: the two solvers take bit-identical iterates except on the rare sub-steps where an r02 device acts (3 of the
  first 128 000 sub-steps of the bench batch), and there both reach the same root;
* TactileInsertion: environment 28 of the per-GPU batch, sub-step 6 — r02 jumps, literal does not.
"""
import numpy as np
import pytest

from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset, push_workload, insertion_workload

from literal_util import compare_with_literal, r02_rollout

S = 5
# "same root to the solver tolerance": both iterates satisfy ||g||_2 < tol = 1e-8 with g = h^2 r, so they differ by at most
# ~ 2 tol ||H^-1||; H ~ the joint-space mass matrix, whose smallest entries are the 3.6e-5 kg finger links of the insertion model
# and ~1e-2 on TactilePush => 1e-6 is a bound, a few 1e-7 is what is seen
ROOT_TOL = 1e-6


def test_literal_is_the_default_and_reads_no_tuned_constant(pusher_model):
    from oracle.oracle import OracleSim
    q0, u, _ = push_workload(4, 6, seed=3)
    o = OracleSim(pusher_model)
    assert o.solver == "literal"
    for e in range(4):
        o.reset(q0[e])
        for t in range(6):
            assert o.forward(u[e, t], S) == 0
    st = o.stats()
    assert st["kicks"] == 0 and st["restarts"] == 0 and st["trust_region"] == 0
    assert st["substeps"] == 4 * 6 * S and st["nonconverged"] == 0
    with pytest.raises(KeyError):
        o.set_solver("tuned")


def test_push_r02_iterates_equal_the_literal_ones_where_no_device_acts(pusher_model):
    B, T = 16, 30
    q0, u, _ = push_workload(B, T, seed=0)                         # the first environments of the bench batch
    q, qd, ok, acted = r02_rollout(pusher_model, q0, u, S)
    assert ok.all()
    dq, ok_l, st = compare_with_literal(pusher_model, q, qd, u, S)
    assert ok_l.all() and st["ls_exhausted"] == 0
    assert (dq[~acted] == 0.0).all()                               # the same arithmetic, bit for bit
    assert dq.max() < ROOT_TOL


@pytest.mark.parametrize("env,t_last", [(166, 1), (105, 58)])
def test_push_substeps_with_a_non_monotone_step_reach_the_literal_root(pusher_model, env, t_last):
    """Two of the three sub-steps among the first 128 000 of the bench batch on which the r02 solver takes a non-monotone step
    (environment 166, env-step 1, sub-step 2; environment 105, env-step 58, sub-step 2): the literal solver gets there by backtracking
    alone, and to the same root."""
    q0, u, _ = push_workload(4096, 100, seed=0)
    q, qd, ok, acted = r02_rollout(pusher_model, q0[env:env + 1], u[env:env + 1, :t_last + 1], S)
    assert ok.all() and acted[0, t_last * S + 2], np.nonzero(acted[0])
    dq, ok_l, _ = compare_with_literal(pusher_model, q, qd, u[env:env + 1, :t_last + 1], S)
    assert ok_l.all()
    assert 0.0 < dq[0, t_last * S + 2] < ROOT_TOL                  # different iterates, the same root
    assert (dq[~acted] == 0.0).all()


def test_insertion_r02_globalisation_reaches_another_root_where_backtracking_does_not():
    """Why the kernels dropped it.  Environment 28 of BASELINE configs[4]'s per-GPU batch, the sub-step in which the closing fingers meet
    the box (env-step 1, sub-step 1): plain backtracking needs step lengths below 1/16 there and converges in a handful of iterations
    with the box at rest; the r02 solver gives up halving after four trials, takes the full Newton step, and ends — with |g| < tol — on a
    root where the box has turned by 0.15 rad and the gripper has moved 2 cm within 5 ms."""
    from oracle.oracle import OracleSim
    m = load_model(asset("tactile_insertion"))
    q0, u = insertion_workload(4096, 14, seed=7)
    e = 28
    q, qd, ok, acted = r02_rollout(m, q0[e:e + 1], u[e:e + 1, :2], S)
    dq, ok_l, st = compare_with_literal(m, q, qd, u[e:e + 1, :2], S)
    k = 6
    assert ok[0, k] and ok_l[0, k] and acted[0, k]                 # both "converged"
    assert dq[0, k] > 0.1                                          # ... 0.15 apart
    assert (dq[0, :k] < ROOT_TOL).all()                            # identical roots up to there
    o = OracleSim(m)
    o.reset(q[0, k], qd[0, k])
    assert o.forward(u[e, k // S], 1) == 0 and o.stats()["newton_iters"] <= 6
    step_literal = np.abs(o.state()[0] - q[0, k]).max()
    step_r02 = np.abs(q[0, k + 1] - q[0, k]).max()
    assert step_literal < 1e-3 and step_r02 > 0.1, (step_literal, step_r02)
