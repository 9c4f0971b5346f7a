"""Edge cases of the C ABI on the GPU: ragged batches (B no multiple of the environments per wavefront), batch-composition
independence, error paths that must fail loudly and leave the batch usable, non-finite inputs confined to their environment."""
import numpy as np
import pytest
import torch

from tactilesimulation_amd.workloads import push_workload

pytestmark = pytest.mark.gpu
S = 5


def _run(model, q0, u, lanes, dtype, grad=True, first=0):
    from tactilesimulation_amd.host.batch import BatchSim
    B, T = u.shape[0], u.shape[1]
    sim = BatchSim(model, B, dtype=dtype, tape_capacity=T * S)
    sim.set_lanes_per_env(lanes)
    sim.reset(torch.tensor(q0), None, backward_flag=grad)
    qs, tacs = [], []
    for t in range(T):
        o = sim.step(torch.tensor(u[:, t]), S)
        qs.append(o["q"].clone()); tacs.append(o["tactile"].clone())
        assert int(o["status"].max()) == 0
    du = None
    if grad:
        w = torch.tensor(np.random.default_rng(3).normal(size=(128, 7))[first:first + B])        # the same seed rows for every batch composition
        du = []
        for t in range(T):                                   # newest first
            du.append(sim.backward_episode(1, S, w[None].to("cuda", dtype), None, None)[0].clone())
    return torch.stack(qs), torch.stack(tacs), (torch.stack(du) if grad else None)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_ragged_batches_and_batch_composition(pusher_model, dtype, lanes):
    """An environment's results do not depend on which other environments share its wavefront or batch: B = 1, 3, 5 and 67
    (no multiples of 4 or 2 environments per wavefront: idle slots in the last block) give bit-identical rows."""
    T = 6
    q0, u, _ = push_workload(67, T, seed=4)
    ref = _run(pusher_model, q0, u, lanes, dtype)
    for B in (1, 3, 5):
        got = _run(pusher_model, q0[:B], u[:B], lanes, dtype)
        for a, b, name in zip(got, ref, ("q", "tactile", "df_du")):
            assert torch.equal(a, b[:, :B]), (name, B)
    # and the last, partly filled block of the big batch is a real result: compare env 66 alone
    solo = _run(pusher_model, q0[66:67], u[66:67], lanes, dtype, first=66)
    for a, b in zip(solo, ref):
        assert torch.equal(a, b[:, 66:67])


def test_errors_fail_loudly_and_leave_the_batch_usable(pusher_model):
    from tactilesimulation_amd.host.batch import BatchSim
    B = 6
    q0, u, _ = push_workload(B, 4, seed=1)
    sim = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=2 * S)
    sim.reset(torch.tensor(q0), None, backward_flag=True)
    with pytest.raises(RuntimeError, match="num_steps"):
        sim.step(torch.tensor(u[:, 0]), 0)
    with pytest.raises(RuntimeError, match="nothing|recorded|tape"):
        sim.backward_steps(S)                                 # nothing recorded yet
    a = sim.step(torch.tensor(u[:, 0]), S)["q"].clone()
    sim.step(torch.tensor(u[:, 1]), S)
    q_before, _ = sim.get_state()
    with pytest.raises(RuntimeError, match="capacity"):
        sim.step(torch.tensor(u[:, 2]), S)                    # the tape holds 2 env-steps
    q_after, _ = sim.get_state()
    assert torch.equal(q_before, q_after) and sim.tape_len() == 2 * S     # the refused call changed nothing
    with pytest.raises(RuntimeError):
        sim.backward_steps(3 * S)                             # more than recorded
    du = sim.backward_steps(2 * S, torch.ones(B, 7))          # still usable
    assert bool(torch.isfinite(du).all()) and sim.tape_len() == 0
    with pytest.raises(RuntimeError, match="recording"):
        sim.reset(torch.tensor(q0), None, backward_flag=True)
        sim.reset_masked(torch.tensor(q0), torch.ones(B, dtype=torch.int32))
    sim.reset(torch.tensor(q0), None, backward_flag=False)
    assert torch.equal(sim.step(torch.tensor(u[:, 0]), S)["q"], a)         # and deterministic after all of that
    with pytest.raises(RuntimeError):
        sim.set_lanes_per_env(48)
    # empty inputs are refused, not launched
    with pytest.raises((RuntimeError, ValueError)):
        BatchSim(pusher_model, 0, dtype=torch.float32)
    with pytest.raises((RuntimeError, ValueError)):
        sim.rollout(torch.zeros(0, B, 6, device="cuda"), S)
    with pytest.raises((RuntimeError, ValueError)):
        sim.backward_episode(0, S, None, None, None)
    assert torch.equal(sim.step(torch.tensor(u[:, 1]), S)["q"].isfinite().all(), torch.tensor(True, device="cuda"))


def test_non_finite_action_is_confined_to_its_environment(pusher_model):
    from tactilesimulation_amd.host.batch import BatchSim
    B, T = 9, 3
    q0, u, _ = push_workload(B, T, seed=2)
    clean = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=0)
    dirty = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=0)
    for s in (clean, dirty):
        s.reset(torch.tensor(q0), None, backward_flag=False)
    ud = u.copy(); ud[4, 1, 0] = np.nan
    for t in range(T):
        a, b = clean.step(torch.tensor(u[:, t]), S), dirty.step(torch.tensor(ud[:, t]), S)
        keep = [e for e in range(B) if e != 4]
        assert torch.equal(a["q"][keep], b["q"][keep]) and torch.equal(a["tactile"][keep], b["tactile"][keep])
        st = b["status"].cpu().numpy()
        assert all(st[e] == 0 for e in keep)
        assert bool(st[4] & (1 << 30)) == (t == 1)           # flagged in the launch that received it, not silently clamped


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("name", ["pusher", "dclaw_position_control", "tactile_insertion", "stable_grasp", "tactile_pad"])
def test_on_demand_readout_equals_the_step_outputs(name, dtype):
    """tsim_readout (k_readout + k_taxels: its own kernels, an fp32 far-test in front of the double-precision taxel position, per-block
    staging of the (sensor, primitive) records — 22 of them for stable_grasp) returns bit for bit what the stepping kernel wrote for the
    same state, on every reference model."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.workloads import asset
    m = load_model(asset(name))
    B, T = 5, 6
    if name in ("dclaw_position_control", "tactile_insertion"):
        from test_gpu_models import _inputs
        T = 12                                                       # the grasps close within ~10 env-steps
        q0, u = _inputs(name, m, B, T)
    elif name == "pusher":
        T = 40
        q0, u, _ = push_workload(B, T, seed=3)
        u[:, :, 0] = 0.9                                             # drive the pad into the box
    elif name == "tactile_pad":
        q0 = np.zeros((B, m.ndof_r)); u = np.tile(np.array([0.0, 0.0, 0.2]), (B, T, 1)); T = 6
    else:                                                            # stable_grasp: close the fingers on the stack
        q0 = np.zeros((B, m.ndof_r)); u = np.zeros((B, T, m.ndof_u)); u[:, :, -2:] = 1.0
    sim = BatchSim(m, B, dtype=dtype, tape_capacity=0)
    # The statement is about the GENERIC kernels: the read-out kernels are generic ones.  A batch on the statically specialised TactilePush
    # kernels (fp32; csrc/tsim_static.h) is another fp32 rounding of the same link poses: there the two agree to 1e-6 of the frame's maximum
    # (second pass below), not bit for bit.  (tolerance 1e-5 of the frame maximum)
    for static in ([False, True] if sim.static_model() else [False]):
        sim.set_static(static)
        sim.reset(torch.tensor(q0), None, backward_flag=False)
        steps = 60 if name == "tactile_pad" else T                   # the pad needs ~100 BDF2 steps to reach the ball: 60 x 2
        for t in range(steps):
            o = sim.step(torch.tensor(u[:, min(t, T - 1)]), 2 if name == "tactile_pad" else S)
        var, tac = sim.readout()
        if not static:
            assert torch.equal(tac, o["tactile"])
            if var is not None:
                assert torch.equal(var, o["var"])
        else:
            assert float((tac - o["tactile"]).abs().max()) <= 1e-5 * float(tac.abs().max())
            assert float((var - o["var"]).abs().max()) <= 1e-6
    if name in ("pusher", "tactile_pad", "tactile_insertion"):       # scenarios known to load taxels (the other two compare zeros and variables)
        assert float(tac.abs().max()) > 0, "the scenario never loaded a taxel: nothing was compared"


@pytest.mark.parametrize("name,static", [("pusher", True), ("pusher", False), ("dclaw_position_control", False), ("tactile_insertion", False)])
def test_scheduling_inside_a_launch_does_not_change_a_bit(name, static, monkeypatch):
    """How the environments of a wavefront are scheduled inside an episode launch is not part of the result: free-running slots with the
    tactile frames evaluated by k_taxels after the launch (the default), slots held together at the frame ends with the in-kernel read-out
    (TSIM_NO_FREE_RUN / TSIM_INKERNEL_READOUT: what closed-loop launches do), and lock-step sub-steps (TSIM_LOCKSTEP: round 3's loop) give
    the same states, tactile frames, Newton work and episode gradients bit for bit — on a ragged batch (67 environments), masked tactile frames."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.workloads import asset
    m = load_model(asset(name))
    B, T = 67, 8
    if name == "pusher":
        q0, u, _ = push_workload(B, T, seed=9)
        u[:, :, 0] = np.abs(u[:, :, 0])                              # towards the box: taxels load within the episode
    else:
        from test_gpu_models import _inputs
        T = 12
        q0, u = _inputs(name, m, B, T)
    mask = torch.ones(T, dtype=torch.bool); mask[1] = False; mask[T - 2] = False
    g = torch.Generator().manual_seed(4)
    wq, wt = torch.randn(T, B, m.ndof_r, generator=g).cuda(), torch.randn(int(mask.sum()), B, m.ndof_tactile, generator=g).cuda()
    wv = torch.randn(T, B, m.ndof_var, generator=g).cuda() if m.ndof_var else None
    def run(env):
        for k in ("TSIM_NO_FREE_RUN", "TSIM_INKERNEL_READOUT", "TSIM_LOCKSTEP"):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, "1")
        sim = BatchSim(m, B, dtype=torch.float32, tape_capacity=T * S)
        sim.set_static(static)
        assert sim.static_model() == int(static)
        sim.reset(torch.tensor(q0, dtype=torch.float32), None, backward_flag=True)
        ro = sim.rollout(torch.tensor(u, dtype=torch.float32).transpose(0, 1).contiguous().cuda(), S, want_qd=True, tactile_mask=mask)
        ev = sim.last_evals().copy()
        du = sim.backward_episode(T, S, wq, wv, wt, tactile_mask=mask)
        return ro, ev, du
    ref = run(())
    assert float(ref[0]["tactile"].abs().max()) > 0
    for env in (("TSIM_NO_FREE_RUN",), ("TSIM_INKERNEL_READOUT",), ("TSIM_NO_FREE_RUN", "TSIM_INKERNEL_READOUT"), ("TSIM_LOCKSTEP",), ("TSIM_LOCKSTEP", "TSIM_INKERNEL_READOUT")):
        got = run(env)
        for k in ("q", "qd", "tactile", "status") + (("var",) if m.ndof_var else ()):
            assert torch.equal(got[0][k], ref[0][k]), (env, k)
        assert (got[1] == ref[1]).all() and torch.equal(got[2], ref[2]), env


def test_large_batch_long_episode_indices_are_64_bit(pusher_model):
    """32 768 environments x 100 env-steps (a 4.5 GB tape, 5 GB of tactile output: every per-environment offset beyond 2^32 bytes): eight
    copies of one 4096-environment batch give eight bit-identical blocks, forward and adjoint, and everything converges."""
    from tactilesimulation_amd.host.batch import BatchSim
    T, copies = 100, 8
    q0, u, _ = push_workload(4096, T, seed=0)
    B = 4096 * copies
    sim = BatchSim(pusher_model, B, dtype=torch.float32, tape_capacity=T * S)
    sim.reset(torch.tensor(np.tile(q0, (copies, 1)), device="cuda", dtype=torch.float32), None, backward_flag=True)
    ro = sim.rollout(torch.tensor(np.tile(u, (copies, 1, 1)), device="cuda", dtype=torch.float32).transpose(0, 1).contiguous(), S)
    assert int((ro["status"] != 0).sum()) == 0
    du = sim.backward_episode(T, S, *[torch.ones(T, B, d, device="cuda") for d in (7, 6, 390)])
    q, tac, g = ro["q"].reshape(T, copies, 4096, 7), ro["tactile"].reshape(T, copies, 4096, 390), du.reshape(T, copies, 4096, 6)
    for c in range(1, copies):
        assert torch.equal(q[:, 0], q[:, c]) and torch.equal(tac[:, 0], tac[:, c]) and torch.equal(g[:, 0], g[:, c]), c
    assert float(tac.abs().max()) > 0 and bool(torch.isfinite(g).all())
