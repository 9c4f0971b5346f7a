"""tsim_readout on a large pad (RollingBall, 200 x 200 taxels) after a forward launch uses the pose records that launch left
(k_forward, end of the launch) instead of a kinematics kernel of its own: same numbers as the read-out that recomputes them, and
every other way of changing the state (reset, masked reset, new model tables) makes it recompute."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _sim(dtype, B=6, lanes=None):
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.host.batch import BatchSim
    m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "tactile_pad.npz"))
    sim = BatchSim(m, B, dtype=dtype, tape_capacity=0)
    if lanes:
        sim.set_lanes_per_env(lanes)
    return sim


def _press(sim, n=100):
    B = sim.B
    u = torch.zeros(B, sim.ndof_u, device=sim.device, dtype=sim.dtype)
    u[:, 2] = 0.2
    u[:, 0] = torch.linspace(-0.05, 0.05, B, device=sim.device, dtype=sim.dtype)     # the environments differ
    for _ in range(n):
        sim.step(u, 1, want_var=False, want_tactile=False)
    return u


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("lanes", [None, 64])
def test_records_left_by_the_forward_launch_equal_recomputed_ones(dtype, lanes):
    sim = _sim(dtype, lanes=lanes)
    sim.reset(torch.zeros(sim.B, sim.ndof_r, device=sim.device, dtype=dtype), None, backward_flag=False)
    _press(sim)
    _, t_fast = sim.readout(want_var=False)                  # after a forward launch: k_taxels alone
    _, t_full = sim.readout(want_var=True)                   # variables asked for: kinematics kernel + k_taxels
    assert (t_full != 0).any()
    # same kinematics code on the same state (double pose chain); instantiations may contract differently in R
    tol = 0.0 if dtype == torch.float64 else 2e-5
    assert (t_fast - t_full).abs().max().item() <= tol * t_full.abs().max().item()
    _, t_again = sim.readout(want_var=False)
    assert torch.equal(t_again, t_fast)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_state_changes_outside_forward_launches_recompute(dtype):
    sim = _sim(dtype)
    z = torch.zeros(sim.B, sim.ndof_r, device=sim.device, dtype=dtype)
    sim.reset(z, None, backward_flag=False)
    _press(sim)
    q, qd = sim.get_state()
    _, t_pressed = sim.readout(want_var=False)
    # reset: the pad is back above the ball, nothing touches
    sim.reset(z, None, backward_flag=False)
    _, t0 = sim.readout(want_var=False)
    assert not (t0 != 0).any()
    # back to the pressed state through reset(q, qd): read-out from recomputed kinematics equals the one the launch left
    sim.reset(q, qd, backward_flag=False)
    _, t1 = sim.readout(want_var=False)
    tol = 0.0 if dtype == torch.float64 else 2e-5
    assert (t1 - t_pressed).abs().max().item() <= tol * t_pressed.abs().max().item()
    # masked reset of half the batch after a step
    u = torch.zeros(sim.B, sim.ndof_u, device=sim.device, dtype=dtype); u[:, 2] = 0.2
    sim.step(u, 1, want_var=False, want_tactile=False)
    mask = torch.zeros(sim.B, dtype=torch.int32, device=sim.device); mask[::2] = 1
    sim.reset_masked(z, mask)
    _, t2 = sim.readout(want_var=False)
    t2 = t2.reshape(sim.B, -1)
    assert not (t2[::2] != 0).any() and (t2[1::2] != 0).any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_a_backward_launch_invalidates_the_pose_records(dtype):
    """After a recorded forward launch and a backward launch the live state is the rolled-back one (tape record t_cur): the tactile read-out
    must describe THAT state, with or without the variables (the pose records of the pre-backward state were reused)."""
    from tactilesimulation_amd.model.compiler import load_model
    from tactilesimulation_amd.host.batch import BatchSim
    m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "tactile_pad.npz"))
    B, n = 3, 130                                          # the pad reaches the ball after ~100 sub-steps (test_sim_speed.py:43)
    sim = BatchSim(m, B, dtype=dtype, tape_capacity=n)
    sim.reset(torch.zeros(B, sim.ndof_r, device=sim.device, dtype=dtype), None, backward_flag=True)
    u = torch.zeros(n, B, sim.ndof_u, device=sim.device, dtype=dtype); u[:, :, 2] = 0.2
    u[:, :, 0] = torch.linspace(-0.05, 0.05, B, device=sim.device, dtype=dtype)
    sim.rollout(u, 1, want_tactile=False, want_var=False)
    _, t_end = sim.readout(want_var=False)
    assert (t_end != 0).any()
    sim.backward_steps(n - 5, torch.zeros(B, sim.ndof_r, device=sim.device, dtype=dtype))       # back to sub-step 5: the pad is still in the air
    _, t_fast = sim.readout(want_var=False)
    _, t_full = sim.readout(want_var=True)
    assert torch.equal(t_fast, t_full)
    assert not torch.equal(t_fast, t_end)                    # ... and it is not the frame of the state the forward launch ended in
