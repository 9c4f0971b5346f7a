"""TactilePush environment arithmetic against golden vectors recorded from the REFERENCE's own
envs/tactile_push_env.py::TactilePushEnv, run in the dev container against a scripted simulator (tools/make_env_fixture.py ->
tests/golden/tactile_push_env.npz): the action handed to the simulator, the observation, the reward and its four terms.
CPU: the plain formulas the GPU tests use as their reference.  GPU: the fused kernels of include/tsim_env.h."""
import math
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tactile_push_env.npz"))


def _formulas(q, var, tac, goal, u, ext):
    """The expressions tests/test_gpu_batched_env.py::test_fused_push_formulas_match_the_reference_formulas uses as its reference."""
    action = np.concatenate([np.tanh(u), ext, np.zeros((len(u), 1))], axis=1)
    th = q[:, 0]
    c, s = np.cos(-th), np.sin(-th)
    gl = np.stack([c * goal[0] - s * goal[1] - q[:, 1], s * goal[0] + c * goal[1] - q[:, 2], goal[2] - th], axis=1)
    obs = np.concatenate([gl, tac], axis=1)
    terms = np.stack([-(((q[:, 3:5] - goal[0:2]) / 0.01) ** 2).sum(1) * 0.01, -(((q[:, 6] - goal[2]) / (math.pi / 36.0)) ** 2) * 0.1,
                      -((var[:, 0:3] - var[:, 3:6]) ** 2).sum(1) / (0.02 ** 2), -(u ** 2).sum(1) * 0.1], axis=1)
    return action, obs, terms


def test_reference_env_facts_recorded_in_the_fixture():
    """What the reference's environment does, as recorded: frame_skip 5, the pre-tanh policy output enters the action penalty, the
    disturbance on the box is redrawn at (almost) every env-step — `current_step` is never incremented (tactile_push_env.py:185) —
    and the initial observation uses the tactile read-out after reset."""
    assert list(G["frame_skip"]) == [5]
    changes = (np.abs(np.diff(G["external_force"], axis=0)).sum(1) > 0).mean()
    assert changes > 0.6, changes                                   # a 10-step hold would give <= 0.1
    assert np.array_equal(G["robot_action"][:, 3:5], G["external_force"]) and np.all(G["robot_action"][:, 5] == 0)
    assert np.allclose(G["obs0"][3:], G["tactile0"]) and G["obs"].shape == (40, 393)


def test_plain_formulas_match_the_reference_env():
    action, obs, terms = _formulas(G["q"], G["var"], G["tactile"], G["goal"], G["u"], G["external_force"])
    assert np.abs(action - G["robot_action"]).max() < 1e-15
    assert np.abs(obs - G["obs"]).max() < 1e-14
    assert np.abs(terms - G["reward_terms"]).max() < 1e-10 * np.abs(G["reward_terms"]).max()
    assert np.abs(terms.sum(1) - G["reward"]).max() < 1e-10 * np.abs(G["reward"]).max()


def test_observation_types_match_the_reference_env():
    """tactile_push_env.py:72-131 builds four observations from the same state; the batched env derives "no_tactile", "privilege" and
    "tactile_map" from the "tactile_flatten" one and q (envs/tactile_push.shape_observation) — checked here against what the reference's
    own class returned for each type on the same scripted episode."""
    import torch
    from tactilesimulation_amd.envs.tactile_push import shape_observation
    obs = torch.tensor(np.concatenate([G["obs0"][None], G["obs"]]))
    q = torch.tensor(np.concatenate([G["q0"][None], G["q"]]))
    nt = shape_observation("no_tactile", obs, q).numpy()
    assert np.array_equal(nt[0], G["obs0_no_tactile"]) and np.array_equal(nt[1:], G["obs_no_tactile"])
    pv = shape_observation("privilege", obs, q).numpy()
    assert pv.shape == (41, 6) and np.abs(pv[0] - G["obs0_privilege"]).max() < 1e-15 and np.abs(pv[1:] - G["obs_privilege"]).max() < 1e-15
    tm, st = shape_observation("tactile_map", obs, q)
    assert np.array_equal(tm.numpy()[0], G["obs0_tactile_map"]) and np.array_equal(tm.numpy()[1:], G["obs_tactile_map"])
    assert np.array_equal(st.numpy()[1:], G["obs_tactile_map_state"])
    assert shape_observation("tactile_flatten", obs, q) is obs
    with pytest.raises(ValueError):
        shape_observation("depth", obs, q)
    # differentiable: the privileged observation carries the box pose's gradient
    qg = q.clone().requires_grad_(True)
    shape_observation("privilege", obs, qg)[:, 0:3].sum().backward()
    assert float(qg.grad[:, 3:5].abs().min()) > 0.0 and float(qg.grad[:, 6].abs().min()) == 1.0


@pytest.mark.gpu
def test_fused_kernels_match_the_reference_env():
    import torch
    from tactilesimulation_amd.envs.push_ops import PushAction, PushObserve, observe_reset
    T = len(G["u"])
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda", dtype=torch.float64)
    goal = dev(np.tile(G["goal"], (T, 1)))
    u = dev(G["u"])
    obs, rew = PushObserve.apply(dev(G["q"]), dev(G["var"]), dev(G["tactile"]), goal, u)
    act = PushAction.apply(u, dev(G["external_force"]))
    assert np.abs(act.cpu().numpy() - G["robot_action"]).max() < 1e-15
    assert np.abs(obs.cpu().numpy() - G["obs"]).max() < 1e-14
    assert np.abs(rew.cpu().numpy() - G["reward"]).max() < 1e-12 * np.abs(G["reward"]).max()
    o0 = observe_reset(dev(G["q0"][None]), dev(G["tactile0"][None]), dev(G["goal"][None]))
    assert np.abs(o0.cpu().numpy()[0] - G["obs0"]).max() < 1e-14
    # fp32 kernels: the tolerance stated for the path
    f = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda", dtype=torch.float32)
    obs32, rew32 = PushObserve.apply(f(G["q"]), f(G["var"]), f(G["tactile"]), f(np.tile(G["goal"], (T, 1))), f(G["u"]))
    assert np.abs(obs32.cpu().numpy() - G["obs"]).max() < 2e-6 * max(1.0, np.abs(G["obs"]).max())
    assert np.abs(rew32.cpu().numpy() - G["reward"]).max() < 2e-5 * np.abs(G["reward"]).max()
