"""D'Claw path through the `redmax_py` shim exactly as envs/dclaw_rotate_env.py drives it: the four
reset-time randomisers (update_joint_damping / update_body_size / update_endeffector_position / update_joint_location,
:173-178), relative position control (:201-207), get_q / get_qdot / get_variables (:94-97) and the 3 x 20 x 20 x 3 tactile flow
images (:103-114) — against the oracle compiled from the identically edited spec."""
import copy
import os
import sys

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _report import rep as _rep
import numpy as np
import pytest
import torch

import tactilesimulation_amd.model.blob as Bl
from tactilesimulation_amd.model import compiler as mc
from tactilesimulation_amd.workloads import asset

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tactilesimulation_amd", "compat"))

DOF_LIMIT = np.array([[-0.45, 1.35], [-2, 2], [1, 2]] * 3, dtype=np.float64)        # envs/dclaw_rotate_env.py:78-88


def _dclaw(tol=1e-12):
    m = mc.load_model(asset("dclaw_position_control"))
    spec = copy.deepcopy(m.spec)
    spec["options"]["tol"] = tol
    return mc.compile_spec(spec)


def _episode(sim_like_step, get_q, n_steps, rng):
    """The env's step(): relative joint-position control, frame_skip 5."""
    for _ in range(n_steps):
        action = np.clip(rng.uniform(-1, 1, 9), -1.0, 1.0)
        cur = get_q()[:9]
        target = np.clip(cur + action * 0.06, DOF_LIMIT[:, 0], DOF_LIMIT[:, 1])     # relative_q_scale 0.06 (:23)
        sim_like_step(target)


def test_reset_randomisers_match_an_oracle_compiled_from_the_edited_model():
    import redmax_py as redmax
    from oracle.oracle import OracleSim
    base = _dclaw()
    damping, radius, dx, dy = 0.31, 0.052, 0.013, -0.017                   # one draw of the ranges at :169-171
    sim = redmax.Simulation(copy.deepcopy(base))
    sim.update_joint_damping("cap", damping)
    sim.update_body_size("cap", np.array([0.03, radius]))
    sim.update_endeffector_position("cap", np.array([radius, 0, 0]))
    sim.update_joint_location("cap", np.array([dx, dy, 0.075]))
    # the same edits on an independent copy of the spec -> oracle model
    spec = copy.deepcopy(base.spec)
    mc.edit_spec(spec, "joint_damping", "cap", damping)
    mc.edit_spec(spec, "body_size", "cap", np.array([0.03, radius]))
    mc.edit_spec(spec, "endeffector_position", "cap", np.array([radius, 0, 0]))
    mc.edit_spec(spec, "joint_location", "cap", np.array([dx, dy, 0.075]))
    edited = mc.compile_spec(spec)
    # the edits land where the kernels read them: dof damping, cylinder shape (radius, half length), variable point, joint frame
    I, F, F0 = edited.I, edited.F, base.F
    cap_dof = edited.meta["dof_of_joint"]["cap"][0]
    assert F[I[Bl.TSIM_IH_FOFF_DOF] + cap_dof * Bl.TSIM_DF_SIZE + Bl.TSIM_DF_DAMPING] == damping
    cap_link = edited.meta["link_of_joint"][edited.meta["joint_names"].index("cap")]
    lf = I[Bl.TSIM_IH_FOFF_LINK] + (cap_link - 1) * Bl.TSIM_LF_SIZE
    assert np.allclose(F[lf + Bl.TSIM_LF_P:lf + Bl.TSIM_LF_P + 2] - F0[lf + Bl.TSIM_LF_P:lf + Bl.TSIM_LF_P + 2], [dx, dy])
    pf = I[Bl.TSIM_IH_FOFF_PAIR]
    assert all(abs(F[pf + p * Bl.TSIM_PF_SIZE + Bl.TSIM_PF_SHAPE] - radius) < 1e-15 for p in range(3))

    rng = np.random.default_rng(3)
    q_init = sim.get_q_init().copy()
    q_init[[1, 4, 7]], q_init[[2, 5, 8]] = -0.5, 0.8                         # :74-77
    q_init[:9] += rng.normal(size=9) * 0.05                                  # :167
    # bring the fingertips onto the cap so that contacts and taxels are active
    q_init[[1, 4, 7]], q_init[[2, 5, 8]] = 0.1 + 0.01 * rng.normal(size=3), 0.97
    sim.set_state_init(q_init, np.zeros(10))
    sim.reset(backward_flag=False)                                         # the four edits are compiled and uploaded here, once
    assert np.array_equal(sim._model.F, F) and np.array_equal(sim._model.I, I)
    o = OracleSim(edited)
    o.reset(q_init, np.zeros(10))
    o_plain = OracleSim(base)
    o_plain.reset(q_init, np.zeros(10))
    rng_a, rng_b, rng_c = (np.random.default_rng(9) for _ in range(3))
    rec = {"sim": [], "orc": [], "plain": []}

    def step_sim(target):
        sim.set_u(target); sim.forward(5, verbose=False, test_derivatives=False)
        rec["sim"].append((sim.get_q().copy(), sim.get_qdot().copy(), sim.get_variables().copy(), sim.get_tactile_force_vector().copy()))

    def step_orc(target):
        assert o.forward(target, 5) == 0
        rec["orc"].append(o.state() + o.outputs())

    def step_plain(target):
        o_plain.forward(target, 5)
        rec["plain"].append(o_plain.state() + o_plain.outputs())
    _episode(step_sim, sim.get_q, 16, rng_a)
    _episode(step_orc, lambda: o.state()[0], 16, rng_b)
    _episode(step_plain, lambda: o_plain.state()[0], 16, rng_c)
    tmax = max(np.abs(r[3]).max() for r in rec["orc"])
    assert tmax > 1e-3                                                     # the fingertips do press on the cap
    for a, b in zip(rec["sim"], rec["orc"]):
        assert np.abs(a[0] - b[0]).max() < 1e-9                            # get_q
        assert np.abs(a[1] - b[1]).max() < 1e-6                            # get_qdot (:94)
        assert np.abs(a[2] - b[2]).max() < 1e-9                            # get_variables: fingertips + the cap marker
        assert np.abs(a[3] - b[3]).max() < 1e-7 * tmax
    # ... and the randomised model is a different system from the XML's
    assert np.abs(rec["orc"][-1][0] - rec["plain"][-1][0]).max() > 1e-4
    assert np.abs(rec["orc"][-1][2][9:12] - rec["plain"][-1][2][9:12]).max() > 5e-3     # the cap marker moved with the radius

    # ---- tactile flow images (:103-114): 3 sensors x 20 x 20 x 3, taxel k of a sensor at its (row, col) of the spec file
    imgs = sim.get_tactile_flow_images()
    tact = np.array(imgs)
    assert tact.shape == (3, 20, 20, 3)
    tac = rec["orc"][-1][3].reshape(3, 302, 3)
    for s, name in enumerate(("one3_link_fingertip", "two3_link_fingertip", "three3_link_fingertip")):
        pos = sim.get_tactile_image_pos(name)
        assert len(pos) == 302
        want = np.zeros((20, 20, 3))
        mask = np.zeros((20, 20), dtype=bool)
        for k, (r, c) in enumerate(pos):                                   # [CHOICE] taxels that share a cell: the last one wins
            want[r, c] = tac[s, k]
            mask[r, c] = True
        assert mask.sum() == 182                                            # cells covered by the 302 taxels (:69-72 builds this mask)
        assert np.abs(tact[s] - want).max() < 1e-7 * tmax
        assert np.all(tact[s][~mask] == 0.0)
    obs = torch.tensor(tact).permute(0, 3, 1, 2).reshape(-1, 20, 20)       # the "tactile" observation layout (:110-112)
    assert obs.shape == (9, 20, 20)


@pytest.mark.parametrize("name,B_,T,S", [("pusher_13x13", 8, 10, 5), ("dclaw_9x9", 4, 8, 5), ("tactile_insertion_32x32", 4, 10, 5)])
@pytest.mark.parametrize("dtype,tq,tt", [(torch.float64, 1e-9, 1e-7), (torch.float32, 5e-5, 2e-3)])
def test_baseline_worded_sizes_run_and_match_the_oracle(name, B_, T, S, dtype, tq, tt):
    """BASELINE.json's synthetic taxel layouts (13 x 13 pad, 9 x 9 per finger, 32 x 32 pads): compiled from the real models
    with only the sensor layout changed (workloads.synthetic_variant), run on the HIP path, compared with the oracle."""
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.workloads import synthetic_variant, push_workload
    from oracle.oracle import OracleSim
    from test_gpu_models import _inputs
    m = synthetic_variant(name)
    m.F[Bl.TSIM_FH_TOL] = 1e-13 if dtype == torch.float64 else 1e-8
    want_tax = {"pusher_13x13": 169, "dclaw_9x9": 243, "tactile_insertion_32x32": 2048}[name]
    assert m.ndof_tactile == 3 * want_tax
    if name == "pusher_13x13":
        q0, u, _ = push_workload(B_, T, seed=41)
    else:
        q0, u = _inputs({"dclaw_9x9": "dclaw_position_control", "tactile_insertion_32x32": "tactile_insertion"}[name], m, B_, T)
    sim = BatchSim(m, B_, dtype=dtype, tape_capacity=T * S)
    sim.reset(torch.tensor(q0), None, backward_flag=True)
    ro = sim.rollout(torch.tensor(u).transpose(0, 1).contiguous(), S, want_qd=True)
    assert int((ro["status"] != 0).sum()) == 0
    rng = np.random.default_rng(1)
    wt = rng.normal(size=(T, m.ndof_tactile))
    tile = lambda w: torch.tensor(np.broadcast_to(w[:, None, :], (T, B_, w.shape[1])).copy(), device="cuda:0", dtype=dtype)
    du = sim.backward_episode(T, S, None, None, tile(wt)).double().cpu().numpy()          # loss on the tactile field only
    o = OracleSim(m)
    tmax = 0.0
    for e in range(B_):
        o.reset(q0[e], record=True)
        for t in range(T):
            assert o.forward(u[e, t], S) == 0
            q, qd = o.state()
            _, tac = o.outputs()
            tmax = max(tmax, np.abs(tac).max())
            assert np.abs(ro["q"][t, e].double().cpu().numpy() - q).max() < tq, (name, e, t)
            assert np.abs(ro["tactile"][t, e].double().cpu().numpy() - tac).max() < tt * max(np.abs(tac).max(), 1e-3), (name, e, t)
        n = T * S
        st = np.zeros((n, m.ndof_tactile)); st[S - 1::S] = wt
        g = o.backward_steps(n, None, None, st).reshape(T, S, m.ndof_u).sum(1)
        _rep("site1_layout_grad", name=name, dtype=str(dtype), env=e, rel=np.abs(du[:, e] - g).max() / max(np.abs(g).max(), 1e-12))
        # BASELINE.json's "gradients within 1e-4 rel of CPU", asserted as stated.  Measured (profiles/r04_fp32_tolerance_sites.md): fp32 1.9e-6
        # (13 x 13 pad), 1.1e-6 (9 x 9 per finger), 3.7e-5 (32 x 32 pads); fp64 1.5e-12.  (Round 3 stated 2e-2 here, from the round-2 solver.)
        assert np.abs(du[:, e] - g).max() < (1e-10 if dtype == torch.float64 else 1e-4) * max(np.abs(g).max(), 1e-12), (name, e)
    assert tmax > 1e-4, "no taxel touched anything in %s" % name


def test_reference_env_call_protocol_replays_on_the_shim():
    """Every call the reference's own DClawRotateEnv makes on its simulator during construction, reset() and three step()s — recorded
    by running that class against a recording stand-in (tools/make_env_protocol_fixtures.py -> tests/golden/dclaw_env_protocol.json) — is
    replayed on this repository's `redmax_py` shim with the recorded argument values: each call is accepted and returns what the
    environment expects (shapes, dtypes, list structure), and the observation it would assemble has the recorded size."""
    import json
    import redmax_py as redmax
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from replay_protocol import replay
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dclaw_env_protocol.json")))
    assert g["frame_skip"] == 5 and g["relative_q_scale"] == 0.06 and np.array_equal(np.array(g["dof_limit"]), DOF_LIMIT)

    def make(path, verbose):
        assert path.endswith("dclaw_rotate/dclaw_position_control.xml")
        sim = redmax.Simulation(_dclaw(tol=1e-8), verbose=verbose, dtype=torch.float32)
        assert (sim.ndof_r, sim.ndof_u, sim.ndof_var, sim.ndof_tactile) == (10, 9, 12, 2718) and abs(sim.options.h - 5e-3) < 1e-15
        return sim

    def on_call(sim, name, out):
        if name == "get_tactile_flow_images":                       # (3, 20, 20, 3) after np.array(), dclaw_rotate_env.py:103-105
            assert np.array(out).shape == (3, 20, 20, 3)
        if name == "get_tactile_image_pos":                         # (row, col) pairs inside the 20 x 20 image
            assert len(out) == 302 and all(len(p) == 2 and 0 <= p[0] < 20 and 0 <= p[1] < 20 for p in out)
    sim, seen = replay(make, g["log"], on_call)
    assert {"update_joint_damping", "update_body_size", "update_endeffector_position", "update_joint_location", "set_state_init", "set_u",
            "forward", "get_qdot", "get_tactile_flow_images", "get_tactile_image_pos"} <= set(seen)
    assert sum(seen.values()) == g["calls"]
    # the observation the environment assembles from these getters (dclaw_rotate_env.py:93-118)
    obs = np.concatenate((sim.get_q()[:9], sim.get_variables()[:9], np.array(sim.get_tactile_flow_images()).transpose(0, 3, 1, 2).reshape(-1)))
    assert list(obs.shape) == g["obs_shape_after_reset"] == [9 + 9 + 3 * 3 * 20 * 20]
    assert sim.last_nonconverged_substeps == 0
