"""D'Claw environment arithmetic against golden vectors recorded from the REFERENCE's own DClawRotateEnv run against
a scripted simulator (tools/make_dclaw_env_fixture.py -> tests/golden/dclaw_env.npz).  CPU: the pure functions of envs/dclaw_rotate.py.
GPU: the batched environment itself (variants == separately compiled models, masked resets, the real simulator)."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dclaw_env.npz"))


def test_pure_functions_match_the_reference_env():
    from tactilesimulation_amd.envs import dclaw_rotate as D
    T = len(G["u"])
    assert np.array_equal(G["dof_limit"], D.DOF_LIMIT) and float(G["relative_q_scale"]) == 0.06
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64)
    # targets handed to set_u at step k come from the state BEFORE the step (q[k]) and the policy output u[k]
    tg = D.joint_targets(t(G["q"][:T]), t(G["u"]))
    assert np.abs(tg.numpy() - G["targets"]).max() < 1e-15
    obs = D.observation(t(G["q"][1:]), t(G["var"][1:]), t(G["flow"][1:]))
    assert np.abs(obs.numpy() - G["obs"]).max() < 1e-15
    assert np.abs(D.observation(t(G["q"][:1]), t(G["var"][:1]), t(G["flow"][:1])).numpy()[0] - G["obs0"]).max() < 1e-15
    # the reward's contact term uses the flow images of the PREVIOUS observation (flow[k] at step k), state and variables of the new one
    r, done, succ = D.reward(t(G["q"][1:]), t(G["var"][1:]), t(G["flow"][:T]), t(G["u"]), float(G["rot_coef"]), float(G["power_coef"]), float(G["cap_top_surface_z"]))
    assert np.abs(r.numpy() - G["reward"]).max() < 1e-12
    assert np.array_equal(done.numpy(), G["done"]) and np.array_equal(succ.numpy(), G["success"])
    assert G["done"].sum() == 2 and G["success"].sum() == 1                                 # both terminal branches are in the fixture
    r_now, _, _ = D.reward(t(G["q"][1:]), t(G["var"][1:]), t(G["flow"][1:]), t(G["u"]))
    assert np.abs(r_now.numpy() - G["reward"]).max() > 0.4                                   # ... and the one-step lag is observable in it


@pytest.mark.gpu
def test_batched_env_on_the_simulator():
    from tactilesimulation_amd.envs.dclaw_rotate import BatchedDClawRotateEnv, joint_targets
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.model import compiler as mc
    B, K, T = 24, 3, 6
    env = BatchedDClawRotateEnv(B, dtype=torch.float64, seed=1, variants=K)
    obs = env.reset()
    assert obs.shape == (B, 3618) and bool(torch.isfinite(obs).all())
    q_start, variant = env.q.clone(), env.variant_of.cpu().numpy()
    rng = np.random.default_rng(0)
    U = torch.tensor(rng.uniform(-1.5, 1.5, size=(T, B, 9)), device="cuda")
    for t in range(T):
        obs, r, done, info = env.step(U[t])
        assert int((info["status"] != 0).sum()) == 0 and bool(torch.isfinite(obs).all()) and r.shape == (B,)
    # every environment equals a one-environment simulator on the separately compiled variant it was assigned (bit for bit)
    for e in (0, 7, B - 1):
        damping, radius, dx, dy = env.variant_params[variant[e]]
        spec = mc.compile_spec(env.model.spec).spec
        mc.edit_spec(spec, "joint_damping", "cap", damping); mc.edit_spec(spec, "body_size", "cap", np.array([0.03, radius]))
        mc.edit_spec(spec, "endeffector_position", "cap", np.array([radius, 0.0, 0.0])); mc.edit_spec(spec, "joint_location", "cap", np.array([dx, dy, 0.075]))
        one = BatchSim(mc.compile_spec(spec), 1, dtype=torch.float64, tape_capacity=0)
        one.reset(q_start[e:e + 1], None, backward_flag=False)
        q = q_start[e:e + 1]
        for t in range(T):
            q = one.step(joint_targets(q, U[t, e:e + 1]), 5)["q"]
        assert torch.equal(q[0], env.q[e]), e
        var, tac = one.readout()
        assert torch.equal(var[0], env.var[e]) and torch.equal(env.flow_images(tac)[0], env.flow[e])
    # masked reset: only the masked environments start over
    mask = torch.zeros(B, dtype=torch.bool, device="cuda"); mask[::5] = True
    q_before = env.q.clone()
    env.reset(mask)
    assert torch.equal(env.q[~mask], q_before[~mask]) and not torch.equal(env.q[mask], q_before[mask])
    assert int(env.steps[mask].max()) == 0 and int(env.steps[~mask].min()) == T
    # flow images == the shim's get_tactile_flow_images (last taxel of a cell wins; 182 of the 400 cells of a finger are covered)
    tac = torch.arange(1, 2719, device="cuda", dtype=torch.float64)[None]
    img = env.flow_images(tac)[0].cpu().numpy()
    want = np.zeros((3, 20, 20, 3))
    for s_, (name, (t0, nt, rows, cols)) in enumerate(zip(env.model.meta["sensor_names"], env.model.meta["sensor_taxels"])):
        for k, (r, c) in enumerate(env.model.meta["image_pos"][name]):
            want[s_, r, c] = tac[0].cpu().numpy().reshape(-1, 3)[t0 + k]
    assert np.array_equal(img, want) and [(img[f].sum(-1) > 0).sum() for f in range(3)] == [182, 182, 182]


@pytest.mark.gpu
def test_graphed_collector_equals_the_eager_loop():
    """The collection step replayed from a HIP graph (policy, env.step, per-environment resets incl. new model variants) yields exactly
    the transitions of the eager loop: same seeds, same default-generator draws."""
    from tactilesimulation_amd.envs.dclaw_rotate import BatchedDClawRotateEnv, GraphedCollector
    B, T = 96, 40

    def run(graphed):
        env = BatchedDClawRotateEnv(B, dtype=torch.float32, seed=3, variants=4)
        env.max_episode_steps = 16                                 # so that per-environment resets happen inside the run
        torch.manual_seed(5)
        W = torch.randn(env.obs_dim, env.act_dim, device="cuda") * 0.02
        policy = lambda obs: torch.tanh(obs @ W) + 0.3 * torch.randn(B, 9, device="cuda")
        out = []
        if graphed:
            col = GraphedCollector(env, policy)
            torch.manual_seed(7); col.next_obs.copy_(env.reset())
            for t in range(T):
                o, u, r, d, n = col.step()
                out.append((o.clone(), u.clone(), r.clone(), d.clone(), n.clone()))
        else:
            env._gen = None                                        # both runs are re-seeded right before their first counted reset
            torch.manual_seed(7); obs = env.reset()
            for t in range(T):
                u = policy(obs)
                o2, r, d, info = env.step(u)
                n = env.reset(d)
                out.append((obs.clone(), u.clone(), r.clone(), d.clone(), n.clone()))
                obs = n
        return out
    a, b = run(False), run(True)
    assert sum(int(x[3].sum()) for x in a) >= 2 * B                  # every environment was reset at least twice
    for t, (x, y) in enumerate(zip(a, b)):
        for k, name in enumerate(("obs", "action", "reward", "done", "next_obs")):
            assert torch.equal(x[k], y[k]), (t, name)


@pytest.mark.gpu
def test_continuous_per_environment_randomisation_on_the_device():
    """randomize=True: every environment its own U(0.01, 0.7) damping, U(0.02, 0.08) cap radius, U(+-0.02)^2 cap location, drawn on the device at
    each (masked) reset as the reference draws them per environment per reset (envs/dclaw_rotate_env.py:164-184).  Eight random environments:
    the table row the device wrote equals the float records of the model compiled on the host from the identically edited XML (1e-12), and their
    trajectories match the fp64 ORACLE running that edited model (q 1e-8, tactile 1e-8 of its scale)."""
    from tactilesimulation_amd.envs.dclaw_rotate import BatchedDClawRotateEnv, joint_targets
    from oracle.oracle import OracleSim
    B, T = 64, 5
    env = BatchedDClawRotateEnv(B, dtype=torch.float64, seed=11, randomize=True)
    env.reset()
    p = env.params.cpu().numpy()
    R = BatchedDClawRotateEnv.RANDOMISER_RANGES
    for j, k in enumerate(("damping", "radius", "dx", "dy")):
        assert (p[:, j] >= R[k][0]).all() and (p[:, j] <= R[k][1]).all() and len(np.unique(p[:, j])) == B       # one draw per environment, inside the reference's ranges
    q_start = env.q.clone()
    tables = env.tables.cpu().numpy()
    rng = np.random.default_rng(2)
    U = torch.tensor(rng.uniform(-1.5, 1.5, size=(T, B, 9)), device="cuda")
    traj_q, traj_tac, targets = [], [], []
    for t in range(T):
        targets.append(joint_targets(env.q, U[t]).cpu().numpy())
        out = env.sim.step(joint_targets(env.q, U[t]), 5)
        env.q.copy_(out["q"])
        assert int((out["status"] != 0).sum()) == 0
        traj_q.append(out["q"].cpu().numpy()); traj_tac.append(out["tactile"].cpu().numpy())
    n = tables.shape[1]
    for e in rng.choice(B, size=8, replace=False):
        m = BatchedDClawRotateEnv.edited_model(env.model, *p[e])
        assert np.array_equal(m.I, env.model.I)
        d = np.abs(tables[e] - m.F[:n])
        assert (d <= 1e-12 * np.maximum(np.abs(m.F[:n]), 1e-3)).all(), (e, d.max())
        o = OracleSim(m)
        o.reset(q_start[e].cpu().numpy())
        for t in range(T):
            assert o.forward(targets[t][e], 5) == 0
            q, _ = o.state()
            _, tac = o.outputs()
            assert np.abs(traj_q[t][e] - q).max() < 1e-8, (e, t, np.abs(traj_q[t][e] - q).max())
            assert np.abs(traj_tac[t][e] - tac).max() < 1e-8 * max(np.abs(tac).max(), 1.0), (e, t)
    # a masked reset redraws the masked environments only
    mask = torch.zeros(B, dtype=torch.bool, device="cuda"); mask[::3] = True
    env.reset(mask)
    p2 = env.params.cpu().numpy()
    mk = mask.cpu().numpy()
    assert np.array_equal(p2[~mk], p[~mk]) and (p2[mk] != p[mk]).all()
