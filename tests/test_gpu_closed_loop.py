"""The closed loop in one launch each way (include/tsim_env.h tsim_push_closed_rollout / tsim_push_closed_backward, csrc/tsim_policy_push.h)
against the loop it replaces: algorithms/batched_gd.rollout_loss on BatchedTactilePushEnv — observation, policy, action mapping and
env-step as separate launches per env-step with torch autograd in between (itself pinned to the reference's TactilePushEnv and GD class on
golden vectors: tests/test_env_golden.py, tests/test_gd_loop_golden.py).  Same episode, same policy: the loss, every frame's state and
policy output, and the gradient of every policy parameter must agree."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _report import rep as _rep
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _episode(B, T, seed):
    rng = np.random.default_rng(seed)
    q0 = np.zeros((B, 7)); q0[:, 1] = -0.001; q0[:, 4] = rng.uniform(-0.02, 0.02, size=B)
    goal = np.zeros((B, 3)); goal[:, 0:2] = rng.uniform([0.15, -0.2], [0.25, 0.2], size=(B, 2))
    goal[:, 2] = rng.uniform(goal[:, 1] * np.pi - np.pi / 16.0, goal[:, 1] * np.pi + np.pi / 16.0)
    dist = rng.uniform(-1.0, 1.0, size=(T, B, 2)) * (rng.uniform(size=(T, B, 1)) < 0.5)
    return q0, goal, dist


@pytest.mark.parametrize("observation_type", ["no_tactile", "privilege"])
@pytest.mark.parametrize("lanes", [16, 64])
@pytest.mark.parametrize("dtype,tol_q,tol_g", [(torch.float64, 1e-9, 1e-7), (torch.float32, 2e-5, 2e-5)])      # policy gradient, fused vs per-step autograd, per parameter: measured 5.3e-7 (fp32), 1.3e-15 (fp64)
def test_fused_episode_with_the_other_observation_types(pusher_model, dtype, tol_q, tol_g, lanes, observation_type):
    """cfg/gd_no_tactile.yaml / gd_privilege.yaml on the fused path: 3 / 6 policy inputs; the privileged observation adds a path from
    the box pose of the state before a frame to the policy, which the adjoint launch returns to that state's adjoint."""
    test_fused_episode_equals_the_per_step_loop(pusher_model, dtype, tol_q, tol_g, lanes, observation_type)


def _randomised_tables(sim, model, B, seed=4):
    """One parameter table per environment, drawn the way the reference's reset-time randomisers draw (envs/tactile_insertion_env.py:238-281: contact /
    tactile stiffness, friction, damping; envs/stable_grasp_env.py:122: density) — here on the TactilePush records."""
    import tactilesimulation_amd.model.blob as BL
    I = model.I
    fp, fs, fd, fl = (int(I[k]) for k in (BL.TSIM_IH_FOFF_PAIR, BL.TSIM_IH_FOFF_SENSOR, BL.TSIM_IH_FOFF_DOF, BL.TSIM_IH_FOFF_LINK))
    tab = sim.base_tables()
    r = torch.rand(B, 6, generator=torch.Generator().manual_seed(seed), dtype=torch.float64).to(tab)
    tab[:, fp + BL.TSIM_PF_SIZE + BL.TSIM_PF_KN] *= 0.7 + 0.6 * r[:, 0]
    tab[:, fp + BL.TSIM_PF_SIZE + BL.TSIM_PF_MU] *= 0.5 + r[:, 1]
    tab[:, fs + BL.TSIM_SF_KN] *= 0.7 + 0.6 * r[:, 2]
    tab[:, fs + BL.TSIM_SF_KT] *= 0.7 + 0.6 * r[:, 3]
    tab[:, fd + 6 * BL.TSIM_DF_SIZE + BL.TSIM_DF_DAMPING] = 0.01 + 0.1 * r[:, 4]
    scale = 0.8 + 0.4 * r[:, 5]
    for e in (BL.TSIM_LF_MASS, BL.TSIM_LF_INERTIA, BL.TSIM_LF_INERTIA + 1, BL.TSIM_LF_INERTIA + 2):
        tab[:, fl + 3 * BL.TSIM_LF_SIZE + e] *= scale
    return tab


@pytest.mark.parametrize("lanes", [16, 64])
@pytest.mark.parametrize("dtype,tol_q,tol_g", [(torch.float64, 1e-9, 1e-7), (torch.float32, 2e-5, 2e-5)])
def test_fused_episode_with_per_environment_tables(pusher_model, dtype, tol_q, tol_g, lanes):
    """the fused closed loop on a batch with one parameter table per environment (tsim_set_env_tables) — round 5 refused it.
    Same check as below: loss, states, policy outputs and every policy parameter's gradient equal the per-step loop with torch autograd on the
    same tables; the fp32 batch at four environments per wavefront runs the structure-static closed-loop kernels (param:pusher)."""
    test_fused_episode_equals_the_per_step_loop(pusher_model, dtype, tol_q, tol_g, lanes, tables=True)


@pytest.mark.parametrize("lanes", [16, 32, 64])
@pytest.mark.parametrize("dtype,tol_q,tol_g", [(torch.float64, 1e-9, 1e-7), (torch.float32, 2e-5, 2e-5)])      # policy gradient, fused vs per-step autograd, per parameter: measured 5.3e-7 (fp32), 1.3e-15 (fp64)
def test_fused_episode_equals_the_per_step_loop(pusher_model, dtype, tol_q, tol_g, lanes, observation_type="tactile_flatten", tables=False):
    from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
    from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode
    from tactilesimulation_amd.algorithms.batched_gd import Actor, rollout_loss
    B, T = 10, 12                                                    # a batch that is no multiple of the slots per wavefront
    q0, goal, dist = (torch.tensor(a, device="cuda", dtype=dtype) for a in _episode(B, T, 3))
    torch.manual_seed(1)
    nin = {"tactile_flatten": 393, "no_tactile": 3, "privilege": 6}[observation_type]
    actor = Actor(obs_dim=nin, dtype=dtype).cuda()
    with torch.no_grad():                                            # a policy that acts (the initial one outputs ~0)
        for p in actor.parameters():
            p.mul_(3.0)
    # ---- the per-step loop with autograd
    env = BatchedTactilePushEnv(pusher_model, B, dtype=dtype, gradient=True, seed=0, tape_steps=T, observation_type=observation_type)
    env.sim.set_lanes_per_env(lanes)
    if tables:
        tab = _randomised_tables(env.sim, pusher_model, B)
        env.sim.set_env_tables(tab)
    obs = env.reset(q0, goal)
    qs, us, total = [], [], obs.new_zeros(())
    for t in range(T):
        u = actor(obs)
        obs, rew, info = env.step(u, dist[t])
        total = total - rew.sum()
        qs.append(info["q"].detach().clone()); us.append(u.detach().clone())
    named = [(n, p) for n, p in actor.named_parameters() if n != "logstd"]
    ref = torch.autograd.grad(total, [p for _, p in named])
    # ---- the fused episode
    env2 = BatchedTactilePushEnv(pusher_model, B, dtype=dtype, gradient=True, seed=0, tape_steps=T, observation_type=observation_type)
    env2.sim.set_lanes_per_env(lanes)
    if tables:
        env2.sim.set_env_tables(tab)
        assert env2.sim.kernel_variant() == ("param:pusher" if dtype == torch.float32 else "generic")
    ep = FusedPushEpisode(env2, actor, T)
    loss = ep.rollout(q0, goal, dist)
    assert int((ep.status != 0).sum()) == 0
    if tables:      # ... and the tables matter: the XML's own parameters give another trajectory
        env3 = BatchedTactilePushEnv(pusher_model, B, dtype=dtype, gradient=True, seed=0, tape_steps=T, observation_type=observation_type)
        env3.sim.set_lanes_per_env(lanes)
        ep3 = FusedPushEpisode(env3, actor, T)
        ep3.rollout(q0, goal, dist)
        assert float((ep3.q - ep.q).abs().max()) > 1e-5
        ep3.backward()
    ep.backward()
    assert env2.sim.tape_len() == 0
    q_ref, u_ref = torch.stack(qs), torch.stack(us)
    assert float((ep.q - q_ref).abs().max()) < tol_q
    assert float((ep.u - u_ref).abs().max()) < 50 * tol_q * max(1.0, float(u_ref.abs().max()))
    assert abs(float(loss) - float(total)) < 100 * tol_q * abs(float(total))
    for (n, p), r in zip(named, ref):
        err = float((p.grad - r).norm()) / max(float(r.norm()), 1e-30)
        _rep("site5_closed_loop", dtype=str(dtype), lanes=lanes, obs=observation_type, param=n, rel=err)
        assert err < tol_g, (n, err)
    assert float(torch.stack([r.norm() for r in ref]).min()) > 0.0   # every parameter does get a gradient


def test_fused_epoch_b4096_trains_and_matches_the_graphed_loop(pusher_model):
    """B = 4096 fp32, the bench's closed-loop leg: the fused episode's policy gradient equals the graphed per-step loop's on the same
    episode, and three Adam epochs reduce the loss."""
    from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
    from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode, train_epoch_fused
    from tactilesimulation_amd.algorithms.batched_gd import Actor, rollout_loss
    B, T, dt = 4096, 20, torch.float32
    q0, goal, dist = (torch.tensor(a, device="cuda", dtype=dt) for a in _episode(B, T, 5))
    torch.manual_seed(0)
    actor = Actor(dtype=dt).cuda()
    env = BatchedTactilePushEnv(pusher_model, B, dtype=dt, gradient=True, seed=0, tape_steps=T)
    total = rollout_loss(env, actor, T, q0=q0, goal=goal, disturbances=dist)
    named = [(n, p) for n, p in actor.named_parameters() if n != "logstd"]
    ref = torch.autograd.grad(total, [p for _, p in named])
    env2 = BatchedTactilePushEnv(pusher_model, B, dtype=dt, gradient=True, seed=0, tape_steps=T)
    ep = FusedPushEpisode(env2, actor, T)
    loss = ep.rollout(q0, goal, dist)
    ep.backward()
    assert abs(float(loss) - float(total)) < 1e-4 * abs(float(total))
    for (n, p), r in zip(named, ref):
        _rep("site5_closed_loop_b4096", param=n, rel=float((p.grad - r).norm()) / float(r.norm()))
        assert float((p.grad - r).norm()) <= 2e-5 * float(r.norm()), (n, float((p.grad - r).norm()) / float(r.norm()))      # measured 2.8e-7
    opt = torch.optim.Adam(actor.parameters(), lr=5e-3, betas=(0.7, 0.95))
    losses = [float(train_epoch_fused(ep, opt, q0, goal, dist, B)) / B for _ in range(4)]
    assert losses[-1] < losses[0], losses


def test_evaluate_is_the_unrecorded_rollout(pusher_model):
    from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
    from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode
    from tactilesimulation_amd.algorithms.batched_gd import Actor
    B, T = 37, 10
    q0, goal, dist = (torch.tensor(a, device="cuda", dtype=torch.float32) for a in _episode(B, T, 5))
    torch.manual_seed(2)
    actor = Actor(dtype=torch.float32).cuda()
    with torch.no_grad():
        for p in actor.parameters():
            p.mul_(3.0)
    env = BatchedTactilePushEnv(pusher_model, B, dtype=torch.float32, gradient=True, seed=0, tape_steps=T)
    ep = FusedPushEpisode(env, actor, T)
    loss = float(ep.rollout(q0, goal, dist))
    q_rec = ep.q.clone()
    ret = ep.evaluate(q0, goal, dist)
    assert env.sim.tape_len() == 0
    assert torch.equal(ep.q, q_rec)                                  # the same launch, no tape
    assert abs(float(-ret.sum()) - loss) <= 1e-6 * abs(loss) and tuple(ret.shape) == (B,)
    with pytest.raises(RuntimeError):
        ep.backward()


def test_simulator_gradients_in_the_regime_a_trained_policy_reaches():
    """The bench workload and the parity tests drive the environments with random open-loop actions.  Here: 40 epochs of the fused GD
    loop at B = 4096, then the 6 actuator inputs per env-step the trained policy produced in one more episode are replayed OPEN LOOP
    through tsim_rollout / tsim_backward_episode with the reward's own partials as seeds, and compared with the fp64 oracle (literal
    solver) on the 16 environments with the largest losses + 32 random ones: pad pressed on the box, sliding contact, |dL/du| up to 1e3
    (tools/trained_regime_grad_check.py; measured after 80 epochs: fp32 q 1.8e-6, all 48 on the oracle's branches, dL/du 1.2e-5;
    fp64 kernels 6e-14 / 2.4e-12)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import trained_regime_grad_check as T
    out = T.run(40)
    f32, f64 = out["f32"], out["f64"]
    # (the trained policy, hence the regime, depends on the summation order of the in-kernel policy: 3.5e-6 with the vector-ALU layers,
    # 5.3e-6 with the MFMA layers of round 4 — another 40-epoch trajectory, the same simulator)
    assert f32["q_err_max"] < 1e-5 and f64["q_err_max"] < 1e-10, (f32, f64)
    # What a trained policy does to 4096 environments depends on the build's fp32 roundings (40 epochs of training amplify them), so the 48
    # environments looked at are a different draw for every build.  Measured on 8 trained policies (two builds x 30 / 36 / 40 / 44 epochs,
    # tools/archive/gpu_r04.sh w2, profiles/r04_trained_regime.md): 45 - 48 of 48 fp32 trajectories on the oracle's contact / friction branches; dL/du
    # within 1.1e-5 ... 7.3e-5 of the oracle on all of them but at most ONE environment per policy (2.4e-3, 2.7e-3, 3.9e-4: an environment
    # that passes a kink inside a sub-step — the branch signature is taken at the sub-steps' ends and does not always see it).  Asserted: what
    # holds for every draw.
    assert f32["branch_agree"] >= out["subset"] - 4 and f64["branch_agree"] == out["subset"], (f32, f64)
    assert f32["grad_err_median"] < 1e-5 and f32["grad_err_within_1e4"] >= out["subset"] - 2 and f32["grad_err_second_largest"] < 1e-3 and f32["grad_err_max_all"] < 2e-2, (f32, f64)
    assert f64["grad_err_max_agreeing"] < 1e-8, (f32, f64)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 3e-5)])
def test_collection_with_the_stochastic_policy_on_normalised_observations(pusher_model, dtype, tol):
    """cfg/ppo_tactile.yaml: the same 393-64-64-3 actor, sampled (u = mean + exp(logstd) eps), on observations normalised and clipped
    with frozen statistics — inside the forward launch (FusedPushEpisode.collect), against the per-step loop in torch."""
    from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
    from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode
    from tactilesimulation_amd.algorithms.batched_gd import Actor
    B, T = 21, 12
    q0, goal, dist = (torch.tensor(a, device="cuda", dtype=dtype) for a in _episode(B, T, 7))
    g = torch.Generator(device="cpu").manual_seed(3)
    eps = torch.randn(T, B, 3, generator=g, dtype=torch.float64).to("cuda", dtype)
    mean = (torch.randn(393, generator=g, dtype=torch.float64) * 0.05).to("cuda", dtype)
    var = (torch.rand(393, generator=g, dtype=torch.float64) * 0.01 + 1e-6).to("cuda", dtype)      # small: the clip at 3 does act
    torch.manual_seed(4)
    actor = Actor(dtype=dtype).cuda()
    with torch.no_grad():
        for p in actor.parameters():
            p.mul_(2.0)
        actor.logstd.fill_(-0.7)
    norm = lambda o: torch.clamp((o - mean) / torch.sqrt(var.double() + 1e-8).to(dtype), -3.0, 3.0)
    env = BatchedTactilePushEnv(pusher_model, B, dtype=dtype, gradient=False, seed=0, tape_steps=T)
    obs = env.reset(q0, goal)
    O, U, R, Q = [], [], [], []
    with torch.no_grad():
        for t in range(T):
            O.append(obs.clone())
            u = actor(norm(obs)) + torch.exp(actor.logstd) * eps[t]
            obs, rew, info = env.step(u, dist[t])
            U.append(u.clone()); R.append(rew.clone()); Q.append(info["q"].clone())
    clipped = float((norm(torch.stack(O)).abs() == 3.0).double().mean())
    assert 0.0 < clipped < 0.9, clipped
    env2 = BatchedTactilePushEnv(pusher_model, B, dtype=dtype, gradient=False, seed=0, tape_steps=T)
    ep = FusedPushEpisode(env2, actor, T)
    out = ep.collect(q0, goal, dist, eps=eps, obs_mean=mean, obs_var=var, obs_clip=3.0)
    assert int((ep.status != 0).sum()) == 0 and env2.sim.tape_len() == 0
    sc = lambda a: max(1.0, float(a.abs().max()))
    assert float((out["q"] - torch.stack(Q)).abs().max()) < tol
    assert float((out["action"] - torch.stack(U)).abs().max()) < 50 * tol * sc(torch.stack(U))
    assert float((out["obs"] - torch.stack(O)).abs().max()) < 50 * tol * sc(torch.stack(O))
    assert float((out["reward"] - torch.stack(R)).abs().max()) < 200 * tol * sc(torch.stack(R))
    # the deterministic, un-normalised collection is evaluate()
    out2 = ep.collect(q0, goal, dist)
    ret = ep.evaluate(q0, goal, dist)
    assert torch.equal(out2["reward"].sum(0), ret)
    # statistics are for collection only: a recorded roll-out refuses them
    ep._norm = (mean.contiguous(), mean.contiguous(), 3.0)
    with pytest.raises(RuntimeError):
        ep.rollout(q0, goal, dist, record=True)
    ep._norm = None
