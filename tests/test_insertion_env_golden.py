"""TactileInsertion environment arithmetic (BASELINE configs[4]) against golden vectors recorded from the REFERENCE's
own TactileInsertionEnv run against a scripted simulator (tools/make_insertion_env_fixture.py -> tests/golden/insertion_env.npz).
CPU: the pure functions of envs/tactile_insertion.py.  GPU: the batched environment on the real simulator."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "insertion_env.npz"))
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64)


def test_pure_functions_match_the_reference_env():
    from tactilesimulation_amd.envs import tactile_insertion as E
    assert list(np.nonzero(G["tactile_masks"])[0]) == list(E.TACTILE_FRAMES) and np.allclose(G["max_error"], E.MAX_ERROR)
    assert np.array_equal(G["action_scale"].astype(np.float64), [E.XY_SCALE, E.XY_SCALE, E.ROT_SCALE]) and np.array_equal(G["working_space_boundary"].astype(np.float64), [E.WORKSPACE_XY] * 2)
    n = len(G["u"])
    for rt in ("absolute", "delta"):
        qi = G[rt + "/q_init"]                                            # pre-grasp state of episode k (0: after reset)
        # reset: the reference pose moved by the recorded noise
        pn = G["reset_noise"]
        # (the reference wraps the two python-float noises in torch.tensor(), i.e. float32: they reach q rounded to float32)
        f32 = lambda v: t(np.float32(v).astype(np.float64))
        q0 = E.apply_relative_motion(t(G["q_init_reference"][None]), t(pn[None, 0:3]), f32(pn[3:4]), f32(pn[4:5]))
        assert np.abs(q0.numpy()[0] - qi[0]).max() < 1e-15
        assert np.array_equal(G[rt + "/state_init"], qi)                  # what EpisodicSimFunction handed to set_state_init
        # step k: action -> clipped relative motion -> new pre-grasp state
        dxy, drot = E.relative_motion_of_action(t(G["u"]), t(qi[:n]))
        q1 = E.apply_relative_motion(t(qi[:n]), dxy, drot)
        assert np.abs(q1.numpy() - qi[1:]).max() < 1e-15
        # the joint-target table of every attempt, the frames it captures
        act = E.insertion_actions(t(qi), 1.0).permute(1, 0, 2)
        assert np.abs(act.numpy() - G[rt + "/actions"]).max() < 1e-15
        assert all(list(r) == list(E.TACTILE_FRAMES) for r in G[rt + "/tactile_substeps"])
        # observation from the scripted tactile frames
        obs = E.observation(t(G["script_tactile"]).permute(1, 0, 2))
        assert np.abs(obs.numpy() - G[rt + "/obs"]).max() < 1e-12
        # reward / success / done
        prev = np.concatenate([qi[:1, [0, 1, 3]], qi[:-1, [0, 1, 3]]])   # pose before each attempt (reset: its own)
        r, succ, _ = E.reward_done(t(qi), t(G["script_qs"][:, -1]), t(prev), rt, True)
        assert np.abs(r.numpy() - G[rt + "/reward"]).max() < 1e-10 and np.array_equal(succ.numpy(), G[rt + "/success"])
        assert np.array_equal(G[rt + "/done"], G[rt + "/success"]) and G[rt + "/success"].sum() == 2


@pytest.mark.gpu
def test_batched_env_on_the_simulator():
    from tactilesimulation_amd.envs import tactile_insertion as E
    from tactilesimulation_amd.host.batch import BatchSim
    B = 16
    env = E.BatchedTactileInsertionEnv(B, dtype=torch.float64, seed=2, reward_type="delta")
    qr = env.q_init_reference[0].cpu().numpy()
    assert abs(qr[2] - 0.229) < 5e-3 and abs(qr[4] - qr[5]) < 1e-6 and -0.03 < qr[4] < 0.0 and 0.0 < qr[8] < 0.06, qr     # a closed, symmetric grasp, lifted
    obs = env.reset()
    assert obs.shape == (B, 2600) and bool(torch.isfinite(obs).all()) and float(obs.abs().max()) <= 30.0 + 1e-6
    u = torch.tensor(np.random.default_rng(0).uniform(-1.2, 1.2, size=(B, 3)), device="cuda")
    q_before = env.current_q_init.clone()
    obs, r, done, info = env.step(u)
    assert int((info["status"] != 0).sum()) == 0 and r.shape == (B,) and bool(torch.isfinite(r).all())
    assert float((env.current_q_init[:, 0:2].abs()).max()) <= E.WORKSPACE_XY + 1e-12
    # one environment alone gives the same attempt bit for bit
    e = 5
    one = BatchSim(env.model, 1, dtype=torch.float64, tape_capacity=0)
    one.reset(env.current_q_init[e:e + 1], None, backward_flag=False)
    ro = one.rollout(E.insertion_actions(env.current_q_init[e:e + 1], 1.0), 1, want_var=False, tactile_mask=env.mask)
    assert torch.equal(E.observation(ro["tactile"])[0], obs[e])
    # masked reset keeps the others' pre-grasp state
    m = torch.zeros(B, dtype=torch.bool, device="cuda"); m[::4] = True
    keep = env.current_q_init.clone()
    env.reset(m)
    assert torch.equal(env.current_q_init[~m], keep[~m]) and not torch.equal(env.current_q_init[m], keep[m])
    # step(u, reset=mask): the masked environments start a new episode inside the same launch, the others take their action
    keep, steps = env.current_q_init.clone(), env.steps.clone()
    o3, r3, d3, _ = env.step(u, reset=m)
    moved = E.apply_relative_motion(keep, *E.relative_motion_of_action(u, keep))
    assert torch.equal(env.current_q_init[~m], moved[~m]) and not torch.equal(env.current_q_init[m], moved[m])
    assert torch.equal(env.steps[~m], steps[~m] + 1) and int(env.steps[m].max()) == 0 and not bool(d3[m].any())
    # domain randomisation: per-environment tables, still converging
    env2 = E.BatchedTactileInsertionEnv(B, dtype=torch.float32, seed=3, domain_randomization=True)
    o2 = env2.reset()
    assert bool(torch.isfinite(o2).all()) and float(env2.grasp_force.min()) >= 0.125 and float(env2.grasp_force.max()) <= 0.8
    c = env2.model.table_offset("pair", ("tactile_pad_left", "box"), "kn")
    assert float(env2.tables[:, c].min()) >= 2e3 and float(env2.tables[:, c].max()) <= 14e3 and float(env2.tables[:, c].std()) > 0
