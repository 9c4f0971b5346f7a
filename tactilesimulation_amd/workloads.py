"""Explicit synthetic inputs shared by the parity tests, smoke() and bench.py.

"Identical seeds" is defined as identical explicit input tables: one numpy.random.default_rng(seed) stream, env-major
draw order, materialised once and fed to both the CPU oracle and the HIP path.
Mirrors the reference's TactilePush episode set-up (envs/tactile_push_env.py:133-193).
"""
import numpy as np


def push_workload(B, T, seed=0, q_init=None):
    """q0 [B,7], u [B,T,6] (already tanh-squashed robot action + random disturbance), goal [B,3]."""
    rng = np.random.default_rng(seed)
    q0 = np.zeros((B, 7)) if q_init is None else np.tile(np.asarray(q_init, dtype=np.float64), (B, 1))
    u = np.zeros((B, T, 6))
    goal = np.zeros((B, 3))
    for e in range(B):
        q0[e, 1] = -0.001                                    # tactile_push_env.py:135
        q0[e, 4] = rng.uniform(-0.02, 0.02)                  # :136
        gxy = rng.uniform([0.15, -0.2], [0.25, 0.2])         # :144
        goal[e, 0:2] = gxy
        goal[e, 2] = rng.uniform(gxy[1] * np.pi - np.pi / 16.0, gxy[1] * np.pi + np.pi / 16.0)   # :146
        ext = np.zeros(2)
        for t in range(T):
            u[e, t, 0:3] = np.tanh(rng.normal(size=3))       # policy-free open loop, :177-183
            if t % 10 == 0:                                  # :185-190
                if rng.uniform(0.0, 1.0) < 0.5:
                    ext = rng.uniform(-1.0, 1.0, 2)
                else:
                    ext = np.zeros(2)
            u[e, t, 3:5] = ext
    return q0, u, goal


def dclaw_workload(B, T, seed=7):
    """A scripted stand-in kept from rounds 1-3 for the parity tests (NOT the bench's inputs since round 4: dclaw_random_workload below is
    what SURVEY.md §8d words): q0 [B, 10], u [B, T, 9] absolute joint targets.
    A grasp of the cap: all three fingertips close on the cylinder and twist it (joint order per finger: base abduction, proximal,
    distal; limits and relative-target stepping as in envs/dclaw_rotate_env.py:23,86-96,201-207)."""
    rng = np.random.default_rng(seed)
    q0 = np.zeros((B, 10)); q0[:, [1, 4, 7]] = 0.1; q0[:, [2, 5, 8]] = 0.97
    q0[:, :9] += 0.01 * rng.normal(size=(B, 9))
    goal = np.zeros(9); goal[[0, 3, 6]] = 0.04; goal[[1, 4, 7]] = 0.1; goal[[2, 5, 8]] = 1.1
    u = np.zeros((B, T, 9))
    cur = q0[:, :9].copy()
    for t in range(T):
        cur = cur + np.clip(goal - cur, -0.02, 0.02) + 0.005 * rng.uniform(-1, 1, size=(B, 9))
        u[:, t] = cur
    return q0, u


# The settled grasp every TactileInsertion episode starts from (envs/tactile_insertion_env.py:126-170, `generate_initial_pose`): fingers open at
# -0.03 at height 0.2, 100 sub-steps to the grasp pose, the closing force ramped 0 -> 1 over 100 sub-steps, held for 300, the state lifted
# by 0.029, 500 sub-steps of settling.  Computed ONCE ("computed once by the oracle"): these are the fp64 CPU
# oracle's numbers (tests/test_reference_pins.py::test_insertion_settled_grasp_is_what_the_reference_script_produces recomputes them to 1e-12;
# the HIP path's own 1000 sub-steps land within 2e-8, tests/test_gpu_reference_pins.py::test_settled_grasp_reproduced_by_the_kernels).  Every sub-step of that script converges in <= 6 evaluations of the XML's Newton loop.
INSERTION_Q_REF = (-1.5060891958816560e-12, -1.4307747385982361e-15, 2.2585783843111251e-01, 3.4983025765399271e-17,
                   -2.2903846143541743e-02, -2.2903846134958005e-02, -5.6189946002572018e-12, 8.1323789918651949e-10,
                   2.5489690750464609e-02, -2.7179132731683355e-09, -1.6785018740271060e-14, 2.3422194581612920e-17)
INSERTION_EXECUTION_STEPS = 45                             # envs/tactile_insertion_env.py:53
INSERTION_TACTILE_FRAMES = (6, 20, 26, 32, 38, 44)         # :75-77 (tactile_initial_frame 15, 5 frames, + the reference frame)


def _rotvec_mul_z(r, angle):
    """utils/torch_utils.py rotvec_mul(r, [0, 0, angle]) for rows of rotation vectors r [B, 3] (numpy, via quaternions)."""
    th = np.linalg.norm(r, axis=1)
    ax = np.where(th[:, None] > 1e-12, r / np.maximum(th, 1e-300)[:, None], np.array([[0.0, 0.0, 1.0]]))
    qa = np.concatenate([np.cos(th / 2)[:, None], ax * np.sin(th / 2)[:, None]], axis=1)
    qb = np.stack([np.cos(angle / 2), 0 * angle, 0 * angle, np.sin(angle / 2)], axis=1)
    w = qa[:, 0] * qb[:, 0] - (qa[:, 1:] * qb[:, 1:]).sum(1)
    v = qa[:, 0:1] * qb[:, 1:] + qb[:, 0:1] * qa[:, 1:] + np.cross(qa[:, 1:], qb[:, 1:])
    n = np.linalg.norm(v, axis=1)
    ang = 2 * np.arctan2(n, w)
    return np.where(n[:, None] > 1e-12, v / np.maximum(n, 1e-300)[:, None], 0.0) * ang[:, None]


DCLAW_DOF_LIMIT = ((-0.45, 1.35), (-2.0, 2.0), (1.0, 2.0)) * 3        # envs/dclaw_rotate_env.py:78-88


def dclaw_random_workload(B, T, seed=0):
    """BASELINE configs[3] inputs as SURVEY.md §8d words them: q0 [B, 10] = the env's q_init (proximal joints -0.5, distal 0.8,
    envs/dclaw_rotate_env.py:74-77) + 0.05 N(0, 1) on the nine hand joints (:163); u [B, T, 9] = absolute joint targets of a random
    policy under relative position control: a ~ U(-1, 1)^9, target <- clip(target + 0.06 a, dof_limit) (:23,201-204).  (The env steps from
    the SIMULATED joint angles; an open-loop table steps from the previous target, which the PD motors track.)  On this walk the fingertips
    do meet the cap: by the env's own criterion — a finger's summed taxel force >= 1.0 (:131-133) — 15 % of the (finger, read-out) pairs of a
    200-step episode are in contact and the cap gets turned (oracle, 16 environments)."""
    rng = np.random.default_rng(seed)
    lim = np.asarray(DCLAW_DOF_LIMIT)
    q0 = np.zeros((B, 10)); q0[:, [1, 4, 7]] = -0.5; q0[:, [2, 5, 8]] = 0.8
    q0[:, :9] += 0.05 * rng.normal(size=(B, 9))
    u = np.zeros((B, T, 9))
    cur = q0[:, :9].copy()
    for t in range(T):
        cur = np.clip(cur + 0.06 * rng.uniform(-1.0, 1.0, size=(B, 9)), lim[:, 0], lim[:, 1])
        u[:, t] = cur
    return q0, u


def insertion_attempt_workload(B, seed=7, max_error=(0.006, 0.006, np.pi / 18.0), grasp_force=1.0):
    """BASELINE configs[4] inputs as SURVEY.md §8d words them: q0 [B, 12] = the settled grasp moved rigidly by U(+-6 mm, +-6 mm, +-0.2 mm)
    and turned by U(+-10 deg), grasp height U(-10 mm, +5 mm) (envs/tactile_insertion_env.py:200-216,174-194); u [B, 45, 6] = the joint-target
    table of ONE insertion attempt (:344-357: z lowered by 1.1 mm over the 45 sub-steps, +3 mm feed-forward, both fingers at the grasp
    force), one row per sub-step (frame_skip 1).  qd0 = 0 (:359 passes zeros)."""
    rng = np.random.default_rng(seed)
    q = np.tile(np.asarray(INSERTION_Q_REF), (B, 1))
    pos = rng.uniform([-max_error[0], -max_error[1], -0.0002], [max_error[0], max_error[1], 0.0002], size=(B, 3))
    rot = rng.uniform(-max_error[2], max_error[2], size=B)
    gh = rng.uniform(-0.01, 0.005, size=B)
    q0 = q.copy()
    q0[:, 0:3] += pos; q0[:, 6:9] += pos
    q0[:, 2] += gh
    q0[:, 3] += rot
    q0[:, 9:12] = _rotvec_mul_z(q[:, 9:12], rot)
    return q0, insertion_attempt_table(q0, grasp_force)


def insertion_attempt_table(q0, grasp_force=1.0):
    """q0 [B, 12] pre-grasp states -> the joint-target table [B, 45, 6] of one insertion attempt (envs/tactile_insertion_env.py:344-357)."""
    init = np.asarray(q0, dtype=np.float64)[:, :6]
    target = init.copy(); target[:, 2] -= 0.0011
    frac = (np.arange(1, INSERTION_EXECUTION_STEPS + 1) / INSERTION_EXECUTION_STEPS)[None, :, None]
    u = (target - init)[:, None, :] * frac + init[:, None, :]
    u[:, :, 2] += 0.003
    u[:, :, 4] = grasp_force; u[:, :, 5] = grasp_force
    return u.astype(np.float32).astype(np.float64)          # the reference assembles the table in a float32 tensor (:345)


def insertion_workload(B, T, seed=7):
    """A HARDER synthetic stand-in kept from rounds 1-3 (NOT the reference's episode, which starts from the settled grasp:
    insertion_attempt_workload above): q0 [B, 12], u [B, T, 6] close the grasp INSIDE the episode — height 0.2, fingers open at -0.03, the
    closing force ramped over 5 env-steps instead of the reference's 100 sub-steps — then drag the gripped box sideways into the hole
    walls.  The two sub-steps in which the fingers meet the box at ~1 m/s are where the XML's Newton loop creeps (parity tests keep it as
    the stress case of the solver)."""
    rng = np.random.default_rng(seed)
    q0 = np.zeros((B, 12)); q0[:, 2] = 0.2; q0[:, 4] = -0.03; q0[:, 5] = -0.03
    q0[:, 6:8] += 5e-4 * rng.normal(size=(B, 2))
    u = np.zeros((B, T, 6))
    for t in range(T):
        a = min(1.0, (t + 1) / 5.0)
        drift = 0.004 * max(0.0, (t - 5) / 8.0)
        u[:, t] = np.array([drift, 0.6 * drift, 0.2, 0.05 * max(0.0, (t - 7) / 6.0), a, a]) + \
            np.concatenate([1e-4 * rng.normal(size=(B, 3)), np.zeros((B, 3))], axis=1)
    return q0, u


import os as _os

ASSETS = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "assets")
PUSHER_BLOB = _os.path.join(ASSETS, "pusher.npz")      # envs/assets/pusher/pusher.xml compiled (tools/make_model_fixtures.py)


def asset(name):
    """Path of a precompiled model blob: pusher, dclaw_position_control, tactile_insertion, stable_grasp, tactile_pad."""
    return _os.path.join(ASSETS, name + ".npz")


def synthetic_variant(name):
    """The sizes BASELINE.json words differently from the reference's assets, as spec edits of the real
    models — same bodies, joints and contacts, only the taxel layout changes:
      pusher_13x13            TactilePush pad with resolution 13 x 13 (the XML has 13 x 10, pusher.xml:61)
      dclaw_9x9               D'Claw with 9 x 9 taxels per fingertip: 81 of the 302 abstract taxels of each finger (evenly
                              strided), image positions re-gridded row-major into 9 x 9
      tactile_insertion_32x32 both pads with resolution 32 x 32 (the XML has 13 x 10, tactile_insertion.xml:98-99)
    Returns a CompiledModel."""
    import copy
    import numpy as _np
    from .model.compiler import load_model, compile_spec
    base = {"pusher_13x13": "pusher", "dclaw_9x9": "dclaw_position_control", "tactile_insertion_32x32": "tactile_insertion"}[name]
    spec = copy.deepcopy(load_model(asset(base)).spec)
    for s in spec["sensors"]:
        if name == "pusher_13x13":
            s["resolution"] = [13, 13]
        elif name == "tactile_insertion_32x32":
            s["resolution"] = [32, 32]
        else:
            keep = _np.round(_np.linspace(0, len(s["taxels"]) - 1, 81)).astype(int)
            s["taxels"] = [dict(s["taxels"][k], img=[i // 9, i % 9]) for i, k in enumerate(keep)]
    return compile_spec(spec)
