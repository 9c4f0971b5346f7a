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
    """BASELINE configs[3] inputs (D'Claw, dclaw_position_control.xml): q0 [B, 10], u [B, T, 9] absolute joint targets.
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


def insertion_workload(B, T, seed=7):
    """BASELINE configs[4] inputs (tactile_insertion.xml): q0 [B, 12], u [B, T, 6].  Grasp as in envs/tactile_insertion_env.py:126-170
    (height 0.2, fingers open at -0.03, closing force ramp), then drag the gripped box sideways into the hole walls; T = 9 env-steps of 5
    sub-steps = the env's 45-sub-step episode (:53)."""
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
