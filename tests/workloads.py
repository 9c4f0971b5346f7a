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
