"""Helpers of the literal-solver tests (test_oracle_literal.py on the CPU, test_gpu_literal.py on the GPU).

The oracle has two Newton drivers (oracle/tsim_oracle.cpp): "literal" (the default) — Newton + monotone backtracking exactly as the
model XML states it (envs/assets/pusher/pusher.xml:4: tol 1e-8, max_iter 100, max_ls 20), which is also the loop the HIP kernels run
since round 3 — and "r02", the globalisation kernels and oracle shared in rounds 1-2 (non-monotone steps across contact / friction
kinks, restart, trust region), kept to document what it did.  These helpers run the literal solver TEACHER-FORCED: every sub-step
starts from the state another solver (the r02 oracle, or the HIP path) was in, so a difference in one sub-step is seen as such
instead of compounding over the roll-out."""
import os
import threading

import numpy as np


def run_threads(n_items, work):
    """work(i, nthr) on min(host threads, n_items) python threads (ctypes releases the GIL inside the oracle)."""
    nthr = max(1, min(len(os.sched_getaffinity(0)), 32, n_items))
    err = []

    def guarded(i):
        try:
            work(i, nthr)
        except Exception as ex:          # surface worker failures in the test thread
            err.append(ex)
    th = [threading.Thread(target=guarded, args=(i,)) for i in range(nthr)]
    [t.start() for t in th]
    [t.join() for t in th]
    if err:
        raise err[0]


def literal_substeps(model, q_before, qd_before, u_sub, q_prev=None, qd_prev=None, has_prev=None):
    """One literal sub-step from every given state.  q_before, qd_before [N, nr] (state BEFORE the sub-step), u_sub [N, nu]; BDF2 models:
    q_prev, qd_prev [N, nr] = the state one sub-step earlier, used where has_prev [N] holds (elsewhere the sub-step is the BDF1 start-up).
    Returns q1 [N, nr], qd1 [N, nr], converged [N] bool, and the summed solver statistics."""
    from oracle.oracle import OracleSim
    N = q_before.shape[0]
    q1, qd1, ok = np.zeros_like(q_before), np.zeros_like(q_before), np.zeros(N, dtype=bool)
    stats = []

    def work(i, nthr):
        o = OracleSim(model, solver="literal")
        for j in range(i, N, nthr):
            o.reset(q_before[j], qd_before[j])
            if has_prev is not None and has_prev[j]:
                o.set_prev(q_prev[j], qd_prev[j])
            ok[j] = o.forward(u_sub[j], 1) == 0
            q1[j], qd1[j] = o.state()
        stats.append(o.stats())
    run_threads(N, work)
    tot = {k: sum(s[k] for s in stats) for k in stats[0]}
    return q1, qd1, ok, tot


def r02_rollout(model, q0, u, S):
    """Roll-out by the oracle's legacy r02 solver that keeps every sub-step's state: q [B, T*S + 1, nr], qd likewise (index 0 = initial
    state), converged [B, T*S], and per sub-step whether one of its globalisation devices acted (kick / restart / trust region)."""
    from oracle.oracle import OracleSim
    B, T = u.shape[0], u.shape[1]
    nr = q0.shape[1]
    q, qd = np.zeros((B, T * S + 1, nr)), np.zeros((B, T * S + 1, nr))
    ok, acted = np.zeros((B, T * S), dtype=bool), np.zeros((B, T * S), dtype=bool)

    def work(i, nthr):
        o = OracleSim(model, solver="r02")
        for e in range(i, B, nthr):
            o.reset(q0[e])
            q[e, 0], qd[e, 0] = o.state()
            for t in range(T):
                for s in range(S):
                    a = o.stats()
                    ok[e, t * S + s] = o.forward(u[e, t], 1) == 0
                    b = o.stats()
                    acted[e, t * S + s] = any(b[k] != a[k] for k in ("kicks", "restarts", "trust_region"))
                    q[e, t * S + s + 1], qd[e, t * S + s + 1] = o.state()
    run_threads(B, work)
    return q, qd, ok, acted


def compare_with_literal(model, q_traj, qd_traj, u, S, bdf2=False):
    """Teacher-forced comparison of a recorded trajectory (q_traj, qd_traj [B, T*S + 1, nr]; any solver) with the literal solver.
    Returns dq [B, T*S] = max_k |q1_recorded - q1_literal|, literal-converged [B, T*S], literal statistics."""
    B, n1, nr = q_traj.shape
    n = n1 - 1
    qb, qdb = q_traj[:, :-1].reshape(B * n, nr), qd_traj[:, :-1].reshape(B * n, nr)
    us = np.repeat(u, S, axis=1).reshape(B * n, -1)
    prev = {}
    if bdf2:                                    # the state one sub-step earlier (none before the first sub-step: BDF1 start-up)
        qp, qdp = np.concatenate([q_traj[:, :1], q_traj[:, :-2]], axis=1), np.concatenate([qd_traj[:, :1], qd_traj[:, :-2]], axis=1)
        hp = np.ones((B, n), dtype=bool); hp[:, 0] = False
        prev = dict(q_prev=qp.reshape(B * n, nr), qd_prev=qdp.reshape(B * n, nr), has_prev=hp.reshape(B * n))
    q1, _, ok, stats = literal_substeps(model, qb, qdb, us, **prev)
    dq = np.abs(q1.reshape(B, n, nr) - q_traj[:, 1:]).max(axis=2)
    return dq, ok.reshape(B, n), stats
