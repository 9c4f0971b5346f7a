"""Why 3 of the 2048 D'Claw environments of BASELINE configs[3] never converge: replay, in the fp64 oracle, the sub-step of
each that ends at max_iter (environments 529, 700, 1855 of workloads.dclaw_random_workload(2048, 50, seed=7)), log the literal Newton loop iteration by
iteration, and look at ||g|| along the Newton direction at the point it ends on.  CPU only (oracle = checker; this is a diagnosis tool, not product).

    python tools/dclaw_nonconv_probe.py [env ...] > profiles/r06_dclaw_nonconv.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import OracleSim                                     # noqa: E402
from tactilesimulation_amd.model.compiler import load_model            # noqa: E402
from tactilesimulation_amd import workloads as W                        # noqa: E402


def literal_loop(o, q0, qd0, u, tol, max_iter, max_ls, log):
    """oracle/tsim_oracle.cpp substep_literal restated on orc_residual (BDF1): returns (q1, converged)."""
    h = o.h
    q1 = q0 + h * qd0
    for it in range(max_iter + 1):
        g, H = o.residual(q1, q0, qd0, u, which=0)
        gn = float(np.linalg.norm(g))
        if gn < tol:
            return q1, True
        if it == max_iter:
            break
        dq = np.linalg.solve(H, -g)
        alpha, ls, gtr = 1.0, 0, None
        trials = []
        while True:
            qn = q1 + alpha * dq
            gtr = float(np.linalg.norm(o.residual(qn, q0, qd0, u)))
            trials.append(gtr)
            if gtr < gn or ls >= max_ls:
                break
            alpha *= 0.5
            ls += 1
        log.append({"it": it, "gnorm": gn, "step_inf": float(np.abs(dq).max()), "step_dof": int(np.abs(dq).argmax()), "ls": ls, "accepted": bool(gtr < gn), "g_after": gtr,
                    "cond_H": float(np.linalg.cond(H))})
        q1 = qn
    return q1, False


def main():
    envs = [int(a) for a in sys.argv[1:]] or [529, 700, 1855]
    model = load_model(W.asset("dclaw_position_control"))
    I = np.asarray(model.I)
    q0s, us = W.dclaw_random_workload(2048, 50, seed=7)
    out = []
    for e in envs:
        o = OracleSim(model)
        o.reset(q0s[e])
        found = None
        for t in range(50):
            for s in range(5):
                q, qd = o.state()
                n0 = o.stats()["nonconverged"]
                ev0 = o.stats()["evals"]
                o.forward(us[e, t], 1)
                if o.stats()["nonconverged"] > n0:
                    found = (t, s, q.copy(), qd.copy(), us[e, t].copy(), o.stats()["evals"] - ev0)
                    break
            if found:
                break
        if not found:
            out.append({"env": e, "nonconverged": False})
            continue
        t, s, q, qd, u, nev = found
        log = []
        q1, ok = literal_loop(o, q, qd, u, 1e-8, 100, 20, log)
        # ||g|| along the last Newton direction, both signs, log-spaced: is the end point a kink minimum of ||g||?
        g, H = o.residual(q1, q, qd, u, which=0)
        dq = np.linalg.solve(H, -g)
        scan = [{"alpha": float(a), "gnorm": float(np.linalg.norm(o.residual(q1 + a * dq, q, qd, u)))} for a in
                [-1e-2, -1e-4, -1e-6, 0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.1, 0.25, 0.5, 1.0, 1.5, 2.0]]
        # one-sided directional derivatives of g along dq (a kink: they differ)
        eps = 1e-7
        gp = (o.residual(q1 + eps * dq, q, qd, u) - g) / eps
        gm = (g - o.residual(q1 - eps * dq, q, qd, u)) / eps
        # the penetrating contact points just before and just after the jump of ||g|| along dq: which point changes its smooth piece?
        h = o.h
        def contacts(a):
            qa = q1 + a * dq
            return o.contact_list(qa, (qa - q) / h)
        jump_lo, jump_hi = 0.0, 1.0
        g_lo = float(np.linalg.norm(o.residual(q1, q, qd, u)))
        for _ in range(60):                      # bisect the first discontinuity of ||g|| on (0, 1]
            mid = 0.5 * (jump_lo + jump_hi)
            gm_ = float(np.linalg.norm(o.residual(q1 + mid * dq, q, qd, u)))
            if abs(gm_ - g_lo) > 0.05 * g_lo: jump_hi = mid
            else: jump_lo = mid
        ca, cb = contacts(jump_lo), contacts(jump_hi)
        changed = [{"pair": a[0], "point": a[1], "branch_before": a[2], "branch_after": b[2], "depth": a[3], "medial_distance_before": a[4], "medial_distance_after": b[4]}
                   for a, b in zip(ca, cb) if a[:2] == b[:2] and a[2] != b[2]] if len(ca) == len(cb) else "contact set changes: %d -> %d points" % (len(ca), len(cb))
        g_jump = o.residual(q1 + jump_hi * dq, q, qd, u) - o.residual(q1 + jump_lo * dq, q, qd, u)
        lsx = sum(1 for r in log if not r["accepted"])
        out.append({"env": e, "env_step": t, "sub_step": s, "oracle_evals_in_substep": nev, "converged_in_replay": ok, "iterations": len(log), "line_searches_exhausted": lsx,
                    "gnorm_first": log[0]["gnorm"], "gnorm_min": min(r["gnorm"] for r in log), "gnorm_last": log[-1]["gnorm"],
                    "q0": q.tolist(), "qd0": qd.tolist(), "u": u.tolist(), "q_end": q1.tolist(),
                    "first_iterations": log[:6], "last_iterations": log[-6:],
                    "gnorm_along_last_newton_direction": scan,
                    "jump": {"alpha_before": jump_lo, "alpha_after": jump_hi, "jump_of_g": g_jump.tolist(), "jump_norm": float(np.linalg.norm(g_jump)),
                             "penetrating_points": len(ca), "points_that_change_branch": changed,
                             "branch_code": "bit 0 = sticking, bits 1.. = face of the primitive (cylinder: 0 side, 1 / 2 caps)"},
                    "Hdq_plus_g_rel": float(np.linalg.norm(H @ dq + g) / max(np.linalg.norm(g), 1e-300)),
                    "directional_derivative_mismatch": float(np.linalg.norm(gp - gm) / max(np.linalg.norm(gp), 1e-300)),
                    "dd_plus_vs_H": float(np.linalg.norm(gp - H @ dq) / max(np.linalg.norm(H @ dq), 1e-300)),
                    "dd_minus_vs_H": float(np.linalg.norm(gm - H @ dq) / max(np.linalg.norm(H @ dq), 1e-300))})
        print("env %d: step %d.%d, %d oracle evals, replay %s after %d iterations (%d exhausted line searches), ||g|| %.3e -> min %.3e -> %.3e" % (
            e, t, s, nev, "converged" if ok else "NOT converged", len(log), lsx, log[0]["gnorm"], min(r["gnorm"] for r in log), log[-1]["gnorm"]), file=sys.stderr)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
