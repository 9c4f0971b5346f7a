#!/usr/bin/env python
"""Capture golden vectors from the REAL DiffRedMax (`redmax_py`, eanswer/DiffHand) — the hook SURVEY.md §8c specifies.

The reference's simulator source is an un-vendored submodule (`externals/DiffHand`, .gitmodules:1-3), so this repository's
parity is UNPINNED until someone with a DiffHand build runs this script once:

    cd <TactileSimulation checkout with externals/DiffHand built>      # README.md:29-34 "Install DiffRedMax"
    python /path/to/tools/capture_diffhand_golden.py --model pusher        # -> diffhand_pusher.npz
    python /path/to/tools/capture_diffhand_golden.py --model tactile_pad   # -> diffhand_tactile_pad.npz (RollingBall, forward only)

and commits the results as `tests/golden/diffhand_pusher.npz` / `diffhand_tactile_pad.npz` (< 400 KB each).  `tests/test_diffhand_golden.py` then compares
the fp64 oracle and the fp64 HIP kernels with it (it skips while the file is absent).  What is recorded, exactly as
SURVEY.md §8c lists it: TactilePush model, q0 = get_q_init() with q[1] = -0.001, q[4] = 0.01
(envs/tactile_push_env.py:134-136), a fixed closed-form 100 x 6 action table (no RNG: independent of numpy versions),
frame_skip 5 (:66); per env-step q, qdot, variables, tactile (envs/redmax_torch_functions.py:131-136); and df_du of
L = sum_t (q_t[3] + q_t[4]) through backward() with the seeding layout of :83-105.

The script talks to the binding only (no gym, no torch), so it also runs against this repository's `redmax_py` shim
(`--shim`: self-test of the hook on a GPU box; such a file is NOT a golden vector and is marked so inside).
"""
import argparse
import os
import sys

import numpy as np

T, S = 100, 5


def action_table():
    """u[t] = (planar force x, y, base torque | box disturbance x, y, z), all inside [-1, 1] (ctrl_range scaling is the
    simulator's, pusher.xml:55-57); the disturbance is piecewise constant over 10 env-steps as in tactile_push_env.py:185-193."""
    t = np.arange(T, dtype=np.float64)
    u = np.zeros((T, 6))
    u[:, 0] = 0.6 * np.sin(0.17 * t) + 0.3
    u[:, 1] = 0.5 * np.cos(0.11 * t)
    u[:, 2] = 0.3 * np.sin(0.07 * t + 1.0)
    k = np.floor(t / 10.0)
    u[:, 3] = 0.5 * np.sin(1.3 * k)
    u[:, 4] = 0.5 * np.cos(2.1 * k)
    return u


def capture(redmax, xml, source):
    sim = redmax.Simulation(xml)
    nr, nu, nv, nt = sim.ndof_r, sim.ndof_u, sim.ndof_var, sim.ndof_tactile
    q0 = np.array(sim.get_q_init(), dtype=np.float64).copy()
    q0[1], q0[4] = -0.001, 0.01
    sim.set_state_init(q0, np.zeros(nr))
    sim.reset(True)
    u = action_table()
    assert u.shape[1] == nu, "model has ndof_u = %d" % nu
    out = {"q": np.zeros((T, nr)), "qdot": np.zeros((T, nr)), "var": np.zeros((T, nv)), "tactile": np.zeros((T, nt))}
    tac0 = np.array(sim.get_tactile_force_vector(), dtype=np.float64).copy()
    for t in range(T):
        sim.set_u(u[t].copy())
        sim.forward(S, verbose=False, test_derivatives=False)
        out["q"][t] = np.array(sim.get_q()).copy()
        out["qdot"][t] = np.array(sim.get_qdot()).copy()
        out["var"][t] = np.array(sim.get_variables()).copy()
        out["tactile"][t] = np.array(sim.get_tactile_force_vector()).copy()
    # L = sum over env-steps of q[3] + q[4], seeded on the last sub-step of each env-step (redmax_torch_functions.py:83-92)
    n = T * S
    df_dq = np.zeros(n * nr)
    for t in range(T):
        df_dq[((t + 1) * S - 1) * nr + 3] = 1.0
        df_dq[((t + 1) * S - 1) * nr + 4] = 1.0
    sim.backward_info.set_flags(flag_q0=True, flag_qdot0=True, flag_p=False, flag_u=True)
    sim.backward_info.df_dq = df_dq
    sim.backward_info.df_dvar = np.zeros(n * nv)
    sim.backward_info.df_dtactile = np.zeros(n * nt)
    sim.backward_info.df_dq0 = np.zeros(nr)
    sim.backward_info.df_dqdot0 = np.zeros(nr)
    sim.backward_info.df_du = np.zeros(n * nu)
    sim.backward()
    out["df_du"] = np.array(sim.backward_results.df_du, dtype=np.float64).reshape(n, nu).copy()      # per sub-step
    out["df_dq0"] = np.array(sim.backward_results.df_dq0, dtype=np.float64).copy()
    out["df_dqdot0"] = np.array(sim.backward_results.df_dqdot0, dtype=np.float64).copy()
    out.update({"q0": q0, "u": u, "tactile0": tac0, "h": np.float64(sim.options.h), "frame_skip": np.int64(S),
                "dims": np.array([nr, nu, nv, nt], dtype=np.int64), "source": np.array(source)})
    return out


TAXEL_STRIDE = 37        # tactile_pad: every 37th of the 40 000 taxels is kept in full (1 082 taxels), plus sums over all of them


def pad_actions():
    """examples/RollingBallExp/test_sim_speed.py:43-48, verbatim as data: 350 actions."""
    acts = [[0, 0, .2]] * 100 + [[.1, 0, .2]] * 50 + [[-.2, 0, .2]] * 50 + [[0, .1, .2]] * 50 + [[0, -.2, .2]] * 100
    return np.asarray(acts, dtype=np.float64)


def capture_pad(redmax, xml, source):
    """RollingBall (BASELINE configs[0]): assets/tactile_pad/tactile_pad.xml — BDF2, free3d-exp sphere, 200 x 200 taxels — driven as
    test_sim_speed.py:63-81 drives it: reset(False), 350 x (set_u, forward(1)), tactile read-out every 5th step.  Forward only (the
    reference never differentiates this model).  Pins the [CHOICE] items the pusher cannot: BDF2 start-up, the rotation-vector joint,
    the sphere-on-plane contact point and the taxel frame of a 40 000-taxel pad.  Per read-out: the kept taxels in full, the sum of
    all forces, the number of taxels with a non-zero force and the index of the largest normal force."""
    sim = redmax.Simulation(xml)
    nr, nu, nt = sim.ndof_r, sim.ndof_u, sim.ndof_tactile
    sim.reset(False)
    A = pad_actions()
    assert A.shape[1] == nu, "model has ndof_u = %d" % nu
    n = len(A)
    reads = list(range(0, n, 5))
    keep = np.arange(0, nt // 3, TAXEL_STRIDE)
    out = {"q": np.zeros((n, nr)), "qdot": np.zeros((n, nr)), "tactile_kept": np.zeros((len(reads), len(keep), 3)),
           "tactile_sum": np.zeros((len(reads), 3)), "tactile_nonzero": np.zeros(len(reads), dtype=np.int64),
           "tactile_argmax": np.zeros(len(reads), dtype=np.int64)}
    for i in range(n):
        sim.set_u(A[i].copy())
        sim.forward(1, verbose=False, test_derivatives=False)
        out["q"][i] = np.array(sim.get_q()).copy()
        out["qdot"][i] = np.array(sim.get_qdot()).copy()
        if i % 5 == 0:
            tac = np.array(sim.get_tactile_force_vector(), dtype=np.float64).reshape(-1, 3)
            k = i // 5
            out["tactile_kept"][k] = tac[keep]
            out["tactile_sum"][k] = tac.sum(0)
            out["tactile_nonzero"][k] = int((np.abs(tac).max(1) > 0).sum())
            out["tactile_argmax"][k] = int(np.argmax(np.abs(tac[:, 2])))
    pos = sim.get_tactile_image_pos("pad")
    out.update({"u": A, "reads": np.array(reads), "kept_taxels": keep, "h": np.float64(sim.options.h), "dims": np.array([nr, nu, 0, nt], dtype=np.int64),
                "image_pos_first_last": np.array([pos[0], pos[-1]], dtype=np.int64), "source": np.array(source)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="pusher", choices=["pusher", "tactile_pad"],
                    help="pusher: SURVEY 8c's TactilePush record with gradients; tactile_pad: the RollingBall test_sim_speed sequence (forward)")
    ap.add_argument("--xml", default=None, help="default: envs/assets/pusher/pusher.xml or assets/tactile_pad/tactile_pad.xml")
    ap.add_argument("--out", default=None, help="default: diffhand_<model>.npz")
    ap.add_argument("--shim", action="store_true", help="use this repository's redmax_py shim (hook self-test; not a golden vector)")
    a = ap.parse_args()
    if a.shim:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        sys.path.insert(0, os.path.join(root, "tactilesimulation_amd", "compat"))
        import redmax_py as redmax
        source = "SHIM SELF-TEST (tactilesimulation_amd.compat.redmax_py) - NOT a DiffRedMax golden vector"
    else:
        import redmax_py as redmax
        if "tactilesimulation_amd" in (getattr(redmax, "__file__", "") or ""):
            raise SystemExit("this is the shim, not DiffRedMax: pass --shim for a self-test, or fix PYTHONPATH")
        source = "DiffRedMax redmax_py " + str(getattr(redmax, "__version__", "(no version attribute)"))
    xml = a.xml or {"pusher": "envs/assets/pusher/pusher.xml", "tactile_pad": "assets/tactile_pad/tactile_pad.xml"}[a.model]
    out = a.out or "diffhand_%s.npz" % a.model
    np.savez_compressed(out, **(capture if a.model == "pusher" else capture_pad)(redmax, xml, source))
    print("wrote", out, os.path.getsize(out), "bytes;", source)


if __name__ == "__main__":
    main()
