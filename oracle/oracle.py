"""ctypes front end of the fp64 CPU oracle (oracle/tsim_oracle.cpp).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by the product package. PARITY UNPINNED: see the header of tsim_oracle.cpp.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libtsim_oracle.so")
_SO_NATIVE = os.path.join(_DIR, "libtsim_oracle_native.so")
_libs = {}
_dp = C.POINTER(C.c_double)


def _src_hash():
    import hashlib
    h = hashlib.sha256()
    for p in (os.path.join(_DIR, "tsim_oracle.cpp"), os.path.join(_DIR, "..", "include", "tsim_blob.h"), os.path.join(_DIR, "Makefile")):
        h.update(open(p, "rb").read())
    return h.hexdigest()


def build(force=False):
    """Rebuild when the content hash of the sources differs from the one in the sidecar (not by file times)."""
    side = _SO + ".buildhash"
    want = _src_hash()
    have = open(side).read().strip() if os.path.exists(side) else None
    if force or not os.path.exists(_SO) or have != want:
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B", "libtsim_oracle.so"])
        open(side, "w").write(want + "\n")
    return _SO


def build_native():
    """-march=native build on the machine that will time it (bench.py cpu_baseline); always rebuilt: a copy made on
    another host must not be trusted."""
    subprocess.check_call(["make", "-C", _DIR, "-s", "-B", "libtsim_oracle_native.so"])
    return _SO_NATIVE


def lib(native=False):
    key = "native" if native else "portable"
    if key not in _libs:
        so = _SO_NATIVE if native else _SO
        if native:
            build_native()
        else:
            build()         # no-op when the sidecar hash matches the sources
        L = C.CDLL(so)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(C.c_int), _dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_reset.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.orc_set_prev.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_forward.argtypes = [C.c_void_p, _dp, C.c_int]
        L.orc_forward.restype = C.c_int
        L.orc_forward_sig.argtypes = [C.c_void_p, _dp, C.c_int, C.POINTER(C.c_uint32)]
        L.orc_forward_sig.restype = C.c_int
        L.orc_get_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_outputs.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_tape_len.argtypes = [C.c_void_p]
        L.orc_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        L.orc_set_solver.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_solver.restype = C.c_int
        L.orc_backward_steps.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
        L.orc_backward_steps.restype = C.c_int
        L.orc_get_adjoint.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_clear_adjoint.argtypes = [C.c_void_p]
        L.orc_residual.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_contact_list.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.POINTER(C.c_int), _dp]
        L.orc_contact_list.restype = C.c_int
        L.orc_inverse_dynamics.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.orc_bench_rollout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        L.orc_bench_rollout.restype = C.c_long
        _libs[key] = L
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _f(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    if n is not None and a.size != n:
        raise ValueError("expected %d values, got %d" % (n, a.size))
    return a


class OracleSim:
    """One environment, fp64, CPU. Mirrors the stepping/adjoint part of redmax_py.Simulation."""

    def __init__(self, model, native=False, solver="literal"):
        """solver: "literal" (default) — Newton + monotone backtracking exactly as the model XML states it (tol / max_iter / max_ls
        of <solver_option>, nothing else; substep_literal in tsim_oracle.cpp) — the loop the HIP kernels run since round 3;
        "r02" — the globalisation kernels and oracle shared in rounds 1-2 (non-monotone steps across kinks, restart, trust region),
        kept only to document what it did (tests/test_oracle_literal.py)."""
        self.model = model
        self._L = lib(native)
        self._I = np.ascontiguousarray(model.I, dtype=np.int32)
        self._F = np.ascontiguousarray(model.F, dtype=np.float64)
        self._h = self._L.orc_create(self._I.ctypes.data_as(C.POINTER(C.c_int)), _p(self._F))
        if not self._h:
            raise RuntimeError("oracle rejected the model blob")
        self.nr, self.nu = model.ndof_r, model.ndof_u
        self.nvar, self.ntac = model.ndof_var, model.ndof_tactile
        self.h = model.h
        self.set_solver(solver)

    def set_solver(self, solver):
        mode = {"r02": 0, "literal": 1}[solver]
        if self._L.orc_set_solver(self._h, mode) != 0:
            raise RuntimeError("oracle: bad solver mode")
        self.solver = solver

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    def reset(self, q, qd=None, record=False):
        q = _f(q, self.nr)
        qd = np.zeros(self.nr) if qd is None else _f(qd, self.nr)
        self._L.orc_reset(self._h, _p(q), _p(qd), int(record))

    def set_prev(self, q_prev, qd_prev):
        """BDF2 models: the state before the previous sub-step (after reset(); without it the next sub-step is a BDF1 start-up step)."""
        self._L.orc_set_prev(self._h, _p(_f(q_prev, self.nr)), _p(_f(qd_prev, self.nr)))

    def forward(self, u, nsub=1):
        u = _f(u, self.nu)
        return self._L.orc_forward(self._h, _p(u), int(nsub))

    def forward_sig(self, u, nsub=1):
        """forward(u, nsub) + the branch signature (count, hash) after each sub-step: (bad, int64 [nsub, 2])."""
        u = _f(u, self.nu)
        sig = np.zeros((nsub, 2), dtype=np.uint32)
        bad = self._L.orc_forward_sig(self._h, _p(u), int(nsub), sig.ctypes.data_as(C.POINTER(C.c_uint32)))
        return bad, sig.astype(np.int64)

    def state(self):
        q, qd = np.zeros(self.nr), np.zeros(self.nr)
        self._L.orc_get_state(self._h, _p(q), _p(qd))
        return q, qd

    def outputs(self, tactile=True):
        var = np.zeros(max(self.nvar, 1))
        tac = np.zeros(max(self.ntac, 1)) if tactile else None
        self._L.orc_outputs(self._h, _p(var), _p(tac))
        return var[:self.nvar], (tac[:self.ntac] if tactile else None)

    def backward_steps(self, n, df_dq=None, df_dvar=None, df_dtac=None):
        a = _f(df_dq, n * self.nr) if df_dq is not None else None
        b = _f(df_dvar, n * self.nvar) if (df_dvar is not None and self.nvar) else None
        c = _f(df_dtac, n * self.ntac) if (df_dtac is not None and self.ntac) else None
        du = np.zeros(max(n * self.nu, 1))
        rc = self._L.orc_backward_steps(self._h, int(n), _p(a), _p(b), _p(c), _p(du))
        if rc != 0:
            raise RuntimeError("oracle backward failed (%d)" % rc)
        return du[:n * self.nu].reshape(n, self.nu)

    def adjoint(self):
        a, b = np.zeros(self.nr), np.zeros(self.nr)
        self._L.orc_get_adjoint(self._h, _p(a), _p(b))
        return a, b

    def clear_adjoint(self):
        self._L.orc_clear_adjoint(self._h)

    def tape_len(self):
        return self._L.orc_tape_len(self._h)

    def stats(self):
        out = (C.c_long * 8)()
        self._L.orc_stats(self._h, out)
        return {"newton_iters": out[0], "substeps": out[1], "nonconverged": out[2], "evals": out[3],
                "kicks": out[4], "restarts": out[5], "trust_region": out[6], "ls_exhausted": out[7]}

    def residual(self, q1, q0, qd0, u, which=-1):
        g = np.zeros(self.nr)
        ncol = self.nu if which == 3 else self.nr
        J = np.zeros((self.nr, max(ncol, 1)))
        self._L.orc_residual(self._h, _p(_f(q1)), _p(_f(q0)), _p(_f(qd0)), _p(_f(u, self.nu)), int(which), _p(g), _p(J))
        return (g, J[:, :ncol]) if which >= 0 else g

    def contact_list(self, q, qd, max_rows=256):
        """Penetrating dynamics contact points of the state (q, qd): [(pair, point, branch, depth, medial distance)] (diagnostics)."""
        oi = np.zeros((max_rows, 3), dtype=np.int32)
        od = np.zeros((max_rows, 2))
        n = self._L.orc_contact_list(self._h, _p(_f(q, self.nr)), _p(_f(qd, self.nr)), max_rows, oi.ctypes.data_as(C.POINTER(C.c_int)), _p(od))
        return [(int(oi[i, 0]), int(oi[i, 1]), int(oi[i, 2]), float(od[i, 0]), float(od[i, 1])) for i in range(min(n, max_rows))]

    def inverse_dynamics(self, q, qd, qdd, u=None):
        r = np.zeros(self.nr)
        u = np.zeros(max(self.nu, 1)) if u is None else _f(u)
        self._L.orc_inverse_dynamics(self._h, _p(_f(q)), _p(_f(qd)), _p(_f(qdd)), _p(u), _p(r))
        return r

    def bench_rollout(self, q0, u_tab, nsub, with_backward):
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        u_tab = np.ascontiguousarray(u_tab, dtype=np.float64)
        nenv, nstep = u_tab.shape[0], u_tab.shape[1]
        cs = C.c_double(0)
        n = self._L.orc_bench_rollout(self._h, nenv, nstep, int(nsub), _p(q0), _p(u_tab), int(with_backward), C.byref(cs))
        return n, cs.value
