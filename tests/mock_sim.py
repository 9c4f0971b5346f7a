"""Recording mock of the `redmax_py.Simulation` surface: logs every call (name, kwargs, array shapes/dtypes) so that the
call protocol of the autograd Functions can be compared with a trace recorded from the reference's own
envs/redmax_torch_functions.py (tests/golden/protocol_trace.json, made by tools/make_protocol_fixture.py)."""
import numpy as np


def _desc(v):
    if isinstance(v, np.ndarray):
        return {"ndarray": list(v.shape), "dtype": str(v.dtype)}
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    return str(type(v).__name__)


class _Info:
    def __init__(self, log, prefix):
        object.__setattr__(self, "_log", log)
        object.__setattr__(self, "_prefix", prefix)
        object.__setattr__(self, "_vals", {})

    def __setattr__(self, k, v):
        self._log.append(["set", self._prefix + "." + k, _desc(np.asarray(v)) if not isinstance(v, np.ndarray) else _desc(v)])
        self._vals[k] = v

    def __getattr__(self, k):
        if k in ("df_dq0", "df_dqdot0", "df_du") and k in self._vals:
            self._log.append(["get", self._prefix + "." + k])
            return self._vals[k]
        raise AttributeError(k)

    def set_flags(self, **kw):
        self._log.append(["call", self._prefix + ".set_flags", {k: bool(v) for k, v in sorted(kw.items())}])


class MockSim:
    def __init__(self, ndof_r, ndof_u, ndof_var, ndof_tactile):
        self.ndof_r, self.ndof_u, self.ndof_var, self.ndof_tactile = ndof_r, ndof_u, ndof_var, ndof_tactile
        self.log = []
        self.backward_info = _Info(self.log, "backward_info")
        self.backward_results = _Info(self.log, "backward_results")
        self._k = 0

    def _rec(self, name, *a, **kw):
        self.log.append(["call", name, [_desc(x) for x in a], {k: _desc(v) for k, v in sorted(kw.items())}])

    def set_state_init(self, q, qdot): self._rec("set_state_init", q, qdot)
    def reset(self, **kw): self._rec("reset", **kw)
    def set_u(self, u): self._rec("set_u", u)
    def forward(self, n, **kw): self._rec("forward", n, **kw); self._k += 1
    def get_q(self): self._rec("get_q"); return np.full(self.ndof_r, 0.1 * self._k)
    def get_variables(self): self._rec("get_variables"); return np.full(self.ndof_var, 0.2 * self._k)
    def get_tactile_force_vector(self): self._rec("get_tactile_force_vector"); return np.full(self.ndof_tactile, 0.3 * self._k)
    def saveBackwardCache(self): self._rec("saveBackwardCache")
    def popBackwardCache(self): self._rec("popBackwardCache")

    def backward_steps(self, n):
        self._rec("backward_steps", n)
        object.__getattribute__(self.backward_results, "_vals")["df_du"] = np.ones(self.ndof_u * n)

    def backward(self):
        self._rec("backward")
        v = object.__getattribute__(self.backward_results, "_vals")
        T = object.__getattribute__(self.backward_info, "_vals")["df_du"].size // max(self.ndof_u, 1)
        v["df_dq0"], v["df_dqdot0"], v["df_du"] = np.ones(self.ndof_r), 2 * np.ones(self.ndof_r), np.ones(self.ndof_u * T)


def run_step(StepFn, torch):
    sim = MockSim(7, 6, 6, 390)
    a = torch.linspace(-0.5, 0.5, 6, dtype=torch.float64, requires_grad=True)
    q, var, tac = StepFn.apply(a, 5, sim, True)
    (q.sum() + 2 * var.sum() + 3 * tac.sum()).backward()
    return sim.log, {"action_grad": a.grad.tolist(), "q": q.tolist()}


def run_episodic(EpiFn, torch):
    sim = MockSim(12, 6, 0, 780)
    q0 = torch.zeros(12, dtype=torch.float64, requires_grad=True)
    qd0 = torch.zeros(12, dtype=torch.float64, requires_grad=True)
    act = torch.zeros(4, 6, dtype=torch.float64, requires_grad=True)
    mask = torch.tensor([False, True, False, True])
    qs, vs, ts = EpiFn.apply(q0, qd0, act, mask, sim, True)
    (qs.sum() + ts.sum()).backward()
    return sim.log, {"q0_grad": q0.grad.tolist(), "qd0_grad": qd0.grad.tolist(), "act_grad_shape": list(act.grad.shape),
                     "qs_shape": list(qs.shape), "vs_shape": list(vs.shape), "ts_shape": list(ts.shape)}
