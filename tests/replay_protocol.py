"""Replay of a recorded environment call protocol (tests/golden/*_env_protocol.json, tools/make_env_protocol_fixtures.py) on a
`redmax_py.Simulation`-like object: every recorded call is issued with the recorded arguments and its return value is checked against the
recorded shape / dtype / list structure.  Run-length-encoded runs keep the arguments of their first and last repetition; the ones in
between are interpolated linearly (the environments' scripted motions are linear ramps)."""
import numpy as np


def _dec(a):
    if isinstance(a, dict) and "values" in a:
        return np.array(a["values"], dtype=a["dtype"]).reshape(a["shape"])
    return a


def _mix(a, b, w):
    a, b = _dec(a), _dec(b)
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.shape == b.shape and a.dtype.kind == "f":
        return (1.0 - w) * a + w * b
    return a


def _check(name, out, want):
    if isinstance(want, dict) and "shape" in want:
        out = np.asarray(out)
        assert list(out.shape) == want["shape"] and str(out.dtype) == want["dtype"], (name, out.shape, out.dtype, want)
        assert np.all(np.isfinite(out)), name
    elif isinstance(want, dict) and "list" in want:
        assert len(out) == want["list"], (name, len(out), want["list"])
    else:
        assert out is None, (name, type(out))


def replay(make_sim, log, on_call=None):
    """make_sim(model_relpath, verbose) -> simulator.  Returns (sim, {call name: count})."""
    sim, seen = None, {}

    def issue(e, args, kw):
        nonlocal sim
        name = e["call"]
        seen[name] = seen.get(name, 0) + 1
        if name == "Simulation":
            sim = make_sim(args[0], kw.get("verbose", False))
            return
        obj, meth = sim, name
        if "." in name:                                            # backward_info.set_flags
            o, meth = name.split("."); obj = getattr(sim, o)
        out = getattr(obj, meth)(*args, **kw)
        _check(name, out, e["returns"])
        if on_call:
            on_call(sim, name, out)
    for e in log:
        if "repeat" not in e:
            issue(e, [_dec(a) for a in e["args"]], {k: _dec(v) for k, v in e["kwargs"].items()})
            continue
        n = e["repeat"]
        for r in range(n):
            w = r / max(n - 1, 1)
            for first, last in zip(e["period"], e["last"]):
                issue(first, [_mix(a, b, w) for a, b in zip(first["args"], last["args"])], {k: _dec(v) for k, v in first["kwargs"].items()})
    return sim, seen
