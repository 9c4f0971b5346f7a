"""Call protocol of the reference's D'Claw environment (BASELINE configs[3]) at the `redmax_py.Simulation` boundary,
recorded by running the REFERENCE's own `envs/dclaw_rotate_env.py::DClawRotateEnv` (position control, relative actions, observation_type
"tactile") here in the dev container against a recording simulator stand-in: every method the environment calls on the simulator during
construction, reset() and step(), with the argument values (small arrays inline), and the shapes / dtypes of what it gets back.

tests/test_gpu_dclaw.py::test_reference_env_call_protocol_replays_on_the_shim replays the recorded calls on this repository's shim and
checks that each one is accepted and returns the recorded shape — the env-level counterpart of tests/golden/protocol_trace.json.
Writes tests/golden/dclaw_env_protocol.json (data only).
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("TSIM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
LOG = []
rng = np.random.default_rng(5)


def _enc(v):
    if isinstance(v, np.ndarray):
        return {"shape": list(v.shape), "dtype": str(v.dtype), "values": v.reshape(-1).tolist()} if v.size <= 32 else {"shape": list(v.shape), "dtype": str(v.dtype)}
    if isinstance(v, (list, tuple)):
        return {"list": len(v), "first": _enc(v[0]) if len(v) else None}
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, (np.floating, np.integer)):
        return float(v)
    return type(v).__name__


def rec(fn):
    def w(self, *a, **kw):
        out = fn(self, *a, **kw)
        LOG.append({"call": fn.__name__, "args": [_enc(np.asarray(x) if isinstance(x, (list, tuple)) and fn.__name__ != "get_tactile_image_pos" else x) for x in a],
                    "kwargs": {k: _enc(v) for k, v in sorted(kw.items())}, "returns": _enc(out)})
        return out
    return w


class Simulation:
    """Recording stand-in with D'Claw's sizes (dclaw_position_control.xml: ndof_r 10, ndof_u 9, 4 end-effectors, 3 x 302 taxels)."""
    def __init__(self, model_path, verbose=False):
        from tactilesimulation_amd.model.compiler import load_model
        from tactilesimulation_amd.workloads import asset
        self._meta = load_model(asset("dclaw_position_control")).meta
        self.ndof_r, self.ndof_u, self.ndof_var, self.ndof_tactile = 10, 9, 12, 2718
        self.options = types.SimpleNamespace(h=5e-3)
        self.viewer_options = types.SimpleNamespace(camera_lookat=np.zeros(3), camera_pos=np.zeros(3))
        self._q = np.zeros(10)
        LOG.append({"call": "Simulation", "args": [os.path.relpath(model_path, REF)], "kwargs": {"verbose": verbose}, "returns": None})

    @rec
    def get_q_init(self): return np.zeros(10)
    @rec
    def set_q_init(self, q): pass
    @rec
    def set_state_init(self, q, qdot): self._q = np.array(q, dtype=np.float64).copy()
    @rec
    def reset(self, backward_flag=False): pass
    @rec
    def set_u(self, u): self._u = np.array(u, dtype=np.float64).copy()
    @rec
    def forward(self, n, verbose=False, test_derivatives=False, save_last_frame_var_only=False): self._q[:9] = 0.7 * self._q[:9] + 0.3 * self._u; self._q[9] += 0.05
    @rec
    def get_q(self): return self._q.copy()
    @rec
    def get_qdot(self): return np.full(10, 0.01)
    @rec
    def get_variables(self): return np.concatenate([np.tile([0.02, 0.01, 0.03], 3), [0.0, 0.0, 0.04]])
    @rec
    def get_tactile_flow_images(self): return [[[[0.5, 0.0, -1.0] for _ in range(20)] for _ in range(20)] for _ in range(3)]
    @rec
    def get_tactile_image_pos(self, name): return [tuple(p) for p in self._meta["image_pos"][name]]
    @rec
    def update_joint_damping(self, name, damping): pass
    @rec
    def update_body_size(self, name, size): pass
    @rec
    def update_endeffector_position(self, name, pos): pass
    @rec
    def update_joint_location(self, name, pos): pass


def _stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    return m


if __name__ == "__main__":
    class _Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None): self.low, self.high, self.shape = low, high, shape
    spaces = _stub("gym.spaces", Box=_Box)
    seeding = _stub("gym.utils.seeding", np_random=lambda seed=None: (np.random.RandomState(seed), seed))
    _stub("gym.utils", seeding=seeding)
    _stub("gym", Env=object, spaces=spaces, utils=sys.modules["gym.utils"], __path__=[])
    _stub("gym.envs", __path__=[])
    _stub("gym.envs.registration", registry=types.SimpleNamespace(env_specs={}), register=lambda **kw: None, make=None, spec=None)
    _stub("cv2")
    _stub("redmax_py", Simulation=Simulation)
    from scipy.spatial.transform import Rotation
    _stub("scipy.spatial.transform.rotation", Rotation=Rotation)
    sys.path.insert(0, REF)
    from envs.dclaw_rotate_env import DClawRotateEnv                 # the reference's class

    marks = {}
    env = DClawRotateEnv(use_torch=False, observation_type="tactile", render_tactile=False, seed=2)
    marks["constructed"] = len(LOG)
    obs = env.reset()
    marks["reset"] = len(LOG)
    outs = []
    for t in range(3):
        o, r, done, info = env.step(rng.uniform(-1.5, 1.5, size=9))        # beyond [-1, 1]: the env clips
        outs.append({"obs_shape": list(np.asarray(o).shape), "reward": float(r), "done": bool(done), "success": bool(info["success"])})
    marks["stepped"] = len(LOG)
    path = os.path.join(ROOT, "tests", "golden", "dclaw_env_protocol.json")
    json.dump({"source": "envs/dclaw_rotate_env.py DClawRotateEnv(use_torch=False, observation_type='tactile', render_tactile=False, seed=2) (reference)",
               "marks": marks, "obs_shape_after_reset": list(np.asarray(obs).shape), "steps": outs, "dof_limit": env.dof_limit.tolist(),
               "relative_q_scale": env.relative_q_scale, "frame_skip": env.frame_skip, "q_init": env.q_init.tolist(), "log": LOG}, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", len(LOG), "calls;", marks, [o["reward"] for o in outs])
    for e in LOG[:marks["reset"]]:
        print(e["call"], json.dumps(e["args"])[:100], json.dumps(e["kwargs"])[:60])
