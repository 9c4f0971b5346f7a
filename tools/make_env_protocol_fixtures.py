"""Call protocols of the reference's environments at the `redmax_py.Simulation` boundary, recorded by running
the REFERENCE's own environment classes here in the dev container against a recording simulator stand-in (gym / cv2 / matplotlib /
redmax_py are stubs in sys.modules): every method the environment calls on the simulator during construction, reset() and step(), with
the argument values (small arrays inline) and the shapes / dtypes of what it gets back.  Runs of identical consecutive calls
(set_u / forward(1) / get_q ... inside the scripted grasps) are run-length encoded.

    dclaw         envs/dclaw_rotate_env.py        DClawRotateEnv(observation_type="tactile")              -> tests/golden/dclaw_env_protocol.json
    stable_grasp  envs/stable_grasp_env.py        StableGraspEnv(observation_type="tactile_map")          -> tests/golden/stable_grasp_env_protocol.json
    insertion     envs/tactile_insertion_env.py   TactileInsertionEnv(observation_type="tactile_flatten") -> tests/golden/insertion_env_protocol.json

The GPU tests replay the recorded calls on this repository's shim (tests/test_gpu_dclaw.py, tests/test_gpu_shim.py) — the env-level
counterpart of tests/golden/protocol_trace.json.  Data only; no reference source travels.
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
SIZES = {"dclaw": ("dclaw_position_control", 10, 9, 12, 2718), "stable_grasp": ("stable_grasp", 12, 6, 0, 780), "insertion": ("tactile_insertion", 12, 6, 0, 780)}
WHICH = sys.argv[1] if len(sys.argv) > 1 else "dclaw"


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
        LOG.append({"call": fn.__name__, "args": [_enc(np.asarray(x) if isinstance(x, (list, tuple)) else x) for x in a],
                    "kwargs": {k: _enc(v) for k, v in sorted(kw.items())}, "returns": _enc(out)})
        return out
    return w


class _Info:                                            # backward_info / backward_results of the stand-in (forward-only envs never read them)
    def set_flags(self, **kw): LOG.append({"call": "backward_info.set_flags", "args": [], "kwargs": {k: bool(v) for k, v in kw.items()}, "returns": None})


class Simulation:
    """Recording stand-in with the sizes of the model the environment loads."""
    def __init__(self, model_path, verbose=False):
        from tactilesimulation_amd.model.compiler import load_model
        from tactilesimulation_amd.workloads import asset
        name, nr, nu, nv, nt = SIZES[WHICH]
        self._meta = load_model(asset(name)).meta
        self.ndof_r, self.ndof_u, self.ndof_var, self.ndof_tactile = nr, nu, nv, nt
        self.options = types.SimpleNamespace(h=5e-3)
        self.viewer_options = types.SimpleNamespace(camera_lookat=np.zeros(3), camera_pos=np.zeros(3))
        self.backward_info, self.backward_results = _Info(), _Info()
        self._q, self._u, self._qi = np.zeros(nr), np.zeros(nu), np.zeros(nr)
        LOG.append({"call": "Simulation", "args": [os.path.relpath(model_path, REF)], "kwargs": {"verbose": verbose}, "returns": None})

    @rec
    def get_q_init(self): return self._qi.copy()
    @rec
    def set_q_init(self, q): self._qi = np.array(q, dtype=np.float64).copy()
    @rec
    def set_state_init(self, q, qdot): self._qi = np.array(q, dtype=np.float64).copy()
    @rec
    def reset(self, backward_flag=False, backward_design_params_flag=False): self._q = self._qi.copy()
    @rec
    def set_u(self, u): self._u = np.array(u, dtype=np.float64).copy()
    @rec
    def forward(self, n, verbose=False, test_derivatives=False, save_last_frame_var_only=False):
        k = min(len(self._u), 9 if WHICH == "dclaw" else 6)
        self._q[:k] = 0.7 * self._q[:k] + 0.3 * self._u[:k]
        if WHICH == "dclaw": self._q[9] += 0.05
    @rec
    def get_q(self): return self._q.copy()
    @rec
    def get_qdot(self): return np.full(self.ndof_r, 0.01)
    @rec
    def get_variables(self): return np.concatenate([np.tile([0.02, 0.01, 0.03], 3), [0.0, 0.0, 0.04]])[:self.ndof_var]
    @rec
    def get_tactile_force_vector(self): return np.tile([0.1, -0.05, -0.6], self.ndof_tactile // 3)
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
    @rec
    def update_body_density(self, name, density): pass
    @rec
    def update_body_color(self, name, color): pass
    @rec
    def update_contact_parameters(self, general_body, primitive_body, kn=None, kt=None, mu=None, damping=None): pass
    @rec
    def update_tactile_parameters(self, sensor_body, kn=None, kt=None, mu=None, damping=None): pass
    @rec
    def update_virtual_object(self, name, data): pass
    @rec
    def saveBackwardCache(self): pass
    @rec
    def popBackwardCache(self): pass
    @rec
    def clearBackwardCache(self): pass


def _stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    return m


def rle(log):
    """Collapse periodic runs: find the shortest period p (<= 8) at each position whose call names repeat >= 3 times; keep the first and the
    last repetition's entries and the count."""
    out, i = [], 0
    names = [e["call"] for e in log]
    while i < len(log):
        best = None
        for p in range(1, 9):
            n = 1
            while i + (n + 1) * p <= len(log) and names[i + n * p:i + (n + 1) * p] == names[i:i + p]:
                n += 1
            if n >= 3 and (best is None or n * p > best[0] * best[1]):
                best = (n, p)
        if best:
            n, p = best
            out.append({"repeat": n, "period": log[i:i + p], "last": log[i + (n - 1) * p:i + n * p]})
            i += n * p
        else:
            out.append(log[i]); i += 1
    return out


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
    _stub("matplotlib", __path__=[]); _stub("matplotlib.pyplot")
    _stub("redmax_py", Simulation=Simulation)
    from scipy.spatial.transform import Rotation
    _stub("scipy.spatial.transform.rotation", Rotation=Rotation)
    sys.path.insert(0, REF)
    rng = np.random.default_rng(5)
    marks, extra = {}, {}
    if WHICH == "dclaw":
        from envs.dclaw_rotate_env import DClawRotateEnv
        env = DClawRotateEnv(use_torch=False, observation_type="tactile", render_tactile=False, seed=2)
        src = "envs/dclaw_rotate_env.py DClawRotateEnv(use_torch=False, observation_type='tactile', render_tactile=False, seed=2)"
        act = lambda: rng.uniform(-1.5, 1.5, size=9)                 # beyond [-1, 1]: the env clips
        extra = {"dof_limit": env.dof_limit.tolist(), "relative_q_scale": env.relative_q_scale, "frame_skip": env.frame_skip, "q_init": env.q_init.tolist()}
    elif WHICH == "stable_grasp":
        from envs.stable_grasp_env import StableGraspEnv
        env = StableGraspEnv(use_torch=False, observation_type="tactile_map", render_tactile=False, seed=2)
        src = "envs/stable_grasp_env.py StableGraspEnv(use_torch=False, observation_type='tactile_map', render_tactile=False, seed=2)"
        act = lambda: rng.uniform(-1.0, 1.0, size=1)
    else:
        from envs.tactile_insertion_env import TactileInsertionEnv
        env = TactileInsertionEnv(use_torch=False, observation_type="tactile_flatten", render_tactile=False, seed=2, domain_randomization=True)
        src = "envs/tactile_insertion_env.py TactileInsertionEnv(use_torch=False, observation_type='tactile_flatten', render_tactile=False, seed=2, domain_randomization=True)"
        act = lambda: rng.uniform(-1.0, 1.0, size=env.ndof_u)
    marks["constructed"] = len(LOG)
    obs = env.reset()
    marks["reset"] = len(LOG)
    outs = []
    for t in range(2 if WHICH != "dclaw" else 3):
        o, r, done, info = env.step(act())
        outs.append({"obs_shape": list(np.asarray(o).shape), "reward": float(r), "done": bool(done)})
    marks["stepped"] = len(LOG)
    name = {"dclaw": "dclaw_env_protocol.json", "stable_grasp": "stable_grasp_env_protocol.json", "insertion": "insertion_env_protocol.json"}[WHICH]
    path = os.path.join(ROOT, "tests", "golden", name)
    comp = rle(LOG)
    d = {"source": src + " (reference)", "calls": len(LOG), "marks": marks, "obs_shape_after_reset": list(np.asarray(obs).shape), "steps": outs, "log": comp}
    d.update(extra)
    json.dump(d, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", len(LOG), "calls ->", len(comp), "entries;", marks)
    import collections
    print(collections.Counter(e["call"] for e in LOG))
