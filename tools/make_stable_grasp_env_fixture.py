"""Golden vectors of the reference's StableGrasp environment arithmetic: the REFERENCE's own
`envs/stable_grasp_env.py::StableGraspEnv` (observation_type "tactile_flatten") run here in the dev container against a SCRIPTED simulator,
recording per env-step (= one five-stage grasp of 180 sub-steps): the grasp position the action leads to, the state the grasp starts from,
the 180 x 6 joint-target table, the sub-step at which the tactile frame is captured, the observation, reward, done, success — and the
block densities drawn at reset.  Writes tests/golden/stable_grasp_env.npz (data only)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("TSIM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
rng = np.random.default_rng(44)
QREF = np.array([0.0, 0.0, 0.2029862, 0.0, -0.03, -0.03, 0.0, 0.0, 0.015, 0.0, 0.0, 0.0])
STATE = {"episode": -1, "k": 0}
LOG = {"set_u": [], "tac_at": [], "state_init": [], "density": [], "color": []}
NSTEP = 5


def script_episode(i):
    qs = np.tile(QREF, (180, 1)) + rng.normal(size=(180, 12)) * 0.001
    qs[60, 9:12] = [0.004, -0.003, 0.002] if i in (2, 4) else [0.08, -0.05, 0.02]       # tilt of the bar at the captured frame
    qs[60, 8] = 0.02 if i != 4 else 0.001                                               # episode 4: level but not lifted -> no success
    tac = rng.normal(size=780) * (rng.uniform(size=780) < 0.4)
    return qs, tac


class Simulation:
    def __init__(self, model_path, verbose=False):
        self.ndof_r, self.ndof_u, self.ndof_var, self.ndof_tactile = 12, 6, 0, 780
        self.options = types.SimpleNamespace(h=5e-3)
        self.viewer_options = types.SimpleNamespace(camera_lookat=np.zeros(3), camera_pos=np.zeros(3))
        self.backward_info = types.SimpleNamespace(set_flags=lambda **kw: None)
        self._qi = np.zeros(12)
    def get_q_init(self): return self._qi.copy()
    def set_q_init(self, q): self._qi = np.array(q, dtype=np.float64).copy()
    def set_state_init(self, q, qdot):
        LOG["state_init"].append(np.array(q, dtype=np.float64).copy())
        STATE["episode"] += 1; STATE["k"] = 0
        STATE["qs"], STATE["tac"] = script_episode(STATE["episode"])
        LOG["set_u"].append([]); LOG["tac_at"].append([])
    def reset(self, backward_flag=False): pass
    def set_u(self, u):
        if STATE["episode"] >= 0: LOG["set_u"][-1].append(np.array(u, dtype=np.float64).copy())
    def forward(self, n, verbose=False, test_derivatives=False, save_last_frame_var_only=False):
        if STATE["episode"] >= 0: STATE["k"] += 1
    def get_q(self): return QREF.copy() if STATE["episode"] < 0 else STATE["qs"][STATE["k"] - 1].copy()
    def get_qdot(self): return np.zeros(12)
    def get_variables(self): return np.zeros(0)
    def get_tactile_force_vector(self):
        LOG["tac_at"][-1].append(STATE["k"] - 1)
        return STATE["tac"].copy()
    def clearBackwardCache(self): pass
    def saveBackwardCache(self): pass
    def update_body_density(self, name, d): LOG["density"].append((name, float(d)))
    def update_body_color(self, name, c): LOG["color"].append((name, np.array(c).tolist()))


def _stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    return m


if __name__ == "__main__":
    import torch

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
    from envs.stable_grasp_env import StableGraspEnv                  # the reference's class
    StableGraspEnv.render = lambda self, mode="once": None            # step() calls render('loop') on success: viewer, out of scope

    env = StableGraspEnv(use_torch=True, observation_type="tactile_flatten", render_tactile=False, seed=9)
    obs0 = env.reset()
    U = np.random.default_rng(45).uniform(-1.5, 1.5, size=(NSTEP, 1))      # its own stream: the scripted episodes are regenerated below
    rec = {"obs": [obs0.numpy().copy()], "grasp_position": [float(env.grasp_position)], "reward": [float(env.reward_buf)], "done": [bool(env.done_buf)], "success": [bool(env.is_success)]}
    for t in range(NSTEP):
        o, r, d, info = env.step(torch.tensor(U[t]))
        rec["obs"].append(o.numpy().copy()); rec["grasp_position"].append(float(env.grasp_position)); rec["reward"].append(float(r)); rec["done"].append(bool(d)); rec["success"].append(bool(info["success"]))
    out = {k: np.array(v) for k, v in rec.items()}
    out["actions"] = np.array(LOG["set_u"]); out["tactile_substeps"] = np.array(LOG["tac_at"]); out["state_init"] = np.array(LOG["state_init"])
    out["density_names"] = np.array([n for n, _ in LOG["density"]]); out["densities"] = np.array([d for _, d in LOG["density"]])
    rng = np.random.default_rng(44)
    eps = [script_episode(i) for i in range(NSTEP + 1)]
    out["script_qs"] = np.array([e[0] for e in eps]); out["script_tactile"] = np.array([e[1] for e in eps])
    out.update({"u": U, "q_ref": QREF, "qpos_init_reference": env.qpos_init_reference.numpy(), "action_scale": np.float64(env.action_scale), "grasp_position_bound": np.float64(env.grasp_position_bound)})
    path = os.path.join(ROOT, "tests", "golden", "stable_grasp_env.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; obs", out["obs"].shape, "success", out["success"], "reward", np.round(out["reward"], 3), "grasp_position", np.round(out["grasp_position"], 4))
    print("tactile sub-steps", out["tactile_substeps"].reshape(-1), "densities", np.round(out["densities"], 1), out["density_names"][:3])
