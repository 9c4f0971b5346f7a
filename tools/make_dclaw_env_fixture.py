"""Golden vectors of the reference's D'Claw environment arithmetic (BASELINE configs[3]): the REFERENCE's own
`envs/dclaw_rotate_env.py::DClawRotateEnv` (position control, relative actions, observation_type "tactile") run here in the dev container
against a SCRIPTED simulator (prescribed q / qdot / variables / flow images per step), recording what the environment makes of them: the
joint targets it hands to set_u (relative control, scaling, clipping to the joint limits), the observation, the reward, `done` and
`success` — including steps where a fingertip rises above the cap and where the cap passes pi/4.
Writes tests/golden/dclaw_env.npz (data only).  tests/test_dclaw_env_golden.py checks envs/dclaw_rotate.py against it."""
import os
import sys
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("TSIM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
T = 24
rng = np.random.default_rng(21)
Q = np.zeros((T + 1, 10)); Q[:, :9] = rng.uniform(-0.4, 1.3, size=(T + 1, 9)); Q[:, 9] = np.linspace(0.0, 0.7, T + 1)
Q[T - 1, 9] = 0.9                                                   # beyond pi / 4: success
VAR = np.tile(np.array([0.02, 0.01, 0.03] * 3 + [0.0, 0.0, 0.04]), (T + 1, 1)) + rng.normal(size=(T + 1, 12)) * 0.003
VAR[10, 5] = 0.06                                                   # a fingertip above the cap's top surface: done, -50
FLOW = rng.normal(size=(T + 1, 3, 20, 20, 3)) * (rng.uniform(size=(T + 1, 3, 20, 20, 1)) < 0.15)
FLOW[5:8, 1] = 0.0                                                  # finger 1 out of contact for three steps
LOG = {"set_u": []}


class Simulation:
    def __init__(self, model_path, verbose=False):
        from tactilesimulation_amd.model.compiler import load_model
        from tactilesimulation_amd.workloads import asset
        self._meta = load_model(asset("dclaw_position_control")).meta
        self.ndof_r, self.ndof_u, self.ndof_var, self.ndof_tactile = 10, 9, 12, 2718
        self.options = types.SimpleNamespace(h=5e-3)
        self.viewer_options = types.SimpleNamespace(camera_lookat=np.zeros(3), camera_pos=np.zeros(3))
        self.k = 0
    def get_q_init(self): return np.zeros(10)
    def set_q_init(self, q): pass
    def set_state_init(self, q, qdot): pass
    def reset(self, backward_flag=False): self.k = 0
    def set_u(self, u): LOG["set_u"].append(np.array(u, dtype=np.float64).copy())
    def forward(self, n, verbose=False, test_derivatives=False, save_last_frame_var_only=False): self.k += 1
    def get_q(self): return Q[self.k].copy()
    def get_qdot(self): return np.full(10, 0.01)
    def get_variables(self): return VAR[self.k].copy()
    def get_tactile_flow_images(self): return FLOW[self.k].tolist()
    def get_tactile_image_pos(self, name): return [tuple(p) for p in self._meta["image_pos"][name]]
    def update_joint_damping(self, name, v): pass
    def update_body_size(self, name, v): pass
    def update_endeffector_position(self, name, v): pass
    def update_joint_location(self, name, v): pass


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

    env = DClawRotateEnv(use_torch=False, observation_type="tactile", render_tactile=False, seed=4)
    obs0 = env.reset()
    U = rng.uniform(-1.6, 1.6, size=(T, 9))                          # beyond [-1, 1]: the env clips
    obs, rew, done, succ = [], [], [], []
    for t in range(T):
        o, r, d, info = env.step(U[t].copy())
        obs.append(np.asarray(o, dtype=np.float64).copy()); rew.append(float(r)); done.append(bool(d)); succ.append(bool(info["success"]))
    out = {"q": Q, "var": VAR, "flow": FLOW, "u": U, "targets": np.array(LOG["set_u"]), "obs0": np.asarray(obs0, dtype=np.float64), "obs": np.array(obs),
           "reward": np.array(rew), "done": np.array(done), "success": np.array(succ), "dof_limit": env.dof_limit, "relative_q_scale": np.float64(env.relative_q_scale),
           "rot_coef": np.float64(env.rot_coef), "power_coef": np.float64(env.power_coef), "cap_top_surface_z": np.float64(env.cap_top_surface_z), "q_init": env.q_init}
    path = os.path.join(ROOT, "tests", "golden", "dclaw_env.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; obs", out["obs"].shape, "done at", np.nonzero(out["done"])[0], "success at", np.nonzero(out["success"])[0], "rewards", np.round(out["reward"][[0, 5, 9, 10, T - 2, T - 1]], 3))
