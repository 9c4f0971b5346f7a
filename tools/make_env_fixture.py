"""Golden vectors of the reference's TactilePush environment formulas, recorded by running the REFERENCE's own
`envs/tactile_push_env.py::TactilePushEnv` (use_torch=True, observation_type="tactile_flatten", gradient=False) here in the dev container
against a SCRIPTED simulator: `gym`, `cv2` and `redmax_py` are stubs in sys.modules, and the stub `Simulation` returns prescribed q /
variables / tactile values (drawn once, recorded) — so what is pinned is exactly the env-side arithmetic the fused kernels of
include/tsim_env.h replace: action mapping (robot_action handed to set_u), observation, reward and its four terms.

Writes tests/golden/tactile_push_env.npz (data only; no reference source travels).
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("TSIM_REFERENCE", "/root/reference")
T = 40
rng = np.random.default_rng(11)
SCRIPT = {"q_init": np.concatenate([[0.0, 0.0, 0.0], rng.normal(size=4) * 0.01]),
          "q": np.cumsum(rng.normal(size=(T, 7)) * 0.02, axis=0) + np.array([0.3, 0.1, -0.05, 0.2, 0.05, 0.0, 0.4]),
          "var": rng.normal(size=(T, 6)) * 0.03, "tactile": rng.normal(size=(T + 1, 390)) * (rng.uniform(size=(T + 1, 390)) < 0.2)}
LOG = {"set_u": [], "forward": []}


class _Opt:
    h = 5e-3


class Simulation:                                      # the scripted stand-in for redmax_py.Simulation
    def __init__(self, model_path, verbose=False):
        self.ndof_r, self.ndof_u, self.ndof_var, self.ndof_tactile = 7, 6, 6, 390
        self.options = _Opt()
        self.viewer_options = types.SimpleNamespace()
        self.k = -1
        self.q_init = SCRIPT["q_init"].copy()

    def get_q_init(self): return self.q_init.copy()
    def set_q_init(self, q): self.q_init = np.array(q, dtype=np.float64).copy()
    def update_virtual_object(self, name, data): pass
    def reset(self, backward_flag=False): self.k = -1
    def set_u(self, u): LOG["set_u"].append(np.array(u, dtype=np.float64).copy())
    def forward(self, n, verbose=False, test_derivatives=False, save_last_frame_var_only=False): LOG["forward"].append(int(n)); self.k += 1
    def get_q(self): return SCRIPT["q"][self.k].copy()
    def get_variables(self): return SCRIPT["var"][self.k].copy()
    def get_tactile_force_vector(self): return SCRIPT["tactile"][self.k + 1].copy()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
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
    _stub("redmax_py", Simulation=Simulation)
    from scipy.spatial.transform import Rotation
    _stub("scipy.spatial.transform.rotation", Rotation=Rotation)          # private path the reference imports
    sys.path.insert(0, REF)
    from envs.tactile_push_env import TactilePushEnv                      # the reference's class

    env = TactilePushEnv(use_torch=True, gradient=False, observation_type="tactile_flatten", seed=3)
    obs0 = env.reset()
    out = {"goal": env.goal.numpy().copy(), "q0": env.state_q.numpy().copy(), "obs0": obs0.numpy().copy(), "tactile0": SCRIPT["tactile"][0],
           "q": SCRIPT["q"], "var": SCRIPT["var"], "tactile": SCRIPT["tactile"][1:], "u": rng.normal(size=(T, 3)) * 0.8}
    obs, rew, ext, terms = [], [], [], []
    for t in range(T):
        o, r, done, info = env.step(torch.tensor(out["u"][t], dtype=torch.float64))
        obs.append(o.numpy().copy()); rew.append(float(r)); ext.append(np.array(env.external_force, dtype=np.float64).copy())
        terms.append([float(info[k]) for k in ("reward_pos", "reward_rot", "reward_touch", "reward_action")])
        assert not done
    out.update({"obs": np.array(obs), "reward": np.array(rew), "external_force": np.array(ext), "reward_terms": np.array(terms),
                "robot_action": np.array(LOG["set_u"]), "frame_skip": np.array(sorted(set(LOG["forward"])))})
    # the other observation types of :72-131 on the same script (same seed: same goal, same disturbances)
    for ot in ("no_tactile", "privilege", "tactile_map"):
        LOG["set_u"].clear(); LOG["forward"].clear()
        e2 = TactilePushEnv(use_torch=True, gradient=False, observation_type=ot, seed=3)
        o0 = e2.reset()
        assert np.array_equal(e2.goal.numpy(), out["goal"])
        rec = []
        for t in range(T):
            o, r, done, info = e2.step(torch.tensor(out["u"][t], dtype=torch.float64))
            rec.append(o)
            assert abs(float(r) - out["reward"][t]) < 1e-12 * abs(out["reward"][t])
        if ot == "tactile_map":
            out["obs0_tactile_map"], out["obs0_tactile_map_state"] = o0[0].numpy().copy(), o0[1].numpy().copy()
            out["obs_tactile_map"] = np.array([o[0].numpy() for o in rec]); out["obs_tactile_map_state"] = np.array([o[1].numpy() for o in rec])
        else:
            out["obs0_" + ot] = o0.numpy().copy(); out["obs_" + ot] = np.array([o.numpy() for o in rec])
    path = os.path.join(ROOT, "tests", "golden", "tactile_push_env.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; obs", out["obs"].shape, "frame_skip", out["frame_skip"], "reward[0:3]", out["reward"][:3])
