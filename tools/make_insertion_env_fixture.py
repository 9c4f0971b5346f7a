"""Golden vectors of the reference's TactileInsertion environment arithmetic (BASELINE configs[4]): the REFERENCE's own
`envs/tactile_insertion_env.py::TactileInsertionEnv` (observation_type "tactile_flatten", translation + rotation actions, relative action
type; reward types "absolute" and "delta") run here in the dev container against a SCRIPTED simulator, recording per env-step (= one insertion
attempt of 45 sub-steps): the pre-grasp state the action leads to, the 45 x 6 joint-target table handed to the simulator, the sub-steps at
which it asks for tactile frames, the observation, the reward, done and success.
Writes tests/golden/insertion_env.npz (data only).  tests/test_insertion_env_golden.py checks envs/tactile_insertion.py against it."""
import os
import sys
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("TSIM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
rng = np.random.default_rng(33)
QREF = np.array([0.0, 0.0, 0.2, 0.0, -0.008, -0.008, 0.0, 0.0, 0.17, 0.0, 0.0, 0.0])     # what the scripted settling "converges" to
STATE = {"episode": -1, "k": 0, "qs": None, "tacs": None}
LOG = {"set_u": [], "tac_at": [], "state_init": []}
NSTEP = 6


def script_episode(i):
    """q per sub-step and tactile per call of one insertion attempt; attempts 3 and 5 end inside the success tolerances."""
    qs = np.tile(QREF, (45, 1)) + rng.normal(size=(45, 12)) * 0.002
    qs[-1, 6:9] = [0.001, -0.0015, 0.02] if i in (3, 5) else [0.004, 0.003, 0.03]
    tacs = rng.normal(size=(8, 780)) * (rng.uniform(size=(8, 780)) < 0.3)
    return qs, tacs


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
        STATE["episode"] += 1; STATE["k"] = 0; STATE["tk"] = 0
        STATE["qs"], STATE["tacs"] = script_episode(STATE["episode"])
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
        LOG["tac_at"][-1].append(STATE["k"] - 1); STATE["tk"] += 1
        return STATE["tacs"][STATE["tk"] - 1].copy()
    def clearBackwardCache(self): pass
    def saveBackwardCache(self): pass
    def update_contact_parameters(self, *a, **kw): pass
    def update_tactile_parameters(self, *a, **kw): pass


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
    from envs.tactile_insertion_env import TactileInsertionEnv        # the reference's class

    out = {}
    U = rng.uniform(-1.4, 1.4, size=(NSTEP, 3))                      # beyond [-1, 1]: the env clips
    for rt in ("absolute", "delta"):
        STATE.update({"episode": -1, "k": 0}); LOG["set_u"].clear(); LOG["tac_at"].clear(); LOG["state_init"].clear()
        rng = np.random.default_rng(33)                              # the same scripted simulator for both reward types
        env = TactileInsertionEnv(use_torch=True, observation_type="tactile_flatten", observation_noise=False, normalize_tactile_obs=True,
                                  allow_translation=True, allow_rotation=True, action_type="relative", reward_type=rt, domain_randomization=False,
                                  seed=6, render_tactile=False)
        obs0 = env.reset(position_noise=np.array([0.004, -0.003, 0.0001]), rotation_noise=0.08, grasp_height_noise=-0.002)
        rec = {"obs": [obs0.numpy().copy()], "q_init": [env.current_q_init.numpy().copy()], "reward": [float(env.reward_buf)], "done": [bool(env.done_buf)],
               "success": [bool(env.info_buf["success"])]}
        for t in range(NSTEP):
            o, r, d, info = env.step(torch.tensor(U[t]))
            rec["obs"].append(o.numpy().copy()); rec["q_init"].append(env.current_q_init.numpy().copy()); rec["reward"].append(float(r)); rec["done"].append(bool(d))
            rec["success"].append(bool(info["success"]))
        for k, v in rec.items():
            out[rt + "/" + k] = np.array(v)
        out[rt + "/actions"] = np.array(LOG["set_u"])                # [episodes, 45, 6]
        out[rt + "/tactile_substeps"] = np.array(LOG["tac_at"])      # [episodes, 6]
        out[rt + "/state_init"] = np.array(LOG["state_init"])
        out[rt + "/qs_last"] = np.array([script_episode_q for script_episode_q in []]) if False else np.zeros(0)
    # the scripted simulator's outputs, regenerated in the same order
    rng = np.random.default_rng(33)
    eps = [script_episode(i) for i in range(NSTEP + 1)]
    out["script_qs"] = np.array([e[0] for e in eps]); out["script_tactile"] = np.array([e[1][:6] for e in eps])
    out.update({"u": U, "q_ref": QREF, "q_init_reference": env.q_init_reference, "max_error": env.max_error, "action_scale": env.action_scale.numpy(),
                "working_space_boundary": env.working_space_boundary.numpy(), "working_rotation_boundary": np.float64(env.working_rotation_boundary),
                "tactile_masks": env.tactile_masks.numpy(), "reset_noise": np.array([0.004, -0.003, 0.0001, 0.08, -0.002])})
    path = os.path.join(ROOT, "tests", "golden", "insertion_env.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; obs", out["absolute/obs"].shape, "success", out["delta/success"], "reward abs", np.round(out["absolute/reward"], 3), "delta", np.round(out["delta/reward"], 3))
    print("tactile sub-steps", out["absolute/tactile_substeps"][0], "mask", np.nonzero(out["tactile_masks"])[0])
