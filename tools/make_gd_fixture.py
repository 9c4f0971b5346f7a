"""Golden vectors of the reference's training loop (algorithms/gd.py: per-epoch linear learning-rate schedule, `num_episodes` episodes run
one after another with loss = -return / num_episodes accumulated by backward(), clip_grad_norm_, Adam(betas) step):
the REFERENCE's own `GD` class trained for a few epochs here in the dev container on a TOY differentiable environment (a linear system with a
tanh action and quadratic reward: no simulator needed — what is pinned is the loop, not the physics), with gym / cv2 / tensorboardX stubbed.
Records the policy's initial parameters, the episodes' initial states, and the parameters after every epoch.
Writes tests/golden/gd_loop.npz (data only).  tests/test_gd_loop_golden.py runs algorithms/batched_gd.train_epoch on the same episodes as
ONE batch and compares the parameter trajectory."""
import os
import sys
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("TSIM_REFERENCE", "/root/reference")
OBS, ACT, HORIZON, EPISODES, EPOCHS = 6, 3, 5, 4, 3
rs = np.random.default_rng(77)
A = np.eye(OBS) * 0.9 + rs.normal(size=(OBS, OBS)) * 0.05
Bm = rs.normal(size=(OBS, ACT)) * 0.3
X0 = rs.normal(size=(EPOCHS * EPISODES + 8, OBS))                    # initial states, consumed in order by reset()
CALLS = {"reset": 0}


def _stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    return m


if __name__ == "__main__":
    import torch

    class ToyEnv:
        """x' = A x + B tanh(u); reward = -|x'|^2 - 0.1 |u|^2; HORIZON steps.  Differentiable in torch; the training env (gradient=True)
        takes its initial states from X0 in order, the render env (gradient=False) is never stepped here."""
        def __init__(self, use_torch=True, gradient=False, verbose=False, render_tactile=False, **kw):
            self.gradient = gradient
            self.observation_space = types.SimpleNamespace(shape=(OBS,))
            self.action_space = types.SimpleNamespace(shape=(ACT,))
            self.A, self.B = torch.tensor(A), torch.tensor(Bm)
        def seed(self, s): pass
        def close(self): pass
        def reset(self):
            self.x = torch.tensor(X0[CALLS["reset"]]); CALLS["reset"] += 1; self.t = 0
            return self.x
        def step(self, u):
            self.x = self.A @ self.x + self.B @ torch.tanh(u)
            self.t += 1
            return self.x, -(self.x ** 2).sum() - 0.1 * (u ** 2).sum(), self.t >= HORIZON, {}
    _stub("gym", make=lambda name, **kw: ToyEnv(**kw), logger=types.SimpleNamespace(set_level=lambda l: None), __path__=[])
    _stub("gym.envs", __path__=[]); _stub("gym.envs.registration", registry=types.SimpleNamespace(env_specs={}), register=lambda **kw: None, make=None, spec=None)
    _stub("cv2", determinant=None)
    _stub("tensorboardX", SummaryWriter=lambda *a, **k: types.SimpleNamespace(add_scalar=lambda *a, **k: None, flush=lambda: None, close=lambda: None))
    _stub("redmax_py", Simulation=object)
    _stub("envs")                                                     # the reference's envs package registers gym ids on import: not needed
    sys.path.insert(0, REF)
    torch.set_default_dtype(torch.float64)
    from algorithms.gd import GD                                      # the reference's class
    cfg = {"params": {"general": {"seed": 3, "device": "cpu", "train": True, "checkpoint": None, "logdir": "/tmp/gd_fixture_log", "save_interval": 0,
                                  "log_interval": 0, "render_interval": 0},
                      "env": {"name": "Toy-v0"},
                      "network": {"actor": "DiagGaussianActor", "actor_mlp": {"layer_sizes": [16, 16], "activation": "elu", "layernorm": False}, "actor_logstd_init": -1.0},
                      "config": {"num_epochs": EPOCHS, "num_episodes": EPISODES, "num_processes": 1, "lr": 0.005, "truncate_grads": True, "grad_norm": 1.0,
                                 "betas": [0.7, 0.95], "lr_schedule": "linear", "obs_rms": False, "gamma": 0.99}}}
    GD.save = lambda self, name="best": None                          # checkpoint files: not part of what is recorded
    gd = GD(cfg)
    out = {"A": A, "B": Bm, "x0": X0[:EPOCHS * EPISODES].reshape(EPOCHS, EPISODES, OBS), "horizon": np.int64(HORIZON),
           "lr": np.float64(0.005), "betas": np.array([0.7, 0.95]), "grad_norm": np.float64(1.0)}
    for k, v in gd.actor.state_dict().items():
        out["init/" + k] = v.detach().numpy().copy()
    traj = []
    step0 = gd.actor_optimizer.step

    def step(*a, **k):
        r = step0(*a, **k)
        traj.append({k_: v.detach().numpy().copy() for k_, v in gd.actor.state_dict().items()})
        return r
    gd.actor_optimizer.step = step
    gd.train()
    assert len(traj) == EPOCHS and CALLS["reset"] == EPOCHS * EPISODES, (len(traj), CALLS)
    for e, sd in enumerate(traj):
        for k, v in sd.items():
            out["epoch%d/%s" % (e, k)] = v
    path = os.path.join(ROOT, "tests", "golden", "gd_loop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; parameters", sum(v.size for k, v in out.items() if k.startswith("init/")), "epochs", EPOCHS,
          "first-layer weight drift per epoch", [float(np.abs(traj[e]["feature_net.body.0.weight"] - out["init/feature_net.body.0.weight"]).max()) for e in range(EPOCHS)])
