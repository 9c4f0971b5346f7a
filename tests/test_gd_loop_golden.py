"""The batched training step (algorithms/batched_gd.train_epoch: one batch of episodes, loss normalised by the global episode count, clip by
global norm, Adam, linear learning-rate schedule) against the parameter trajectory of the REFERENCE's own algorithms/gd.py::GD trained on a
toy differentiable environment (tools/make_gd_fixture.py -> tests/golden/gd_loop.npz): the reference runs its `num_episodes` episodes one
after another and accumulates gradients; here they are the rows of one batch.  CPU only (no simulator involved: the loop is what is pinned)."""
import os

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "gd_loop.npz"))


class BatchedToyEnv:
    """The fixture's environment, batched: x' = A x + B tanh(u), reward -|x'|^2 - 0.1 |u|^2."""
    def __init__(self):
        self.A, self.Bm = torch.tensor(G["A"]), torch.tensor(G["B"])
        self.B = G["x0"].shape[1]

    def reset(self, q0=None, goal=None):
        self.x = q0
        return self.x

    def step(self, u, disturbance=None):
        self.x = self.x @ self.A.T + torch.tanh(u) @ self.Bm.T
        return self.x, -(self.x ** 2).sum(1) - 0.1 * (u ** 2).sum(1), {}


def test_train_epoch_follows_the_reference_gd_loop():
    from tactilesimulation_amd.algorithms.batched_gd import Actor, train_epoch
    epochs, episodes, obs_dim = G["x0"].shape
    actor = Actor(obs_dim=obs_dim, act_dim=G["B"].shape[1], hidden=(16, 16), dtype=torch.float64)
    actor.load_reference_state_dict({k[len("init/"):]: G[k] for k in G.files if k.startswith("init/")})
    lr0 = float(G["lr"])
    opt = torch.optim.Adam(actor.parameters(), lr=lr0, betas=tuple(G["betas"]))
    env = BatchedToyEnv()
    for e in range(epochs):
        for g in opt.param_groups:
            g["lr"] = (1e-5 - lr0) * float(e / epochs) + lr0            # algorithms/gd.py:146-149
        train_epoch(env, actor, opt, int(G["horizon"]), episodes, grad_clip=float(G["grad_norm"]), q0=torch.tensor(G["x0"][e]))
        got = actor.reference_state_dict()
        for k, v in got.items():
            want = G["epoch%d/%s" % (e, k)]
            assert np.abs(v.numpy() - want).max() < 1e-12, (e, k, np.abs(v.numpy() - want).max())
    # deferred weight gradients (the graphed loop's mode) give the same trajectory
    actor2 = Actor(obs_dim=obs_dim, act_dim=G["B"].shape[1], hidden=(16, 16), dtype=torch.float64)
    actor2.load_reference_state_dict({k[len("init/"):]: G[k] for k in G.files if k.startswith("init/")})
    actor2.defer_weight_grads = True
    opt2 = torch.optim.Adam(actor2.parameters(), lr=lr0, betas=tuple(G["betas"]))
    for e in range(epochs):
        for g in opt2.param_groups:
            g["lr"] = (1e-5 - lr0) * float(e / epochs) + lr0
        train_epoch(env, actor2, opt2, int(G["horizon"]), episodes, grad_clip=float(G["grad_norm"]), q0=torch.tensor(G["x0"][e]))
    for k, v in actor2.reference_state_dict().items():
        assert np.abs(v.numpy() - G["epoch%d/%s" % (epochs - 1, k)]).max() < 1e-12, k
