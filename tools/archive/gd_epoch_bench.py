"""Closed-loop throughput of the batched GD loop (algorithms/batched_gd.py): policy MLP -> env-step -> ... -> BPTT, one
optimiser step per epoch, B environments x horizon env-steps, per-step launches (a policy sits between env-steps).
GPU box.  Not the bench.py headline (that one times the simulator alone)."""
import json, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.algorithms.batched_gd import Actor, train_epoch, GraphedRollout, train_epoch_graphed

def run(B=4096, T=100, dtype=torch.float32, epochs=3):
    m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
    env = BatchedTactilePushEnv(m, B, dtype=dtype, gradient=True, seed=0, tape_steps=T)
    torch.manual_seed(0)
    actor = Actor(dtype=dtype).cuda()
    opt = torch.optim.Adam(actor.parameters(), lr=1e-3)
    train_epoch(env, actor, opt, T, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [train_epoch(env, actor, opt, T, B) for _ in range(epochs)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"B": B, "horizon": T, "dtype": str(dtype), "epochs": epochs, "env_steps_per_s": B * T * epochs / dt, "s_per_epoch": dt / epochs,
            "loss_per_episode": losses}

def run_graphed(B=4096, T=100, dtype=torch.float32, epochs=3):
    import numpy as np
    m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
    env = BatchedTactilePushEnv(m, B, dtype=dtype, gradient=True, seed=0, tape_steps=T)
    env.reset()                                               # draws q0 / goal like the eager run (same seed)
    q0, goal = env.q0.clone(), env.goal.clone()
    rng = np.random.default_rng(1)
    dist_ = torch.tensor(rng.uniform(-1, 1, size=(T, B, 2)) * (rng.uniform(size=(T, B, 1)) < 0.5), device="cuda", dtype=dtype)
    torch.manual_seed(0)
    actor = Actor(dtype=dtype).cuda()
    opt = torch.optim.Adam(actor.parameters(), lr=1e-3)
    gr = GraphedRollout(env, actor, T, q0, goal, dist_)
    train_epoch_graphed(gr, opt, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [train_epoch_graphed(gr, opt, B).detach().clone() for _ in range(epochs)]      # gr.loss is a static tensor
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"B": B, "horizon": T, "dtype": str(dtype), "epochs": epochs, "mode": "HIP graph replay", "env_steps_per_s": B * T * epochs / dt,
            "s_per_epoch": dt / epochs, "loss_per_episode": [float(l) / B for l in losses]}


if __name__ == "__main__":
    res = [run(), run(dtype=torch.float64), run_graphed(), run_graphed(dtype=torch.float64)]
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gd_epoch_bench.json"), "w"), indent=1)
