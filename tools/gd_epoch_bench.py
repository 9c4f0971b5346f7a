"""Closed-loop throughput of the batched GD loop (algorithms/batched_gd.py): policy MLP -> env-step -> ... -> BPTT, one
optimiser step per epoch, B environments x horizon env-steps, per-step launches (a policy sits between env-steps).
GPU box.  Not the bench.py headline (that one times the simulator alone)."""
import json, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.algorithms.batched_gd import Actor, train_epoch

def run(B=4096, T=100, dtype=torch.float32, epochs=3):
    m = load_model(os.path.join(ROOT, "tests", "golden", "models", "pusher.npz"))
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

if __name__ == "__main__":
    res = [run(), run(dtype=torch.float64)]
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gd_epoch_bench.json"), "w"), indent=1)
