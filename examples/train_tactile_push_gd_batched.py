#!/usr/bin/env python
"""Batched analytic-gradient training of the TactilePush policy — the MI355X counterpart of the reference's
examples/TactilePushExp/train_tactile_push_gd.py + algorithms/gd.py (cfg/gd_tactile.yaml: 393 -> 64 -> 64 -> 3 actor,
Adam lr 0.005 betas (0.7, 0.95) with the linear decay of gd.py:146-149, gradient-norm clip 1.0, 100-step episodes).  There, `num_episodes` episodes run one after the other through one
Simulation; here they are one batch per GPU.

    python examples/train_tactile_push_gd_batched.py --batch 4096 --epochs 20                      # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           examples/train_tactile_push_gd_batched.py --batch 4096 --epochs 20                      # 8 GPUs, RCCL over xGMI

Per epoch and rank: one episode of `--batch` environments (policy -> env-step -> ... -> BPTT through the simulator) with the
policy evaluated INSIDE the simulator's episode launches (one launch each way per episode, envs/push_closed_loop.py; `--graphed`: one launch
per env-step replayed from ONE HIP graph, `--eager`: the plain loop); the ranks exchange only the flat policy gradient (118 KB), once
per epoch.  Every epoch draws new goals / box offsets / disturbances per environment (envs/tactile_push_env.py:133-190) and
writes them into the graph's static inputs.  The model path defaults to the compiled TactilePush model of the test fixtures;
pass the reference's envs/assets/pusher/pusher.xml to compile it afresh.
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv                      # noqa: E402
from tactilesimulation_amd.algorithms.batched_gd import Actor, GraphedRollout, train_epoch, train_epoch_graphed   # noqa: E402
from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode, train_epoch_fused          # noqa: E402


def draw_episode(rng, B, T, device, dtype, period=1):
    """q0 [B, 7], goal [B, 3], disturbances [T, B, 2] as in tactile_push_env.py:133-190: a new force on the box, on with p = 0.5, every
    `period` env-steps.  The reference's line reads "every 10 steps" (`current_step % 10 == 0`, :185) but current_step is never
    incremented there, so it redraws at every step: period = 1 is the reference's behaviour."""
    q0 = np.zeros((B, 7)); q0[:, 1] = -0.001; q0[:, 4] = rng.uniform(-0.02, 0.02, size=B)
    goal = np.zeros((B, 3))
    goal[:, 0:2] = rng.uniform([0.15, -0.2], [0.25, 0.2], size=(B, 2))
    goal[:, 2] = rng.uniform(goal[:, 1] * math.pi - math.pi / 16.0, goal[:, 1] * math.pi + math.pi / 16.0)
    d = np.zeros((T, B, 2))
    for t0 in range(0, T, period):
        d[t0:t0 + period] = (rng.uniform(size=(B, 1)) < 0.5) * rng.uniform(-1.0, 1.0, size=(B, 2))
    t = lambda a: torch.tensor(a, device=device, dtype=dtype)
    return t(q0), t(goal), t(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=os.path.join(ROOT, "tactilesimulation_amd", "assets", "pusher.npz"))
    ap.add_argument("--batch", type=int, default=4096, help="environments (= episodes per epoch) per GPU")
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--horizon", type=int, default=100)
    # optimiser of cfg/gd_tactile.yaml (algorithms/gd.py:146-151): Adam lr 0.005, betas (0.7, 0.95), linear decay to 1e-5
    ap.add_argument("--lr", type=float, default=5e-3, help="the reference's 0.005 is tuned for 16 episodes per epoch; with thousands per batch two of six seeds leave the descent near "
                    "epoch 95 (exploding BPTT gradient, profiles/r06_training_300_epochs.md): 0.002 trains every seed; --save-best keeps the best policy either way")
    ap.add_argument("--betas", type=float, nargs=2, default=(0.7, 0.95))
    ap.add_argument("--lr-schedule", default="linear", choices=["linear", "constant"])
    ap.add_argument("--grad-clip", type=float, default=1.0)
    ap.add_argument("--log", default=None, help="write the loss curve (JSON) here")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--eager", action="store_true", help="plain python loop, one launch per env-step")
    ap.add_argument("--graphed", action="store_true", help="one launch per env-step, the episode replayed from one HIP graph (algorithms/batched_gd.GraphedRollout)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + TSIM_SHARE_GPU=1: several ranks on one GPU (plumbing tests)")
    ap.add_argument("--observation-type", default="tactile_flatten", choices=["tactile_flatten", "no_tactile", "privilege"],
                    help="cfg/gd_tactile.yaml / gd_no_tactile.yaml / gd_privilege.yaml (tactile_push_env.py:72-131)")
    ap.add_argument("--disturbance-period", type=int, default=1, help="env-steps between new random forces on the box (the reference: 1, see draw_episode)")
    ap.add_argument("--save-best", default=None, help="torch.save the best policy (lowest loss, as algorithms/gd.py:187-189 keeps it) here")
    args = ap.parse_args()

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("TSIM_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    dev, dtype = "cuda:%d" % local, (torch.float32 if args.dtype == "f32" else torch.float64)
    B, T = args.batch, args.horizon
    env = BatchedTactilePushEnv(args.model, B, device=dev, dtype=dtype, gradient=True, seed=args.seed + rank, tape_steps=T, observation_type=args.observation_type)
    torch.manual_seed(args.seed)                                   # identical initial policy on every rank
    actor = Actor(obs_dim=env.obs_dim, dtype=dtype).to(dev)
    opt = torch.optim.Adam(actor.parameters(), lr=args.lr, betas=tuple(args.betas))
    curve = []
    rng = np.random.default_rng(args.seed + 1000 * rank)
    q0, goal, dist_ = draw_episode(rng, B, T, dev, dtype, args.disturbance_period)
    gr = GraphedRollout(env, actor, T, q0, goal, dist_) if args.graphed else None
    ep = FusedPushEpisode(env, actor, T) if not (args.eager or args.graphed) else None
    best = (float("inf"), -1, None)
    prev_state = {k: v.detach().clone() for k, v in actor.state_dict().items()}
    for epoch in range(args.epochs):
        if args.lr_schedule == "linear":                           # gd.py:146-149
            for g in opt.param_groups:
                g["lr"] = (1e-5 - args.lr) * float(epoch / args.epochs) + args.lr
        nq0, ngoal, ndist = draw_episode(rng, B, T, dev, dtype, args.disturbance_period)
        q0.copy_(nq0); goal.copy_(ngoal); dist_.copy_(ndist)       # static inputs of the graph
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if ep is not None:
            loss = float(train_epoch_fused(ep, opt, q0, goal, dist_, B * world, grad_clip=args.grad_clip).detach()) / B
        elif gr is None:
            loss = train_epoch(env, actor, opt, T, B * world, grad_clip=args.grad_clip, q0=q0, goal=goal, disturbances=dist_)
        else:
            loss = float(train_epoch_graphed(gr, opt, B * world, grad_clip=args.grad_clip).detach()) / B
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        curve.append({"epoch": epoch, "loss_per_episode": loss, "ms": dt * 1e3, "lr": opt.param_groups[0]["lr"]})
        if loss < best[0]:                                             # gd.py:187-189: keep the best policy, not the last
            # (the loss of epoch e belongs to the parameters BEFORE that epoch's update)
            best = (loss, epoch, prev_state)
        prev_state = {k: v.detach().clone() for k, v in actor.state_dict().items()}
        if rank == 0:
            print("epoch %3d  loss/episode (rank 0) %10.3f  %6.1f ms  %.2f M env-steps/s (all ranks)" % (epoch, loss, dt * 1e3, B * T * world / dt / 1e6), flush=True)
    if rank == 0 and args.log:
        import json
        json.dump({"args": vars(args), "world": world, "curve": curve, "best": {"loss_per_episode": best[0], "epoch": best[1]},
                   "note": "loss = -sum of rewards / episodes of rank 0's batch; every epoch draws new goals, offsets and disturbances"},
                  open(args.log, "w"), indent=1)
    if rank == 0 and args.save_best and best[2] is not None:
        torch.save({"actor": best[2], "loss_per_episode": best[0], "epoch": best[1]}, args.save_best)
        print("best policy: epoch %d, loss/episode %.3f -> %s" % (best[1], best[0], args.save_best))
    print("rank %d: parameter checksum %.10e" % (rank, float(sum(p.detach().double().abs().sum() for p in actor.parameters()))), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
