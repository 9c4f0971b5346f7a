"""Sharded == unsharded on the REAL simulator (SURVEY.md §4 "multi-GPU", §8e): two ranks (one process each, gloo, both on
cuda:0 — the only multi-rank evidence obtainable on a 1-GPU box) run BatchedTactilePushEnv + the policy on their `env_shard`
slice of one global batch.
  * open loop (fixed action table): the concatenated simulator outputs (q, qd, variables, tactile) and the episode adjoint
    dL/du are BIT-IDENTICAL to the single-process batch — the simulator exchanges nothing and a row does not know its batch;
  * closed loop (policy between env-steps): the all-reduced, normalised policy gradient equals the single-process gradient.
    (Here the trajectories agree to round-off only: the policy's GEMM picks its blocking by the batch size, so torch's own
    `actor(obs)` is not bit-reproducible across shard sizes — the simulator is.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
T, LANES = 6, 16


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _tables(B):
    from tactilesimulation_amd.workloads import push_workload
    q0, u, goal = push_workload(B, T, seed=77)
    return q0, goal, u[:, :, 3:5].transpose(1, 0, 2).copy()            # disturbances [T, B, 2]


def _run(lo, hi, B, dtype, dev="cuda:0"):
    """Episode + backward on the environments [lo, hi) of the global batch; returns per-step outputs and the actor."""
    from tactilesimulation_amd.algorithms.batched_gd import Actor
    from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
    from tactilesimulation_amd.workloads import PUSHER_BLOB
    q0, goal, dist_tab = _tables(B)
    env = BatchedTactilePushEnv(PUSHER_BLOB, hi - lo, device=dev, dtype=dtype, gradient=True, tape_steps=T)
    env.sim.set_lanes_per_env(LANES)            # same launch shape whatever the shard size: same summation order
    torch.manual_seed(5)
    actor = Actor(dtype=dtype).to(dev)
    obs = env.reset(q0[lo:hi], goal[lo:hi])
    rec = {"obs": [obs.detach().clone()], "q": [], "rew": []}
    total = obs.new_zeros(())
    dtab = torch.tensor(dist_tab[:, lo:hi], device=dev, dtype=dtype)
    for t in range(T):
        obs, rew, info = env.step(actor(obs), dtab[t])
        total = total - rew.sum()
        rec["obs"].append(obs.detach().clone()); rec["q"].append(info["q"].detach().clone()); rec["rew"].append(rew.detach().clone())
    total.backward()
    return {k: torch.stack(v).cpu() for k, v in rec.items()}, actor


def _open_loop(lo, hi, B, dtype, dev="cuda:0"):
    """BatchSim alone on the environments [lo, hi): episode launch forward, episode adjoint backward."""
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.workloads import push_workload, PUSHER_BLOB
    from tactilesimulation_amd.model.compiler import load_model
    q0, u, _ = push_workload(B, T, seed=78)
    sim = BatchSim(load_model(PUSHER_BLOB), hi - lo, device=dev, dtype=dtype, tape_capacity=T * 5)
    sim.set_lanes_per_env(LANES)
    sim.reset(torch.tensor(q0[lo:hi], device=dev, dtype=dtype), None, backward_flag=True)
    ro = sim.rollout(torch.tensor(u[lo:hi], device=dev, dtype=dtype).transpose(0, 1).contiguous(), 5, want_qd=True)
    g = torch.Generator().manual_seed(3)
    w = {k: torch.randn(T, B, d, generator=g, dtype=torch.float64)[:, lo:hi].to(dev, dtype) for k, d in (("q", 7), ("var", 6), ("tactile", 390))}
    du = sim.backward_episode(T, 5, w["q"], w["var"], w["tactile"])
    out = {k: ro[k].cpu() for k in ("q", "qd", "var", "tactile")}
    out["du"] = du.cpu()
    return out


def _worker(rank, world, port, B, dtype, out_dir):
    import torch.distributed as dist
    from tactilesimulation_amd.dist import env_shard, allreduce_policy_grad_
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lo, hi = env_shard(B, rank, world)
    rec, actor = _run(lo, hi, B, dtype)
    open_loop = _open_loop(lo, hi, B, dtype)
    from types import SimpleNamespace                    # gloo: reduce host copies of the gradients, as bench.py --backend gloo does
    params = [SimpleNamespace(grad=p.grad.cpu()) for p in actor.parameters() if p.grad is not None]
    flat = allreduce_policy_grad_(params, B)
    torch.save({"rec": rec, "open": open_loop, "flat": flat.cpu(), "lo": lo, "hi": hi}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
def test_two_ranks_on_one_gpu_equal_the_unsharded_batch(tmp_path, dtype, tol):
    B, world = 22, 2                                    # 11 environments per rank: the last wavefront has idle slots
    mp.spawn(_worker, args=(world, _free_port(), B, dtype, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    assert [(p["lo"], p["hi"]) for p in parts] == [(0, 11), (11, 22)]
    ref_open = _open_loop(0, B, B, dtype)
    for k in ("q", "qd", "var", "tactile", "du"):
        got = torch.cat([p["open"][k] for p in parts], dim=1)
        assert torch.equal(got, ref_open[k]), k       # the simulator exchanges nothing: a row does not know its batch
    ref, actor = _run(0, B, B, dtype)
    for k in ("q", "obs", "rew"):
        got = torch.cat([p["rec"][k] for p in parts], dim=1)
        assert float((got - ref[k]).abs().max()) <= tol * max(float(ref[k].abs().max()), 1.0), k
    g_ref = torch.cat([p.grad.reshape(-1) for p in actor.parameters() if p.grad is not None]).cpu() / B
    assert torch.equal(parts[0]["flat"], parts[1]["flat"])                  # every rank holds the same reduced gradient
    err = float((parts[0]["flat"] - g_ref).abs().max()) / float(g_ref.abs().max())
    assert err < tol, err                               # sum over two partial sums vs one sum: round-off only
    assert float(g_ref.abs().max()) > 0


def test_bench_gpus_2_self_launched_on_one_gpu():
    """`python bench.py --gpus 2` with NO launcher (the shape of the driver's N = 1 command with another N): two ranks are started, both
    run the real simulator (sharing cuda:0, collectives over gloo — the only multi-rank run a 1-GPU box allows) and the line says so."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TSIM_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--batch", "512",
                        "--episode", "6", "--no-cpu-baseline", "--no-pmc"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks"]["ranks_in_first_allreduce"] == 2 and j["ranks"]["shared_gpu"] is True
    assert j["config"]["global_batch"] == 1024 and j["value"] > 0 and j["roofline"]["per_kernel"]["k_forward"]["ms"] > 0
    assert len(lines[0].encode()) < 6144 and len(j["per_rank"]) == 2 and j["per_rank"][1]["kernel_ms_total"] > 0


def test_two_ranks_train_the_fused_loop_to_identical_parameters(tmp_path):
    """examples/train_tactile_push_gd_batched.py under torch.distributed.run with two ranks sharing this GPU (gloo): every rank rolls out
    its own environments through the fused closed loop, the flat policy gradient is all-reduced once per epoch, and both ranks must end
    on the same parameters — which differ from a single rank's (other environments entered the gradient)."""
    import re
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, TSIM_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    ex = os.path.join(root, "examples", "train_tactile_push_gd_batched.py")
    common = ["--batch", "256", "--epochs", "3", "--horizon", "20", "--backend", "gloo"]
    out2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29531", ex] + common, env=env, capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    sums = dict((int(r), float(v)) for r, v in re.findall(r"rank (\d+): parameter checksum ([0-9.e+-]+)", out2.stdout))
    assert set(sums) == {0, 1} and sums[0] == sums[1], sums
    out1 = subprocess.run([sys.executable, ex] + common, env=env, capture_output=True, text=True, timeout=600)
    assert out1.returncode == 0, out1.stderr[-2000:]
    one = float(re.findall(r"rank 0: parameter checksum ([0-9.e+-]+)", out1.stdout)[0])
    assert one != sums[0]
