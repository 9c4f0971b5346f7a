"""RCCL on this stack, as far as a 1-GPU box allows: a one-rank `nccl` process group (backend "nccl" IS RCCL on ROCm) is initialised with
the device binding bench.py uses, and the GD outer loop's only collective — the flat policy-gradient all-reduce of
`tactilesimulation_amd.dist.allreduce_policy_grad_` (118 296 B for the gd_tactile actor) — runs through it on `cuda:0`, followed by the
barrier / MAX-reduce / destroy sequence of bench.py's timed region.  World size 1 makes every collective the identity, so this checks
library loading, communicator creation and stream ordering, not the exchange itself (the exchange arithmetic is covered on gloo with two ranks:
tests/test_distributed_cpu.py, tests/test_gpu_sharded.py).  The second test IS the exchange on real RCCL — min(device_count, 8) ranks, one GPU each,
the sharded simulator against the unsharded batch — and enables itself as soon as two GPUs are visible (round 5; skipped on a 1-GPU box)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_group_runs_the_policy_gradient_allreduce():
    import torch.distributed as dist
    from tactilesimulation_amd.algorithms.batched_gd import Actor
    from tactilesimulation_amd.dist import allreduce_policy_grad_
    if not dist.is_nccl_available():
        pytest.skip("torch was built without the nccl (RCCL) backend")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(0)
        actor = Actor().cuda()
        n = sum(p.numel() for p in actor.parameters())
        assert n == 29574
        for p in actor.parameters():
            p.grad = torch.randn_like(p)
        ref = torch.cat([p.grad.reshape(-1) for p in actor.parameters()]).clone()
        flat = allreduce_policy_grad_(list(actor.parameters()), global_episodes=4096)
        torch.cuda.synchronize()
        assert flat.numel() == n and torch.allclose(flat, ref / 4096.0)
        t = torch.tensor([1.25], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                  # bench.py: max over ranks of the timed region
        dist.barrier()
        torch.cuda.synchronize()
        assert float(t) == 1.25
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------- N ranks, one GPU each (skipped on a 1-GPU box)
def _nrank_worker(rank, world, port, B, out_dir):
    """One rank = one process = one GPU, RCCL between them: the real simulator on this rank's `env_shard` of one global batch (open loop: the
    simulator exchanges nothing), then the closed loop's policy gradient reduced by `allreduce_policy_grad_` over xGMI."""
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_sharded import _open_loop, _run
    from tactilesimulation_amd.dist import env_shard, allreduce_policy_grad_
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        lo, hi = env_shard(B, rank, world)
        dev = "cuda:%d" % rank
        open_loop = _open_loop(lo, hi, B, torch.float32, dev=dev)
        rec, actor = _run(lo, hi, B, torch.float32, dev=dev)
        flat = allreduce_policy_grad_([p for p in actor.parameters() if p.grad is not None], B)      # CUDA tensors on the nccl backend: RCCL
        torch.cuda.synchronize()
        torch.save({"rec": rec, "open": open_loop, "flat": flat.cpu(), "lo": lo, "hi": hi, "device": torch.cuda.get_device_name(rank)},
                   os.path.join(out_dir, "rank%d.pt" % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_n_rank_rccl_sharded_simulator_equals_the_unsharded_batch(tmp_path):
    """min(device_count, 8) `nccl` ranks, one GPU each (skipped on a 1-GPU box — the logic is the two-ranks-on-one-GPU gloo test's,
    tests/test_gpu_sharded.py, on real RCCL): the concatenated simulator outputs and episode adjoints of the shards are BIT-EQUAL to the
    single-process batch, every rank ends with the same reduced policy gradient, and it equals the single-process gradient to round-off.
    Reference counterpart of the sharding: SubprocVecEnv workers (examples/TactilePushExp/cfg/ppo_tactile.yaml:24)."""
    import sys
    import torch.distributed as dist
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("needs >= 2 GPUs (found %d): RCCL with N > 1 ranks cannot run here" % torch.cuda.device_count())
    if not dist.is_nccl_available():
        pytest.skip("torch was built without the nccl (RCCL) backend")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_sharded import _open_loop, _run, _free_port
    B = 11 * world                                      # 11 environments per rank: the last wavefront of every shard has idle slots
    mp.spawn(_nrank_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    assert [(p["lo"], p["hi"]) for p in parts] == [(11 * r, 11 * r + 11) for r in range(world)]
    ref_open = _open_loop(0, B, B, torch.float32)
    for k in ("q", "qd", "var", "tactile", "du"):
        assert torch.equal(torch.cat([p["open"][k] for p in parts], dim=1), ref_open[k]), k      # the simulator exchanges nothing: a row does not know its batch (nor its GPU)
    ref, actor = _run(0, B, B, torch.float32)
    for k in ("q", "obs", "rew"):
        got = torch.cat([p["rec"][k] for p in parts], dim=1)
        assert float((got - ref[k]).abs().max()) <= 2e-5 * max(float(ref[k].abs().max()), 1.0), k
    g_ref = torch.cat([p.grad.reshape(-1) for p in actor.parameters() if p.grad is not None]).cpu() / B
    for p in parts[1:]:
        assert torch.equal(p["flat"], parts[0]["flat"])                     # every rank holds the same reduced gradient
    assert float((parts[0]["flat"] - g_ref).abs().max()) / float(g_ref.abs().max()) < 2e-5
