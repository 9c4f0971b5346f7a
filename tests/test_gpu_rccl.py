"""RCCL on this stack, as far as a 1-GPU box allows: a one-rank `nccl` process group (backend "nccl" IS RCCL on ROCm) is initialised with
the device binding bench.py uses, and the GD outer loop's only collective — the flat policy-gradient all-reduce of
`tactilesimulation_amd.dist.allreduce_policy_grad_` (118 296 B for the gd_tactile actor) — runs through it on `cuda:0`, followed by the
barrier / MAX-reduce / destroy sequence of bench.py's timed region.  World size 1 makes every collective the identity, so this checks
library loading, communicator creation and stream ordering, not the exchange itself (no multi-GPU box is available to this repository;
the exchange arithmetic is covered on gloo with two ranks: tests/test_distributed_cpu.py, tests/test_gpu_sharded.py)."""
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
