"""N > 1 path on CPU: 2 processes, gloo. Environments shard with no exchange; the all-reduced policy gradient equals the
single-process gradient (the GD outer loop's only collective)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tactilesimulation_amd.dist import env_shard, allreduce_policy_grad_


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _policy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(5, 8), torch.nn.ELU(), torch.nn.Linear(8, 3)).double()


def _episode_reward(policy, obs):
    """stand-in for a differentiable roll-out: any per-environment function of the policy output"""
    a = torch.tanh(policy(obs))
    return -((a - 0.3) ** 2).sum(dim=1) - 0.1 * (obs[:, :3] * a).sum(dim=1)


def _worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(123)
    obs = torch.randn(B, 5, dtype=torch.float64)            # same table on every rank, each takes its slice
    lo, hi = env_shard(B, rank, world)
    pol = _policy()
    loss = -_episode_reward(pol, obs[lo:hi]).sum()           # un-normalised local sum; normalisation after the reduce
    loss.backward()
    flat = allreduce_policy_grad_(list(pol.parameters()), B)
    torch.nn.utils.clip_grad_norm_(pol.parameters(), 1.0)
    if rank == 0:
        torch.save({"flat": flat, "clipped": torch.cat([p.grad.reshape(-1) for p in pol.parameters()])}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_env_shard_partition():
    for B in (4096, 10, 7):
        for world in (1, 2, 3, 8):
            cuts = [env_shard(B, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gradient_equals_single_process(tmp_path):
    B, world = 11, 2
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    got = torch.load(out)
    torch.manual_seed(123)
    obs = torch.randn(B, 5, dtype=torch.float64)
    pol = _policy()
    (-_episode_reward(pol, obs).sum() / B).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in pol.parameters()])
    assert torch.allclose(got["flat"], ref, rtol=1e-12, atol=1e-14)
    torch.nn.utils.clip_grad_norm_(pol.parameters(), 1.0)
    assert torch.allclose(got["clipped"], torch.cat([p.grad.reshape(-1) for p in pol.parameters()]), rtol=1e-12, atol=1e-14)
