"""Multi-GPU plumbing: environments shard across ranks with NO data-path exchange; the only collective is the GD
outer loop's all-reduce(sum) of the flat policy gradient (RCCL over xGMI when the backend is "nccl").
The reference has no counterpart (single process; algorithms/gd.py:224-259 runs episodes serially)."""
import torch
import torch.distributed as dist


def env_shard(global_batch, rank, world):
    """Contiguous slice [lo, hi) of the global environment index range owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(int(global_batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_policy_grad_(params, global_episodes, group=None):
    """In-place: sum the gradients of `params` over all ranks as ONE flat buffer (118 KB for the gd_tactile actor —
    latency-bound, so a single collective), then normalise by the global episode count so that the result equals the
    single-process gradient of  -sum(reward) / num_episodes  (algorithms/gd.py:258).  Clip-by-global-norm must come
    AFTER this call (gd.py:157-159) so that the norm matches the single-GPU one."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return None
    flat = torch.cat([g.reshape(-1) for g in grads])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":      # CPU plumbing runs (several ranks sharing one GPU): through the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= float(global_episodes)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return flat
