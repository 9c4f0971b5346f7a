"""autograd Functions over the fused TactilePush kernels of include/tsim_env.h (one HIP launch each way).

The formulas are the reference's (envs/tactile_push_env.py:84-114 observation, :175-193 action mapping, :202-211 reward);
tests/test_gpu_batched_env.py checks values and gradients against those formulas written in plain torch.  No fallback:
CUDA tensors and the built libtsim_hip.so are required.
"""
import ctypes as C

import torch

from ..host import capi

_DT = {torch.float32: capi.TSIM_F32, torch.float64: capi.TSIM_F64}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _c(t, B, dim, name):
    if not t.is_cuda or t.dtype not in _DT:
        raise RuntimeError("%s: a float32 / float64 CUDA tensor is required (no CPU fallback)" % name)
    if tuple(t.shape) != (B, dim):
        raise ValueError("%s: expected [%d, %d], got %s" % (name, B, dim, tuple(t.shape)))
    return t.contiguous()


class PushAction(torch.autograd.Function):
    """(u [B, 3] policy output, ext [B, 2] external force on the box) -> robot_action [B, 6] = [tanh(u), ext, 0]"""

    @staticmethod
    def forward(ctx, u, ext):
        B = u.shape[0]
        u, ext = _c(u.detach(), B, 3, "u"), _c(ext.detach().to(u.dtype), B, 2, "ext")
        a = torch.empty((B, 6), device=u.device, dtype=u.dtype)
        capi.check(capi.lib().tsim_push_action(B, _DT[u.dtype], _p(u), _p(ext), _p(a), _st(u)))
        ctx.save_for_backward(u)
        return a

    @staticmethod
    def backward(ctx, da):
        (u,) = ctx.saved_tensors
        B = u.shape[0]
        da = _c(da, B, 6, "d_action")
        du = torch.empty_like(u)
        capi.check(capi.lib().tsim_push_action_backward(B, _DT[u.dtype], _p(u), _p(da), _p(du), _st(u)))
        return du, None


class PushObserve(torch.autograd.Function):
    """(q [B, 7], var [B, 6], tactile [B, ntac], goal [B, 3], u [B, 3]) -> obs [B, 3 + ntac], reward [B].
    goal is data (no gradient)."""

    @staticmethod
    def forward(ctx, q, var, tactile, goal, u):
        B, ntac = q.shape[0], tactile.shape[1]
        q, var, tactile = _c(q.detach(), B, 7, "q"), _c(var.detach(), B, 6, "var"), _c(tactile.detach(), B, ntac, "tactile")
        goal, u = _c(goal.detach().to(q.dtype), B, 3, "goal"), _c(u.detach(), B, 3, "u")
        obs = torch.empty((B, 3 + ntac), device=q.device, dtype=q.dtype)
        rew = torch.empty((B,), device=q.device, dtype=q.dtype)
        capi.check(capi.lib().tsim_push_observe(B, ntac, _DT[q.dtype], _p(q), _p(var), _p(tactile), _p(goal), _p(u), _p(obs), _p(rew), _st(q)))
        ctx.save_for_backward(q, var, goal, u)
        ctx.ntac = ntac
        return obs, rew

    @staticmethod
    def backward(ctx, dobs, drew):
        q, var, goal, u = ctx.saved_tensors
        B, ntac = q.shape[0], ctx.ntac
        if dobs is None:
            dobs = torch.zeros((B, 3 + ntac), device=q.device, dtype=q.dtype)
        dobs = _c(dobs, B, 3 + ntac, "d_obs")
        new = lambda d: torch.empty((B, d), device=q.device, dtype=q.dtype)
        dq, dtac = new(7), new(ntac)
        if drew is None:
            dvar = du = None
            stride = 0
        else:
            if tuple(drew.shape) != (B,) or drew.dtype != q.dtype:
                raise ValueError("d_reward: expected [%d] %s" % (B, q.dtype))
            stride = drew.stride(0)             # 0: the broadcast scalar that the gradient of a sum is; read in place
            dvar, du = new(6), new(3)
        capi.check(capi.lib().tsim_push_observe_backward(B, ntac, _DT[q.dtype], _p(q), _p(var), _p(goal), _p(u), _p(dobs), _p(drew), stride,
                                                          _p(dq), _p(dvar), _p(dtac), _p(du), _st(q)))
        return dq, dvar, dtac, None, du


def observe_reset(q, tactile, goal):
    """The observation after reset (no reward, no gradient): [B, 3 + ntac]."""
    B, ntac = q.shape[0], tactile.shape[1]
    q, tactile, goal = _c(q.detach(), B, 7, "q"), _c(tactile.detach(), B, ntac, "tactile"), _c(goal.detach().to(q.dtype), B, 3, "goal")
    obs = torch.empty((B, 3 + ntac), device=q.device, dtype=q.dtype)
    capi.check(capi.lib().tsim_push_observe(B, ntac, _DT[q.dtype], _p(q), None, _p(tactile), _p(goal), None, _p(obs), None, _st(q)))
    return obs
