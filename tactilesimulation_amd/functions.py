"""Autograd surface of the hot path.

* `StepSimFunction`, `EpisodicSimFunction` — same names, argument order, outputs and simulator call protocol as the
  reference's envs/redmax_torch_functions.py:11-174 (one environment, a `redmax_py.Simulation`-like object; numpy
  float64 across the binding).  The reference's own file also runs unmodified on top of compat/redmax_py.py; these
  re-statements exist so that the package is usable without the reference checkout and so that the call protocol is
  pinned by tests/test_protocol.py against a trace recorded from the reference's functions.
* `BatchedStepSimFunction` — the MI355X-native counterpart: B environments per call, tensors stay on the device, one
  forward kernel launch per env-step and one adjoint launch per env-step (tactilesimulation_amd.host.BatchSim).
* `BatchedEpisodicSimFunction` — batched counterpart of EpisodicSimFunction: the whole open-loop episode of B
  environments in ONE forward launch and ONE adjoint launch (tsim_rollout / tsim_backward_episode).
"""
import numpy as np
import torch
from torch import autograd


def _to_torch(x, dtype, device, requires_grad):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


class StepSimFunction(autograd.Function):
    """(action[ndof_u], num_steps, sim, grad_mode) -> q[ndof_r], var[ndof_var], tactile[ndof_tactile]
    Holds `action` for num_steps sub-steps and returns the last frame.  State-to-state gradient flow lives inside the
    simulator (tape + carried adjoint); only `action` is a differentiable input."""

    @staticmethod
    def forward(ctx, action, num_steps, sim, grad_mode):
        ctx.sim, ctx.num_steps = sim, num_steps
        ctx.need_du = action.requires_grad
        ctx.device, ctx.dtype = action.device, action.dtype
        sim.set_u(action.detach().cpu().numpy())
        sim.forward(num_steps, verbose=False, test_derivatives=False, save_last_frame_var_only=True)
        q = _to_torch(sim.get_q().copy(), ctx.dtype, ctx.device, grad_mode)
        var = _to_torch(sim.get_variables().copy(), ctx.dtype, ctx.device, grad_mode)
        tactile = _to_torch(sim.get_tactile_force_vector(), ctx.dtype, ctx.device, grad_mode)
        return q, var, tactile

    @staticmethod
    def backward(ctx, df_dq, df_dvar, df_dtactile):
        sim, n = ctx.sim, ctx.num_steps
        sim.backward_info.set_flags(flag_q0=False, flag_qdot0=False, flag_p=False, flag_u=ctx.need_du)

        def last_block(g, dim):          # partials are w.r.t. the LAST of the n sub-steps; earlier blocks are zero
            full = np.zeros(dim * n)
            if dim:
                full[-dim:] = g.reshape(-1).detach().cpu().numpy()
            return full
        sim.backward_info.df_dq = last_block(df_dq, sim.ndof_r)
        sim.backward_info.df_dvar = last_block(df_dvar, sim.ndof_var)
        sim.backward_info.df_dtactile = last_block(df_dtactile, sim.ndof_tactile)
        sim.backward_info.df_du = np.zeros(sim.ndof_u * n)
        sim.backward_steps(n)
        if not ctx.need_du:
            return None, None, None, None
        du = sim.backward_results.df_du.copy().reshape(n, sim.ndof_u)    # autograd sum-reduces (n, nu) -> (nu,)
        return _to_torch(du, ctx.dtype, ctx.device, True), None, None, None


class EpisodicSimFunction(autograd.Function):
    """(q0, qdot0, actions[T, ndof_u], tactile_masks[T] bool, sim, grad_mode) -> qs[T, ndof_r], vars[T, ndof_var],
    tactiles[sum(mask), ndof_tactile].  One sub-step per action."""

    @staticmethod
    def forward(ctx, q0, qdot0, actions, tactile_masks, sim, grad_mode):
        T = actions.shape[0]
        ctx.sim, ctx.T = sim, T
        ctx.need_q0, ctx.need_qdot0, ctx.need_du = q0.requires_grad, qdot0.requires_grad, actions.requires_grad
        ctx.device, ctx.dtype = q0.device, q0.dtype
        ctx.tactile_masks = tactile_masks
        a_np = actions.detach().cpu().numpy()
        sim.set_state_init(q0.detach().cpu().numpy(), qdot0.detach().cpu().numpy())
        sim.reset(backward_flag=grad_mode)
        qs, vs, ts = [], [], []
        for t in range(T):
            sim.set_u(a_np[t])
            sim.forward(1, verbose=False, test_derivatives=False)
            qs.append(_to_torch(sim.get_q().copy(), ctx.dtype, ctx.device, grad_mode))
            vs.append(_to_torch(sim.get_variables().copy(), ctx.dtype, ctx.device, grad_mode))
            if tactile_masks[t]:
                ts.append(_to_torch(sim.get_tactile_force_vector().copy(), ctx.dtype, ctx.device, grad_mode))
        if grad_mode:
            sim.saveBackwardCache()
        return torch.stack(qs, dim=0), torch.stack(vs, dim=0), torch.stack(ts, dim=0)

    @staticmethod
    def backward(ctx, df_dq, df_dvar, df_dtactile):
        sim, T = ctx.sim, ctx.T
        sim.popBackwardCache()
        sim.backward_info.set_flags(flag_q0=ctx.need_q0, flag_qdot0=ctx.need_qdot0, flag_p=False, flag_u=ctx.need_du)
        sim.backward_info.df_dq = df_dq.reshape(-1).detach().cpu().numpy()
        sim.backward_info.df_dvar = df_dvar.reshape(-1).detach().cpu().numpy()
        # As the reference does (envs/redmax_torch_functions.py:87): only the frames the mask selected, sum(mask) x ndof_tactile values.
        # The simulator knows which sub-steps had their tactile frame read (compat/redmax_py.py records every get_tactile_force_vector()
        # after a forward) and puts each block on its sub-step.
        sim.backward_info.df_dtactile = df_dtactile.reshape(-1).detach().cpu().numpy()
        sim.backward_info.df_dq0 = np.zeros(sim.ndof_r)
        sim.backward_info.df_dqdot0 = np.zeros(sim.ndof_r)
        sim.backward_info.df_du = np.zeros(sim.ndof_u * T)
        sim.backward()
        res = sim.backward_results
        g_q0 = _to_torch(res.df_dq0.copy(), ctx.dtype, ctx.device, True) if ctx.need_q0 else None
        g_qd0 = _to_torch(res.df_dqdot0.copy(), ctx.dtype, ctx.device, True) if ctx.need_qdot0 else None
        g_u = _to_torch(res.df_du.copy().reshape(T, sim.ndof_u), ctx.dtype, ctx.device, True) if ctx.need_du else None
        return g_q0, g_qd0, g_u, None, None, None


class BatchedStepSimFunction(autograd.Function):
    """(action[B, ndof_u], num_steps, batch_sim, grad_mode) -> q[B, ndof_r], var[B, ndof_var], tactile[B, ndof_tactile]

    Batched, device-resident counterpart of StepSimFunction: no numpy hop, one HIP launch forward and one backward per
    env-step for all B environments.  As in the reference, outputs are fresh leaves of the autograd graph (the
    state-to-state chain is carried inside the simulator), so backward calls must arrive newest-first — which autograd
    guarantees whenever action_{t+1} depends on the outputs of step t, and which callers of open-loop roll-outs get by
    summing a loss over steps in order."""

    @staticmethod
    def forward(ctx, action, num_steps, sim, grad_mode):
        ctx.sim, ctx.num_steps = sim, num_steps
        ctx.need_du = action.requires_grad
        ctx.in_dtype = action.dtype
        out = sim.step(action.detach(), num_steps)
        q, var, tac = out["q"], out.get("var"), out.get("tactile")
        if var is None:
            var = q.new_zeros((sim.B, 0))
        if tac is None:
            tac = q.new_zeros((sim.B, 0))
        ctx.status = out["status"]
        res = tuple(t.to(action.dtype) for t in (q, var, tac))
        if not grad_mode:
            ctx.mark_non_differentiable(*res)
        return res

    @staticmethod
    def backward(ctx, df_dq, df_dvar, df_dtactile):
        sim, n = ctx.sim, ctx.num_steps
        # one frame of n sub-steps: the kernel sums df_du over the frame's sub-steps itself (include/tsim.h tsim_backward_episode)
        f = lambda t, on: t.to(sim.dtype).unsqueeze(0) if (on and t is not None) else None
        du = sim.backward_episode(1, n, f(df_dq, True), f(df_dvar, sim.ndof_var), f(df_dtactile, sim.ndof_tactile))
        if not ctx.need_du:
            return None, None, None, None
        return du[0].to(ctx.in_dtype), None, None, None


class BatchedEpisodicSimFunction(autograd.Function):
    """(q0[B, ndof_r], qdot0[B, ndof_r], actions[T, B, ndof_u], tactile_masks bool[T], batch_sim, grad_mode, num_steps=1)
        -> qs[T, B, ndof_r], vars[T, B, ndof_var], tactiles[sum(mask), B, ndof_tactile]

    Batched, device-resident counterpart of EpisodicSimFunction (envs/redmax_torch_functions.py:11-109): same argument
    order and meaning with a batch axis after the time axis, the reference's `forward(1)` per action generalised to
    `num_steps` sub-steps per action.  Like the reference it saves the tape on the backward cache in forward and pops it
    in backward, so several episodes may be forwarded before their backward passes (newest first)."""

    @staticmethod
    def forward(ctx, q0, qdot0, actions, tactile_masks, sim, grad_mode, num_steps=1):
        ctx.sim, ctx.T, ctx.num_steps = sim, int(actions.shape[0]), int(num_steps)
        ctx.need = (q0.requires_grad, qdot0.requires_grad, actions.requires_grad)
        ctx.in_dtype = actions.dtype
        ctx.tactile_masks = tactile_masks
        sim.reset(q0.detach(), qdot0.detach(), backward_flag=grad_mode)
        out = sim.rollout(actions.detach(), ctx.num_steps, tactile_mask=tactile_masks)
        qs, vars_, tacs = out["q"], out.get("var"), out.get("tactile")
        if vars_ is None:
            vars_ = qs.new_zeros((ctx.T, sim.B, 0))
        if tacs is None:
            tacs = qs.new_zeros((0, sim.B, 0))
        ctx.status = out["status"]
        if grad_mode:
            sim.cache_save()
        res = tuple(t.to(actions.dtype) for t in (qs, vars_, tacs))
        if not grad_mode:
            ctx.mark_non_differentiable(*res)
        return res

    @staticmethod
    def backward(ctx, df_dq, df_dvar, df_dtactile):
        sim = ctx.sim
        sim.cache_pop()
        du = sim.backward_episode(ctx.T, ctx.num_steps, df_dq, df_dvar if sim.ndof_var else None,
                                  df_dtactile if (sim.ndof_tactile and df_dtactile.shape[0] > 0) else None,
                                  tactile_mask=ctx.tactile_masks)
        lq, lv = sim.get_adjoint()
        g = (lq.to(ctx.in_dtype) if ctx.need[0] else None, lv.to(ctx.in_dtype) if ctx.need[1] else None,
             du.to(ctx.in_dtype) if ctx.need[2] else None)
        return g + (None, None, None, None)
