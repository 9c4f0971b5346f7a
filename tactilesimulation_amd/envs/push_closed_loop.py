"""FusedPushEpisode — the TactilePush GD epoch with the policy INSIDE the simulator's episode launches.

algorithms/gd.py:224-259 (cfg/gd_tactile.yaml) alternates policy and env-step.  Run as one launch per env-step that makes every env-step
wait for the slowest of the batch's environments (algorithms/batched_gd.GraphedRollout: 4.0 M env-steps/s at B = 4096 against 7.1 M for
the open-loop episode launch).  Here `tsim_push_closed_rollout` / `tsim_push_closed_backward` (include/tsim_env.h) evaluate the
observation, the 393-64-64-3 ELU policy and the action mapping of an environment inside its slot of the episode launch: one launch each
way per episode.  What stays in torch: the reward (elementwise, envs/tactile_push_env.py:202-211), its partial derivatives (the seeds of
the backward launch) and the weight gradients — one batched GEMM per layer over the whole episode from the per-frame (input,
pre-activation gradient) records the backward launch leaves.  Same numbers as rollout_loss(...).backward() on BatchedTactilePushEnv
(tests/test_gpu_closed_loop.py); the dependence of the first observation on the initial state is not propagated (nothing uses it).
"""
import ctypes as C
import math

import torch

from ..host import capi

_p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


OBS_MODES = {"tactile_flatten": (0, 393), "no_tactile": (1, 3), "privilege": (2, 6)}      # include/tsim_env.h tsim_push_policy.obs_mode


class FusedPushEpisode:
    def __init__(self, env, actor, horizon):
        """env: BatchedTactilePushEnv (its BatchSim, dtype, device, observation_type: tactile_flatten / no_tactile / privilege — the
        three GD configurations of examples/TactilePushExp/cfg); actor: algorithms.batched_gd.Actor (obs -> 64 -> 64 -> 3)."""
        self.env, self.actor, self.T = env, actor, int(horizon)
        self.sim, self.B, self.dt, self.dev = env.sim, env.B, env.dtype, env.device
        self.obs_type = getattr(env, "observation_type", "tactile_flatten")
        if self.obs_type not in OBS_MODES:
            raise ValueError("the policy inside the episode launches takes the observation types %s (not %r)" % (tuple(OBS_MODES), self.obs_type))
        self.mode, self.nin = OBS_MODES[self.obs_type]
        lin = [m for m in actor.mu_net if isinstance(m, torch.nn.Linear)]
        assert [tuple(m.weight.shape) for m in lin] == [(64, self.nin), (64, 64), (3, 64)], "the fused policy is the GD actor (obs-64-64-3)"
        self.lin = lin
        T, B = self.T, self.B
        new = lambda *d: torch.empty(d, device=self.dev, dtype=self.dt)
        tactile = self.mode == 0
        self.q, self.var = new(T, B, 7), new(T, B, 6)
        self.tac = new(T, B, 390) if tactile else None              # without the tactile observation the launch skips the read-out
        self.u, self.gl, self.h1, self.h2 = new(T, B, 3), new(T, B, 3), new(T, B, 64), new(T, B, 64)
        self.g1, self.g2, self.g3 = new(T, B, 64), new(T, B, 64), new(T, B, 3)
        self.dobs_tac = new(T, B, 390) if tactile else None
        self.status = torch.empty(B, device=self.dev, dtype=torch.int32)
        self.w1s = (self.nin + 3) // 4 * 4
        self.W1p = torch.zeros(64, self.w1s, device=self.dev, dtype=self.dt)
        self._pol = capi.PushPolicyStruct()

    def _weights(self):
        """The layouts the kernels read (tiny copies; the parameters themselves stay torch's)."""
        l1, l2, l3 = self.lin
        with torch.no_grad():
            # snapshots, all eight: .contiguous() on an already-contiguous parameter returns the parameter itself, and an optimizer step
            # between rollout() and backward() would then pair new W2 / W3 / biases with the old transposes and the old forward records
            self._w = [l1.weight.t().contiguous(), l1.bias.clone(), l2.weight.t().contiguous(), l2.bias.clone(),
                       l3.weight.clone(), l3.bias.clone(), self.W1p, l2.weight.clone()]
            self.W1p[:, :self.nin].copy_(l1.weight)
        for n, t in zip(("W1T", "b1", "W2T", "b2", "W3", "b3", "W1p", "W2"), self._w):
            assert t.dtype == self.dt and t.device == self.dev
            setattr(self._pol, n, t.data_ptr())
        self._pol.w1_stride = self.w1s
        self._pol.obs_mode = self.mode
        sample, norm = getattr(self, "_sample", None), getattr(self, "_norm", None)
        self._pol.eps, self._pol.logstd = (sample[0].data_ptr(), sample[1].data_ptr()) if sample else (None, None)
        self._pol.obs_mean, self._pol.obs_istd, self._pol.obs_clip = (norm[0].data_ptr(), norm[1].data_ptr(), norm[2]) if norm else (None, None, 0.0)
        return C.byref(self._pol)

    def collect(self, q0, goal, disturbances, eps=None, obs_mean=None, obs_var=None, obs_clip=10.0, var_eps=1e-8):
        """One episode of roll-out collection as cfg/ppo_tactile.yaml runs it (the same actor, stochastic, on normalised observations),
        without a tape: u = mean + exp(logstd) eps with eps [T, B, 3] standard-normal draws (None: the deterministic mean), the
        observation normalised as VecNormalize does — clamp((obs - mean) / sqrt(var + 1e-8), +-clip) — with the statistics frozen for
        the episode (None: raw observations).  Returns dict(obs [T, B, nin] raw observations the policy saw before each env-step,
        action [T, B, 3] (pre-tanh, as the env takes it), reward [T, B], q, var)."""
        self._sample = None
        if eps is not None:
            eps = eps.to(self.dev, self.dt).contiguous()
            assert tuple(eps.shape) == (self.T, self.B, 3)
            self._sample = (eps, self.actor.logstd.detach().to(self.dev, self.dt).contiguous())
        self._norm = None
        if obs_mean is not None:
            m = obs_mean.to(self.dev, self.dt).reshape(-1).contiguous()
            istd = (1.0 / torch.sqrt(obs_var.to(self.dev, torch.float64).reshape(-1) + var_eps)).to(self.dt).contiguous()
            assert m.numel() == self.nin and istd.numel() == self.nin
            self._norm = (m, istd, float(obs_clip))
        try:
            self.rollout(q0, goal, disturbances, record=False)
        finally:
            self._sample = self._norm = None
        return {"obs": self.observations(), "action": self.u, "reward": self.rewards, "q": self.q, "var": self.var}

    def observations(self):
        """[T, B, nin]: the (raw) observation in front of every env-step of the last roll-out, assembled from its records."""
        if self.mode == 0:
            tprev = torch.cat([self.tac0.unsqueeze(0), self.tac[:-1]], dim=0)
            return torch.cat([self.gl, tprev], dim=2)
        if self.mode == 1:
            return self.gl
        qb = torch.cat([self.q0.unsqueeze(0), self.q[:-1]], dim=0)
        th = qb[:, :, 0]
        c, s_ = torch.cos(th), torch.sin(th)
        obj = torch.stack([c * qb[:, :, 3] + s_ * qb[:, :, 4] - qb[:, :, 1], -s_ * qb[:, :, 3] + c * qb[:, :, 4] - qb[:, :, 2], qb[:, :, 6] - th], dim=2)
        return torch.cat([obj, self.gl], dim=2)

    def evaluate(self, q0, goal, disturbances):
        """The episode without a tape (algorithms/gd.py:265-290 evaluates the deterministic policy this way between epochs): per-environment
        return [B] (sum of rewards).  backward() is not available afterwards."""
        self.rollout(q0, goal, disturbances, record=False)
        return self.returns

    def rollout(self, q0, goal, disturbances, record=True):
        """One episode forward.  q0 [B, 7], goal [B, 3], disturbances [T, B, 2].  Returns -sum of rewards (a 0-d tensor; its
        partials w.r.t. the frames' outputs are kept for backward())."""
        env, sim = self.env, self.sim
        self.goal = goal.to(self.dev, self.dt).contiguous()
        dist = disturbances.to(self.dev, self.dt).contiguous()
        self.q0 = q0.to(self.dev, self.dt).contiguous()
        sim.reset(self.q0, None, backward_flag=bool(record))
        self._recorded = bool(record)
        tac0 = sim.readout(want_var=False)[1] if self.mode == 0 else None
        self.tac0 = tac0
        pol = self._weights()
        st = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        capi.check(capi.lib().tsim_push_closed_rollout(sim._h, pol, _p(self.goal), _p(dist), _p(tac0), self.T, env.frame_skip,
                                                       _p(self.q), None, _p(self.var), _p(self.tac), _p(self.u), _p(self.gl), _p(self.h1), _p(self.h2),
                                                       _p(self.status), st))
        # reward of every frame and its partials (envs/tactile_push_env.py:202-211), elementwise over [T, B, .]
        g = self.goal.unsqueeze(0)
        dp = self.q[:, :, 3:5] - g[:, :, 0:2]
        dr = self.q[:, :, 6] - g[:, :, 2]
        dt_ = self.var[:, :, 0:3] - self.var[:, :, 3:6]
        k = (36.0 / math.pi) ** 2
        rew = -(dp ** 2).sum(2) * 100.0 - dr ** 2 * (0.1 * k) - (dt_ ** 2).sum(2) * 2500.0 - (self.u ** 2).sum(2) * 0.1
        self.df_dq = torch.zeros_like(self.q)                       # d(-sum rew)/dq, /dvar, /du
        self.df_dq[:, :, 3:5] = 200.0 * dp
        self.df_dq[:, :, 6] = (0.2 * k) * dr
        self.df_dvar = torch.cat([5000.0 * dt_, -5000.0 * dt_], dim=2)
        self.du_direct = 0.2 * self.u
        self.rewards = rew                                          # [T, B]
        self.returns = rew.sum(0)                                   # per environment
        self.loss = -rew.sum()
        return self.loss

    def backward(self):
        """Adjoint launch of the episode rollout() ran, then the weight gradients: sets .grad of the actor's weights and biases
        (assigned, not accumulated; un-normalised, as rollout_loss(...).backward() would)."""
        sim = self.sim
        if not getattr(self, "_recorded", False):
            raise RuntimeError("FusedPushEpisode.backward: the last roll-out was not recorded (evaluate())")
        st = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        capi.check(capi.lib().tsim_push_closed_backward(sim._h, C.byref(self._pol), _p(self.goal), self.T, self.env.frame_skip,
                                                        _p(self.df_dq), _p(self.df_dvar), _p(self.du_direct), _p(self.u), _p(self.h1), _p(self.h2),
                                                        _p(self.g1), _p(self.g2), _p(self.g3), _p(self.dobs_tac), None, st))
        l1, l2, l3 = self.lin
        g1, g2, g3 = self.g1, self.g2, self.g3
        T = self.T
        w1 = torch.empty_like(l1.weight)
        go = 3 if self.mode == 2 else 0
        w1[:, go:go + 3] = torch.bmm(g1.transpose(1, 2), self.gl).sum(0)
        if self.mode == 0:
            # x = [gl, tactile frame before the env-step]: tac0 for frame 0, the previous frame's read-out otherwise (no concatenated copy)
            w1[:, 3:] = g1[0].t() @ self.tac0
            if T > 1:
                w1[:, 3:] += torch.bmm(g1[1:].transpose(1, 2), self.tac[:-1]).sum(0)
        elif self.mode == 2:
            # x = [box pose in the gripper frame of the state before the env-step, gl] (tactile_push_env.py:96-106)
            qb = torch.cat([self.q0.unsqueeze(0), self.q[:-1]], dim=0)
            th = qb[:, :, 0]
            c, s_ = torch.cos(th), torch.sin(th)
            obj = torch.stack([c * qb[:, :, 3] + s_ * qb[:, :, 4] - qb[:, :, 1], -s_ * qb[:, :, 3] + c * qb[:, :, 4] - qb[:, :, 2], qb[:, :, 6] - th], dim=2)
            w1[:, 0:3] = torch.bmm(g1.transpose(1, 2), obj).sum(0)
        l1.weight.grad, l1.bias.grad = w1, g1.sum((0, 1))
        l2.weight.grad, l2.bias.grad = torch.bmm(g2.transpose(1, 2), self.h1).sum(0), g2.sum((0, 1))
        l3.weight.grad, l3.bias.grad = torch.bmm(g3.transpose(1, 2), self.h2).sum(0), g3.sum((0, 1))
        return self.loss


def train_epoch_fused(ep, optimizer, q0, goal, disturbances, global_episodes, grad_clip=1.0):
    """algorithms/batched_gd.train_epoch with the episode and its adjoint as one launch each (FusedPushEpisode)."""
    from ..dist import allreduce_policy_grad_
    for p in ep.actor.parameters():
        p.grad = None
    loss = ep.rollout(q0, goal, disturbances)
    ep.backward()
    params = list(ep.actor.parameters())
    allreduce_policy_grad_(params, global_episodes)
    if grad_clip:
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], grad_clip)
    optimizer.step()
    return loss
