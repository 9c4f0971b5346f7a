"""BatchedTactilePushEnv — B TactilePush-v1 environments stepped as one batch on one GPU.

Vectorised counterpart of the reference's envs/tactile_push_env.py (use_torch; all four observation types of :72-131 —
"tactile_flatten" of cfg/gd_tactile.yaml, "no_tactile" of gd_no_tactile.yaml, "privilege" of gd_privilege.yaml, "tactile_map" of
ppo_tactile.yaml):
same action mapping (:175-193), observation (:72-131) and reward (:202-211), evaluated for all B environments in torch
on the device by the fused kernels of include/tsim_env.h (envs/push_ops.py: one launch each way for the action mapping, one for
observation + reward); the simulator step is BatchedStepSimFunction (one HIP launch forward, one backward per env-step). The
per-environment gym wrapper of the reference keeps working through compat/redmax_py.py; this class is what SURVEY.md
§8(f).1 calls the batched counterpart.
"""
import math
import os

import numpy as np
import torch

from ..functions import BatchedStepSimFunction
from .push_ops import PushAction, PushObserve, observe_reset
from ..host.batch import BatchSim
from ..model.compiler import load_model


OBSERVATION_TYPES = ("tactile_flatten", "no_tactile", "privilege", "tactile_map")


def shape_observation(observation_type, obs393, q, rows=13, cols=10):
    """The observation of tactile_push_env.py:72-131 for `observation_type`, from the "tactile_flatten" one ([B, 3 + 390]: goal pose in
    the gripper frame, tactile frame) and the state q [B, 7] — plain torch, differentiable:
      tactile_flatten  [B, 393]                                 (:113-118)
      no_tactile       [B, 3]     the goal in the gripper frame (:126-131)
      privilege        [B, 6]     box position and yaw in the gripper frame, then the goal (:104-111)
      tactile_map      ([B, 3, rows, cols], [B, 3])             the taxel forces as three images, and the goal (:78-81, :119-125)"""
    if observation_type == "tactile_flatten":
        return obs393
    gl = obs393[:, 0:3]
    if observation_type == "no_tactile":
        return gl
    if observation_type == "privilege":
        th = q[:, 0]
        c, s = torch.cos(-th), torch.sin(-th)
        ox, oy = q[:, 3], q[:, 4]
        obj = torch.stack([c * ox - s * oy - q[:, 1], s * ox + c * oy - q[:, 2], q[:, 6] - th], dim=1)
        return torch.cat([obj, gl], dim=1)
    if observation_type == "tactile_map":
        B = obs393.shape[0]
        return obs393[:, 3:].reshape(B, rows, cols, 3).permute(0, 3, 1, 2), gl
    raise ValueError("observation_type must be one of %s" % (OBSERVATION_TYPES,))


class BatchedTactilePushEnv:
    tactile_rows, tactile_cols = 13, 10           # envs/tactile_push_env.py:31-32
    frame_skip = 5                                # :66
    max_episode_steps = 100                       # envs/__init__.py:9-13

    def __init__(self, model, batch_size, device="cuda:0", dtype=torch.float32, gradient=True, seed=0, tape_steps=None,
                 observation_type="tactile_flatten"):
        if observation_type not in OBSERVATION_TYPES:
            raise ValueError("observation_type must be one of %s" % (OBSERVATION_TYPES,))
        self.observation_type = observation_type
        if isinstance(model, str):
            model = load_model(model)
        self.B, self.device, self.dtype, self.gradient = int(batch_size), torch.device(device), dtype, bool(gradient)
        T = tape_steps if tape_steps is not None else self.max_episode_steps
        self.sim = BatchSim(model, self.B, device=device, dtype=dtype, tape_capacity=T * self.frame_skip)
        assert (self.sim.ndof_r, self.sim.ndof_u, self.sim.ndof_var, self.sim.ndof_tactile) == (7, 6, 6, 390)
        self.rng = np.random.default_rng(seed)
        self.obs_dim, self.act_dim = {"tactile_flatten": 3 + 390, "no_tactile": 3, "privilege": 6, "tactile_map": (3 * 130, 3)}[observation_type], 3
        self.dt = self.sim.h * self.frame_skip
        self.current_step = 0
        # The reference tests `current_step % 10 == 0` before drawing a new disturbance (tactile_push_env.py:185) but never increments
        # current_step (it is only ever set to 0, :62 and :167), so it draws a new force at EVERY env-step.  1 reproduces that; 10 is
        # what the line reads like.
        self.disturbance_period = 1

    # ------------------------------------------------------------------ helpers
    def _t(self, a):
        if torch.is_tensor(a):                       # device tensors pass through (no host round trip: graph capture)
            return a.to(device=self.device, dtype=self.dtype)
        return torch.as_tensor(np.asarray(a), device=self.device, dtype=self.dtype)

    # ------------------------------------------------------------------ gym-like API, batched
    def reset(self, q0=None, goal=None):
        """Per-environment draws of tactile_push_env.py:133-172 (box y offset, goal xy, goal yaw), or explicit tables."""
        B = self.B
        if q0 is None:
            q0 = np.zeros((B, 7))
            q0[:, 1] = -0.001
            q0[:, 4] = self.rng.uniform(-0.02, 0.02, size=B)
        if goal is None:
            goal = np.zeros((B, 3))
            goal[:, 0:2] = self.rng.uniform([0.15, -0.2], [0.25, 0.2], size=(B, 2))
            goal[:, 2] = self.rng.uniform(goal[:, 1] * math.pi - math.pi / 16.0, goal[:, 1] * math.pi + math.pi / 16.0)
        self.q0, self.goal = self._t(q0), self._t(goal)
        self.sim.reset(self.q0, None, backward_flag=self.gradient)
        _, tac = self.sim.readout(want_var=False)
        self.external_force = torch.zeros(B, 2, device=self.device, dtype=self.dtype)
        self.current_step = 0
        return shape_observation(self.observation_type, observe_reset(self.q0, tac, self.goal), self.q0, self.tactile_rows, self.tactile_cols)

    def step(self, u, disturbance=None):
        """u: policy output [B, 3] (pre-tanh). Returns the observation (shape_observation), reward [B], info dict."""
        if disturbance is not None:
            self.external_force = disturbance.to(self.device, self.dtype)
        elif self.current_step % self.disturbance_period == 0:              # :185-190
            on = self._t(self.rng.uniform(0.0, 1.0, size=self.B) < 0.5).unsqueeze(1)
            self.external_force = on * self._t(self.rng.uniform(-1.0, 1.0, size=(self.B, 2)))
        robot_action = PushAction.apply(u, self.external_force)                                    # [tanh(u), force on the box, 0]
        q, var, tactile = BatchedStepSimFunction.apply(robot_action, self.frame_skip, self.sim, self.gradient)
        self.current_step += 1
        obs, rew = PushObserve.apply(q, var, tactile, self.goal, u)
        return shape_observation(self.observation_type, obs, q, self.tactile_rows, self.tactile_cols), rew, {"q": q, "var": var}

    def reward_terms(self, q, var, u):
        """The four terms of the reward (tactile_push_env.py:202-211) in plain torch, for logging; step() computes their sum in
        the fused kernel."""
        r_pos = -(((q[:, 3:5] - self.goal[:, 0:2]) / 0.01) ** 2).sum(1) * 0.01
        r_rot = -(((q[:, 6] - self.goal[:, 2]) / (math.pi / 36.0)) ** 2) * 0.1
        r_touch = -((var[:, 0:3] - var[:, 3:6]) ** 2).sum(1) / (0.02 ** 2)
        r_act = -(u ** 2).sum(1) * 0.1
        return {"reward_pos": r_pos, "reward_rot": r_rot, "reward_touch": r_touch, "reward_action": r_act}
