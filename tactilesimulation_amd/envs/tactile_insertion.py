"""BatchedTactileInsertionEnv — B Insertion-v3 environments (the reference's envs/tactile_insertion_env.py, observation_type
"tactile_flatten") as one batch on one GPU: BASELINE configs[4].  One env-step is one whole insertion ATTEMPT: the policy moves the
pre-grasp pose, then 45 open-loop sub-steps lower the gripped box towards the hole while six tactile frames are captured (one launch:
tsim_rollout with a tactile mask, the batched EpisodicSimFunction.forward).  Forward-only (the reference trains this task with PPO).

The environment-side arithmetic is the reference's, pinned on golden vectors recorded from its own class against a scripted simulator
(tools/make_insertion_env_fixture.py -> tests/golden/insertion_env.npz, tests/test_insertion_env_golden.py):
  * action -> relative motion of the pre-grasp pose, clipped to the working space (:300-318; incl. the reference's asymmetric upper bound
    of the rotation, `working_rotation_boundary` rather than `working_rotation_boundary - q[3]`)
  * apply_relative_motion (:174-194): gripper and box move together, the box's rotation vector is composed with utils.rotvec_mul
  * the 45 x 6 joint-target table of an attempt (:344-357) and the tactile frames it captures: sub-steps 6 (reference), 20, 26, 32, 38, 44
  * observation (:359-383): shear components of the five frames relative to the reference frame, per-environment normalisation to a
    largest vector of 30, flattened (5, 2, 13, 10, 2)
  * success (:387-391), reward "absolute" / "delta" (:402-410), done = success.
Domain randomisation (:238-281: contact and tactile parameters, grasp force) becomes per-environment tables and a per-environment grasp
force.
"""
import math

import numpy as np
import torch

from ..host.batch import BatchSim
from ..model import compiler as mc
from ..utils.torch_utils import rotvec_mul
from ..workloads import asset

EXECUTION_STEPS = 45                                                  # :53
TACTILE_FRAMES = (6, 20, 26, 32, 38, 44)                              # :76-77 with tactile_initial_frame 15, 5 frames
MAX_ERROR = (0.006, 0.006, math.pi / 18.0)                            # :36
# The reference keeps three of its constants in float32 tensors (torch.tensor([...]) of python floats: action_scale :104-107,
# working_space_boundary :33), so their float32 roundings are the values it computes with — also when the state is float64.
_f32 = lambda v: float(np.float32(v))
WORKSPACE_XY, WORKSPACE_ROT = _f32(0.015), math.pi / 12.0             # :33-34
XY_SCALE, ROT_SCALE = _f32(0.02), _f32(math.pi / 18.0)                # action_xy_scale, action_rot_scale (:24-25)


def apply_relative_motion(q, dpos, drot, grasp_height_noise=None):
    """q [B, 12]; dpos [B, 2] (xy) or [B, 3]; drot [B] -> the pre-grasp state moved rigidly (gripper dofs 0:4 and box dofs 6:12)."""
    new = q.clone()
    n = dpos.shape[1]
    new[:, 0:n] += dpos
    new[:, 6:6 + n] += dpos
    if grasp_height_noise is not None:
        new[:, 2] += grasp_height_noise
    new[:, 3] = new[:, 3] + drot
    z = torch.zeros_like(drot)
    new[:, 9:12] = rotvec_mul(q[:, 9:12], torch.stack([z, z, drot], dim=1))
    return new


def relative_motion_of_action(u, q_init, xy_scale=XY_SCALE, rot_scale=ROT_SCALE):
    """u [B, 3] policy output -> (dxy [B, 2], drot [B]) after scaling and clipping to the working space (action_type "relative")."""
    a = torch.clamp(u, -1.0, 1.0) * torch.tensor([xy_scale, xy_scale, rot_scale], dtype=u.dtype, device=u.device)
    dxy = torch.minimum(torch.maximum(a[:, 0:2], -WORKSPACE_XY - q_init[:, 0:2]), WORKSPACE_XY - q_init[:, 0:2])
    drot = torch.minimum(torch.maximum(a[:, 2], -WORKSPACE_ROT - q_init[:, 3]), torch.full_like(a[:, 2], WORKSPACE_ROT))
    return dxy, drot


def insertion_actions(q_init, grasp_force):
    """q_init [B, 12], grasp_force [B] or float -> joint targets [45, B, 6] of one attempt."""
    init = q_init[:, :6]
    target = init.clone()
    target[:, 2] -= 0.0011
    frac = torch.arange(1, EXECUTION_STEPS + 1, dtype=q_init.dtype, device=q_init.device)[:, None, None] / EXECUTION_STEPS
    act = (target - init)[None] * frac + init[None]
    act[:, :, 2] += 0.003                                             # feed-forward term
    act[:, :, 4] = grasp_force
    act[:, :, 5] = grasp_force
    return act.to(torch.float32).to(q_init.dtype)                     # the reference assembles the table in a float32 tensor (:345)


def observation(tactiles, normalize=True):
    """tactiles [6, B, 780] (frame 0 = reference) -> [B, 2600]."""
    B = tactiles.shape[1]
    rel = (tactiles[1:] - tactiles[0:1]).permute(1, 0, 2).reshape(B, 5, 2, 13, 10, 3)[..., 0:2]
    if normalize:
        mx = rel.norm(dim=-1).reshape(B, -1).max(dim=1).values + 1e-5
        rel = rel / (mx / 30.0)[:, None, None, None, None, None]
    return rel.reshape(B, -1)


def reward_done(q_init, q_last, prev_pose, reward_type="absolute", allow_rotation=True):
    """-> reward [B], success [B] (= done), current pose [B, 3]."""
    cur = torch.stack([q_init[:, 0], q_init[:, 1], q_init[:, 3]], dim=1)
    success = (q_last[:, 8] < 0.0247) if allow_rotation else ((q_last[:, 6].abs() <= 0.0022) & (q_last[:, 7].abs() <= 0.0022))
    if reward_type == "absolute":
        r = -(q_init[:, 0:2] ** 2).sum(1) * 10000.0 - q_init[:, 3] ** 2 * 20.0
    elif reward_type == "delta":
        me = torch.tensor(MAX_ERROR, dtype=q_init.dtype, device=q_init.device)
        r = ((prev_pose / me).norm(dim=1) - (cur / me).norm(dim=1)) * 10.0 + torch.where(success, 20.0, -1.0).to(q_init.dtype)
    else:
        raise NotImplementedError(reward_type)
    return r, success, cur


class BatchedTactileInsertionEnv:
    max_episode_steps = 15                                            # envs/__init__.py

    def __init__(self, batch_size, model=None, device="cuda:0", dtype=torch.float32, seed=0, reward_type="absolute", allow_rotation=True,
                 domain_randomization=False):
        self.model = mc.load_model(asset("tactile_insertion")) if model is None else model
        self.B, self.device, self.dtype = int(batch_size), torch.device(device), dtype
        self.reward_type, self.allow_rotation, self.domain_randomization = reward_type, allow_rotation, domain_randomization
        self.sim = BatchSim(self.model, self.B, device=device, dtype=dtype, tape_capacity=0)
        assert (self.sim.ndof_r, self.sim.ndof_u, self.sim.ndof_var, self.sim.ndof_tactile) == (12, 6, 0, 780)
        self.gen = torch.Generator(device=self.device); self.gen.manual_seed(seed)
        self.obs_dim, self.act_dim = 2600, 3 if allow_rotation else 2
        self.mask = torch.zeros(EXECUTION_STEPS, dtype=torch.bool)
        self.mask[list(TACTILE_FRAMES)] = True
        self.q_init_reference = self._generate_initial_pose()
        self.grasp_force = torch.ones(self.B, device=self.device, dtype=dtype)
        self.current_q_init = self.q_init_reference.repeat(self.B, 1)
        self.prev_pose = torch.zeros(self.B, 3, device=self.device, dtype=dtype)
        self.steps = torch.zeros(self.B, device=self.device, dtype=torch.long)

    def _generate_initial_pose(self):
        """The settled grasp every episode starts from (:126-170): close the fingers on the box in three scripted stages, lift the state by
        the object height, hold for 500 sub-steps.  The same for all environments: computed once, on a one-environment batch."""
        one = BatchSim(self.model, 1, device=str(self.device), dtype=torch.float64, tape_capacity=0)
        q = np.zeros(12); q[2], q[4], q[5] = 0.2, -0.03, -0.03
        one.reset(torch.tensor(q[None]), None, backward_flag=False)
        tq = [np.array([q[0], q[1], q[2], q[4], 0.0, 0.0]), np.array([0.0, 0.0, 0.2, 0.0, 0.0, 0.0]), np.array([0.0, 0.0, 0.2, 0.0, 1.0, 1.0]), np.array([0.0, 0.0, 0.2, 0.0, 1.0, 1.0])]
        rows = []
        for stage, n in enumerate((100, 100, 300)):
            rows += [(tq[stage + 1] - tq[stage]) / n * (i + 1) + tq[stage] for i in range(n)]
        one.rollout(torch.tensor(np.array(rows)[:, None, :], device=self.device), 1, want_tactile=False)
        qs, _ = one.get_state()
        qs = qs[0].clone()
        qs[2] += 0.029; qs[8] += 0.029                                # initial_object_height 0.026 + 0.003
        one.reset(qs[None], None, backward_flag=False)
        u = qs[:6].clone(); u[4:6] = 1.0
        one.step(u[None], 500, want_tactile=False)
        q_ref, _ = one.get_state()
        return q_ref.to(self.dtype)

    def _randomize(self, m):
        """Per-environment contact / tactile parameters and grasp force for the environments in m (:238-281)."""
        if getattr(self, "tables", None) is None:
            self.tables = self.sim.base_tables()
        U = lambda lo, hi: lo + (hi - lo) * torch.rand(self.B, device=self.device, dtype=self.dtype, generator=self.gen)
        draws = {"pair": {"kn": U(2e3, 14e3), "kt": U(20.0, 140.0), "mu": U(0.5, 2.5), "damping": U(1e3, 1e3)},
                 "sensor": {"kn": U(50.0, 450.0), "kt": U(0.2, 2.3), "mu": U(0.5, 2.5), "damping": U(0.0, 100.0)}}
        for pad in ("tactile_pad_left", "tactile_pad_right"):
            for f, v in draws["pair"].items():
                c = self.model.table_offset("pair", (pad, "box"), f)
                self.tables[:, c] = torch.where(m, v, self.tables[:, c])
            for f, v in draws["sensor"].items():
                c = self.model.table_offset("sensor", pad, f)
                self.tables[:, c] = torch.where(m, v, self.tables[:, c])
        self.grasp_force = torch.where(m, U(1.0 / 8.0, 0.8), self.grasp_force)
        self.sim.set_env_tables(self.tables)

    def _execute(self):
        """One insertion attempt of every environment from its current pre-grasp state: ONE forward launch."""
        self.sim.reset(self.current_q_init, None, backward_flag=False)
        ro = self.sim.rollout(insertion_actions(self.current_q_init, self.grasp_force), 1, want_var=False, tactile_mask=self.mask)
        obs = observation(ro["tactile"])
        r, success, cur = reward_done(self.current_q_init, ro["q"][-1], self.prev_pose, self.reward_type, self.allow_rotation)
        self.prev_pose = cur
        return obs, r, success, ro["status"]

    def _new_episodes(self, m):
        """Random pre-grasp states for the environments in m (:200-216): the reference pose moved by U(+-6 mm, +-6 mm, +-0.2 mm), rotated by
        U(+-10 deg) when rotation is allowed, grasp height U(-10 mm, 5 mm); new contact / tactile parameters when randomisation is on."""
        B = self.B
        U = lambda lo, hi: lo + (hi - lo) * torch.rand(B, device=self.device, dtype=self.dtype, generator=self.gen)
        pos = torch.stack([U(-MAX_ERROR[0], MAX_ERROR[0]), U(-MAX_ERROR[1], MAX_ERROR[1]), U(-0.0002, 0.0002)], dim=1)
        rot = U(-MAX_ERROR[2], MAX_ERROR[2]) if self.allow_rotation else torch.zeros(B, device=self.device, dtype=self.dtype)
        new = apply_relative_motion(self.q_init_reference.repeat(B, 1), pos, rot, U(-0.01, 0.005))
        self.current_q_init = torch.where(m[:, None], new, self.current_q_init)
        self.prev_pose = torch.where(m[:, None], torch.stack([new[:, 0], new[:, 1], new[:, 3]], dim=1), self.prev_pose)
        self.steps = torch.where(m, torch.zeros_like(self.steps), self.steps)
        if self.domain_randomization:
            self._randomize(m)

    def reset(self, mask=None):
        """New episodes for the environments in mask (all when None), then one attempt from there (the first observation, as the
        reference's reset() does).  A collector that resets single environments should pass them to step(u, reset=done) instead: their
        first attempt then shares the launch with the others' next attempt."""
        m = torch.ones(self.B, dtype=torch.bool, device=self.device) if mask is None else torch.as_tensor(mask, device=self.device).bool()
        self._new_episodes(m)
        obs, _, _, _ = self._execute()
        return obs

    def step(self, u, reset=None):
        """u [B, 3] (or [B, 2] without rotation) -> obs [B, 2600], reward [B], done [B], info.  Environments in `reset` (bool [B], e.g. the
        previous step's done) ignore their action and start a new episode instead: what they return is their first observation (reward
        and done of that attempt carry no meaning, as for the reference's reset())."""
        u = u.to(self.device, self.dtype)
        if not self.allow_rotation:
            u = torch.cat([u, torch.zeros(self.B, 1, device=self.device, dtype=self.dtype)], dim=1)
        dxy, drot = relative_motion_of_action(u, self.current_q_init)
        if not self.allow_rotation:
            drot = torch.zeros_like(drot)
        moved = apply_relative_motion(self.current_q_init, dxy, drot)
        if reset is None:
            self.current_q_init = moved
            self.steps += 1
        else:
            m = torch.as_tensor(reset, device=self.device).bool()
            self.current_q_init = torch.where(m[:, None], self.current_q_init, moved)
            self.steps += (~m).long()
            self._new_episodes(m)
        obs, r, success, status = self._execute()
        done = success | (self.steps >= self.max_episode_steps)
        if reset is not None:
            done = done & ~m
        return obs, r, done, {"success": success, "status": status}
