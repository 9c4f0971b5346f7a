"""BatchedStableGraspEnv — B StableGrasp-v1 environments (the reference's envs/stable_grasp_env.py, observation_type "tactile_flatten") as
one batch on one GPU.  One env-step is one five-stage GRASP of 180 sub-steps (move to the grasp position, close, lift and capture one
tactile frame, put down, open: :197-246), run as ONE launch with a tactile mask; the policy shifts the grasp position along the bar until the
bar stays level when lifted.  Forward-only.

The environment-side arithmetic is the reference's, pinned on golden vectors recorded from its own class against a scripted simulator
(tools/make_stable_grasp_env_fixture.py -> tests/golden/stable_grasp_env.npz, tests/test_stable_grasp_env_golden.py):
  * action -> grasp position: clip(position + clip(u, -1, 1) * 0.05, +-0.11)                                   (:146-152)
  * the 180 x 6 joint-target table of a grasp from the state the previous grasp ended in (:197-229), tactile frame at sub-step 60
  * observation: shear components of that frame, normalised to a largest vector of 30, flattened (1, 2, 13, 10, 2)  (:246-262)
  * success: the bar's rotation vector at the captured frame shorter than 0.02 and the bar lifted (q[8] > 0.005): reward 100 and done;
    otherwise reward -10 |rotation vector|                                                                       (:264-283)
The reset-time randomisation of the eleven block densities (:68-129) changes the composite inertia of the bar: per-environment tables, either
drawn from a pool of compiled variants (`variants=K`, rounds 3-4) or — `randomize=True`, round 5: what the reference does, a fresh draw per
environment per reset — computed on the device from per-environment density draws: the bar's mass, first moment and second moment about the
link origin are LINEAR in the eleven densities (coefficients fitted once from host compilations, checked to 1e-13), its centre of mass and
inertia about it follow from them.
"""
import numpy as np
import torch

from ..host.batch import BatchSim
from ..model import compiler as mc
from ..workloads import asset

GRASP_STEPS = (20, 10, 50, 20, 50, 10, 20)                            # :224
CAPTURE_FRAME = 60                                                    # :236
LIFT_HEIGHT, GRASP_HEIGHT, FINGER = 0.2029862 + 0.03, 0.2029862, -0.008   # :198-200
BOX_IDS = (9, 8, 1, 2, 3, 4, 5, 6, 7, 10, 11)                         # :75


def grasp_position_of_action(pos, u, action_scale=0.05, bound=0.11):
    return torch.clamp(pos + torch.clamp(u, -1.0, 1.0) * action_scale, -bound, bound)


def grasp_actions(q, grasp_position):
    """q [B, 12] (the state the previous grasp ended in), grasp_position [B] -> (start state [B, 12], joint targets [180, B, 6])."""
    q = q.clone()
    q[:, 1] = grasp_position
    B, z = q.shape[0], torch.zeros_like(grasp_position)
    f = torch.full_like(grasp_position, FINGER)
    row = lambda h, a, b: torch.stack([z, grasp_position, torch.full_like(z, h), z, a, b], dim=1)
    t = [q[:, :6], row(GRASP_HEIGHT, f, f), row(GRASP_HEIGHT, f, f), row(LIFT_HEIGHT, f, f), row(LIFT_HEIGHT, f, f), row(GRASP_HEIGHT, f, f), row(GRASP_HEIGHT, f, f),
         row(GRASP_HEIGHT, q[:, 4], q[:, 5])]
    rows = []
    for s, n in enumerate(GRASP_STEPS):
        frac = torch.arange(1, n + 1, dtype=q.dtype, device=q.device)[:, None, None] / n
        rows.append((t[s + 1] - t[s])[None] * frac + t[s][None])
    return q, torch.cat(rows, dim=0)


def observation(tactile):
    """tactile [B, 780] (the captured frame) -> [B, 520]."""
    B = tactile.shape[0]
    sh = tactile.reshape(B, 1, 2, 13, 10, 3)[..., 0:2]
    mx = sh.norm(dim=-1).reshape(B, -1).max(dim=1).values + 1e-5
    return (sh / (mx / 30.0)[:, None, None, None, None, None]).reshape(B, -1)


def reward_done(q_capture):
    """q at the captured frame [B, 12] -> reward [B], success [B] (= done)."""
    angle = q_capture[:, 9:12].norm(dim=1)
    success = (angle < 0.02) & (q_capture[:, 8] > 0.005)
    return torch.where(success, torch.full_like(angle, 100.0), -angle * 10.0), success


def draw_block_densities(rng):
    """The reference's density draw (:68-115) with a numpy RandomState-like rng: 11 densities whose centre of mass is uniform along the bar."""
    density_range, num_blocks = [600.0, 700.0], 11
    com_y = rng.uniform(1, num_blocks - 1, 1)
    nl = int(com_y[0]); nr = num_blocks - 1 - nl
    mid_left = com_y - nl
    mid = rng.uniform(density_range[0], density_range[1], 1)[0]
    if mid_left < 0.5:
        right = rng.uniform(density_range[0] * nr, density_range[1] * nr, 1)[0]
        left = right + (1 - mid_left * 2) * mid
    else:
        left = rng.uniform(density_range[0] * nl, density_range[1] * nl, 1)[0]
        right = left + (mid_left * 2 - 1) * mid
    lr = rng.random(nl) + 0.1; lr /= lr.sum()
    rr = rng.random(nr) + 0.1; rr /= rr.sum()
    d = (np.atleast_1d(left) * lr).reshape(-1).tolist()
    if mid_left > 0:
        d.append(mid)
    d.extend((right * rr).tolist())
    d = np.array(d, dtype=np.float64).reshape(-1)
    return d / d.sum() * np.clip(d.sum(), 3000, 7000)


class BatchedStableGraspEnv:
    max_episode_steps = 10                                            # envs/__init__.py

    def __init__(self, batch_size, model=None, device="cuda:0", dtype=torch.float32, seed=0, variants=8, randomize=False):
        self.model = mc.load_model(asset("stable_grasp")) if model is None else model
        self.B, self.device, self.dtype = int(batch_size), torch.device(device), dtype
        self.sim = BatchSim(self.model, self.B, device=device, dtype=dtype, tape_capacity=0)
        assert (self.sim.ndof_r, self.sim.ndof_u, self.sim.ndof_var, self.sim.ndof_tactile) == (12, 6, 0, 780)
        self.rng = np.random.RandomState(seed)
        self.gen = torch.Generator(device=self.device); self.gen.manual_seed(seed)
        self.obs_dim, self.act_dim = 520, 1
        self.mask = torch.zeros(sum(GRASP_STEPS), dtype=torch.bool)
        self.mask[CAPTURE_FRAME] = True
        self.q_reference = self._generate_initial_state()
        self.randomize = bool(randomize)
        if self.randomize:
            self._build_randomiser()
        else:
            self._build_variants(int(variants))
        self.current_q = self.q_reference.repeat(self.B, 1)
        self.grasp_position = torch.zeros(self.B, device=self.device, dtype=dtype)
        self.steps = torch.zeros(self.B, device=self.device, dtype=torch.long)

    def _generate_initial_state(self):
        """The settled open gripper above the bar (:165-186): 500 sub-steps holding the initial targets."""
        one = BatchSim(self.model, 1, device=str(self.device), dtype=torch.float64, tape_capacity=0)
        q = np.zeros(12); q[2], q[4], q[5] = 0.2, -0.03, -0.03
        one.reset(torch.tensor(q[None]), None, backward_flag=False)
        u = q[:6].copy(); u[2] += 0.003
        one.step(torch.tensor(u[None]), 500, want_tactile=False)
        return one.get_state()[0].to(self.dtype)

    def _build_variants(self, K):
        n, rows, self.variant_densities = self.sim.base_tables().shape[1], [], []
        for _ in range(K):
            d = draw_block_densities(self.rng)
            spec = mc.compile_spec(self.model.spec).spec
            for i, b in enumerate(BOX_IDS):
                mc.edit_spec(spec, "body_density", "box_%d" % b, float(d[i]))
            m = mc.compile_spec(spec)
            assert np.array_equal(m.I, self.model.I) and np.array_equal(m.F[n:], self.model.F[n:])
            rows.append(m.F[:n]); self.variant_densities.append(d)
        self._variant_rows = torch.tensor(np.array(rows), device=self.device, dtype=self.dtype)
        self.variant_of = torch.zeros(self.B, device=self.device, dtype=torch.long)

    # ------------------------------------------------------------------ continuous density draws on the device (randomize=True)
    @staticmethod
    def edited_model(model, densities):
        """The model with the eleven block densities set on the host (update_body_density, envs/stable_grasp_env.py:117-129), compiled."""
        spec = mc.compile_spec(model.spec).spec
        for i, b in enumerate(BOX_IDS):
            mc.edit_spec(spec, "body_density", "box_%d" % b, float(densities[i]))
        return mc.compile_spec(spec)

    def _build_randomiser(self):
        """The bar is ONE link made of eleven blocks: its mass M, first moment S = M c and second moment about the link origin
        J = I_c + M (|c|^2 1 - c c^T) are linear in the densities.  The 10 x 11 coefficients are fitted from host compilations of drawn density
        vectors (no assumption about the blocks' geometry), checked on held-out draws to 1e-13; the ten float records of that link — mass, centre of
        mass, inertia about it — are then written per environment on the device."""
        from ..model import blob as Bl
        n = self.sim.base_tables().shape[1]
        D = np.array([draw_block_densities(self.rng) for _ in range(16)])
        rows = []
        for d in D:
            m = self.edited_model(self.model, d)
            assert np.array_equal(m.I, self.model.I) and np.array_equal(m.F[n:], self.model.F[n:])
            rows.append(m.F[:n])
        R = np.array(rows)
        changed = np.nonzero((R != R[0]).any(0))[0]
        fl = int(self.model.I[Bl.TSIM_IH_FOFF_LINK])
        link = (int(changed[0]) - fl) // Bl.TSIM_LF_SIZE                        # the bar's link (0-based record)
        base = fl + link * Bl.TSIM_LF_SIZE
        rec = np.concatenate([[base + Bl.TSIM_LF_MASS], base + Bl.TSIM_LF_COM + np.arange(3), base + Bl.TSIM_LF_INERTIA + np.arange(6)])
        assert set(changed) <= set(rec), "the densities change records outside the bar's mass / centre of mass / inertia"
        Y = np.array([self._moments(r[rec]) for r in R])
        A = np.concatenate([D, np.ones((len(D), 1))], axis=1)
        coef = np.linalg.lstsq(A[:13], Y[:13], rcond=None)[0]
        assert np.abs(A[13:] @ coef - Y[13:]).max() <= 1e-13 * np.abs(Y).max()
        self._rec = torch.tensor(rec, device=self.device, dtype=torch.long)
        self._coef = torch.tensor(coef, device=self.device, dtype=torch.float64)          # [12, 10]: rows = eleven densities + intercept
        self.densities = torch.zeros(self.B, 11, device=self.device, dtype=torch.float64)
        self.tables = self.sim.base_tables().contiguous()

    @staticmethod
    def _moments(r):
        """(mass, com[3], inertia about com xx yy zz xy xz yz) -> (M, S[3], J[6] about the origin)."""
        M, c, (xx, yy, zz, xy, xz, yz) = r[0], r[1:4], r[4:10]
        cc = float(c @ c)
        return np.array([M, M * c[0], M * c[1], M * c[2], xx + M * (cc - c[0] ** 2), yy + M * (cc - c[1] ** 2), zz + M * (cc - c[2] ** 2),
                         xy - M * c[0] * c[1], xz - M * c[0] * c[2], yz - M * c[1] * c[2]])

    def _draw_densities(self):
        """draw_block_densities (the reference's :68-115) for every environment at once, on the device: [B, 11] float64."""
        B, dev, f64 = self.B, self.device, torch.float64
        u = lambda *shape: torch.rand(*shape, device=dev, dtype=f64, generator=self.gen)
        com = 1.0 + 9.0 * u(B)
        nl = torch.floor(com).clamp(max=9.0)
        nr = 10.0 - nl
        mid_left = com - nl
        mid = 600.0 + 100.0 * u(B)
        side = 600.0 + 100.0 * u(B)
        small = mid_left < 0.5
        right = torch.where(small, side * nr, side * nl + (mid_left * 2 - 1) * mid)
        left = torch.where(small, side * nr + (1 - mid_left * 2) * mid, side * nl)
        i = torch.arange(11, device=dev, dtype=f64)[None]
        wl = (u(B, 11) + 0.1) * (i < nl[:, None])
        wr = (u(B, 11) + 0.1) * (i > nl[:, None])
        d = left[:, None] * wl / wl.sum(1, keepdim=True).clamp_min(1e-300) + right[:, None] * wr / wr.sum(1, keepdim=True).clamp_min(1e-300) + mid[:, None] * (i == nl[:, None])
        tot = d.sum(1, keepdim=True)
        return d / tot * tot.clamp(3000.0, 7000.0)

    def _randomise(self, mask):
        self.densities.copy_(torch.where(mask[:, None], self._draw_densities(), self.densities))
        Y = torch.cat([self.densities, torch.ones(self.B, 1, device=self.device, dtype=torch.float64)], dim=1) @ self._coef      # M, S, J
        M, c = Y[:, 0:1], Y[:, 1:4] / Y[:, 0:1]
        cc = (c * c).sum(1, keepdim=True)
        diag = Y[:, 4:7] - M * (cc - c * c)
        off = Y[:, 7:10] + M * torch.stack([c[:, 0] * c[:, 1], c[:, 0] * c[:, 2], c[:, 1] * c[:, 2]], dim=1)
        self.tables[:, self._rec] = torch.cat([M, c, diag, off], dim=1).to(self.dtype)
        self.sim.set_env_tables(self.tables)

    def _grasp(self):
        start, act = grasp_actions(self.current_q, self.grasp_position)
        self.sim.reset(start, None, backward_flag=False)
        ro = self.sim.rollout(act, 1, want_var=False, tactile_mask=self.mask)
        self.current_q = ro["q"][-1].clone()
        r, success = reward_done(ro["q"][CAPTURE_FRAME])
        self.last_obs = observation(ro["tactile"][0])
        return self.last_obs, r, success, ro["status"]

    def reset(self, mask=None):
        """New episodes for the environments in mask (all when None): a new bar (density variant), grasp position 0, the settled open
        gripper; then one grasp (the first observation)."""
        m = torch.ones(self.B, dtype=torch.bool, device=self.device) if mask is None else torch.as_tensor(mask, device=self.device).bool()
        if self.randomize:
            self._randomise(m)
        else:
            new = torch.randint(0, self._variant_rows.shape[0], (self.B,), device=self.device, generator=self.gen)
            self.variant_of = torch.where(m, new, self.variant_of)
            self.sim.set_env_tables(self._variant_rows[self.variant_of].contiguous())
        self.current_q = torch.where(m[:, None], self.q_reference.repeat(self.B, 1), self.current_q)
        self.grasp_position = torch.where(m, torch.zeros_like(self.grasp_position), self.grasp_position)
        self.steps = torch.where(m, torch.zeros_like(self.steps), self.steps)
        if mask is None or getattr(self, "last_obs", None) is None:
            obs, _, _, _ = self._grasp()
            return obs
        # Only the masked environments take the reset-time grasp (the reference resets one environment at a time,
        # envs/stable_grasp_env.py:136-163): the others keep the state and the observation their last step() left — the launch runs the
        # whole batch, its results are kept for the masked rows only.
        keep_q, keep_obs = self.current_q, self.last_obs
        obs, _, _, _ = self._grasp()
        self.current_q = torch.where(m[:, None], self.current_q, keep_q)
        self.last_obs = torch.where(m[:, None], obs, keep_obs)
        return self.last_obs

    def step(self, u):
        """u [B, 1] -> obs [B, 520], reward [B], done [B], info."""
        u = u.to(self.device, self.dtype).reshape(self.B)
        self.grasp_position = grasp_position_of_action(self.grasp_position, u)
        obs, r, success, status = self._grasp()
        self.steps += 1
        return obs, r, success | (self.steps >= self.max_episode_steps), {"success": success, "status": status}
