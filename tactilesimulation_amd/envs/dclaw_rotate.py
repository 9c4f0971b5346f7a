"""BatchedDClawRotateEnv — B TactileRotation-v1 environments (the reference's envs/dclaw_rotate_env.py, position control with relative
actions, observation_type "tactile") stepped as one batch on one GPU: the forward-only roll-out collection that BASELINE configs[3]
describes (the reference trains this task with PPO over SubprocVecEnv, one simulator per process).

The environment-side arithmetic is the reference's, pinned on golden vectors recorded from its own class
(tools/make_dclaw_env_fixture.py -> tests/golden/dclaw_env.npz, tests/test_dclaw_env_golden.py):
  * action -> joint targets: clip(q[:9] + clip(u, -1, 1) * 0.06, dof limits)                                   (:199-207)
  * observation: [q[:9], three fingertip positions (variables[:9]), flow images (3, 20, 20, 3) as (9, 20, 20)]    (:93-118)
  * reward: -0.5 per finger whose summed |flow| is below 1, -(min(cap angle - pi/4, 0))^2, -0.005 |u|^2, -50 and done when a fingertip
    rises above the cap's top surface, +50, done and success when the cap has turned pi/4                          (:122-162)
    — with the contact term computed from the flow images of the PREVIOUS observation, as the reference does (its step() calls
    _get_reward before _get_obs refreshes tactile_force_buf, :211-222).
The reset-time randomisers (cap joint damping, cap radius, end-effector offset, cap location: :165-178) edit constants of the model; for a
batch they become per-environment parameter tables (include/tsim.h tsim_set_env_tables): `randomize=True` draws every environment its OWN
damping ~ U(0.01, 0.7), radius ~ U(0.02, 0.08), (dx, dy) ~ U(+-0.02)^2 at each (masked) reset, on the device, and writes the float records
those four edits touch — what the reference does per environment per reset; `variants=K` (rounds 3-4) assigns environments to a pool of K
host-compiled variants instead.
Forward-only: PPO needs no simulator gradients.
"""
import math

import numpy as np
import torch

from ..host.batch import BatchSim
from ..model import compiler as mc
from ..workloads import asset

DOF_LIMIT = np.array([[-0.45, 1.35], [-2, 2], [1, 2]] * 3, dtype=np.float64)          # dclaw_rotate_env.py:78-88
_LIMITS = {}


def joint_targets(q, u, relative_q_scale=0.06, limits=DOF_LIMIT):
    """u [B, 9] policy output -> absolute joint targets [B, 9] (relative position control)."""
    key = (id(limits), q.dtype, q.device)
    lim = _LIMITS.get(key)
    if lim is None:                                                 # once per device / dtype: no host copy in the stepping loop
        lim = _LIMITS[key] = torch.as_tensor(np.asarray(limits), dtype=q.dtype, device=q.device)
    return torch.minimum(torch.maximum(q[:, :9] + torch.clamp(u, -1.0, 1.0) * relative_q_scale, lim[:, 0]), lim[:, 1])


def observation(q, var, flow):
    """q [B, 10], var [B, 12], flow [B, 3, 20, 20, 3] -> [B, 9 + 9 + 3 * 3 * 20 * 20]."""
    B = q.shape[0]
    return torch.cat([q[:, :9], var[:, :9], flow.permute(0, 1, 4, 2, 3).reshape(B, -1)], dim=1)


def reward(q, var, flow_prev, u, rot_coef=1.0, power_coef=0.005, cap_top_surface_z=0.05):
    """-> reward [B], done [B] bool, success [B] bool.  flow_prev: the flow images of the previous observation."""
    force = flow_prev.norm(dim=-1).sum(dim=(-1, -2))                                    # [B, 3] per finger
    cap = q[:, -1]
    r = -0.5 * (force < 1.0).sum(1).to(q.dtype) - rot_coef * torch.clamp(cap - math.pi / 4, max=0.0) ** 2 - power_coef * (u ** 2).sum(1)
    high = (var[:, 2:9:3] > cap_top_surface_z).any(1)
    success = cap > math.pi / 4
    r = r - 50.0 * high.to(q.dtype) + 50.0 * success.to(q.dtype)
    return r, high | success, success


class BatchedDClawRotateEnv:
    frame_skip = 5                                                                        # :59
    max_episode_steps = 200                                                               # envs/__init__.py

    def __init__(self, batch_size, model=None, device="cuda:0", dtype=torch.float32, seed=0, variants=0, randomize=False):
        self.model = mc.load_model(asset("dclaw_position_control")) if model is None else model
        self.B, self.device, self.dtype = int(batch_size), torch.device(device), dtype
        self.sim = BatchSim(self.model, self.B, device=device, dtype=dtype, tape_capacity=0)
        assert (self.sim.ndof_r, self.sim.ndof_u, self.sim.ndof_var, self.sim.ndof_tactile) == (10, 9, 12, 2718)
        self.rng = np.random.default_rng(seed)
        self.obs_dim, self.act_dim = 9 + 9 + 3 * 3 * 20 * 20, 9
        # taxel -> image cell: flow image [3, 20, 20, 3] from the taxel-major tactile vector (get_tactile_flow_images, :103-105)
        # (several of the 302 abstract taxels of a fingertip share an image cell: 182 cells are covered; like the shim's
        # get_tactile_flow_images, the last taxel of a cell — in taxel order — is the one shown.  As a gather, that is deterministic.)
        src = np.full(3 * 400, -1, dtype=np.int64)
        for s, (name, (t0, nt, rows, cols)) in enumerate(zip(self.model.meta["sensor_names"], self.model.meta["sensor_taxels"])):
            assert (rows, cols) == (20, 20)
            for k, (r, c) in enumerate(self.model.meta["image_pos"][name]):
                src[s * 400 + r * 20 + c] = t0 + k
        self._src = torch.tensor(np.maximum(src, 0), device=self.device, dtype=torch.long)
        self._covered = torch.tensor(src >= 0, device=self.device)
        q_init = np.zeros(10)
        q_init[[1, 4, 7]], q_init[[2, 5, 8]] = -0.5, 0.8                                   # :74-77
        self.q_init = q_init
        self._q_init = torch.tensor(q_init, device=self.device, dtype=self.dtype)[None]
        self._gen = torch.Generator(device=self.device); self._gen.manual_seed(seed)
        self.tables = None
        self.randomize = bool(randomize)
        if variants and randomize:
            raise ValueError("variants=K (a pool of compiled models) and randomize=True (continuous draws per environment) exclude each other")
        if variants:
            self._build_variants(int(variants))
        if randomize:
            self._build_randomisers()
        # persistent state buffers, updated in place: a collector's step can be captured in a HIP graph and replayed
        z = lambda *shape: torch.zeros(*shape, device=self.device, dtype=self.dtype)
        self.q, self.var, self.flow = z(self.B, 10), z(self.B, 12), z(self.B, 3, 20, 20, 3)
        self.steps = torch.zeros(self.B, device=self.device, dtype=torch.long)

    # ------------------------------------------------------------------ domain randomisation (reset-time randomisers)
    def _build_variants(self, K):
        """K compiled variants of the model with the reference's four reset-time draws (:167-178); environments are assigned to variants at
        reset.  Only the model's float records differ between variants, which is what per-environment tables hold."""
        n, rows, self.variant_params = self.sim.base_tables().shape[1], [], []
        for _ in range(K):
            damping, radius = self.rng.uniform(0.01, 0.7), self.rng.uniform(0.02, 0.08)
            dx, dy = self.rng.uniform(-0.02, 0.02, size=2)
            spec = mc.compile_spec(self.model.spec).spec
            mc.edit_spec(spec, "joint_damping", "cap", damping)
            mc.edit_spec(spec, "body_size", "cap", np.array([0.03, radius]))
            mc.edit_spec(spec, "endeffector_position", "cap", np.array([radius, 0.0, 0.0]))
            mc.edit_spec(spec, "joint_location", "cap", np.array([dx, dy, 0.075]))
            m = mc.compile_spec(spec)
            assert np.array_equal(m.I, self.model.I) and np.array_equal(m.F[n:], self.model.F[n:]), "a randomiser changed more than the float records"
            rows.append(m.F[:n]); self.variant_params.append((damping, radius, dx, dy))
        self._variant_rows = torch.tensor(np.array(rows), device=self.device, dtype=self.dtype)
        self.variant_of = torch.zeros(self.B, device=self.device, dtype=torch.long)
        self.tables = self._variant_rows[self.variant_of].contiguous()          # persistent: rewritten in place at every reset

    RANDOMISER_RANGES = {"damping": (0.01, 0.7), "radius": (0.02, 0.08), "dx": (-0.02, 0.02), "dy": (-0.02, 0.02)}        # :169-178

    @staticmethod
    def edited_model(model, damping, radius, dx, dy):
        """The model with the reference's four reset-time edits applied on the host (update_joint_damping, update_body_size,
        update_endeffector_position, update_joint_location: envs/dclaw_rotate_env.py:169-178), compiled: what one environment's table must equal."""
        spec = mc.compile_spec(model.spec).spec
        mc.edit_spec(spec, "joint_damping", "cap", float(damping))
        mc.edit_spec(spec, "body_size", "cap", np.array([0.03, float(radius)]))
        mc.edit_spec(spec, "endeffector_position", "cap", np.array([float(radius), 0.0, 0.0]))
        mc.edit_spec(spec, "joint_location", "cap", np.array([float(dx), float(dy), 0.075]))
        return mc.compile_spec(spec)

    def _build_randomisers(self):
        """Which float records the four randomisers touch, and how: found by compiling edited models on the host ONCE (no assumption about the
        compiler's formulas: the cap's mass ~ r^2 and inertia ~ r^4, r^2 come out of it as polynomials in the radius of degree <= 4, the
        primitive radius of the three fingertip-cap pairs and the end-effector offset as r itself, damping / dx / dy as themselves), then
        written per environment on the device at every reset.  Checked here against three more host compilations to 1e-13."""
        n = self.sim.base_tables().shape[1]
        lo_r, hi_r = self.RANDOMISER_RANGES["radius"]
        mid = {"damping": 0.3, "radius": 0.05, "dx": 0.0, "dy": 0.0}
        row = lambda **kw: self._row(n, **dict(mid, **kw))
        base = row()
        self._rand_single = {}                                     # parameter -> indices of the records that ARE the parameter
        for k, v in (("damping", 0.55), ("dx", 0.013), ("dy", -0.017)):
            r_ = row(**{k: v})
            idx = np.nonzero(r_ != base)[0]
            assert len(idx) >= 1 and np.all(r_[idx] == v), (k, idx, r_[idx])
            self._rand_single[k] = torch.tensor(idx, device=self.device, dtype=torch.long)
        nodes = 0.5 * (lo_r + hi_r) + 0.5 * (hi_r - lo_r) * np.cos(np.pi * (np.arange(7) + 0.5) / 7)            # Chebyshev nodes: a well-conditioned fit
        rows = np.array([row(radius=r) for r in nodes])
        idx = np.nonzero((rows != rows[0]).any(0))[0]
        t = (nodes - 0.05) / 0.03                                   # the polynomial's variable: radius mapped to [-1, 1]
        coef = np.linalg.solve(np.vander(t, 7, increasing=True), rows[:, idx])[:5]              # degree <= 4 (the higher coefficients must vanish)
        for r in (0.0231, 0.0507, 0.0789):
            got = (((r - 0.05) / 0.03) ** np.arange(5)) @ coef
            want = row(radius=r)[idx]
            assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max() + 1e-300, (r, np.abs(got - want).max())
        self._rand_ridx = torch.tensor(idx, device=self.device, dtype=torch.long)
        self._rand_rcoef = torch.tensor(coef, device=self.device, dtype=torch.float64)          # [5, len(idx)]
        self.params = torch.zeros(self.B, 4, device=self.device, dtype=torch.float64)           # damping, radius, dx, dy of every environment
        self._rand_lo = torch.tensor([self.RANDOMISER_RANGES[k][0] for k in ("damping", "radius", "dx", "dy")], device=self.device, dtype=torch.float64)
        self._rand_hi = torch.tensor([self.RANDOMISER_RANGES[k][1] for k in ("damping", "radius", "dx", "dy")], device=self.device, dtype=torch.float64)
        self.tables = self.sim.base_tables().contiguous()                                       # persistent: rewritten in place at every reset

    def _row(self, n, damping, radius, dx, dy):
        m = self.edited_model(self.model, damping, radius, dx, dy)
        assert np.array_equal(m.I, self.model.I) and np.array_equal(m.F[n:], self.model.F[n:]), "a randomiser changed more than the float records"
        return m.F[:n].copy()

    def _randomise(self, mask):
        """New draws for the environments with mask set; every environment's table rewritten from `params` (device only, no synchronisation)."""
        lo, hi = self._rand_lo, self._rand_hi                      # (made once: nothing here touches the host, a collector's reset is captured in a HIP graph)
        draw = lo + (hi - lo) * torch.rand(self.B, 4, device=self.device, dtype=torch.float64, generator=self._gen)
        self.params.copy_(torch.where(mask[:, None], draw, self.params))
        t = (self.params[:, 1:2] - 0.05) / 0.03
        pw = torch.cat([torch.ones_like(t), t, t * t, t * t * t, t * t * t * t], dim=1)          # [B, 5]
        self.tables[:, self._rand_ridx] = (pw @ self._rand_rcoef).to(self.dtype)
        for j, k in ((0, "damping"), (2, "dx"), (3, "dy")):
            self.tables[:, self._rand_single[k]] = self.params[:, j:j + 1].to(self.dtype)
        self.sim.set_env_tables(self.tables)

    # ------------------------------------------------------------------ read-out helpers
    def flow_images(self, tactile):
        t = tactile.reshape(tactile.shape[0], -1, 3)
        img = t[:, self._src] * self._covered[None, :, None].to(t.dtype)
        return img.reshape(-1, 3, 20, 20, 3)

    def _observe(self):
        q, _ = self.sim.get_state()
        var, tac = self.sim.readout()
        self.q.copy_(q); self.var.copy_(var); self.flow.copy_(self.flow_images(tac))
        return observation(self.q, self.var, self.flow)

    # ------------------------------------------------------------------ gym-like API, batched
    def reset(self, mask=None):
        """New episodes for the environments with mask set (all when None): q_init + 0.05 N(0, 1) on the nine hand joints (:164-166), and a
        new variant of the model when a pool was built.  Everything stays on the device (no synchronisation): a collector can call
        reset(done) after every step."""
        B = self.B
        m = torch.ones(B, dtype=torch.bool, device=self.device) if mask is None else torch.as_tensor(mask, device=self.device).bool()
        q0 = self._q_init.repeat(B, 1)
        q0[:, :9] += 0.05 * torch.randn(B, 9, device=self.device, dtype=self.dtype, generator=self._gen)
        if self.randomize:
            self._randomise(m)
        elif self.tables is not None:
            new = torch.randint(0, self._variant_rows.shape[0], (B,), device=self.device, generator=self._gen)
            self.variant_of.copy_(torch.where(m, new, self.variant_of))
            self.tables.copy_(self._variant_rows[self.variant_of])
            self.sim.set_env_tables(self.tables)
        if mask is None:
            self.sim.reset(q0, None, backward_flag=False)
        else:
            self.sim.reset_masked(q0, m.to(torch.int32))
        self.steps.masked_fill_(m, 0)
        return self._observe()

    def step(self, u):
        """u [B, 9] -> obs [B, 3618], reward [B], done [B], info.  Finished environments are NOT reset here (call reset(done))."""
        u = u.to(self.device, self.dtype)
        flow_prev = self.flow.clone()
        out = self.sim.step(joint_targets(self.q, u), self.frame_skip)
        self.q.copy_(out["q"]); self.var.copy_(out["var"]); self.flow.copy_(self.flow_images(out["tactile"]))
        obs = observation(self.q, self.var, self.flow)
        r, done, success = reward(self.q, self.var, flow_prev, u)
        self.steps += 1
        done = done | (self.steps >= self.max_episode_steps)
        return obs, r, done, {"success": success, "status": out["status"]}


class GraphedCollector:
    """One collection step — policy, env.step, reset of the environments that finished — captured in ONE HIP graph and replayed.  The
    eager loop issues ~80 small launches per env-step from Python and is host-bound at a few hundred thousand env-steps/s; a replay is one
    launch (tools/dclaw_graph_collector_probe.py: 0.5 -> 1.8 M env-steps/s at B = 2048, bit-identical transitions).  `policy(obs) -> u`
    must be capturable torch code (no host round trips; torch.randn with the default generator is fine).  After step(): `obs` holds the
    observation the policy acted on, `action`, `reward`, `done`, `success` the transition, `next_obs` the observation after the
    per-environment resets (it is also the next step's input)."""

    def __init__(self, env, policy, warmup=2):
        self.env, self.policy = env, policy
        # Host-side batch state travels as kernel ARGUMENTS and is frozen at its capture-time value in a replayed graph.  For BDF1
        # models none of it changes a launch; a BDF2 model's has_prev flag (history of the previous sub-step) would: every replayed
        # step would silently drop the BDF2 history.  Refuse instead.
        from ..model import blob as _blob
        if int(env.sim.model.I[_blob.TSIM_IH_INTEGRATOR]) != 1:
            raise RuntimeError("GraphedCollector: BDF2 models cannot be captured (the integrator's history flag is host state baked into the "
                               "captured launch); use the eager loop")
        env._gen = None                                            # the default generator is the one graph capture knows how to advance
        self.next_obs = env.reset().clone()
        self.obs = torch.empty_like(self.next_obs)
        side = torch.cuda.Stream(env.device)
        side.wait_stream(torch.cuda.current_stream(env.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(env.device).wait_stream(side)
        self.next_obs.copy_(env.reset())
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self._body()
        self.next_obs.copy_(env.reset())                           # capture does not execute: start from fresh episodes

    def _body(self):
        self.obs.copy_(self.next_obs)
        self.action = self.policy(self.obs)
        _, self.reward, self.done, info = self.env.step(self.action)
        self.success, self.status = info["success"], info["status"]
        self.next_obs.copy_(self.env.reset(self.done))

    def step(self):
        self.graph.replay()
        return self.obs, self.action, self.reward, self.done, self.next_obs
