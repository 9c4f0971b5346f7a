"""`redmax_py` drop-in: the python surface of the reference's pybind11 module, backed by the HIP path.

Put this directory on PYTHONPATH and the reference's `envs/*.py`, `envs/redmax_torch_functions.py`,
`algorithms/gd.py` and `examples/RollingBallExp/test_sim_speed.py` run unmodified:  `import redmax_py as redmax`.

Every member used anywhere in the reference is here (list reconstructed in SURVEY.md §8b; call sites cited inline).
One `Simulation` == one environment == a B = 1 `tsim_batch` (include/tsim.h).  Arrays cross this boundary as
float64 numpy, exactly like the pybind11/Eigen binding (the reference computes in float64:
examples/TactilePushExp/train_tactile_push_gd.py:13).  There is no CPU fallback: a GPU and the built
libtsim_hip.so are required.  Viewer members are accepted and ignored (rendering is out of scope).
"""
import numpy as np
import torch

from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model import compiler as _mc


class _Options:
    def __init__(self, h):
        self.h = h


class _ViewerOptions:
    """envs/redmax_torch_env.py:52-70, utils/renderer.py:7-30 — stored, never used."""
    def __init__(self):
        self.fps, self.speed, self.loop, self.infinite = 30, 1.0, False, False
        self.record, self.record_folder = False, ""
        self.camera_pos, self.camera_lookat = np.zeros(3), np.zeros(3)


class _BackwardInfo:
    """envs/redmax_torch_functions.py:83-90,151-165"""
    def __init__(self):
        self.flag_q0 = self.flag_qdot0 = self.flag_p = False
        self.flag_u = True
        self.df_dq = self.df_dvar = self.df_dtactile = None
        self.df_dq0 = self.df_dqdot0 = self.df_du = self.df_dp = None

    def set_flags(self, flag_q0=False, flag_qdot0=False, flag_p=False, flag_u=False):
        if flag_p:
            raise NotImplementedError("design-parameter gradients (flag_p) are not part of this path; the reference "
                                      "always passes flag_p=False (envs/redmax_torch_functions.py:83,151)")
        self.flag_q0, self.flag_qdot0, self.flag_p, self.flag_u = bool(flag_q0), bool(flag_qdot0), False, bool(flag_u)


class _BackwardResults:
    """envs/redmax_torch_functions.py:95-105,170"""
    def __init__(self):
        self.df_dq0 = self.df_dqdot0 = self.df_du = None


class Simulation:
    def __init__(self, model_path, verbose=False, device="cuda:0", dtype=torch.float64, tape_capacity=1024):
        self._model = _mc.load_model(model_path) if isinstance(model_path, str) else model_path
        self._sim = BatchSim(self._model, 1, device=device, dtype=dtype, tape_capacity=tape_capacity)
        self._dev, self._dtype = self._sim.device, dtype
        self.ndof_r, self.ndof_u = self._sim.ndof_r, self._sim.ndof_u
        self.ndof_var, self.ndof_tactile = self._sim.ndof_var, self._sim.ndof_tactile
        self.ndof_p = 0
        self.options = _Options(self._sim.h)
        self.viewer_options = _ViewerOptions()
        self.backward_info = _BackwardInfo()
        self.backward_results = _BackwardResults()
        self._q_init = np.zeros(self.ndof_r)
        self._qdot_init = np.zeros(self.ndof_r)
        self._u = np.zeros(self.ndof_u)
        self._backward_flag = False
        self._dirty_outputs = True
        self._q = self._qdot = self._var = self._tac = None
        # forward() I/O: the action goes up and q, qdot, variables (and tactile, unless it is a high-resolution sensor read on demand) come
        # back through ONE pinned staging buffer each way and one stream synchronisation per call — a B = 1 step is ~70 us of kernel; five
        # pageable copies with a synchronisation each cost more than that (RollingBall test_sim_speed: 4.7 k -> FPS in BASELINE.md §6)
        self._lazy_tac = self.ndof_tactile > 4096
        nr, nv, nt = self.ndof_r, self.ndof_var, (0 if self._lazy_tac else self.ndof_tactile)
        self._dev_out = torch.empty(1, 2 * nr + nv + nt, device=self._dev, dtype=dtype)
        self._host_out = torch.empty(1, 2 * nr + nv + nt, dtype=dtype).pin_memory()
        o = self._dev_out
        self._out = {"q": o[:, :nr], "qd": o[:, nr:2 * nr], "status": torch.empty(1, device=self._dev, dtype=torch.int32)}
        if nv:
            self._out["var"] = o[:, 2 * nr:2 * nr + nv]
        if nt:
            self._out["tactile"] = o[:, 2 * nr + nv:]
        self._host_status = torch.empty(1, dtype=torch.int32).pin_memory()
        self._dev_u = torch.empty(1, self.ndof_u, device=self._dev, dtype=dtype)
        self._host_u = torch.empty(1, self.ndof_u, dtype=dtype).pin_memory()
        self.reset(False)

    # ------------------------------------------------------------------ helpers
    def _t(self, a, n, name):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1)   # pybind11/Eigen copied too; grads from
        if a.size != n:                                                          # sum() arrive as stride-0 views
            raise RuntimeError("%s: expected %d values, got %d" % (name, n, a.size))
        return torch.from_numpy(a.copy()).to(device=self._dev, dtype=self._dtype).reshape(1, n)

    def _np(self, t):
        return t.detach().to(torch.float64).cpu().numpy().reshape(-1)

    def _refresh(self):
        self._sync_model()
        if self._dirty_outputs:
            q, qd = self._sim.get_state()
            var, tac = self._sim.readout()
            self._q, self._qdot = self._np(q), self._np(qd)
            self._var = self._np(var) if var is not None else np.zeros(0)
            self._tac = self._np(tac) if tac is not None else np.zeros(0)
            self._dirty_outputs = False

    # ------------------------------------------------------------------ initial state
    def get_q_init(self):
        return self._q_init.copy()

    def get_qdot_init(self):
        return self._qdot_init.copy()

    def set_q_init(self, q):
        self._q_init = np.array(q, dtype=np.float64).reshape(-1).copy()
        if self._q_init.size != self.ndof_r:
            raise RuntimeError("set_q_init: expected %d values" % self.ndof_r)

    def set_qdot_init(self, qdot):
        self._qdot_init = np.array(qdot, dtype=np.float64).reshape(-1).copy()

    def set_state_init(self, q, qdot):
        self.set_q_init(q)
        self.set_qdot_init(qdot)

    # ------------------------------------------------------------------ stepping
    def reset(self, backward_flag=False, backward_design_params_flag=False):
        self._sync_model()
        self._backward_flag = bool(backward_flag)
        self._sim.reset(self._t(self._q_init, self.ndof_r, "q_init"), self._t(self._qdot_init, self.ndof_r, "qdot_init"),
                        backward_flag=self._backward_flag)
        self._dirty_outputs = True
        # which taped sub-steps had their tactile frame read: the reference's EpisodicSimFunction.backward hands over the gradients of
        # exactly those frames (envs/redmax_torch_functions.py:55-57,87: sum(mask) x ndof_tactile values, "TODO: change c++ for tactile masks")
        self._nsub, self._tac_reads = 0, []

    def set_u(self, u):
        u = np.asarray(u, dtype=np.float64).reshape(-1)
        if u.size != self.ndof_u:
            raise RuntimeError("set_u: expected %d values, got %d" % (self.ndof_u, u.size))
        self._u = u.copy()          # callers mutate their array afterwards (envs/tactile_insertion_env.py:160-163)

    def forward(self, num_steps, verbose=False, test_derivatives=False, save_last_frame_var_only=False):
        self._sync_model()
        if test_derivatives:
            self._check_derivatives(int(num_steps), verbose)
        # high-resolution sensors (RollingBall: 120 000 values) are read out on demand by the read-out kernels instead of after every
        # step (test_sim_speed.py:79 asks for them every 5th step only)
        lazy = self._lazy_tac
        if self._u.size != self.ndof_u:
            raise RuntimeError("u: expected %d values, got %d" % (self.ndof_u, self._u.size))
        self._host_u.copy_(torch.from_numpy(self._u).reshape(1, -1))
        self._dev_u.copy_(self._host_u, non_blocking=True)
        out = self._sim.step(self._dev_u, int(num_steps), want_qd=True, want_tactile=not lazy, out=self._out)
        self._host_out.copy_(self._dev_out, non_blocking=True)
        self._host_status.copy_(out["status"], non_blocking=True)
        torch.cuda.current_stream(self._dev).synchronize()
        st = int(self._host_status[0])
        if st & (1 << 30):
            raise RuntimeError("simulation produced non-finite values")
        h = self._host_out.numpy()[0].astype(np.float64)            # a fresh float64 array (callers keep what the getters return)
        nr, nv = self.ndof_r, self.ndof_var
        self._q, self._qdot = h[:nr], h[nr:2 * nr]
        self._var = h[2 * nr:2 * nr + nv] if nv else np.zeros(0)
        self._tac = None if lazy else (h[2 * nr + nv:] if self.ndof_tactile else np.zeros(0))
        self._dirty_outputs = False
        self.last_nonconverged_substeps = st
        self._nsub = getattr(self, "_nsub", 0) + int(num_steps)

    def _check_derivatives(self, num_steps, verbose):
        """forward(..., test_derivatives=True) (envs/redmax_torch_functions.py:49,132 pass the flag through): DiffRedMax checks its analytic
        derivatives against finite differences inside forward() and prints the errors.  Here: the adjoint of the `num_steps` sub-steps
        about to be taken — dL/du, dL/dq0, dL/dqdot0 of a fixed random linear functional of the final q, variables and tactile frame —
        against central differences of the same kernels, on scratch batches (the simulation's own state and tape are not touched).
        The report is printed and kept in `last_derivative_check` (relative errors; a contact / friction kink inside the step shows as a
        large one, as it does in the reference)."""
        sim = self._sim
        nr, nu, nv, nt = self.ndof_r, self.ndof_u, self.ndof_var, self.ndof_tactile
        q, qd = sim.get_state()
        u = torch.from_numpy(self._u.copy()).to(self._dev, self._dtype).reshape(1, nu)
        g = torch.Generator().manual_seed(0)
        w = lambda n: (torch.rand(1, n, generator=g, dtype=torch.float64) - 0.5).to(self._dev, self._dtype) if n else None
        wq, wv, wt = w(nr), w(nv), w(nt)
        a = BatchSim(self._model, 1, device=str(self._dev), dtype=self._dtype, tape_capacity=num_steps)
        a.reset(q, qd, backward_flag=True)
        a.step(u, num_steps)
        du = a.backward_steps(num_steps, wq, wv, wt).double().sum(1)[0].cpu().numpy()
        lq, lv = (x.double()[0].cpu().numpy() for x in a.get_adjoint())
        n = nu + 2 * nr
        eps = 1e-6 if self._dtype == torch.float64 else 1e-3
        Q, QD, U = q.repeat(2 * n, 1), qd.repeat(2 * n, 1), u.repeat(2 * n, 1)
        for k in range(n):
            tgt, j = (U, k) if k < nu else ((Q, k - nu) if k < nu + nr else (QD, k - nu - nr))
            tgt[2 * k, j] += eps; tgt[2 * k + 1, j] -= eps
        f = BatchSim(self._model, 2 * n, device=str(self._dev), dtype=self._dtype, tape_capacity=0)
        f.reset(Q, QD, backward_flag=False)
        o = f.step(U, num_steps)
        L = (o["q"].double() * wq.double()).sum(1)
        if nv:
            L = L + (o["var"].double() * wv.double()).sum(1)
        if nt:
            L = L + (o["tactile"].double() * wt.double()).sum(1)
        fd = ((L[0::2] - L[1::2]) / (2 * eps)).cpu().numpy()
        an = np.concatenate([du, lq, lv])
        rel = lambda x, y: float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30))
        rep = {"df_du": rel(an[:nu], fd[:nu]), "df_dq0": rel(an[nu:nu + nr], fd[nu:nu + nr]), "df_dqdot0": rel(an[nu + nr:], fd[nu + nr:]),
               "eps": eps, "num_steps": num_steps}
        self.last_derivative_check = rep
        print("[redmax_py shim] test_derivatives over %d sub-step(s): relative error of the adjoint vs central differences: df_du %.3e, "
              "df_dq0 %.3e, df_dqdot0 %.3e (eps %.0e)" % (num_steps, rep["df_du"], rep["df_dq0"], rep["df_dqdot0"], eps))

    def get_q(self):
        self._refresh()
        return self._q

    def get_qdot(self):
        self._refresh()
        return self._qdot

    def get_variables(self):
        self._refresh()
        return self._var

    def get_tactile_force_vector(self):
        self._refresh()
        if self._backward_flag and getattr(self, "_nsub", 0) > 0 and (not self._tac_reads or self._tac_reads[-1] != self._nsub - 1):
            self._tac_reads.append(self._nsub - 1)          # the frame of taped sub-step _nsub - 1 has been handed out
        if self._tac is None:
            _, tac = self._sim.readout(want_var=False)
            if getattr(self, "_host_tac", None) is None:
                self._host_tac = torch.empty(tac.shape, dtype=tac.dtype).pin_memory()
            self._host_tac.copy_(tac, non_blocking=True)
            torch.cuda.current_stream(self._dev).synchronize()
            self._tac = self._host_tac.numpy().reshape(-1).astype(np.float64)
        return self._tac

    def get_tactile_image_pos(self, name):
        """list of (row, col) per taxel (examples/RollingBallExp/test_sim_speed.py:57-61)."""
        return [tuple(p) for p in self._model.meta["image_pos"][name]]

    def get_tactile_flow_images(self):
        """list[n_sensor] of [rows][cols][3] (envs/dclaw_rotate_env.py:103-105); empty cells are zero.  One fancy-index scatter per sensor
        (the (row, col) index arrays are built once per model: D'Claw has 3 x 302 taxels and the env calls this every step)."""
        tac = self.get_tactile_force_vector().reshape(-1, 3)
        idx = getattr(self, "_flow_idx", None)
        if idx is None:
            idx = self._flow_idx = [(t0, nt, rows, cols, np.asarray(self._model.meta["image_pos"][name], dtype=np.intp).reshape(-1, 2))
                                    for name, (t0, nt, rows, cols) in zip(self._model.meta["sensor_names"], self._model.meta["sensor_taxels"])]
        imgs = []
        for t0, nt, rows, cols, rc in idx:
            img = np.zeros((rows, cols, 3))
            img[rc[:, 0], rc[:, 1]] = tac[t0:t0 + rc.shape[0]]
            imgs.append(img)
        return imgs

    # ------------------------------------------------------------------ differentiation
    def _seeds(self, n):
        bi = self.backward_info
        nr, nv, nt = self.ndof_r, self.ndof_var, self.ndof_tactile

        def seed(a, dim, name):
            if a is None or dim == 0:
                return None
            a = np.asarray(a, dtype=np.float64).reshape(-1)
            if name == "df_dtactile" and a.size != n * dim:
                # masked frames only (envs/redmax_torch_functions.py:85-90 with tactile_masks): one block per tactile frame that was READ
                # among the newest n sub-steps, in the order they were read — scattered to their sub-steps here
                first = self._nsub - n
                reads = [t - first for t in self._tac_reads if t >= first]
                if a.size != len(reads) * dim:
                    raise RuntimeError("backward_info.df_dtactile has %d values: neither num_steps * %d = %d nor one block for each of the %d "
                                       "tactile frames read in these sub-steps" % (a.size, dim, n * dim, len(reads)))
                full = np.zeros((n, dim))
                full[reads] = a.reshape(len(reads), dim)
                a = full.reshape(-1)
            if a.size != n * dim:
                raise RuntimeError("backward_info.%s has %d values, expected num_steps * %d = %d" % (name, a.size, dim, n * dim))
            return a
        parts = [seed(bi.df_dq, nr, "df_dq"), seed(bi.df_dvar, nv, "df_dvar"), seed(bi.df_dtactile, nt, "df_dtactile")]
        # the three seeds go up through ONE pinned staging buffer and one asynchronous copy (three pageable torch.from_numpy(...).to(device)
        # copies cost three synchronisations: more than the B = 1 adjoint kernel itself)
        total = sum(p.size for p in parts if p is not None)
        if total == 0:
            return None, None, None
        if getattr(self, "_seed_host", None) is None or self._seed_host.numel() < total:
            self._seed_host = torch.empty(total, dtype=self._dtype).pin_memory()
            self._seed_dev = torch.empty(total, device=self._dev, dtype=self._dtype)
        hv = self._seed_host.numpy()
        out, o = [], 0
        for p in parts:
            if p is None:
                out.append(None)
                continue
            hv[o:o + p.size] = p                      # (casts to the batch's real type)
            out.append(self._seed_dev[o:o + p.size].reshape(1, p.size))
            o += p.size
        self._seed_dev[:total].copy_(self._seed_host[:total], non_blocking=True)      # stream-ordered before the adjoint launch that reads it
        return tuple(out)

    def backward_steps(self, num_steps):
        """envs/redmax_torch_functions.py:167 — newest num_steps sub-steps, continuing the carried adjoint."""
        self._sync_model()
        n = int(num_steps)
        a, b, c = self._seeds(n)
        du = self._sim.backward_steps(n, a, b, c, all_steps=True)
        self.backward_results.df_du = self._np(du)
        self._dirty_outputs = True
        self._nsub -= n
        self._tac_reads = [t for t in self._tac_reads if t < self._nsub]

    def backward(self):
        """envs/redmax_torch_functions.py:92 — the whole tape."""
        n = self._sim.tape_len()
        if n == 0:
            raise RuntimeError("backward(): nothing recorded (reset(backward_flag=True) + forward first)")
        self.backward_steps(n)
        lq, lv = self._sim.get_adjoint()
        self.backward_results.df_dq0 = self._np(lq)
        self.backward_results.df_dqdot0 = self._np(lv)

    def saveBackwardCache(self):
        self._sim.cache_save()
        # the simulation goes on from its current state on a spare tape, at the same sub-step index (tsim_cache_save)
        self._cache = getattr(self, "_cache", []) + [(self._nsub, list(self._tac_reads), self._backward_flag)]

    def popBackwardCache(self):
        self._sim.cache_pop()
        self._dirty_outputs = True
        if getattr(self, "_cache", None):
            self._nsub, self._tac_reads, self._backward_flag = self._cache.pop()

    def clearBackwardCache(self):
        self._sim.cache_clear()
        self._cache = []

    # ------------------------------------------------------------------ model edits (domain randomisation)
    def _edit(self, what, name, *args, **kw):
        """Edits only touch the spec; the blob is recompiled and uploaded ONCE, by the next call that needs the model on the
        device (envs/dclaw_rotate_env.py:173-178 makes four edits in a row at every reset)."""
        _mc.edit_spec(self._model.spec, what, name, *args, **kw)
        if what != "virtual_object":
            self._model_dirty = True

    def _sync_model(self):
        if getattr(self, "_model_dirty", False):
            self._model = _mc.compile_spec(self._model.spec)
            self._sim.update_model(self._model)
            self._model_dirty = False
            self._dirty_outputs = True

    def update_virtual_object(self, name, data):            # envs/tactile_push_env.py:148-152 (render only)
        self._edit("virtual_object", name, data)

    def update_joint_damping(self, name, damping):          # envs/dclaw_rotate_env.py:173
        self._edit("joint_damping", name, damping)

    def update_joint_location(self, name, pos):             # envs/dclaw_rotate_env.py:178
        self._edit("joint_location", name, pos)

    def update_body_size(self, name, size):                 # envs/dclaw_rotate_env.py:175
        self._edit("body_size", name, size)

    def update_endeffector_position(self, name, pos):       # envs/dclaw_rotate_env.py:176
        self._edit("endeffector_position", name, pos)

    def update_body_density(self, name, density):           # envs/stable_grasp_env.py:122
        self._edit("body_density", name, density)

    def update_body_color(self, name, color):               # envs/stable_grasp_env.py:128 (render only)
        pass

    def update_contact_parameters(self, general_body, primitive_body, kn=None, kt=None, mu=None, damping=None):
        self._edit("contact_parameters", (general_body, primitive_body), kn=kn, kt=kt, mu=mu, damping=damping)

    def update_tactile_parameters(self, sensor_body, kn=None, kt=None, mu=None, damping=None):
        self._edit("tactile_parameters", sensor_body, kn=kn, kt=kt, mu=mu, damping=damping)

    # ------------------------------------------------------------------ viewer (out of scope)
    def replay(self):
        pass

    def print_ctrl_info(self):
        pass

    def print_design_params_info(self):
        pass
