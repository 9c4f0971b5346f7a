"""BatchSim — B environments of one model resident on one MI355X, driven through the C ABI (include/tsim.h).

Device memory, streams and dtype plumbing come from PyTorch-ROCm; all simulation arithmetic runs in the
hand-written HIP kernels (tactilesimulation_amd/csrc). Host-side mirror of the stepping/adjoint part of the
reference's `redmax_py.Simulation`, batched.
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from ..model.compiler import CompiledModel, load_model


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class BatchSim:
    def __init__(self, model, batch_size, device="cuda:0", dtype=torch.float32, tape_capacity=512):
        if isinstance(model, str):
            model = load_model(model)
        assert isinstance(model, CompiledModel)
        self.model = model
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchSim needs a GPU device (got %s); the HIP path has no CPU fallback" % device)
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be float32 or float64")
        self.dtype = dtype
        self.B = int(batch_size)
        self.tape_capacity = int(tape_capacity)
        L = capi.lib()
        self._I = np.ascontiguousarray(model.I, dtype=np.int32)
        self._F = np.ascontiguousarray(model.F, dtype=np.float64)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        capi.check(L.tsim_batch_create(self._I.ctypes.data_as(capi._ip), self._F.ctypes.data_as(C.POINTER(C.c_double)), self.B,
                                       self.tape_capacity, capi.TSIM_F32 if dtype == torch.float32 else capi.TSIM_F64, idx,
                                       C.byref(h)))
        self._h = h
        self.ndof_r, self.ndof_u = L.tsim_ndof_r(h), L.tsim_ndof_u(h)
        self.ndof_var, self.ndof_tactile = L.tsim_ndof_var(h), L.tsim_ndof_tactile(h)
        self.h = L.tsim_timestep(h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                capi.lib().tsim_batch_destroy(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t, dim, name):
        if t is None:
            return None
        if t.device != self.device or t.dtype != self.dtype:
            t = t.to(device=self.device, dtype=self.dtype)
        t = t.contiguous()
        # env-major [B, dim] (or [B, n, d] with n * d == dim for per-sub-step seeds): a transposed or otherwise mis-shaped
        # tensor must not be reinterpreted silently
        if t.dim() < 2 or t.shape[0] != self.B or t.numel() != self.B * dim:
            raise ValueError("%s: expected shape [%d, %d], got %s" % (name, self.B, dim, tuple(t.shape)))
        return t

    def empty(self, *dims):
        return torch.empty((self.B,) + dims, device=self.device, dtype=self.dtype)

    # ------------------------------------------------------------------ stepping
    def update_model(self, model):
        self.model = model
        self._I = np.ascontiguousarray(model.I, dtype=np.int32)
        self._F = np.ascontiguousarray(model.F, dtype=np.float64)
        capi.check(capi.lib().tsim_update_model(self._h, self._I.ctypes.data_as(capi._ip),
                                                self._F.ctypes.data_as(C.POINTER(C.c_double)), self._stream()))

    # ------------------------------------------------------------------ per-environment parameters
    def base_tables(self):
        """[B, table_size] copy of the model's numeric tables, one row per environment, ready to be edited."""
        n = capi.lib().tsim_table_size(self._h)
        row = torch.tensor(self.model.F[:n], device=self.device, dtype=self.dtype)
        return row.unsqueeze(0).repeat(self.B, 1)

    def set_env_tables(self, tables):
        """Per-environment numeric tables (domain randomisation); None reverts to the shared model."""
        if tables is not None:
            tables = self._chk(tables, capi.lib().tsim_table_size(self._h), "tables")
        capi.check(capi.lib().tsim_set_env_tables(self._h, _ptr(tables), self._stream()))

    def reset(self, q0, qd0=None, backward_flag=False):
        q0 = self._chk(q0, self.ndof_r, "q0")
        qd0 = self._chk(qd0, self.ndof_r, "qd0")
        capi.check(capi.lib().tsim_reset(self._h, _ptr(q0), _ptr(qd0), int(bool(backward_flag)), self._stream()))

    def reset_masked(self, q0, mask, qd0=None):
        """New state for the environments with mask != 0 only (forward-only batches: roll-out collection)."""
        q0 = self._chk(q0, self.ndof_r, "q0")
        qd0 = self._chk(qd0, self.ndof_r, "qd0")
        m = torch.as_tensor(mask).to(device=self.device).reshape(-1)
        if m.numel() != self.B:
            raise ValueError("mask: expected %d entries" % self.B)
        m = (m != 0).to(torch.int32).contiguous()
        capi.check(capi.lib().tsim_reset_masked(self._h, _ptr(q0), _ptr(qd0), _ptr(m), self._stream()))

    def step(self, u, num_steps=1, want_qd=False, want_var=True, want_tactile=True, out=None):
        """One env-step for all B environments. Returns dict(q, qd, var, tactile, status)."""
        u = self._chk(u, self.ndof_u, "u")
        o = out if out is not None else {}
        if "q" not in o:
            o["q"] = self.empty(self.ndof_r)
        if want_qd and "qd" not in o:
            o["qd"] = self.empty(self.ndof_r)
        if want_var and self.ndof_var and "var" not in o:
            o["var"] = self.empty(self.ndof_var)
        if want_tactile and self.ndof_tactile and "tactile" not in o:
            o["tactile"] = self.empty(self.ndof_tactile)
        if "status" not in o:
            o["status"] = torch.empty(self.B, device=self.device, dtype=torch.int32)
        capi.check(capi.lib().tsim_step(self._h, _ptr(u), int(num_steps), _ptr(o["q"]), _ptr(o.get("qd")), _ptr(o.get("var")),
                                        _ptr(o.get("tactile")), _ptr(o["status"]), self._stream()))
        return o

    def _tactile_slots(self, T, tactile_mask):
        """tactile_masks (bool[T]) of EpisodicSimFunction -> (device int32[T] slot map, number of masked frames)."""
        if tactile_mask is None:
            return None, T
        m = torch.as_tensor(tactile_mask).to(device=self.device).reshape(-1).bool()
        if m.numel() != T:
            raise ValueError("tactile_mask: expected %d entries, got %d" % (T, m.numel()))
        slots = torch.where(m, torch.cumsum(m.int(), 0) - 1, torch.full_like(m, -1, dtype=torch.int64)).to(torch.int32).contiguous()
        return slots, int(m.sum().item())

    def rollout(self, u, num_steps=1, want_qd=False, want_var=True, want_tactile=True, tactile_mask=None):
        """Open-loop episode in one launch (include/tsim.h tsim_rollout): u [T, B, ndof_u] -> dict of [T, B, dim] outputs
        (+ status [B]); the same results as T calls of step().  tactile_mask (bool[T]): tactile only for the masked
        frames, output [n_masked, B, ndof_tactile]."""
        if u.dim() != 3 or u.shape[1] != self.B or u.shape[2] != self.ndof_u:
            raise ValueError("rollout: u must be [T, %d, %d], got %s" % (self.B, self.ndof_u, tuple(u.shape)))
        T = int(u.shape[0])
        u = u.to(device=self.device, dtype=self.dtype).contiguous()
        new = lambda d: torch.empty((T, self.B, d), device=self.device, dtype=self.dtype)
        o = {"q": new(self.ndof_r), "status": torch.empty(self.B, device=self.device, dtype=torch.int32)}
        if want_qd:
            o["qd"] = new(self.ndof_r)
        if want_var and self.ndof_var:
            o["var"] = new(self.ndof_var)
        slots, nm = self._tactile_slots(T, tactile_mask)
        if want_tactile and self.ndof_tactile:
            o["tactile"] = torch.empty((nm, self.B, self.ndof_tactile), device=self.device, dtype=self.dtype)
        capi.check(capi.lib().tsim_rollout(self._h, _ptr(u), T, int(num_steps), _ptr(slots), _ptr(o["q"]), _ptr(o.get("qd")),
                                           _ptr(o.get("var")), _ptr(o.get("tactile")), _ptr(o["status"]), self._stream()))
        return o

    def backward_episode(self, num_frames, num_steps, df_dq=None, df_dvar=None, df_dtactile=None, tactile_mask=None):
        """Adjoint of the newest num_frames env-steps (num_steps sub-steps each) in one launch. Seeds [T, B, dim] or None
        (df_dtactile [n_masked, B, dim] with tactile_mask); returns df_du [T, B, ndof_u] (gradient w.r.t. the action of
        each frame)."""
        T = int(num_frames)
        slots, nm = self._tactile_slots(T, tactile_mask)

        def chk(t, dim, name, n=T):
            if t is None or dim == 0:
                return None
            t = t.to(device=self.device, dtype=self.dtype).contiguous()
            if tuple(t.shape) != (n, self.B, dim):
                raise ValueError("%s: expected [%d, %d, %d], got %s" % (name, n, self.B, dim, tuple(t.shape)))
            return t
        a, b = chk(df_dq, self.ndof_r, "df_dq"), chk(df_dvar, self.ndof_var, "df_dvar")
        c = chk(df_dtactile, self.ndof_tactile, "df_dtactile", nm) if nm > 0 else None
        du = torch.empty((T, self.B, self.ndof_u), device=self.device, dtype=self.dtype)
        capi.check(capi.lib().tsim_backward_episode(self._h, T, int(num_steps), _ptr(slots), _ptr(a), _ptr(b), _ptr(c), _ptr(du),
                                                    self._stream()))
        return du

    def get_state(self):
        q, qd = self.empty(self.ndof_r), self.empty(self.ndof_r)
        capi.check(capi.lib().tsim_get_state(self._h, _ptr(q), _ptr(qd), self._stream()))
        return q, qd

    def readout(self, want_var=True, want_tactile=True):
        var = self.empty(self.ndof_var) if (want_var and self.ndof_var) else None
        tac = self.empty(self.ndof_tactile) if (want_tactile and self.ndof_tactile) else None
        if var is not None or tac is not None:
            capi.check(capi.lib().tsim_readout(self._h, _ptr(var), _ptr(tac), self._stream()))
        return var, tac

    # ------------------------------------------------------------------ adjoint
    def backward_steps(self, n, df_dq=None, df_dvar=None, df_dtactile=None, all_steps=False):
        """Adjoint of the newest n recorded sub-steps. Seeds are [B, dim] (last sub-step only) or, with
        all_steps=True, [B, n, dim]. Returns df_du [B, n, ndof_u]."""
        mul = n if all_steps else 1
        a = self._chk(df_dq, mul * self.ndof_r, "df_dq")
        b = self._chk(df_dvar, mul * self.ndof_var, "df_dvar") if self.ndof_var else None
        c = self._chk(df_dtactile, mul * self.ndof_tactile, "df_dtactile") if self.ndof_tactile else None
        du = self.empty(n, self.ndof_u)
        capi.check(capi.lib().tsim_backward_steps(self._h, int(n), int(bool(all_steps)), _ptr(a), _ptr(b), _ptr(c), _ptr(du),
                                                  self._stream()))
        return du

    def get_adjoint(self):
        a, b = self.empty(self.ndof_r), self.empty(self.ndof_r)
        capi.check(capi.lib().tsim_get_adjoint(self._h, _ptr(a), _ptr(b), self._stream()))
        return a, b

    def tape_len(self):
        return capi.lib().tsim_tape_len(self._h)

    def cache_save(self):
        capi.check(capi.lib().tsim_cache_save(self._h, self._stream()))

    def cache_pop(self):
        capi.check(capi.lib().tsim_cache_pop(self._h, self._stream()))

    def cache_clear(self):
        capi.check(capi.lib().tsim_cache_clear(self._h))

    def cache_reserve(self, depth):
        """Pre-allocate tape buffers for `depth` saved episodes, so that cache_save / cache_pop never allocate."""
        capi.check(capi.lib().tsim_cache_reserve(self._h, int(depth)))

    def cache_depth(self):
        return capi.lib().tsim_cache_depth(self._h)

    # ------------------------------------------------------------------ diagnostics
    def debug_eval(self, q1, q0, qd0, u, cycles=False):
        q1, q0, qd0 = (self._chk(x, self.ndof_r, "q") for x in (q1, q0, qd0))
        u = self._chk(u, self.ndof_u, "u")
        g, H = self.empty(self.ndof_r), self.empty(self.ndof_r, self.ndof_r)
        cyc = torch.zeros(self.B, 32, device=self.device, dtype=torch.int64) if cycles else None
        capi.check(capi.lib().tsim_debug_eval(self._h, _ptr(q1), _ptr(q0), _ptr(qd0), _ptr(u), _ptr(g), _ptr(H), _ptr(cyc),
                                              self._stream()))
        return (g, H, cyc) if cycles else (g, H)

    def branch_signature(self, t_first=0, n=None):
        """[n, B, 2] int64 (count, hash) of the contact / friction branches of the taped sub-steps t_first+1 .. t_first+n
        (include/tsim.h tsim_debug_signature); needs reset(backward_flag=True)."""
        n = self.tape_len() - t_first if n is None else int(n)
        out = torch.empty((n, self.B, 2), device=self.device, dtype=torch.int32)
        capi.check(capi.lib().tsim_debug_signature(self._h, int(t_first), n, _ptr(out), self._stream()))
        return out.to(torch.int64) & 0xFFFFFFFF

    def last_evals(self):
        out = np.zeros(self.B, dtype=np.int32)
        capi.check(capi.lib().tsim_last_evals(self._h, out.ctypes.data_as(capi._ip)))
        return out

    def last_helper_trials(self):
        """Line-search trials of the most recent forward launch that a helper slot evaluated, per environment (include/tsim.h TSIM_OPT_TRIAL_HELPERS)."""
        out = np.zeros(self.B, dtype=np.int32)
        capi.check(capi.lib().tsim_last_helper_trials(self._h, out.ctypes.data_as(capi._ip)))
        return out

    def last_gnorm(self):
        """Largest ||g|| any sub-step of the most recent forward launch ended with, per environment (float32 [B])."""
        out = np.zeros(self.B, dtype=np.float32)
        capi.check(capi.lib().tsim_last_gnorm(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def set_solver_options(self, cross_kinks=None, eval_budget=0):
        """include/tsim.h tsim_set_solver_options: the two options around the XML-stated Newton loop.  cross_kinks None = the library's
        default (on for fp32 batches, off for fp64 ones, which then run the literal loop)."""
        if cross_kinks is None:
            cross_kinks = self.dtype == torch.float32
        capi.check(capi.lib().tsim_set_solver_options(self._h, int(bool(cross_kinks)), int(eval_budget)))

    def set_lanes_per_env(self, lanes):
        """16 / 32 / 64 lanes per environment, 0 = automatic (include/tsim.h tsim_set_lanes_per_env)."""
        capi.check(capi.lib().tsim_set_lanes_per_env(self._h, int(lanes)))

    def static_model(self):
        """Id of the statically specialised kernel instantiation the next launch uses (include/tsim.h tsim_static_model; 0: generic)."""
        return capi.lib().tsim_static_model(self._h)

    def set_static(self, allow):
        capi.check(capi.lib().tsim_set_static(self._h, int(bool(allow))))

    def kernel_variant(self):
        """Name of the kernel instantiation the next launch uses: "generic", "static:<model>", "param:<model>" (include/tsim.h)."""
        return capi.lib().tsim_kernel_variant(self._h).decode()

    OPT_PAIR_CULL, OPT_VALUE_TRIALS, OPT_TRIAL_HELPERS, OPT_VALUE_FIRST, OPT_CROSS_KINKS, OPT_EVAL_BUDGET, OPT_ALL_DEFAULT = 1, 2, 3, 4, 5, 6, 7

    def set_option(self, option, value):
        """include/tsim.h tsim_set_option (TSIM_OPT_*)."""
        capi.check(capi.lib().tsim_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        return capi.lib().tsim_get_option(self._h, int(option))

    KERNEL_KINDS = ("k_forward", "k_taxels", "k_backward")

    def kernel_timing(self, enable=True):
        """HIP events around every simulation-kernel launch of this batch, on the launching stream (include/tsim.h tsim_kernel_timing)."""
        capi.check(capi.lib().tsim_kernel_timing(self._h, int(bool(enable))))

    def kernel_times(self):
        """{kernel: (summed ms, launches)} since the previous call; waits for the recorded events (include/tsim.h tsim_kernel_times)."""
        ms, n = (C.c_double * 3)(), (C.c_int32 * 3)()
        capi.check(capi.lib().tsim_kernel_times(self._h, ms, n))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.KERNEL_KINDS)}

    def launch_info(self):
        out = (C.c_int32 * 4)()
        capi.lib().tsim_launch_info(self._h, out)
        return {"lds_bytes": out[0], "threads": out[1], "blocks": out[2], "lanes_per_env": out[3]}
