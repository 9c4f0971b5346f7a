"""ctypes binding of the C ABI in include/tsim.h and include/tsim_env.h (libtsim_hip.so, built in-tree by __graft_entry__.build()).

There is NO fallback: if the HIP library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csrc")
LIB_PATH = os.environ.get("TSIM_HIP_LIB") or os.path.abspath(os.path.join(_DIR, "libtsim_hip.so"))   # override: A/B builds
_lib = None

TSIM_F32, TSIM_F64 = 0, 1
_vp, _ip = C.c_void_p, C.POINTER(C.c_int32)

_SIGS = {
    "tsim_batch_create": (C.c_int, [_ip, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "tsim_batch_destroy": (None, [_vp]),
    "tsim_ndof_r": (C.c_int, [_vp]), "tsim_ndof_u": (C.c_int, [_vp]), "tsim_ndof_var": (C.c_int, [_vp]),
    "tsim_ndof_tactile": (C.c_int, [_vp]), "tsim_batch_size": (C.c_int, [_vp]), "tsim_dtype": (C.c_int, [_vp]),
    "tsim_timestep": (C.c_double, [_vp]), "tsim_tape_len": (C.c_int, [_vp]),
    "tsim_update_model": (C.c_int, [_vp, _ip, C.POINTER(C.c_double), _vp]),
    "tsim_reset": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "tsim_set_env_tables": (C.c_int, [_vp, _vp, _vp]), "tsim_table_size": (C.c_int, [_vp]),
    "tsim_reset_masked": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "tsim_step": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tsim_get_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "tsim_readout": (C.c_int, [_vp, _vp, _vp, _vp]),
    "tsim_backward_steps": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "tsim_rollout": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tsim_backward_episode": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tsim_get_adjoint": (C.c_int, [_vp, _vp, _vp, _vp]),
    "tsim_cache_save": (C.c_int, [_vp, _vp]), "tsim_cache_pop": (C.c_int, [_vp, _vp]), "tsim_cache_clear": (C.c_int, [_vp]),
    "tsim_cache_reserve": (C.c_int, [_vp, C.c_int]), "tsim_cache_depth": (C.c_int, [_vp]),
    "tsim_debug_signature": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "tsim_debug_eval": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tsim_debug_stamps": (C.c_int, [_vp, _vp]),
    "tsim_launch_info": (C.c_int, [_vp, _ip]),
    "tsim_set_lanes_per_env": (C.c_int, [_vp, C.c_int]),
    "tsim_kernel_timing": (C.c_int, [_vp, C.c_int]), "tsim_kernel_times": (C.c_int, [_vp, C.POINTER(C.c_double), _ip]),
    "tsim_static_model": (C.c_int, [_vp]), "tsim_set_static": (C.c_int, [_vp, C.c_int]),
    "tsim_kernel_variant": (C.c_char_p, [_vp]),
    "tsim_set_option": (C.c_int, [_vp, C.c_int, C.c_int]), "tsim_get_option": (C.c_int, [_vp, C.c_int]),
    "tsim_last_evals": (C.c_int, [_vp, _ip]),
    "tsim_last_helper_trials": (C.c_int, [_vp, _ip]),
    "tsim_set_solver_options": (C.c_int, [_vp, C.c_int, C.c_int]),
    "tsim_last_gnorm": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "tsim_last_error": (C.c_char_p, []),
    # include/tsim_model.h — the model loader (host code)
    "tsim_model_load": (C.c_int, [C.c_char_p, C.POINTER(_vp)]), "tsim_model_free": (None, [_vp]),
    "tsim_model_blob": (C.c_int, [_vp, C.POINTER(_ip), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int)]),
    "tsim_model_save_blob": (C.c_int, [_vp, C.c_char_p]), "tsim_model_load_blob": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "tsim_batch_create_from_model": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "tsim_model_image_pos": (C.c_int, [_vp, C.c_char_p, _ip, C.c_int]),
    "tsim_model_update": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_int]),
    "tsim_model_table_offset": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_char_p, C.c_int]),
    # include/tsim_env.h — TactilePush per-step formulas
    "tsim_push_action": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "tsim_push_action_backward": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "tsim_push_observe": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tsim_push_closed_rollout": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tsim_push_closed_backward": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tsim_push_observe_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _vp, _vp, _vp, _vp, _vp]),
}
EXPORTS = sorted(_SIGS)


class PushPolicyStruct(C.Structure):
    """include/tsim_env.h tsim_push_policy"""
    _fields_ = [(n, C.c_void_p) for n in ("W1T", "b1", "W2T", "b2", "W3", "b3", "W1p", "W2")] + [("w1_stride", C.c_int), ("obs_mode", C.c_int)] + \
               [(n, C.c_void_p) for n in ("eps", "logstd", "obs_mean", "obs_istd")] + [("obs_clip", C.c_double)]


def lib():
    """Load libtsim_hip.so (after torch, so that both share one HIP runtime)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
                               "g.build()'`). There is no CPU fallback." % LIB_PATH)
        _check_fresh()
        import torch  # noqa: F401  (loads libamdhip64 first)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("TSIM_HIP_LIB"):       # A/B against an older build: entry points it lacks stay unbound
                    continue
                raise RuntimeError("%s does not export %s (stale build? run __graft_entry__.build())" % (LIB_PATH, name))
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check_fresh():
    """The in-tree library must have been built from the sources it sits next to (content hash in its sidecar, written by
    __graft_entry__.build()); an A/B library named by TSIM_HIP_LIB is exempt.  The recipe (sources, flags) is resolved relative to this
    package (host/buildhash.py), never through an import that may fail: a check that cannot be performed is an error, not a pass."""
    if os.environ.get("TSIM_HIP_LIB"):
        return
    from . import buildhash
    try:
        want = buildhash.hip_digest()
    except OSError as e:       # sources not shipped next to the library (a packaged install): say so, loudly, and go on
        import warnings
        warnings.warn("tsim: cannot check %s against its sources (%s): a stale library would go unnoticed" % (LIB_PATH, e), RuntimeWarning)
        return
    have = buildhash.read(LIB_PATH)
    if have != want:
        raise RuntimeError("%s is stale: it was not built from the sources next to it (sidecar %s, sources %s). Run "
                           "`python -c 'import __graft_entry__ as g; g.build()'`." % (LIB_PATH, have and have[:12], want[:12]))


def check(rc):
    if rc != 0:
        raise RuntimeError("tsim: " + lib().tsim_last_error().decode())
