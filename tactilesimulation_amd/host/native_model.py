"""ctypes face of the model loader behind the C ABI (include/tsim_model.h, csrc/tsim_model.cpp): what a C / C++ host calls in place of
`redmax_py.Simulation(model_path)` (envs/redmax_torch_env.py:33) and the update_* family.  The Python hosts of this package compile models
with model/compiler.py; this wrapper exists for tests (the two compilers are held against each other) and as the reference binding of the
loader for other languages."""
import ctypes as C

import numpy as np

from . import capi

UPD = {"joint_damping": 0, "joint_location": 1, "body_density": 2, "body_size": 3, "endeffector_position": 4, "contact_parameters": 5,
       "tactile_parameters": 6, "virtual_object": 7}
TAB = {"pair": 0, "sensor": 1, "dof": 2}
PAIR_FIELDS = {"kn": 0, "kt": 1, "mu": 2, "damping": 3, "shape0": 4, "shape1": 5, "shape2": 6, "shape3": 7}


class NativeModel:
    def __init__(self, path):
        """path: a redmax XML, or a blob file written by save_blob (any other extension than .xml)."""
        self._L = capi.lib()
        self._h = C.c_void_p()
        fn = self._L.tsim_model_load if path.endswith(".xml") else self._L.tsim_model_load_blob
        capi.check(fn(path.encode(), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.tsim_model_free(self._h)
            self._h = C.c_void_p()

    def blob(self):
        """(I int32[nI], F float64[nF]) copies of the compiled blob"""
        pi, pf, ni, nf = C.POINTER(C.c_int32)(), C.POINTER(C.c_double)(), C.c_int(), C.c_int()
        capi.check(self._L.tsim_model_blob(self._h, C.byref(pi), C.byref(ni), C.byref(pf), C.byref(nf)))
        return np.ctypeslib.as_array(pi, (ni.value,)).copy(), np.ctypeslib.as_array(pf, (nf.value,)).copy()

    def save_blob(self, path):
        capi.check(self._L.tsim_model_save_blob(self._h, path.encode()))

    def image_pos(self, sensor):
        n = self._L.tsim_model_image_pos(self._h, sensor.encode(), None, 0)
        if n < 0:
            raise RuntimeError("tsim: " + self._L.tsim_last_error().decode())
        out = np.zeros((n, 2), dtype=np.int32)
        self._L.tsim_model_image_pos(self._h, sensor.encode(), out.ctypes.data_as(capi._ip), n)
        return [tuple(int(x) for x in rc) for rc in out]

    def update(self, what, name, values, name2=None):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
        capi.check(self._L.tsim_model_update(self._h, UPD[what], name.encode(), name2.encode() if name2 is not None else None,
                                             v.ctypes.data_as(C.POINTER(C.c_double)), len(v)))

    def table_offset(self, kind, key0, key1=None, field=0):
        if kind == "pair":
            field = PAIR_FIELDS[field] if isinstance(field, str) else field
        elif kind == "sensor":
            field = PAIR_FIELDS[field] if isinstance(field, str) else field
        r = self._L.tsim_model_table_offset(self._h, TAB[kind], key0.encode(), key1.encode() if key1 is not None else None, int(field))
        if r < 0:
            raise KeyError(self._L.tsim_last_error().decode())
        return r

    def create_batch_handle(self, B, tape_capacity, dtype=capi.TSIM_F32, device=0):
        """tsim_batch_create_from_model -> raw tsim_batch* (c_void_p); the caller owns it (tsim_batch_destroy)"""
        h = C.c_void_p()
        capi.check(self._L.tsim_batch_create_from_model(self._h, B, tape_capacity, dtype, device, C.byref(h)))
        return h
