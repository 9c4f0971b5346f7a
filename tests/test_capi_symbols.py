"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU, no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    inc = os.path.join(ROOT, "include")
    txt = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsim_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tactilesimulation_amd.host import capi
    so = capi.LIB_PATH
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(capi.EXPORTS) == names, "host binding and header disagree"


def test_host_refuses_to_run_without_gpu_or_library(monkeypatch, pusher_model):
    """No CPU fallback: a CPU device is rejected, and a missing library raises."""
    from tactilesimulation_amd.host import capi
    from tactilesimulation_amd.host.batch import BatchSim
    with pytest.raises(RuntimeError):
        BatchSim(pusher_model, 4, device="cpu")
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libtsim_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.lib()


def test_create_rejects_bad_blobs_without_touching_the_gpu(pusher_model):
    import numpy as np
    from tactilesimulation_amd.host import capi
    L = capi.lib()
    I = np.ascontiguousarray(pusher_model.I, dtype=np.int32).copy()
    F = np.ascontiguousarray(pusher_model.F, dtype=np.float64)
    I[0] = 0
    h = ctypes.c_void_p()
    rc = L.tsim_batch_create(I.ctypes.data_as(capi._ip), F.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 4, 8, 0, 0, ctypes.byref(h))
    assert rc != 0 and b"magic" in L.tsim_last_error()


def test_kernel_table_names_the_instantiations_a_batch_launches():
    """bench.py's `kernel` records look an instantiation up by its mangled name in the table host/buildhash.py writes at build time from the code
    object's metadata: every (kernel, dtype, model size, launch shape, variant) a batch can launch must be in it, with its registers and code bytes."""
    import json
    import os
    from tactilesimulation_amd.host import buildhash
    if not os.path.exists(buildhash.KERNELS_JSON):
        import pytest
        pytest.skip("library not built yet (python -c 'import __graft_entry__ as g; g.build()')")
    table = json.load(open(buildhash.KERNELS_JSON))
    combos = [(k, dt, nr, False, l, "generic", False) for k in ("k_forward", "k_backward") for dt in ("f32", "f64") for nr in (7, 12) for l in (16, 32, 64)]
    combos += [(k, dt, 9, True, 64, "generic", False) for k in ("k_forward", "k_backward") for dt in ("f32", "f64")]                      # rotation-vector joint
    combos += [(k, "f32", 7, False, l, v, False) for k in ("k_forward", "k_backward") for l in (16, 32, 64) for v in ("static:pusher", "param:pusher")]
    combos += [(k, "f64", 7, False, l, v, False) for k in ("k_forward", "k_backward") for l in (32, 64) for v in ("static:pusher", "param:pusher")]
    combos += [(k, "f32", 7, False, 16, v, True) for k in ("k_forward", "k_backward") for v in ("static:pusher", "param:pusher")]         # closed loop
    combos += [("k_forward", "f32", 7, False, 16, v, pol, True) for v in ("static:pusher", "param:pusher") for pol in (False, True)]                              # every option at its default: TsDefaultOpts<>
    for c in combos:
        mangled, readable = buildhash.kernel_name(*c)
        assert mangled in table, (readable, mangled)
        rec = table[mangled]
        assert rec["vgpr_count"] > 0 and rec["code_bytes"] > 0 and rec["max_flat_workgroup_size"] == 64, (readable, rec)


def test_public_headers_are_plain_c99_and_cxx17(tmp_path):
    """include/*.h is the drop-in boundary: it must parse for a C host (strict ISO C99, -pedantic -Werror) and a C++ one alike, on its own —
    no HIP, torch or project-internal header behind it"""
    import shutil
    import subprocess
    inc = os.path.join(ROOT, "include")
    hdrs = sorted(f for f in os.listdir(inc) if f.endswith(".h"))
    assert {"tsim.h", "tsim_env.h", "tsim_model.h", "tsim_blob.h"} <= set(hdrs)
    src = "".join('#include "%s"\n' % h for h in hdrs) + "int main(void) { return TSIM_IH_SIZE == 40 && TSIM_KT_COUNT == 3 && TSIM_UPD_VIRTUAL_OBJECT == 7 ? 0 : 1; }\n"
    for cc, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++17", "cpp")):
        if shutil.which(cc) is None:
            pytest.skip("no %s on this machine" % cc)
        p = tmp_path / ("hdr." + ext)
        p.write_text(src)
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + inc, str(p), "-o", str(tmp_path / ("hdr_" + ext))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert subprocess.run([str(tmp_path / ("hdr_" + ext))]).returncode == 0
