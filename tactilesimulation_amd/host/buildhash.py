"""Content hash of a native library's inputs, kept in a sidecar `<lib>.buildhash` next to it.

__graft_entry__.build() rebuilds when the sidecar does not match sha256(sources + flags) — file times play no part (a checkout can
restore sources older than a stray .so) — and host/capi.py refuses to load a library that is stale with respect to the sources
it sits next to."""
import hashlib
import os


def digest(paths, flags=()):
    h = hashlib.sha256()
    for f in flags:
        h.update(f.encode() + b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def sidecar(lib):
    return lib + ".buildhash"


def read(lib):
    try:
        return open(sidecar(lib)).read().strip()
    except OSError:
        return None


def write(lib, value):
    with open(sidecar(lib), "w") as fh:
        fh.write(value + "\n")


# ---------------------------------------------------------------------------------------------------- the HIP library's recipe
# One description of what libtsim_hip.so is built from, resolved from THIS file's location: __graft_entry__.build() compiles with it and
# host/capi.py checks the sidecar against it without importing anything from the repository root (an import that fails silently would
# skip the check — exactly the stale-library situation the sidecar exists to catch).
_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(_PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(_PKG), "include")
HIP_SO = os.path.join(CSRC, "libtsim_hip.so")
# -amdgpu-function-calls=false: every device function is inlined into its kernel (LDS pointers stay LDS pointers; out-of-line calls
#   would pass them as flat pointers, which is slower and trips a backend assertion here)
# -Os, not -O3: the simulation kernels are one 8 - 9 k-instruction body per wavefront, four wavefronts of a CU at four places of it;
#   13 % fewer static instructions measured 1 - 2 % faster on every bench leg (profiles/r03_pmc_wait_decomposition.md)
# -fno-slp-vectorize: packing scalar fp32 math into v_pk_* pairs costs more v_mov than it saves FMAs here and pushes the kernels over
#   256 registers (measured: +9 %, profiles/r01_launch_shape_ab.txt)
# -amdgpu-sched-strategy=iterative-ilp: the kernels are long straight-line code executed by ONE wavefront per SIMD, i.e. bound by the latency of
#   dependent instructions; the ILP-first list scheduler instead of the occupancy-first default (occupancy is one wavefront either way) measured
#   +3 ... +6 % on every leg (TactileInsertion 2.22 -> 2.30 M, fp64 3.02 -> 3.16 M, closed loop 17.0 -> 18.1 M, static k_backward 0.865 -> 0.835 ms)
HIP_FLAGS = ["--offload-arch=gfx950", "-Os", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-amdgpu-function-calls=false",
             "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
# Translation units of the library and the extra flags of each.  -ffinite-math-only -fno-signed-zeros (x * 0 -> 0 and x + 0 -> x may be folded:
# exact for every finite x, only the sign of a zero can differ) go ONLY to the kernels instantiated for a statically known model
# (csrc/tsim_static.h): there they turn the generic link sweep into the handful of operations the model's structure leaves; the generic
# kernels — whose fp64 instantiations walk the oracle's iterates to round-off — are built without them.
# -O2 for the static unit: its kernels are small (42 KB at -Os, 59 KB at -O2: both inside the 64 KB instruction cache) and -O2's scheduling
# is worth 4 % there (k_forward 3.30 -> 3.17 ms, k_backward 0.91 -> 0.87 ms per 20-step launch); the generic unit keeps -Os (its NRM = 16
# kernels are 63 KB already; at -O2: D'Claw -5 %, TactileInsertion -2 %, fp64 -2 %).
HIP_UNITS = [("tsim_hip.hip", []), ("tsim_static_pusher.hip", ["-ffinite-math-only", "-fno-signed-zeros", "-O2"]),
             ("tsim_param_pusher.hip", ["-ffinite-math-only", "-fno-signed-zeros", "-O2"]),      # the same kernels, parameters at run time (tsim_static.h ts_F)
             ("tsim_static_pusher_policy.hip", ["-ffinite-math-only", "-fno-signed-zeros"]),      # (closed-loop instantiations: 76 KB at -O2, stay at -Os)
             ("tsim_param_pusher_policy.hip", ["-ffinite-math-only", "-fno-signed-zeros"]),       # ... of the structure-static kernels (round 6: closed loop with per-environment tables)
             ("tsim_model.cpp", ["-ffp-contract=off"])]                                           # host only: the model loader (include/tsim_model.h); plain double arithmetic, no contraction
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC"]      # units that are not .hip: no device code


def hip_build_commands(hipcc, out_so=None):
    """[(argv, cwd)] that build libtsim_hip.so: one compile per translation unit into csrc/build/, then the link."""
    out_so = out_so or HIP_SO
    bdir = os.path.join(CSRC, "build")
    cmds, objs = [], []
    for src, extra in HIP_UNITS:
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        cmds.append(([hipcc] + (HIP_FLAGS if src.endswith(".hip") else HOST_FLAGS) + extra + ["-c", os.path.join(CSRC, src), "-o", obj], CSRC))
        objs.append(obj)
    cmds.append(([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_so] + objs, CSRC))
    return bdir, cmds


def hip_sources():
    """Every source next to the library (a new header cannot be forgotten) + the three public headers."""
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    return [os.path.join(CSRC, f) for f in srcs] + [os.path.join(INCLUDE, f) for f in ("tsim.h", "tsim_blob.h", "tsim_env.h", "tsim_model.h")]


def hip_digest():
    return digest(hip_sources(), HIP_FLAGS + HOST_FLAGS + [u + ":" + " ".join(f) for u, f in HIP_UNITS])


# ---------------------------------------------------------------------------------------------------- what the built kernels use
KERNELS_JSON = HIP_SO + ".kernels.json"
_LLVM = "/opt/rocm/lib/llvm/bin"


def write_kernel_table(lib=None, out=None):
    """Registers, spills, LDS and code bytes of every kernel of the built library, from the code object's own metadata (llvm-readelf --notes /
    -s on the gfx950 image inside the .so), as JSON next to the library: bench.py names the instantiation it timed and quotes these."""
    import json
    import re
    import shutil
    import subprocess
    import tempfile
    lib, out = lib or HIP_SO, out or KERNELS_JSON
    tmp = tempfile.mkdtemp(prefix="tsim_kt_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(_LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        table = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(_LLVM, "llvm-readelf"), "--notes", f], cwd=tmp, capture_output=True, text=True, check=True).stdout
            syms = subprocess.run([os.path.join(_LLVM, "llvm-readelf"), "-sW", f], cwd=tmp, capture_output=True, text=True, check=True).stdout
            size = {}
            for line in syms.splitlines():
                t = line.split()
                if len(t) >= 8 and t[3] == "FUNC":
                    size[t[7]] = int(t[2])
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
                name = g("name")
                if not name:
                    continue
                name = name.group(1)
                rec = {k: int(g(k).group(1)) for k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
                                                        "private_segment_fixed_size", "max_flat_workgroup_size") if g(k)}
                rec["code_bytes"] = size.get(name)
                table[name] = rec
        with open(out, "w") as fh:
            json.dump(table, fh, indent=0, sort_keys=True)
        return table
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_name(kernel, dtype, nr, has_exp, lanes, variant, policy=False, default_opts=False):
    """Mangled name of the instantiation of k_forward / k_backward a batch launches (csrc/tsim_hip.hip TS_LAUNCH) and a readable form of it.
    default_opts: every solver / scheduling option of the batch is at its default (tsim_get_option TSIM_OPT_ALL_DEFAULT): the fp32 forward launch of a
    compiled-in model at 16 lanes per environment then runs the TsDefaultOpts<> instantiation (csrc/tsim_static.h)."""
    nrm, expj, lpe = (16, True, 64) if has_exp else ((8 if nr <= 8 else 16), False, lanes)
    if variant != "generic":
        nrm = 8
    ms = {"generic": ("v", "void"), "static:pusher": ("14TsStaticPusher", "TsStaticPusher"), "param:pusher": ("7TsParamI14TsStaticPusherE", "TsParam<TsStaticPusher>")}[variant]
    if default_opts and kernel == "k_forward" and variant != "generic" and dtype == "f32" and lpe == 16:
        ms = ("13TsDefaultOptsI%sE" % ms[0], "TsDefaultOpts<%s>" % ms[1])
    r = {"f32": ("f", "float"), "f64": ("d", "double")}[dtype]
    args = {"k_forward": "7FwdArgs", "k_backward": "7BwdArgs"}[kernel]
    mangled = "_Z%d%sI%sLi%dELb%dELi%dELb%dE%sEv%sIT_E" % (len(kernel), kernel, r[0], nrm, int(expj), lpe, int(policy), ms[0], args)
    return mangled, "%s<%s, NRM=%d, EXPJ=%s, LPE=%d, POLICY=%s, %s>" % (kernel, r[1], nrm, str(expj).lower(), lpe, str(policy).lower(), ms[1])
