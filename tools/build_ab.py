"""Build an A/B variant of the HIP library into tactilesimulation_amd/csrc/ab/libtsim_<name>.so with extra compiler flags (usually -D switches:
the ones tests/test_ab_macros_compile.py lists: TS_FINE_STAMPS, TS_SOLVE_PIVOT_ONLY, TS_ROUND_STATS, TS_PP_TIME, TS_WAVES_PER_EU, TS_ISA_MARKS ...), by the same recipe as the shipped library (one compile per
translation unit, host/buildhash.py).  Use it on the GPU box through TSIM_HIP_LIB=<path>.

    python tools/build_ab.py fine -DTS_FINE_STAMPS
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.host import buildhash      # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out = os.path.join(buildhash.CSRC, "ab", "libtsim_%s.so" % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    bdir = os.path.join(buildhash.CSRC, "build", "ab_" + name)
    os.makedirs(bdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for src, unit_flags in buildhash.HIP_UNITS:
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        procs.append(subprocess.Popen([hipcc] + buildhash.HIP_FLAGS + unit_flags + extra + ["-c", os.path.join(buildhash.CSRC, src), "-o", obj], cwd=buildhash.CSRC))
        objs.append(obj)
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("hipcc failed")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, cwd=buildhash.CSRC)
    print(out)


if __name__ == "__main__":
    main()
