"""The compile-time A/B switches that still live in the product kernels (tools/build_ab.py builds a library variant with them; profiles/ quotes
what each measured) must keep compiling: one semantic-analysis pass (`hipcc -fsyntax-only`, host and gfx950 passes, every template
instantiated; no GPU) of the library's translation units with all of them defined.  Switches whose losing side was decided are gone
(TS_TAX_PLAIN / ZEROS / UNROLL / CH, TS_PP_SKIP_*, TS_PP_NO_MFMA, TS_NO_COLD_HINTS, TS_STATIC_UNFUSED, TS_STATIC_BRANCH_BLOCKS)."""
import os
import re
import shutil
import subprocess

import pytest

from tactilesimulation_amd.host import buildhash

SURVIVING = {                      # switch -> what it is for
    "TS_FINE_STAMPS": "extra shader-clock stamps inside the phases of an evaluation (tools/fine_stamps.py)",
    "TS_SOLVE_PIVOT_ONLY": "the pivoted solve only, no pivot-free DPP attempt first",
    "TS_ROUND_STATS": "rounds and shader clocks per wavefront, left in status / gnorm (tools/round_stats.py)",
    "TS_PP_TIME": "share of a closed-loop launch spent in the policy call (tools/closed_loop_breakdown.py)",
    "TS_WAVES_PER_EU": "ask for n wavefronts per SIMD in the simulation kernels",
    "TS_REAL_BARRIERS": "__syncthreads() instead of the compiler-only barrier",
    "TS_LIBM_SINCOS": "libm sincos instead of the kernels' own",
    "TS_BWD_REEVAL": "k_backward evaluates the Newton matrix of the taped point again instead of reading it from the tape: the cost of a tape without it (profiles/r06_tape_ab.md)",
    "TS_BWD_TWO_WAVES": "the adjoint kernel without the one-wavefront-per-SIMD attribute (with TSIM_BWD_LPE=32: two two-environment wavefronts per SIMD; profiles/r06_static_kernel_levers.md)",
    "TS_ISA_MARKS": "every stamp site as a unique s_sleep marker in the ISA (static analysis)",
}


def _hipcc():
    return os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_every_switch_in_the_sources_is_a_known_one():
    found = set()
    for f in os.listdir(buildhash.CSRC):
        if f.endswith((".h", ".hip")):
            found |= set(re.findall(r"^\s*#\s*if(?:n?def)?\s+(?:defined\s*\(\s*)?(TS_[A-Z0-9_]+)", open(os.path.join(buildhash.CSRC, f)).read(), flags=re.M))
    assert found == set(SURVIVING), (sorted(found - set(SURVIVING)), sorted(set(SURVIVING) - found))


@pytest.mark.parametrize("defs", [["-DTS_FINE_STAMPS", "-DTS_SOLVE_PIVOT_ONLY", "-DTS_ROUND_STATS", "-DTS_PP_TIME", "-DTS_WAVES_PER_EU=1", "-DTS_REAL_BARRIERS", "-DTS_LIBM_SINCOS", "-DTS_BWD_REEVAL", "-DTS_BWD_TWO_WAVES"],
                                  ["-DTS_ISA_MARKS"]])
def test_the_surviving_switches_compile(defs):
    if not os.path.exists(_hipcc()):
        pytest.skip("no hipcc")
    for src, unit_flags in buildhash.HIP_UNITS:
        flags = [f for f in buildhash.HIP_FLAGS + unit_flags if f not in ("-fPIC",)]
        r = subprocess.run([_hipcc()] + flags + defs + ["-fsyntax-only", os.path.join(buildhash.CSRC, src)], cwd=buildhash.CSRC, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (src, defs, r.stderr[-3000:])
