"""Measured-error log of the GPU tests: with TSIM_TEST_REPORT=<file> every call of rep() appends {key, value} to that JSON-lines file, so
that one run of the suite yields the numbers the stated tolerances are set from (profiles/r04_fp32_tolerance_sites.md)."""
import json
import os


def rep(key, **values):
    f = os.environ.get("TSIM_TEST_REPORT")
    if f:
        with open(f, "a") as fh:
            fh.write(json.dumps({"key": key, **{k: (float(v) if not isinstance(v, (str, int)) else v) for k, v in values.items()}}) + "\n")
