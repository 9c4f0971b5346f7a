"""Flat model-blob layout — Python mirror of include/tsim_blob.h (data format only).

The numeric constants are parsed out of the C header at import time so that there is exactly one
definition of the layout.
"""
import os
import re

_HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "include", "tsim_blob.h")


def _parse_header(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    consts = {}
    for m in re.finditer(r"^#define[ \t]+(TSIM_\w+)[ \t]+(\S+)", txt, flags=re.M):
        try:
            consts[m.group(1)] = int(m.group(2), 0)
        except ValueError:
            consts[m.group(1)] = float(m.group(2))          # e.g. TSIM_STEP_MAX
    for m in re.finditer(r"enum\s*\{(.*?)\}", txt, flags=re.S):
        val = -1
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [s.strip() for s in item.split("=")]
                val = int(v, 0) if not v.startswith("TSIM") else consts[v]
            else:
                name = item
                val += 1
            consts[name] = val
    return consts


C = _parse_header(_HDR)
globals().update(C)

JOINT_TYPES = {
    "revolute": C["TSIM_J_REVOLUTE"], "prismatic": C["TSIM_J_PRISMATIC"], "planar": C["TSIM_J_PLANAR"],
    "translational": C["TSIM_J_TRANSLATIONAL"], "free3d-euler": C["TSIM_J_FREE3D_EULER"],
    "free3d-exp": C["TSIM_J_FREE3D_EXP"], "spherical-exp": C["TSIM_J_SPHERICAL_EXP"],
}
JOINT_NDOF = {"fixed": 0, "revolute": 1, "prismatic": 1, "planar": 2, "translational": 3,
              "free3d-euler": 6, "free3d-exp": 6}
