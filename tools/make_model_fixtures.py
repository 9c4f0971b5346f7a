"""Compile the reference's model XMLs into self-contained blobs (tactilesimulation_amd/assets/*.npz).

Run in the dev container only (needs /root/reference). The blobs are DATA: flat int/float arrays + a JSON spec in
which every mesh is reduced to its unit-density mass properties; no reference source text is stored. The GPU box
(no /root/reference) loads these blobs.
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model  # noqa: E402

REF = os.environ.get("TSIM_REFERENCE", "/root/reference")
MODELS = {
    "pusher": "envs/assets/pusher/pusher.xml",
    "tactile_pad": "assets/tactile_pad/tactile_pad.xml",
    "tactile_insertion": "envs/assets/tactile_insertion/tactile_insertion.xml",
    "dclaw_position_control": "envs/assets/dclaw_rotate/dclaw_position_control.xml",
    "stable_grasp": "envs/assets/stable_grasp/stable_grasp.xml",
}

if __name__ == "__main__":
    out = os.path.join(ROOT, "tactilesimulation_amd", "assets")
    os.makedirs(out, exist_ok=True)
    for name, rel in MODELS.items():
        m = load_model(os.path.join(REF, rel))
        m.save(os.path.join(out, name + ".npz"))
        print(name, "nr", m.ndof_r, "nu", m.ndof_u, "nvar", m.ndof_var, "ntac", m.ndof_tactile, "I", len(m.I), "F", len(m.F))
