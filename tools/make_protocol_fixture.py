"""Record the simulator-call protocol of the REFERENCE's autograd Functions (envs/redmax_torch_functions.py) against the
recording mock in tests/mock_sim.py, and write it to tests/golden/protocol_trace.json.

Dev-container only: imports the reference's python from /root/reference with `gym` / `redmax_py` stubbed in sys.modules.
Only the recorded trace (data) is committed; no reference source travels.
"""
import json
import os
import sys
import types

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("TSIM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
for name in ("gym", "redmax_py"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["redmax_py"].Simulation = object

import torch  # noqa: E402
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_redmax_torch_functions", os.path.join(REF, "envs", "redmax_torch_functions.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)        # the reference's file, loaded without its envs/__init__.py (which needs gym)
StepSimFunction, EpisodicSimFunction = _mod.StepSimFunction, _mod.EpisodicSimFunction
from tests.mock_sim import run_step, run_episodic  # noqa: E402

if __name__ == "__main__":
    log_s, res_s = run_step(StepSimFunction, torch)
    log_e, res_e = run_episodic(EpisodicSimFunction, torch)
    out = {"source": "envs/redmax_torch_functions.py (reference), torch %s" % torch.__version__,
           "step": {"log": log_s, "results": res_s}, "episodic": {"log": log_e, "results": res_e}}
    with open(os.path.join(ROOT, "tests", "golden", "protocol_trace.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("step calls:", len(log_s), "episodic calls:", len(log_e))
