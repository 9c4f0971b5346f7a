"""The package's StepSimFunction / EpisodicSimFunction must drive a simulator exactly like the reference's
(envs/redmax_torch_functions.py): same calls, order, kwargs, array shapes and dtypes, same returned gradients.
Golden trace: tests/golden/protocol_trace.json, recorded from the REFERENCE's functions by tools/make_protocol_fixture.py."""
import json
import os

import torch

from tactilesimulation_amd.functions import StepSimFunction, EpisodicSimFunction
from tests.mock_sim import run_step, run_episodic

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "protocol_trace.json")))


def _norm(log):
    return json.loads(json.dumps(log))


def test_step_function_protocol():
    log, res = run_step(StepSimFunction, torch)
    assert _norm(log) == GOLD["step"]["log"]
    assert res == GOLD["step"]["results"]          # (5, 6) df_du of ones is sum-reduced by autograd to 5 * ones


def test_episodic_function_protocol():
    """Call for call what the reference's function does — including the masked-frames-only df_dtactile of :87 (2 x 780 values for the mask
    [F, T, F, T]); round 3 scattered it to all frames on this side of the boundary, now the shim does (tests/test_gpu_shim.py)."""
    log, res = run_episodic(EpisodicSimFunction, torch)
    assert _norm(log) == GOLD["episodic"]["log"]
    assert res == GOLD["episodic"]["results"]
