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
    log, res = run_episodic(EpisodicSimFunction, torch)
    gold = GOLD["episodic"]["log"]
    assert len(log) == len(gold)
    diffs = [(a, b) for a, b in zip(_norm(log), gold) if a != b]
    # the single documented deviation: df_dtactile is scattered to all T frames (T * 780 = 3120) instead of the
    # reference's masked-frames-only vector (2 * 780 = 1560), see functions.py and the reference's TODO at :69
    assert len(diffs) == 1
    mine, ref = diffs[0]
    assert mine[:2] == ref[:2] == ["set", "backward_info.df_dtactile"]
    assert mine[2]["ndarray"] == [4 * 780] and ref[2]["ndarray"] == [2 * 780]
    assert res == GOLD["episodic"]["results"]
