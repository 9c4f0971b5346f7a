import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


MODELS = os.path.join(ROOT, "tactilesimulation_amd", "assets")


@pytest.fixture(scope="session")
def pusher_model():
    from tactilesimulation_amd.model.compiler import load_model
    return load_model(os.path.join(MODELS, "pusher.npz"))
