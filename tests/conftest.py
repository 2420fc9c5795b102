import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth_weights():
    from clair_amd import weights
    return weights.synthetic_weights(seed=20250928, head_gain=4.0)


@pytest.fixture(scope="session")
def engine(synth_weights):
    """One engine for the whole GPU session (HIP extension; raises if it cannot be created)."""
    from clair_amd import _capi
    e = _capi.Engine(device=0, max_batch=1024, n_slots=2)
    e.load_weights(synth_weights)
    yield e
    e.close()
