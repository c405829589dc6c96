"""pytest wiring: marker registration + repo root on sys.path.

``-m "not gpu"`` runs here (no GPU): oracle vs golden vectors, host logic,
C-ABI symbol checks, gloo world_size-2 tests.  ``-m gpu`` runs on the MI355X
box and calls the HIP path through the C-ABI.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def decode_array(obj):
    """Inverse of make_golden.b64: {"dtype", "shape", "b64"} -> numpy array."""
    import base64
    import numpy as np
    return np.frombuffer(base64.b64decode(obj["b64"]), dtype=np.dtype(obj["dtype"])).reshape(
        obj["shape"]).copy()


@pytest.fixture(scope="session")
def golden():
    return load_golden
