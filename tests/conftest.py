import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elem_err(a, b):
    """Element-wise form of the parity bound: max over elements of |a - b| / (|b| + rms(b)).  rel_err above bounds the error by the
    LARGEST reference entry; this one holds every element to its own size, with the tensor's RMS as the floor that keeps entries near
    zero from dividing by nothing (rms <= max, so elem_err >= rel_err / 2 always: it is the stricter of the two)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    rms = float(np.sqrt(np.mean(b * b)))
    return float((np.abs(a - b) / (np.abs(b) + max(rms, 1e-30))).max())
