import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "scikit-dsp-comm_amd")
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_err(y, ref):
    """max-abs error / max-abs reference and relative L2 (SURVEY.md section 8c tolerances)."""
    import numpy as np
    y = np.asarray(y)
    ref = np.asarray(ref)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    if ref.size == 0:
        return 0.0, 0.0
    d = np.abs(y.astype(np.complex128) - ref.astype(np.complex128))
    peak = float(np.max(np.abs(ref)))
    l2 = float(np.sqrt(np.sum(np.abs(ref.astype(np.complex128)) ** 2)))
    return float(np.max(d)) / (peak if peak > 0 else 1.0), float(np.sqrt(np.sum(d ** 2))) / (l2 if l2 > 0 else 1.0)
