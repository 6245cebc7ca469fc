import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def qh():
    """The product library; built in-tree if missing.  GPU tests fail loudly without it."""
    try:
        import torch  # noqa: F401  -- before libquiskhip: one HIP runtime per process (torch's own copy), see quisk_amd/lib.py
    except ImportError:
        pass
    from quisk_amd import build as qbuild
    qbuild.build()
    import quisk_amd
    quisk_amd.load()
    return quisk_amd


def rel_rms(a, b):
    import numpy as np
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / max(np.sum(np.abs(b) ** 2), 1e-300)))
