"""pytest configuration: registers the ``gpu`` marker and shared fixtures."""

from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "reference_kernels.npz"))


@pytest.fixture(scope="session")
def visium49():
    return np.load(os.path.join(GOLDEN, "visium49.npz"))
