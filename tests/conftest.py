"""Shared fixtures.  Mirrors the reference's tests/conftest.py:14-41 (seeds 0,1,2; Bernoulli(0.05)
10 x 1000 int8 data; default 16-state model with theta = rho = 1e-2)."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer-running CPU test")


@pytest.fixture(params=[0, 1, 2])
def rng(request):
    return np.random.default_rng(request.param)


@pytest.fixture
def data(rng):
    return (rng.uniform(size=(10, 1000)) < 0.05).astype(np.int8)


@pytest.fixture
def missing_data(data, rng):
    # reference tests/test_gpu.py:16-20
    inds = rng.integers(0, data.size, size=int(0.01 * data.size))
    data.flat[inds] = -1
    return data.clip(-1, 1)


@pytest.fixture
def psmcfa_file():
    # data fixture held by the reference's tests (tests/fixtures/sample.psmcfa, used by
    # tests/test_data.py:31-38); a data file, copied verbatim as a golden input
    return os.path.join(GOLDEN, "sample.psmcfa")
