import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REPORT_DIR = os.path.join(ROOT, "gpurun_out")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def report():
    """Appends 'name err tol' lines to gpurun_out/parity_report.txt so a failing GPU run still tells
    which comparisons were off and by how much."""
    os.makedirs(REPORT_DIR, exist_ok=True)
    path = os.path.join(REPORT_DIR, "parity_report.txt")

    def rec(name, err, tol):
        with open(path, "a") as f:
            f.write(f"{name:60s} err={err:.3e} tol={tol:.1e} {'OK' if err <= tol else 'FAIL'}\n")
        return err <= tol
    return rec


@pytest.fixture
def tune():
    """Planner overrides for one test (svs_tuning_set); every switch is back at its default afterwards."""
    from svs_unet_pytorch_amd import _lib

    def set_(name, value):
        _lib.tuning(name, value)
    yield set_
    _lib.tuning("*", -1)
