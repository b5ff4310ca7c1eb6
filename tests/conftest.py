import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def hostsim():
    """CPU logic harness built from the kernels' own sources (tests/hostsim; test infra)."""
    import ctypes

    so = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
    src = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call([os.path.join(ROOT, "tests", "hostsim", "build.sh")])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
