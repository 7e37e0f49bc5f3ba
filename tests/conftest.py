"""Test configuration.  `-m "not gpu"`: oracle pin against the golden fixtures, host logic, ABI
surface, the test-only CPU emulation of the device code.  `-m gpu`: parity tests proper, through
the C ABI of libhpt.so on cuda:0 (an MI355X)."""
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.util import CASES, load_case  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session")
def cases():
    return {n: load_case(n) for n in CASES}
