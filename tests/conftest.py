import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def gpu(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    pkg.hip.device_check(0)  # raises unless gfx950 + library present: no silent fallback
    return torch.device("cuda:0")


@pytest.fixture
def lib_option(pkg):
    """Set a tuning switch of the library (svr2_set_option) for one test; restored to the default afterwards."""
    touched = []

    def set_(name, value):
        pkg.hip.set_option(name, value)
        touched.append(name)

    yield set_
    for name in touched:
        pkg.hip.set_option(name, None)
