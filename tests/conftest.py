import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_1d():
    return json.load(open(os.path.join(GOLDEN, "ref_1d.json")))


@pytest.fixture(scope="session")
def golden_2d():
    return json.load(open(os.path.join(GOLDEN, "ref_2d.json")))


@pytest.fixture(scope="session")
def golden_wide():
    return json.load(open(os.path.join(GOLDEN, "ref_wide.json")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.load()
    return O


@pytest.fixture(scope="session")
def capi():
    """The product C ABI; the GPU tests call the kernels only through it."""
    from nanorepeat_amd import _capi
    _capi.load()
    return _capi
