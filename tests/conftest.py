import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from irotavg_amd import capi
        return capi.lib().irotavg_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly rather than skip: the product has no CPU path.
    pass


@pytest.fixture(scope="session")
def fixture_graph():
    from irotavg_amd import graphio
    return graphio.read_ravg_input(os.path.join(ROOT, "tests", "golden", "ravg_input.txt"))
