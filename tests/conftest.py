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
    # -m gpu on a box without a usable GPU (or without the built HIP library) must fail loudly rather than skip or
    # pass on some other path: the product has no CPU path. The check runs once, before the first selected gpu test.
    if not any(it.get_closest_marker("gpu") for it in items):
        return
    sel = config.getoption("-m") or ""
    if "gpu" not in sel or "not gpu" in sel:
        return
    if not _has_gpu():
        raise pytest.UsageError("-m gpu was asked for, but irotavg_amd/libirotavg_hip.so does not load or reports no HIP "
                                "device: the product has no CPU path, the gpu tests cannot pass here")


@pytest.fixture(scope="session")
def fixture_graph():
    from irotavg_amd import graphio
    return graphio.read_ravg_input(os.path.join(ROOT, "tests", "golden", "ravg_input.txt"))
