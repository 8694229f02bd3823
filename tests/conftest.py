import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """a plain `pytest` on a box without a GPU skips the gpu-marked tests; when they are SELECTED (`-m gpu`, the GPU CI
    job) or OSOT_REQUIRE_GPU=1 is set, a missing device is a hard failure (gpu_device fixture): there is no fallback"""
    if "gpu" in (config.getoption("-m") or "") or os.environ.get("OSOT_REQUIRE_GPU") == "1":
        return
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU visible (select with -m gpu to make that an error)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible: the HIP path must run, there is no fallback")
    return 0
