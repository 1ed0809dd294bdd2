import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # an experiment build selected for the whole session would be tested instead of the product (tests that want another
    # build select it for a child process of their own)
    if os.environ.get("DVI_ESKF_LIB"):
        raise pytest.UsageError("DVI_ESKF_LIB is set: the tests would not run against dvi_ekf_amd/libdvi_eskf.so")


def _has_gpu() -> bool:
    """Counting devices does not initialise the HIP runtime (torch.cuda.is_available() does): the process that
    collects the tests can still launch GPU programs as children before it touches the GPU itself."""
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
