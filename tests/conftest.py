import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the driver records which
    # native libraries the run actually loaded.  Without -m, GPU tests are skipped on CPU boxes.
    if config.getoption("-m"):
        return
    if not _has_gpu():
        skip = pytest.mark.skip(reason="no GPU in this container")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)
