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


@pytest.fixture
def diag_build():
    """The measurement build of the library (nmfgpu_amd/lib/libnmfgpu64_diag.so, `python -m nmfgpu_amd.build --diag`): the only build in which the A/B switches that select a
    kernel form (NMFAMD_GRAM_KSPLIT, NMFAMD_X3_COLSPLIT, NMFAMD_TRI_RIDE, NMFAMD_SHARD_NO_DIRECT, ...) are read -- the shipped library ignores them (csrc/tuning.h,
    tests/test_abi.py).  Inside the test, nmfgpu_amd binds to that build."""
    import nmfgpu_amd as na
    from nmfgpu_amd import _lib
    from nmfgpu_amd import build as _b
    if not os.path.exists(_b.DIAG_LIB):
        pytest.fail(f"{_b.DIAG_LIB} missing: __graft_entry__.build() builds it")
    with _lib.use_library(_b.DIAG_LIB):
        assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
        na.set_verbosity(na.Verbosity.Nothing)
        yield
        na.finalize()
