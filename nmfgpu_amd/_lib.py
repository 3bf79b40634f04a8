"""Locates and loads libnmfgpu64.so.  Fails loudly: there is no fallback implementation."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# NMFAMD_LIBRARY: another build of the SAME library (A/B measurements of kernel variants); never a fallback
_PATH = os.environ.get("NMFAMD_LIBRARY") or os.path.join(_HERE, "lib", "libnmfgpu64.so")
_lib = None


class LibraryMissing(RuntimeError):
    pass


def library_path() -> str:
    return _PATH


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch's ROCm wheels bundle their own copy of libamdhip64.so (file name without
    the version, SONAME libamdhip64.so.7).  If torch is imported FIRST, libnmfgpu64.so's dependency on libamdhip64.so.7
    binds to that copy and everything shares one runtime (streams can be passed between the two).  The other way round
    the system copy is loaded for us, torch later loads its own by file name, and the second runtime in the process finds
    no device (torch.cuda.is_available() turns False).  So when a torch installation is present and not yet imported,
    its copy is loaded here, before our library -- without importing torch.  NMFAMD_NO_TORCH_HIP_PRELOAD=1 skips this."""
    if "torch" in sys.modules or os.environ.get("NMFAMD_NO_TORCH_HIP_PRELOAD"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def library() -> C.CDLL:
    """The loaded shared library (built by `python -m nmfgpu_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise LibraryMissing(
                f"{_PATH} not found: build it with `python -m nmfgpu_amd.build` (needs hipcc). "
                "nmfgpu_amd has no CPU or PyTorch fallback.")
        _share_torch_hip_runtime()
        _lib = C.CDLL(_PATH, mode=C.RTLD_LOCAL)
    return _lib


class use_library:
    """Context manager for tests and measurements: another BUILD of the same library (the measurement build, nmfgpu_amd/lib/libnmfgpu64_diag.so) becomes what library()
    returns inside the block.  Objects created inside keep the library they were created with.  Never a fallback: the file must exist."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise LibraryMissing(f"{path} not found (python -m nmfgpu_amd.build --diag)")
        self._path = path

    def __enter__(self):
        global _lib
        library()                          # (the default one first: it decides which HIP runtime the process shares)
        self._saved = _lib
        _lib = C.CDLL(self._path, mode=C.RTLD_LOCAL)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False
