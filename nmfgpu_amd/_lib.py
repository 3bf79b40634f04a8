"""Locates and loads libnmfgpu64.so.  Fails loudly: there is no fallback implementation."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NMFAMD_LIBRARY: another build of the SAME library (A/B measurements of kernel variants); never a fallback
_PATH = os.environ.get("NMFAMD_LIBRARY") or os.path.join(_HERE, "lib", "libnmfgpu64.so")
_lib = None


class LibraryMissing(RuntimeError):
    pass


def library_path() -> str:
    return _PATH


def library() -> C.CDLL:
    """The loaded shared library (built by `python -m nmfgpu_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise LibraryMissing(
                f"{_PATH} not found: build it with `python -m nmfgpu_amd.build` (needs hipcc). "
                "nmfgpu_amd has no CPU or PyTorch fallback.")
        _lib = C.CDLL(_PATH, mode=C.RTLD_LOCAL)
    return _lib
