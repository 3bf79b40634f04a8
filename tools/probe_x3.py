"""Accuracy and speed of the split-operand (3 x bf16) fp32 factor product against the native fp32 MFMA kernel."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmfgpu_amd._lib import library
from nmfgpu_amd import engine as eng


def x3(A, F, reps=0):
    lib = library()
    X, Y = A.shape
    r = F.shape[0]
    out = np.zeros((r, X), dtype=np.float32, order="F")
    us = C.c_double(0)
    st = lib.nmfamd_op_factor_product_x3(C.c_void_p(A.ctypes.data), C.c_long(X), X, Y, C.c_void_p(F.ctypes.data), C.c_long(r), r,
                                         C.c_void_p(out.ctypes.data), C.c_long(r), reps, C.byref(us))
    assert st == 0, st
    return out, us.value


def main():
    rs = np.random.RandomState(0)
    for (X, Y, r) in ((300, 500, 64), (1000, 777, 40), (10000, 5000, 64), (5000, 10000, 64)):
        A = np.asfortranarray(rs.random_sample((X, Y)).astype(np.float32))
        F = np.asfortranarray(rs.random_sample((r, Y)).astype(np.float32))
        exact = F.astype(np.float64) @ A.astype(np.float64).T          # r x X
        o3, us = x3(A, F, reps=20 if X >= 5000 else 0)
        of = eng.op_factor_product(A, F)
        of = of[0] if isinstance(of, tuple) else of
        e3 = np.abs(o3 - exact).max() / np.abs(exact).max()
        ef = np.abs(of - exact).max() / np.abs(exact).max()
        r3 = np.sqrt(((o3 - exact) ** 2).mean()) / np.sqrt((exact ** 2).mean())
        rf = np.sqrt(((of - exact) ** 2).mean()) / np.sqrt((exact ** 2).mean())
        print(f"{X}x{Y} r={r}: x3 max {e3:.3e} rms {r3:.3e} | fp32 mfma max {ef:.3e} rms {rf:.3e} | x3 product {us:.1f} us", flush=True)
    # signed, wide-dynamic-range operands
    A = np.asfortranarray((rs.standard_normal((2000, 3000)) * np.exp(rs.standard_normal((2000, 3000)) * 3)).astype(np.float32))
    F = np.asfortranarray(rs.standard_normal((64, 3000)).astype(np.float32))
    exact = F.astype(np.float64) @ A.astype(np.float64).T
    scale = np.abs(F.astype(np.float64)) @ np.abs(A.astype(np.float64)).T
    o3, _ = x3(A, F)
    of = eng.op_factor_product(A, F)
    of = of[0] if isinstance(of, tuple) else of
    print(f"signed/wide: x3 max err/|a||b| {np.abs(o3 - exact).max() / scale.max():.3e}, componentwise {(np.abs(o3 - exact) / scale).max():.3e} | "
          f"fp32 mfma {np.abs(of - exact).max() / scale.max():.3e}, componentwise {(np.abs(of - exact) / scale).max():.3e}")


if __name__ == "__main__":
    main()
