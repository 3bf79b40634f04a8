"""Accuracy and speed of the split-operand product reading the image tiled along the reduction index."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmfgpu_amd._lib import library


def run(name, A, F, reps=0):
    lib = library()
    X, Y = A.shape
    r = F.shape[0]
    out = np.zeros((r, X), dtype=np.float32, order="F")
    us = C.c_double(0)
    st = getattr(lib, name)(C.c_void_p(A.ctypes.data), C.c_long(X), X, Y, C.c_void_p(F.ctypes.data), C.c_long(r), r,
                            C.c_void_p(out.ctypes.data), C.c_long(r), reps, C.byref(us))
    assert st == 0, st
    return out, us.value


rs = np.random.RandomState(0)
reps = int(os.environ.get("REPS", "40"))
for (X, Y, r) in ((300, 500, 64), (1000, 777, 40), (130, 2049, 33), (257, 1111, 256), (5000, 10000, 64), (10000, 5000, 64)):
    A = np.asfortranarray((rs.random_sample((X, Y)) - 0.2).astype(np.float32))
    F = np.asfortranarray(rs.random_sample((r, Y)).astype(np.float32))
    exact = F.astype(np.float64) @ A.astype(np.float64).T
    big = X >= 5000
    o1, us1 = run("nmfamd_op_factor_product_x3", A, F, reps if big else 0)
    o2, us2 = run("nmfamd_op_factor_product_x3_ytiled", A, F, reps if big else 0)
    e = lambda o: np.abs(o - exact).max() / np.abs(exact).max()
    print(f"{X}x{Y} r={r}: x-tiled err {e(o1):.3e} {us1:.1f} us | y-tiled err {e(o2):.3e} {us2:.1f} us | equal {np.array_equal(o1, o2)}", flush=True)
