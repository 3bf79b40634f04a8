"""Ranks <= 32: the products compute 32 panel columns instead of 64 (NMFAMD_FP_FULL_WIDTH=1 restores the full width)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
Vd = np.asfortranarray(np.random.RandomState(1).random_sample((5000, 10000)).T)
for dtype in (np.float32, np.float64):
    V = np.asfortranarray(Vd.astype(dtype))
    for r in (16, 32):
        W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((r, 10000))).T.astype(dtype))
        H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((5000, r))).T.astype(dtype))
        for full in (True, False):
            if full:
                os.environ["NMFAMD_FP_FULL_WIDTH"] = "1"
            else:
                os.environ.pop("NMFAMD_FP_FULL_WIDTH", None)
            eng = na.Engine(10000, 5000, r, "mu", dtype=dtype)
            eng.upload(V); eng.set_factors(W, H)
            eng.iterate(10, first_iteration=1); eng.synchronize()
            t0 = time.perf_counter(); eng.iterate(100, first_iteration=11); eng.synchronize(); dt = time.perf_counter() - t0
            width = "64 columns" if full else "32 columns"
            print(f"{np.dtype(dtype).name} r={r} {width}: {dt / 100 * 1e6:8.1f} us/iteration  frobenius {eng.frobenius:.9f}")
