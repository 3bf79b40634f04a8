"""fp64 engine at BASELINE config 2 shape (the double instantiation the R binding calls)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
V = np.asfortranarray(np.random.RandomState(1).random_sample((5000, 10000)).T)
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((64, 10000))).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((5000, 64))).T)
for alg in ("mu", "als"):
    eng = na.Engine(10000, 5000, 64, alg, dtype=np.float64)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(3, first_iteration=1); eng.synchronize()
    t0 = time.perf_counter(); eng.iterate(20, first_iteration=4); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"fp64 {alg:4s} {dt / 20 * 1e6:9.1f} us/iteration  {20 / dt:7.1f} it/s  frobenius {eng.frobenius:.6f}")
