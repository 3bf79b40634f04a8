"""BASELINE config 2 shape with bf16 MFMA operands (extension; NOT the benchmark's fp32 number)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
V = np.asfortranarray(np.random.RandomState(1).random_sample((5000, 10000)).astype(np.float32).T)
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((64, 10000))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((5000, 64))).astype(np.float32).T)
for prec in ("native", "bf16"):
    eng = na.Engine(10000, 5000, 64, "mu", precision=prec)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(20, first_iteration=1); eng.synchronize()
    t0 = time.perf_counter(); eng.iterate(200, first_iteration=21); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"{prec:7s} {dt / 200 * 1e6:8.1f} us/iteration  {200 / dt:8.1f} it/s  frobenius {eng.frobenius:.4f}")
