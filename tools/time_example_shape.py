"""The configuration of the reference's example program (ref example/main.cpp:30-32,89-131): 4096 x 165, r = 158,
nsNMF theta = 0.5, double precision -- per-iteration time of the resident engine and the kernel mix."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
m, n, r = 4096, 165, 158
rng = np.random.default_rng(1)
for dtype in (np.float64, np.float32):
    V = np.asfortranarray((rng.integers(0, 255, (m, n)) / 255.0).astype(dtype))
    W = np.asfortranarray((1.0 - rng.random((m, r))).astype(dtype)); H = np.asfortranarray((1.0 - rng.random((r, n))).astype(dtype))
    eng = na.Engine(m, n, r, "nsnmf", dtype=dtype, theta=0.5)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(20, first_iteration=1); eng.synchronize()
    t0 = time.perf_counter(); eng.iterate(500, first_iteration=21); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"{np.dtype(dtype).name}: {dt / 500 * 1e6:8.1f} us/iteration  {500 / dt:8.1f} it/s  frobenius {eng.frobenius:.6f}")
