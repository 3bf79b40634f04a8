"""End-to-end nmfgpu::compute at BASELINE config 2 through the drop-in boundary (host buffers in and out):
the PCIe-inclusive rate next to the resident-engine rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na

V = np.asfortranarray(np.random.RandomState(1).random_sample((5000, 10000)).astype(np.float32).T)
W0 = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((64, 10000))).astype(np.float32).T)
H0 = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((5000, 64))).astype(np.float32).T)
na.initialize(); na.set_verbosity(na.Verbosity.Nothing)
for iters in (200, 2000):
    W, H = W0.copy(order="F"), H0.copy(order="F")
    s = na.Summary()
    t0 = time.perf_counter()
    res = na.compute(V, W, H, iterations=iters, summary=s)
    dt = time.perf_counter() - t0
    rec = s.record(0)
    print(f"compute(): {iters} iterations in {dt * 1e3:.1f} ms wall = {iters / dt:.0f} it/s end to end "
          f"(loop itself {rec.elapsedTime * 1e3:.0f} ms = {iters / max(rec.elapsedTime, 1e-9):.0f} it/s), frobenius {rec.frobenius:.3f}, result {res.name}")
na.finalize()
