"""Per-iteration time of every algorithm at BASELINE config 2 / 5 shape (GPU box, not a test)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na

V = np.asfortranarray(np.random.RandomState(1).random_sample((5000, 10000)).astype(np.float32).T)
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((64, 10000))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((5000, 64))).astype(np.float32).T)
for alg, kw in [("mu", {}), ("nsnmf", dict(theta=0.5)), ("gdcls", dict(lam=0.01)), ("als", {}), ("acls", dict(lambda_w=0.01, lambda_h=0.01)),
                ("ahcls", dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01))]:
    eng = na.Engine(10000, 5000, 64, alg, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(10, first_iteration=1); eng.synchronize()
    t0 = time.perf_counter(); eng.iterate(100, first_iteration=11); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"{alg:6s} {dt / 100 * 1e6:8.1f} us/iteration  {100 / dt:8.1f} it/s  frobenius {eng.frobenius:.3f}")
