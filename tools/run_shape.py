"""Run ITERS iterations of one algorithm at an arbitrary shape (timing + rocprofv3 kernel traces).
usage: run_shape.py M N R ALG [native|bf16] [ITERS]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
m, n, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
alg = sys.argv[4] if len(sys.argv) > 4 else "mu"
prec = sys.argv[5] if len(sys.argv) > 5 else "native"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
kw = {"mu": {}, "nsnmf": dict(theta=0.5), "gdcls": dict(lam=0.01), "als": {}, "acls": dict(lambda_w=0.01, lambda_h=0.01),
      "ahcls": dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01)}[alg]
rs = np.random.RandomState(1)
V = np.empty((m, n), dtype=np.float32, order="F")
for j0 in range(0, n, 512):
    V[:, j0:j0 + 512] = rs.random_sample((m, min(512, n - j0))).astype(np.float32)
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((r, m))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((n, r))).astype(np.float32).T)
eng = na.Engine(m, n, r, alg, precision=prec, **kw)
eng.upload(V); eng.set_factors(W, H)
eng.iterate(3, first_iteration=1); eng.synchronize()
t0 = time.perf_counter(); eng.iterate(iters, first_iteration=4); eng.synchronize(); dt = time.perf_counter() - t0
print(f"{alg} {m}x{n} r={r} {prec}: {dt / iters * 1e6:.1f} us/iteration  {iters / dt:.1f} it/s  frobenius {eng.frobenius:.4f}")
