"""Run N iterations of one algorithm at config-2 shape (for rocprofv3 kernel traces)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
alg = sys.argv[1] if len(sys.argv) > 1 else "ahcls"
kw = {"mu": {}, "nsnmf": dict(theta=0.5), "gdcls": dict(lam=0.01), "als": {}, "acls": dict(lambda_w=0.01, lambda_h=0.01),
      "ahcls": dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01)}[alg]
V = np.asfortranarray(np.random.RandomState(1).random_sample((5000, 10000)).astype(np.float32).T)
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((64, 10000))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((5000, 64))).astype(np.float32).T)
eng = na.Engine(10000, 5000, 64, alg, **kw)
eng.upload(V); eng.set_factors(W, H)
eng.iterate(50, first_iteration=1); eng.synchronize()
print(alg, eng.frobenius)
