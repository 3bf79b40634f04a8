import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench, nmfgpu_amd as na
from oracle import oracle
F = np.asfortranarray
m, n, r = 100000, 20000, 128
val, ptr, idx, W0, H0 = bench.make_sparse_problem()
eng = na.Engine(m, n, r, "mu", divergence="kl")
eng.upload_sparse(1, val, ptr, idx, 0)
eng.set_factors(W0, H0)
eng.iterate(1, first_iteration=1, error_every=0)
W1, H1 = eng.get_factors()
nrm = np.linalg.norm(W1.astype(np.float64), axis=0)
print("W1 col norms: min %.8f max %.8f" % (nrm.min(), nrm.max()))
W64, H64 = F(W0.astype(np.float64)), F(H0.astype(np.float64))
oracle.run_kl_csr(m, n, val.astype(np.float64), ptr, idx, W64, H64, 1)
print("it1 rel W %.3e H %.3e" % (np.linalg.norm(W1 - W64) / np.linalg.norm(W64), np.linalg.norm(H1 - H64) / np.linalg.norm(H64)))
eng.iterate(1, first_iteration=2, error_every=0)
W2, H2 = eng.get_factors()
oracle.run_kl_csr(m, n, val.astype(np.float64), ptr, idx, W64, H64, 1)
print("it2 rel W %.3e H %.3e" % (np.linalg.norm(W2 - W64) / np.linalg.norm(W64), np.linalg.norm(H2 - H64) / np.linalg.norm(H64)))
ratio = (H2.astype(np.float64) / H64)
print("H2/H64 per row c: mean", ratio.mean(axis=1)[:8], "spread", (ratio.max(axis=1) - ratio.min(axis=1)).max())
sW = W1.astype(np.float64).sum(axis=0)
print("colsum W1 (fp64 of gpu W1)", sW[:4], "oracle-side", )
