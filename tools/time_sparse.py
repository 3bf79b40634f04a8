"""BASELINE config 3 timing: sparse CSR V 100 000 x 20 000 at 1 % nnz, r = 128, MU with the KL divergence
(and the Frobenius objective on the same sparse path).  GPU box only; not a test."""
import os, sys, time
only = sys.argv[1] if len(sys.argv) > 1 else None
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na

m, n, r = 100000, int(os.environ.get("SPARSE_N", 20000)), 128
rng = np.random.default_rng(1)
t0 = time.perf_counter()
# fast generator: every row gets k = 1 % of n distinct columns (an arithmetic progression modulo n with a stride
# coprime to n and a random start), values uniform in {1..5} (ratings-like)
k = int(os.environ.get("SPARSE_K", n // 100))
start = rng.integers(0, n, size=m)
cols = np.sort((start[:, None] + 7919 * np.arange(k)[None, :]) % n, axis=1).astype(np.int32)
vals = rng.integers(1, 6, size=m * k).astype(np.float32)
s = sp.csr_matrix((vals, cols.ravel(), np.arange(0, m * k + 1, k, dtype=np.int32)), shape=(m, n))
W = np.asfortranarray((1.0 - rng.random((m, r))).astype(np.float32)); H = np.asfortranarray((1.0 - rng.random((r, n))).astype(np.float32))
print(f"generated nnz={s.nnz} in {time.perf_counter() - t0:.1f}s", flush=True)
for div in ("kl", "frobenius"):
    if only and div != only:
        continue
    eng = na.Engine(m, n, r, "mu", divergence=div, sparse_compute=True)
    t0 = time.perf_counter(); eng.upload_sparse(1, s.data, s.indptr, s.indices, 0); up = time.perf_counter() - t0
    eng.set_factors(W, H)
    eng.iterate(10, first_iteration=1, error_every=10); eng.synchronize()
    f0 = (eng.kl_divergence if div == "kl" else eng.frobenius)
    t0 = time.perf_counter(); eng.iterate(50, first_iteration=11, error_every=10); eng.synchronize(); dt = (time.perf_counter() - t0) / 50
    f1 = (eng.kl_divergence if div == "kl" else eng.frobenius)
    passes = 4 if div == "kl" else 2
    gather = passes * s.nnz * r * 4 / dt / 1e12
    flops = (8 if div == "kl" else 4) * s.nnz * r / dt / 1e12
    print(f"{div:10s} upload {up:.2f}s  {dt * 1e3:.3f} ms/iteration  {1 / dt:.1f} it/s  gather {gather:.2f} TB/s  {flops:.2f} TFLOP/s  objective {f0:.4g} -> {f1:.4g}", flush=True)
    del eng
