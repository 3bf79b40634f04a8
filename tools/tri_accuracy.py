"""Rank-256 bf16 mode against the fp64 oracle: relative error of W, H and the Frobenius value after `iters` iterations (nsNMF theta = 0.5).
usage: [NMFAMD_TRI_FP32_DEN=1] python tools/tri_accuracy.py   (needs a GPU; test infrastructure only)"""
# (the switches this tool sets are read by the measurement build only: csrc/tuning.h)
import os as _os
_os.environ.setdefault("NMFAMD_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nmfgpu_amd", "lib", "libnmfgpu64_diag.so"))
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
from oracle import oracle
F = np.asfortranarray
def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64))
assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
for m, n, iters in [(33000, 140, 10), (33000, 140, 40), (40000, 600, 20)]:
    r, theta = 256, 0.5
    rng = np.random.default_rng(m)
    V = F(rng.random((m, n)).astype(np.float32)); W = F((1.0 - rng.random((m, r))).astype(np.float32)); H = F((1.0 - rng.random((r, n))).astype(np.float32))
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("nsnmf", V64, W64, H64, iters, theta=theta)
    out = []
    for prec in ("bf16", "fp32"):
        eng = na.Engine(m, n, r, "nsnmf", theta=theta, **({"precision": "bf16"} if prec == "bf16" else {}))
        eng.upload(V); eng.set_factors(W, H)
        eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
        Wg, Hg = eng.get_factors()
        out.append(f"{prec}: W {rel(Wg, W64):.2e} H {rel(Hg, H64):.2e} frob {abs(eng.frobenius - ref['frobenius']) / ref['frobenius']:.2e}")
        eng.close()
    print(m, n, iters, " | ".join(out), flush=True)
