"""Config 3 (CSR 100 000 x 20 000, 1 % stored, r = 128, KL divergence) through nmfgpu::compute (host CSR in, factors out): the drop-in loop next to the resident rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import nmfgpu_amd as na
val, ptr, idx, W, H = bench.make_sparse_problem()
m, n, r = W.shape[0], H.shape[1], W.shape[1]
na.initialize(); na.set_verbosity(na.Verbosity.Nothing)
for rep in range(2):
    Wc, Hc = W.copy(order="F"), H.copy(order="F")
    s = na.Summary(); t0 = time.perf_counter()
    from nmfgpu_amd.api import sparse_description
    Vd = sparse_description(na.StorageFormat.CSR, m, n, val, ptr, idx)
    res = na.compute(Vd, Wc, Hc, iterations=300, summary=s, parameters={"divergence": 1.0})
    dt = time.perf_counter() - t0
    rec = s.record(0)
    print(f"config 3 through compute(): whole call {dt*1e3:.0f} ms, loop {rec.elapsedTime*1e3:.0f} ms = {rec.elapsedTime/300*1e6:.0f} us/iteration ({res.name})")
na.finalize()
