"""Pairs of identical engines of DIFFERENT kinds on one device at once (a stream and a host thread each): the two engines of a pair must agree bit for
bit while every other kind's kernels run beside them.  usage: python tools/shared_device_mix.py   (ITERS=8; needs a GPU; test infrastructure only)"""
import os, sys, threading
import numpy as np
import scipy.sparse as sp
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmfgpu_amd.engine import Engine

iters = int(os.environ.get("ITERS", "8"))
rng = np.random.default_rng(9)
F = np.asfortranarray
def dense(m, n): return F(rng.random((m, n), dtype=np.float32))
def start(m, n, r, dt=np.float32): return F((1.0 - rng.random((m, r))).astype(dt)), F((1.0 - rng.random((r, n))).astype(dt))
kinds = {}
def add(name, m, n, r, V, dtype=np.float32, sparse=None, **kw):
    W0, H0 = start(m, n, r, dtype)
    def make(stream):
        e = Engine(m, n, r, dtype=dtype, stream=stream, **kw)
        if sparse is not None: e.upload_sparse(1, sparse.data.astype(dtype), sparse.indptr, sparse.indices, 0)
        else: e.upload(V)
        e.set_factors(W0, H0); e.synchronize()
        return e
    kinds[name] = (make, m, n, r)
add("nsnmf r256 bf16", 50000, 1024, 256, dense(50000, 1024), algorithm="nsnmf", theta=0.5, precision="bf16")
add("mu r64", 10000, 5000, 64, dense(10000, 5000), algorithm="mu")
add("ahcls r64", 5000, 2000, 64, dense(5000, 2000), algorithm="ahcls", lambda_w=0.1, lambda_h=0.1, alpha_w=0.5, alpha_h=0.5)
add("gdcls r64", 5000, 2000, 64, dense(5000, 2000), algorithm="gdcls", lam=0.1)
add("mu r200 fp32", 8000, 3000, 200, dense(8000, 3000), algorithm="mu")
add("mu r48 fp64", 3000, 2000, 48, F(rng.random((3000, 2000))), dtype=np.float64, algorithm="mu")
S = sp.random(20000, 4000, density=0.02, format="csr", random_state=3, dtype=np.float32); S.data = np.abs(S.data) + 0.1
add("kl sparse r64", 20000, 4000, 64, None, sparse=S, algorithm="mu", divergence="kl", sparse_compute=True)
add("mu sparse r64", 20000, 4000, 64, None, sparse=S, algorithm="mu", sparse_compute=True)
torch.cuda.set_device(0)
engines = []      # (kind, engine)
for name, (make, m, n, r) in kinds.items():
    for _ in range(2):
        engines.append((name, make(torch.cuda.Stream().cuda_stream)))
streams_alive = []
def work(k, barrier, it):
    torch.cuda.set_device(0)
    barrier.wait()
    engines[k][1].iterate(2, first_iteration=2 * it + 1, error_every=1000); engines[k][1].synchronize()
bad = 0
for it in range(iters):
    barrier = threading.Barrier(len(engines))
    ts = [threading.Thread(target=work, args=(k, barrier, it)) for k in range(len(engines))]
    for t in ts: t.start()
    for t in ts: t.join()
    line = []
    for p in range(0, len(engines), 2):
        name = engines[p][0]; _, m, n, r = kinds[name]; rp = (r + 63) // 64 * 64
        for which, cnt in ((1, rp * n), (0, rp * m)):
            a = engines[p][1].debug_read(which, cnt).view(np.uint8); b = engines[p + 1][1].debug_read(which, cnt).view(np.uint8)
            d = np.flatnonzero(a != b)
            if d.size: line.append(f"{name}: {'WH'[which]} differs in {d.size} bytes"); bad += 1
    print(f"round {it + 1}: " + ("all pairs identical" if not line else "; ".join(line)), flush=True)
print("DIFFERENCES" if bad else "IDENTICAL", flush=True)
