"""How far the least-squares family drifts from the fp64 oracle at config 5 (10 000 x 5 000, r = 64), per iteration count:
the HIP path, the float restatement of the reference's QR arithmetic (oracle on float32 data), cond(W^T W + reg).
usage: python tools/ls_family_drift.py   (needs a GPU; test infrastructure only)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
from oracle import oracle

F = np.asfortranarray
def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64))

V = F(np.random.RandomState(1).random_sample((5000, 10000)).astype(np.float32).T)
W = F((1.0 - np.random.RandomState(2).random_sample((64, 10000))).astype(np.float32).T)
H = F((1.0 - np.random.RandomState(3).random_sample((5000, 64))).astype(np.float32).T)
assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
for alg, kw in [("ahcls", dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01)), ("gdcls", dict(lam=0.01)), ("acls", dict(lambda_w=0.01, lambda_h=0.01)), ("als", {})]:
    for iters in (1, 2, 5, 10, 20):
        V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
        oracle.run(alg, V64, W64, H64, iters, **kw)
        W32, H32 = W.copy(order="F"), H.copy(order="F")
        oracle.run(alg, V, W32, H32, iters, **kw)
        eng = na.Engine(10000, 5000, 64, alg, **kw)
        eng.upload(V); eng.set_factors(W, H)
        eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
        Wg, Hg = eng.get_factors(); eng.close()
        lam = kw.get("lambda_h", kw.get("lam", 0.0))
        cond = np.linalg.cond(W64.T @ W64 + lam * np.eye(64)); condh = np.linalg.cond(H64 @ H64.T + kw.get("lambda_w", 0.0) * np.eye(64))
        print(f"{alg:6s} it={iters:2d}  gpu W {rel(Wg, W64):.2e} H {rel(Hg, H64):.2e} | f32 restatement W {rel(W32, W64):.2e} H {rel(H32, H64):.2e} | cond WtW {cond:.2e} HHt {condh:.2e}", flush=True)
