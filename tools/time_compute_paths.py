import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import nmfgpu_amd as na
na.initialize(); na.set_verbosity(na.Verbosity.Nothing)
def run(name, m, n, r, alg, dtype, iters, **kw):
    rng = np.random.default_rng(1)
    V = np.asfortranarray(rng.random((m, n)).astype(dtype)); W = np.asfortranarray((1 - rng.random((m, r))).astype(dtype)); H = np.asfortranarray((1 - rng.random((r, n))).astype(dtype))
    for rep in range(2):
        Wc, Hc = W.copy(order="F"), H.copy(order="F")
        s = na.Summary(); t0 = time.perf_counter()
        res = na.compute(V, Wc, Hc, algorithm=alg, iterations=iters, summary=s, **kw)
        dt = time.perf_counter() - t0
    rec = s.record(0)
    print(f"{name}: whole call {dt*1e3:.1f} ms, loop {rec.elapsedTime*1e3:.0f} ms = {rec.elapsedTime/iters*1e6:.1f} us/iteration ({res.name})")
run("example shape, nsNMF, double", 4096, 165, 158, na.NmfAlgorithm.nsNMF, np.float64, 2000, parameters={"theta": 0.5})
run("config 5 AHCLS", 10000, 5000, 64, na.NmfAlgorithm.AHCLS, np.float32, 1000, parameters={"lambdaW": 0.01, "lambdaH": 0.01, "alphaW": 0.01, "alphaH": 0.01})
run("config 5 GDCLS", 10000, 5000, 64, na.NmfAlgorithm.GDCLS, np.float32, 1000, parameters={"lambda": 0.01})
run("config 2 MU", 10000, 5000, 64, na.NmfAlgorithm.Multiplicative, np.float32, 2000)
na.finalize()
