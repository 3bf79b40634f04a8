"""Config 4 as stated (8 shards on one device, 50 000 x 8 192, r = 256, nsNMF, bf16): the single-engine run once, each shard mode several times;
prints every run's distance from the single-engine result -- a deterministic path prints the same number every time.
usage: python tools/c4_modes_repeat.py   (needs a GPU; test infrastructure only)"""
# (the switches this tool sets are read by the measurement build only: csrc/tuning.h)
import os as _os
_os.environ.setdefault("NMFAMD_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nmfgpu_amd", "lib", "libnmfgpu64_diag.so"))
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
F = np.asfortranarray
def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64))
assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
na.set_verbosity(na.Verbosity.Nothing)
m, n, r, iters = 50000, 8 * 1024, 256, int(os.environ.get("ITERS", "4"))
rng = np.random.default_rng(4)
V = np.empty((m, n), dtype=np.float32, order="F")
for j0 in range(0, n, 1024):
    V[:, j0:j0 + 1024] = rng.random((m, 1024), dtype=np.float32)
W0 = F((1.0 - rng.random((m, r))).astype(np.float32)); H0 = F((1.0 - rng.random((r, n))).astype(np.float32))
base = {"theta": 0.5, "precision": 1}
def run(extra):
    W, H = W0.copy(order="F"), H0.copy(order="F")
    s = na.Summary()
    assert na.compute(V, W, H, algorithm=na.NmfAlgorithm.nsNMF, iterations=iters, parameters=dict(base, **extra), summary=s) == na.ResultType.Success
    return W, H, s.record(0).frobenius
Ws, Hs, fs = run({})
W2, H2, f2 = run({})
print("single again: W %.3e H %.3e" % (rel(W2, Ws), rel(H2, Hs)), flush=True)
for key, extra in [(k, e) for k, e in (("rows", {"numGpus": 8, "shardMode": 0}), ("repl", {"numGpus": 8, "shardMode": 1}), ("rows2", {"numGpus": 2, "shardMode": 0}), ("repl2", {"numGpus": 2, "shardMode": 1})) if not os.environ.get("ONLY") or k in os.environ["ONLY"].split(",")]:
    for k in range(int(os.environ.get("REPS", "4"))):
        W, H, f = run(extra)
        print(f"{key:6s} run {k}: W {rel(W, Ws):.3e} H {rel(H, Hs):.3e} frob {abs(f - fs) / fs:.2e}", flush=True)
