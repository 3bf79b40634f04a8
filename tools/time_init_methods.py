import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import nmfgpu_amd as na
na.initialize(); na.set_verbosity(na.Verbosity.Nothing)
m, n, r = 10000, 5000, 64
rng = np.random.default_rng(1)
V = np.asfortranarray(rng.random((m, n)).astype(np.float32))
for name in [x for x in dir(na.NmfInitializationMethod) if not x.startswith("_")]:
    meth = getattr(na.NmfInitializationMethod, name)
    if name == "CopyExisting": continue
    W = np.zeros((m, r), dtype=np.float32, order="F"); H = np.zeros((r, n), dtype=np.float32, order="F")
    t0 = time.perf_counter()
    try:
        res = na.compute(V, W, H, init=meth, iterations=10, seed=3)
    except Exception as e:
        print(name, "ERR", e); continue
    print(f"{name}: compute() with 10 iterations {1e3*(time.perf_counter()-t0):.0f} ms ({res.name})", flush=True)
for v, nm in ((0, "nndsvd"), (1, "nndsvda"), (2, "nndsvdar")):
    W = np.zeros((m, r), dtype=np.float32, order="F"); H = np.zeros((r, n), dtype=np.float32, order="F")
    t0 = time.perf_counter()
    res = na.compute(V, W, H, iterations=10, seed=3, parameters={"nndsvd": float(v)})
    print(f"{nm}: {1e3*(time.perf_counter()-t0):.0f} ms ({res.name})", flush=True)
na.finalize()
