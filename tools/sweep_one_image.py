"""Odd shapes through the one-resident-image path (forced) against the two-image path: factors must be bit-identical."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na

shapes = [(5, 7, 3), (17, 300, 2), (129, 2, 1), (1000, 15, 8), (16, 16, 16), (4097, 33, 64), (33, 4097, 33), (130, 131, 64), (257, 513, 40),
          (2000, 3000, 64), (640, 17, 64), (17, 640, 64), (300, 300, 100), (100, 1000, 200)]
bad = 0
for alg, kw in (("mu", {}), ("als", {}), ("nsnmf", dict(theta=0.3))):
    for (m, n, r) in shapes:
        if r >= min(m, n) and alg == "als":
            continue
        rng = np.random.default_rng(m * 7 + n)
        V = np.asfortranarray(rng.random((m, n)).astype(np.float32))
        W = np.asfortranarray((1 - rng.random((m, r))).astype(np.float32))
        H = np.asfortranarray((1 - rng.random((r, n))).astype(np.float32))
        res = {}
        for one in ("0", "1"):
            os.environ["NMFAMD_ONE_IMAGE"] = one
            e = na.Engine(m, n, r, alg, **kw)
            g = e.geometry()
            e.upload(V); e.set_factors(W, H)
            e.iterate(6, last_iteration=6)
            res[one] = e.get_factors() + (e.frobenius, g["product_kernel"], g["resident_images"])
            e.close()
        same = np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1]) and res["0"][2] == res["1"][2]
        fin = np.isfinite(res["1"][0]).all() and np.isfinite(res["1"][1]).all()
        if not (same and fin):
            bad += 1
        print(f"{alg:6s} {m:5d}x{n:5d} r={r:3d} kernel {res['1'][3]} images {res['0'][4]}/{res['1'][4]} identical {same} finite {fin} frob {res['1'][2]:.6f}", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
