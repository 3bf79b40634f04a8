"""Run-to-run determinism of the default path at config 2's shape: identical factors after 1500 iterations, five times."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na

m, n, r = 10000, 5000, 64
rng = np.random.default_rng(1)
V = np.asfortranarray(rng.random((m, n)).astype(np.float32))
W = np.asfortranarray((1 - rng.random((m, r))).astype(np.float32))
H = np.asfortranarray((1 - rng.random((r, n))).astype(np.float32))
sigs = []
for alg, kw in (("mu", {}), ("als", {}), ("nsnmf", dict(theta=0.5))):
    for rep in range(4):
        e = na.Engine(m, n, r, alg, **kw)
        e.upload(V); e.set_factors(W, H)
        e.iterate(1500 if alg == "mu" else 400, last_iteration=0)
        Wg, Hg = e.get_factors()
        sig = hashlib.sha1(Wg.tobytes() + Hg.tobytes()).hexdigest()[:16]
        print(alg, rep, sig, f"frob {e.frobenius:.6f}", e.geometry()["product_kernel"], e.geometry()["resident_images"], flush=True)
        sigs.append((alg, sig))
        e.close()
bad = sum(1 for alg in ("mu", "als", "nsnmf") if len({s for a, s in sigs if a == alg}) != 1)
print("NONDETERMINISTIC ALGORITHMS:", bad)
sys.exit(1 if bad else 0)
