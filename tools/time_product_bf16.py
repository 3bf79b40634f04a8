"""Times the two bf16 factor products of one shape through the engine's kernel-event brackets.
usage: time_product_bf16.py M N R"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
m, n, r = (int(x) for x in sys.argv[1:4])
rs = np.random.RandomState(1)
V = np.empty((m, n), dtype=np.float32, order="F")
for j0 in range(0, n, 512):
    V[:, j0:j0 + 512] = rs.random_sample((m, min(512, n - j0))).astype(np.float32)
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((r, m))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((n, r))).astype(np.float32).T)
eng = na.Engine(m, n, r, "nsnmf", precision="bf16", theta=0.5)
eng.upload(V); eng.set_factors(W, H)
eng.iterate(3, error_every=0); eng.synchronize()
res = []
for which in ("h", "w"):
    # one product per timed "iteration": the sharded entry points run the H step / the W products alone
    ex = torch.zeros(eng.geometry()["exchange_count"], dtype=torch.float32, device="cuda")
    eng.kernel_timing(1)
    for _ in range(6):
        if which == "h":
            eng.h_step(False)
        else:
            eng.h_step(False) if False else None
            eng.w_products(ex.data_ptr())
    eng.synchronize()
    ms, cnt = eng.kernel_timing_read()
    res.append(f"K_{which.upper()} {ms / max(cnt, 1) * 1e3:.1f} us ({cnt})")
print(f"diag={os.environ.get('NMFAMD_BF_DIAG', '0')}: " + "  ".join(res))
