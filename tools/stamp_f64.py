"""usage: NMFAMD_LIBRARY=.../libnmfgpu64_diag.so python tools/stamp_f64.py [m n r alg]  (measurement build)
Wall-clock stamps (100 MHz) of the LAST iteration of the four-launch double-precision iteration: per launch, when its workgroups entered and left and where the
time between went.  Launch 0 / 2: the two product launches (product workgroups: entry, loop end, exit; Gram passengers: entry, partial block stored, counted in,
level-1 reduction done).  Launch 1 / 3: the H and the W update (wide kernel: entry, slabs summed, H-side preparation, MFMA loop end, epilogue, exit)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(tempfile.gettempdir(), "f64_stamps.bin")
os.environ["NMFAMD_F64_STAMPS"] = path
import nmfgpu_amd as na  # noqa: E402

m, n, r = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 165, 158)
alg = sys.argv[4] if len(sys.argv) > 4 else "nsnmf"
rng = np.random.default_rng(1)
V = np.asfortranarray(rng.random((m, n)))
W = np.asfortranarray(1.0 - rng.random((m, r)))
H = np.asfortranarray(1.0 - rng.random((r, n)))
kw = dict(theta=0.5) if alg == "nsnmf" else {}
eng = na.Engine(m, n, r, alg, dtype=np.float64, **kw)
eng.upload(V); eng.set_factors(W, H)
eng.iterate(300, first_iteration=1, error_every=0, last_iteration=0)
eng.synchronize()
eng.close() if hasattr(eng, "close") else None
del eng
st = np.fromfile(path, dtype=np.uint64).reshape(4, 4096, 8).astype(np.int64)
t0 = st[st > 0].min()
us = lambda v: (v - t0) / 100.0


def show(title, rows, names):
    rows = rows[rows[:, 0] > 0]
    if len(rows) == 0:
        return
    print(f"  {title}: {len(rows)} workgroups")
    for k, nm in enumerate(names):
        col = rows[:, k]
        col = col[col > 0]
        if len(col):
            print(f"    {nm:28s} min {us(col.min()):7.2f}  median {us(np.median(col)):7.2f}  max {us(col.max()):7.2f} us")


for L, name in enumerate(["W^T V + W^T W passengers", "H update", "V H^T + H H^T passengers", "W update"]):
    print(f"launch {L}: {name}")
    if L in (0, 2):
        show("product workgroups", st[L, :2048], ["entry", "loop end", "exit", "round 0 in LDS", "round 1 in LDS", "round 2 in LDS", "round 3 in LDS"])
        show("Gram passengers", st[L, 2048:], ["entry", "partial block stored", "counted in", "level-1 reduction done"])
    else:
        show("update workgroups", st[L], ["entry", "slabs summed", "prepared", "MFMA loop end", "epilogue", "exit"])
