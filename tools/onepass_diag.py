"""Which old column did a wrong new column of H come from?  (one iteration, one-pass vs two-pass)"""
import os, sys
os.environ.setdefault("NMFAMD_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmfgpu_amd", "lib", "libnmfgpu64_diag.so"))  # (the one-pass kernel lives in the measurement build)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
m, n, r = 10000, 5000, 64
rs = np.random.RandomState(1)
V = np.asfortranarray(rs.random_sample((m, n)).astype(np.float32))
W0 = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((r, m))).astype(np.float32).T)
H0 = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((n, r))).astype(np.float32).T)
def run(two):
    if two: os.environ.pop("NMFAMD_ONE_PASS", None)
    else: os.environ["NMFAMD_ONE_PASS"] = "1"
    e = na.Engine(m, n, r, "mu"); e.upload(V); e.set_factors(W0, H0); e.iterate(1, first_iteration=1, last_iteration=0); e.synchronize()
    W, H = e.get_factors(); e.close(); return W, H
W1, H1 = run(False); W2, H2 = run(True)
G = (W0.astype(np.float64).T @ W0.astype(np.float64))
eps = np.finfo(np.float32).eps
den0 = G @ H0.astype(np.float64) + eps
num = H2.astype(np.float64) * den0 / H0                      # numerators implied by the two-pass result
bad = np.nonzero(np.abs(H1 - H2).max(axis=0) > 1e-4 * np.abs(H2).max())[0]
print("bad columns", len(bad), bad[:30])
for j in bad[:12]:
    # candidates: old column j' of the same slot in other panels, or the NEW column j (updated twice)
    best = None
    for jp in list(range(j % 32, n, 32)):
        cand = H0[:, jp].astype(np.float64) * num[:, j] / (G @ H0[:, jp].astype(np.float64) + eps)
        err = np.abs(cand - H1[:, j]).max() / np.abs(H1[:, j]).max()
        if best is None or err < best[0]: best = (err, jp, "old column")
    cand = H2[:, j].astype(np.float64) * num[:, j] / (G @ H2[:, j].astype(np.float64) + eps)
    err = np.abs(cand - H1[:, j]).max() / np.abs(H1[:, j]).max()
    if err < best[0]: best = (err, j, "NEW column (updated twice)")
    # mixed: hcur from one, den from another?
    print(f"column {j} (panel {j // 32}, slot {j % 32}): best explanation {best[2]} {best[1]} (panel {best[1] // 32}) residual {best[0]:.2e};  rows wrong: {np.nonzero(np.abs(H1[:, j] - H2[:, j]) > 1e-4 * np.abs(H2[:, j]).max())[0][:20]}")
