"""Mapping probe of the one-pass kernel: W = [I; 0], H = 1, V(i, j) = i + j / 1024 for i < 64  =>  after one iteration H(c, j) = V(c, j)."""
import os, sys
os.environ.setdefault("NMFAMD_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmfgpu_amd", "lib", "libnmfgpu64_diag.so"))  # (the one-pass kernel lives in the measurement build)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NMFAMD_ONE_PASS"] = "1"
import nmfgpu_amd as na
m, n, r = 10000, 5000, 64
V = np.zeros((m, n), dtype=np.float32, order="F")
V[:256, :] = (np.arange(256, dtype=np.float32)[:, None] + 1.0) + np.arange(n, dtype=np.float32)[None, :] / 8192.0
V[256:, :] = 0.5
W = np.zeros((m, r), dtype=np.float32, order="F"); W[4 * np.arange(64), np.arange(64)] = 1.0
H = np.ones((r, n), dtype=np.float32, order="F")
eng = na.Engine(m, n, r, "mu")
print(eng.geometry())
eng.upload(V); eng.set_factors(W, H)
eng.iterate(1, first_iteration=1, last_iteration=0); eng.synchronize()
W1, H1 = eng.get_factors()
exp = V[0:256:4, :]
err = np.abs(H1 - exp)
print("max err", err.max())
np.set_printoptions(linewidth=200, precision=3, suppress=True)
print("H1[:16, :4]\n", H1[:16, :4])
print("expected\n", exp[:16, :4])
bad = np.argwhere(err > 1e-3)
print("bad entries", len(bad), bad[:20])
# which source value did each wrong entry get?  (value = c' + 1 + j' / 8192)
if len(bad):
    for c, j in bad[:24]:
        v = H1[c, j]; cs = int(np.floor(v + 1e-4)) - 1; js = (v - np.floor(v + 1e-4)) * 8192
        print(f"H({c},{j}) = {v:.5f}  ~ V({cs}, {js:.1f})")
