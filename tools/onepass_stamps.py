"""Where a tick of the one-pass kernel spends its cycles (diagnostic build: python -m nmfgpu_amd.build --diag).
usage: NMFAMD_LIBRARY=nmfgpu_amd/lib/libnmfgpu64_diag.so python tools/onepass_stamps.py [M N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "onepass_stamps.bin")
os.environ["NMFAMD_ONEPASS_STAMPS"] = path
import nmfgpu_amd as na
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 5000)
r = 64
rs = np.random.RandomState(1)
V = np.asfortranarray(rs.random_sample((m, n)).astype(np.float32))
W0 = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((r, m))).astype(np.float32).T)
H0 = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((n, r))).astype(np.float32).T)
eng = na.Engine(m, n, r, "mu")
eng.upload(V); eng.set_factors(W0, H0)
eng.iterate(30, first_iteration=1, last_iteration=30); eng.synchronize()
f = eng.frobenius
eng.get_factors()
s = np.fromfile(path, dtype=np.uint64).reshape(-1, 16).astype(np.float64)
s = s[s[:, 11] > 0]
T = s[:, 15]
names = ["A", "wait BAR_a", "book+publish", "wait H cols", "B", "owner ld/panel->LDS/prefetch/wait", "wait BAR_b", "owner math"]
print(f"waves {len(s)}  ticks per group {sorted(set(T.astype(int)))}  frobenius {f:.4f}")
clk = np.median((s[:, 10] + s[:, 11] + s[:, 12]) / (s[:, 13] * 10.0)) * 1000
print(f"kernel: median {np.median(s[:, 13]) / 100:.1f} us per wave, clock ~{clk:.0f} MHz; before loop {np.median(s[:, 10]):.0f} cyc, loop {np.median(s[:, 11]):.0f}, after {np.median(s[:, 12]):.0f}")
per = s[:, :8] / (T[:, None] + 3)
for i, nme in enumerate(names):
    print(f"  {nme:36s} median {np.median(per[:, i]):8.0f} cyc/tick   p10 {np.percentile(per[:, i], 10):8.0f}  p90 {np.percentile(per[:, i], 90):8.0f}")
print(f"  sum {np.median(per.sum(axis=1)):.0f} cyc/tick; retries per wave: H cols {np.median(s[:, 8]):.0f} (max {s[:, 8].max():.0f}), partials {np.median(s[:, 9]):.0f} (max {s[:, 9].max():.0f})")
for w in range(4):
    sel = s[w::4]
    print(f"  wave {w}: " + "  ".join(f"{np.median(sel[:, i] / (sel[:, 15] + 3)):7.0f}" for i in range(8)))
