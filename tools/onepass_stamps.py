"""Where a tick of the one-pass kernel spends its cycles (diagnostic build: python -m nmfgpu_amd.build --diag).
usage: NMFAMD_LIBRARY=nmfgpu_amd/lib/libnmfgpu64_diag.so python tools/onepass_stamps.py [M N]"""
import os, sys
os.environ.setdefault("NMFAMD_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmfgpu_amd", "lib", "libnmfgpu64_diag.so"))  # (the one-pass kernel lives in the measurement build)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "onepass_stamps.bin")
os.environ["NMFAMD_ONEPASS_STAMPS"] = path
os.environ["NMFAMD_ONE_PASS"] = "1"
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
role = (np.arange(len(s)) % 8) // 4
ok = s[:, 11] > 0
print(f"waves {ok.sum()}  ticks per group {sorted(set(s[ok, 15].astype(int)))}  frobenius {f:.4f}")
for r, nm, names in ((0, "A waves", ["A", "exchange: wait free, write", "wait B waves", "wait exchange full", "publish", "-", "next panel requested", "first split (drain)"]),
                     (1, "B waves", ["owner: wait partials", "owner: new column", "H operand: wait + repack", "H H^T + B", "next denominator", "-", "-", "loads issued"])):
    q = s[ok & (role == r)]
    T = q[:, 15]
    clk = np.median((q[:, 10] + q[:, 11] + q[:, 12]) / (q[:, 13] * 10.0)) * 1000
    print(f"{nm}: kernel median {np.median(q[:, 13]) / 100:.1f} us per wave, clock ~{clk:.0f} MHz; before loop {np.median(q[:, 10]):.0f} cyc, loop {np.median(q[:, 11]):.0f}, after {np.median(q[:, 12]):.0f}")
    per = q[:, :8] / T[:, None]
    for i, nme in enumerate(names):
        if nme != "-":
            print(f"  {nme:36s} median {np.median(per[:, i]):8.0f} cyc/panel   p10 {np.percentile(per[:, i], 10):8.0f}  p90 {np.percentile(per[:, i], 90):8.0f}")
    print(f"  sum {np.median(per.sum(axis=1)):.0f} cyc/panel; retries per wave: H cols {np.median(q[:, 8]):.0f} (max {q[:, 8].max():.0f}), partials {np.median(q[:, 9]):.0f} (max {q[:, 9].max():.0f})")
