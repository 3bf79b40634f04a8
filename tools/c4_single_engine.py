import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import nmfgpu_amd as na
m, n, r = 50000, 6250, 256
rs = np.random.RandomState(1)
V = np.asfortranarray(rs.random_sample((m, n)).astype(np.float32))
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((m, r))).astype(np.float32))
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((r, n))).astype(np.float32))
e = na.Engine(m, n, r, "nsnmf", theta=0.5, precision="bf16")
e.upload(V); e.set_factors(W, H)
e.iterate(80, first_iteration=1, error_every=10); e.synchronize()
t0 = time.perf_counter(); e.iterate(150, first_iteration=81, error_every=10); e.synchronize()
print(f"{os.environ.get('NMFAMD_TRI_FUSE_W', 'default')}: {(time.perf_counter() - t0) / 150 * 1e6:.1f} us/iteration, frobenius {e.frobenius:.4f}")
