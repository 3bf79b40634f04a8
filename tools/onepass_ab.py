"""One-pass iteration (kernels_onepass.hip) against the two-pass iteration on the same inputs: agreement of the factors and of
the reported error after ITERS iterations, and time per iteration of both.
usage: onepass_ab.py [M N R ITERS]"""
import os, sys, time
os.environ.setdefault("NMFAMD_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmfgpu_amd", "lib", "libnmfgpu64_diag.so"))  # (the one-pass kernel lives in the measurement build)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na

m, n, r = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10000, 5000, 64)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rs = np.random.RandomState(1)
V = np.empty((m, n), dtype=np.float32, order="F")
for j0 in range(0, n, 512):
    V[:, j0:j0 + 512] = rs.random_sample((m, min(512, n - j0))).astype(np.float32)
W0 = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((r, m))).astype(np.float32).T)
H0 = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((n, r))).astype(np.float32).T)


def run(two_pass: bool, timed: int = 200):
    if two_pass:
        os.environ.pop("NMFAMD_ONE_PASS", None)
    else:
        os.environ["NMFAMD_ONE_PASS"] = "1"
    eng = na.Engine(m, n, r, "mu")
    g = eng.geometry()
    eng.upload(V); eng.set_factors(W0, H0)
    eng.iterate(iters, first_iteration=1, last_iteration=iters); eng.synchronize()
    frob = eng.frobenius
    W, H = eng.get_factors()
    eng.iterate(10, first_iteration=iters + 1); eng.synchronize()
    t0 = time.perf_counter(); eng.iterate(timed, first_iteration=iters + 11); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"{'two-pass' if two_pass else 'one-pass'}: one_pass={g['one_pass']} images={g['resident_images']}  frobenius {frob:.6f}  "
          f"{dt / timed * 1e6:.1f} us/iteration  {timed / dt:.0f} it/s  (one_pass after: {eng.geometry()['one_pass']})", flush=True)
    eng.close()
    return W, H, frob


W1, H1, f1 = run(False)
W2, H2, f2 = run(True)
rw = np.linalg.norm(W1 - W2) / np.linalg.norm(W2)
rh = np.linalg.norm(H1 - H2) / np.linalg.norm(H2)
print(f"relative difference after {iters} iterations: W {rw:.3e}  H {rh:.3e}  frobenius {abs(f1 - f2) / f2:.3e}")
print("max |dW|", float(np.abs(W1 - W2).max()), "max |dH|", float(np.abs(H1 - H2).max()), "finite", bool(np.isfinite(W1).all() and np.isfinite(H1).all()))
if not (rw < 1e-5 and rh < 1e-5):
    # where do they differ?  (columns of H by panel of 32, rows of W by tile of 16)
    rel = np.abs(H1 - H2) / (np.abs(H2) + 1e-30)
    print("H rel err by column mod 32:", np.round(rel.max(axis=0)[: (n // 32) * 32].reshape(-1, 32).max(axis=0), 4))
    print("H rel err by factor row c:", np.round(rel.max(axis=1), 4))
    print("H rel err by panel (first 40):", np.round(rel.max(axis=0)[: (n // 32) * 32].reshape(-1, 32).max(axis=1)[:40], 4))
    relw = np.abs(W1 - W2) / (np.abs(W2) + 1e-30)
    print("W rel err by row mod 16:", np.round(relw.max(axis=1)[: (m // 16) * 16].reshape(-1, 16).max(axis=0), 4))
    print("W rel err by tile of 16 rows (first 48):", np.round(relw.max(axis=1)[: (m // 16) * 16].reshape(-1, 16).max(axis=1)[:48], 4))
    print("W rel err by c:", np.round(relw.max(axis=0), 4))
    dh = np.abs(H1 - H2).max(axis=0)
    bad = np.nonzero(dh > 1e-4 * np.abs(H2).max())[0]
    print("H columns off:", len(bad), bad[:40])
    dw = np.abs(W1 - W2).max(axis=1)
    badw = np.nonzero(dw > 1e-4 * np.abs(W2).max())[0]
    print("W rows off:", len(badw), badw[:40])
    sys.exit(1)
