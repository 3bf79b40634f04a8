"""usage (GPU box): python tools/fuzz_fused.py [CASES] [SEED]
Random shapes through the fused launch sequences (fp32 rank <= 64 MU / nsNMF, fp32 wide ranks, double precision) against the generic sequence of the same library
(NMFAMD_NO_FUSED_MU=1): factors and the reported error must agree to rounding (fp32 2e-5 / 1e-5, fp64 1e-11 / 1e-11).  Prints one line per case and the worst
differences; exit code 1 on a disagreement.  A robustness sweep beside the parity tests (tests/test_gpu_fused_*.py hold the fixed cases)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nmfgpu_amd as na  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300))


def run(V, W, H, alg, dt, iters, theta, fused, downloads):
    if fused:
        os.environ.pop("NMFAMD_NO_FUSED_MU", None)
    else:
        os.environ["NMFAMD_NO_FUSED_MU"] = "1"
    m, n = V.shape
    kw = dict(theta=theta) if alg == "nsnmf" else {}
    eng = na.Engine(m, n, W.shape[1], alg, dtype=dt, **kw)
    eng.upload(V); eng.set_factors(W, H)
    if downloads:
        for k in range(1, iters + 1):
            eng.iterate(1, first_iteration=k, error_every=3, last_iteration=iters)
            if k % 2 == 0:
                eng.get_factors()
    else:
        eng.iterate(iters, first_iteration=1, error_every=3, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    g = eng.geometry()
    return Wg, Hg, eng.frobenius, g["fused_launches"]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst = 0.0
    bad = 0
    for c in range(cases):
        dt = np.float64 if rng.random() < 0.35 else np.float32
        alg = "nsnmf" if rng.random() < 0.5 else "mu"
        kind = rng.integers(0, 4)
        m = int(rng.integers(40, 5000)); n = int(rng.integers(40, 5000))
        if kind == 0:
            m, n = int(rng.integers(20000, 40000)), int(rng.integers(60, 400))      # long and narrow
        elif kind == 1:
            m, n = int(rng.integers(60, 400)), int(rng.integers(20000, 40000))
        r = int(rng.integers(2, min(m, n, 330) + 1))
        theta = float(rng.random())
        iters = int(rng.integers(3, 9))
        downloads = bool(rng.random() < 0.3)
        V = np.asfortranarray(rng.random((m, n)).astype(dt))
        W = np.asfortranarray((1.0 - rng.random((m, r))).astype(dt))
        H = np.asfortranarray((1.0 - rng.random((r, n))).astype(dt))
        Wf, Hf, ff, launches = run(V, W, H, alg, dt, iters, theta, True, downloads)
        Wg, Hg, fg, _ = run(V, W, H, alg, dt, iters, theta, False, False)
        dw, dh, df = rel(Wf, Wg), rel(Hf, Hg), abs(ff - fg) / max(abs(fg), 1e-300)
        tol = (1e-11, 1e-11) if dt == np.float64 else (2e-5, 1e-5)
        ok = dw < tol[0] and dh < tol[0] and df < tol[1] and np.isfinite(Wf).all() and np.isfinite(Hf).all()
        worst = max(worst, (dw + dh) / (tol[0]))
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} {alg:5s} {np.dtype(dt).name:7s} {m:6d} x {n:6d} r = {r:3d} theta {theta:.2f} iters {iters} downloads {int(downloads)} fused launches {launches}: "
              f"dW {dw:.2e} dH {dh:.2e} dE {df:.2e}", flush=True)
    print(f"{cases - bad} of {cases} agree; worst (dW + dH) / tolerance = {worst:.3f}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
