"""nsNMF in float at ranks up to 64 on the four-launch iteration (round 6: csrc/kernels_mu64.hip k_mu64_update32<.., NS>; Engine::iterate_mu64): the smoothing matrix
S = (1 - theta) I + theta / r 1 1^T goes AROUND the H update's r x r product -- the products against V run on the unsmoothed W image and on the image of S H that the
H update leaves -- instead of over the panels (ref source/nmf/AlgorithmNonSmoothNMF.h:174-218: W S and S H as gemms of their own; the error's
tr((S H)(S H)^T W^T W) with the unsmoothed W^T W, :201-202).  The generic sequence of the same library (NMFAMD_NO_FUSED_MU=1: 14.4 launches) is the cross-check.

Tolerance: the fp32 engine tests' 2e-4 on the factors against the fp64 oracle, 1e-5 on the reported error; 2e-5 against the generic sequence (fp32 rounding only)."""
import numpy as np
import pytest

import nmfgpu_amd as na
from oracle import oracle

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(a)


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300))


def problem(m, n, r, seed=1):
    rng = np.random.default_rng(seed)
    return F(rng.random((m, n)).astype(np.float32)), F((1.0 - rng.random((m, r))).astype(np.float32)), F((1.0 - rng.random((r, n))).astype(np.float32))


def run_engine(V, W, H, iters, theta, error_every=10, **kw):
    m, n = V.shape
    eng = na.Engine(m, n, W.shape[1], "nsnmf", theta=theta, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=error_every, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    return eng, Wg, Hg


@pytest.mark.parametrize("m,n,r,theta", [
    (900, 700, 64, 0.5),          # the padded rank exactly
    (1500, 610, 40, 0.3),         # zero rows behind r in the 64-row panels
    (777, 333, 33, 0.9),
    (200, 150, 5, 0.8),           # a problem the plan keeps on the native-fp32 products: the generic sequence (no split images to smooth around)
    (2600, 5000, 64, 0.5),        # several split-K slabs on either side
    (3000, 2100, 17, 0.0),        # S = I; rank <= 32 on the native-fp32 products
    (1100, 900, 57, 0.0),         # S = I on the fused path
    (300, 40000, 48, 0.6),        # a long H panel, a short W panel
])
def test_ns64_iteration_against_the_oracle_and_the_generic_sequence(m, n, r, theta, monkeypatch):
    iters = 12
    V, W, H = problem(m, n, r, seed=7 * r + m)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("nsnmf", V64, W64, H64, iters, theta=theta)
    eng, Wg, Hg = run_engine(V, W, H, iters, theta, error_every=4)
    g = eng.geometry()
    assert g["padded_rank"] == 64 and g["fused_launches"] == (4 if g["product_kernel"] == 2 else 0)
    assert g["fused_launches"] == 4 or r <= 32       # (ranks up to 32 outside the cache window: one 32-column block, native fp32 products -- the plan's choice, engine.cpp)
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4, (rel(Wg, W64), rel(Hg, H64))
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    monkeypatch.setenv("NMFAMD_NO_FUSED_MU", "1")
    gen, Wn, Hn = run_engine(V, W, H, iters, theta, error_every=4)
    assert gen.geometry()["fused_launches"] == 0
    assert rel(Wg, Wn) < 2e-5 and rel(Hg, Hn) < 2e-5, (rel(Wg, Wn), rel(Hg, Hn))
    assert eng.frobenius == pytest.approx(gen.frobenius, rel=1e-5)


def test_ns64_with_theta_zero_is_the_multiplicative_update_bit_for_bit():
    """S = I: every sum the smoothing adds is multiplied by 0 and every value by 1 -- the factors must come out as the multiplicative update's (the same launches, the same
    order of additions), and get_factors' W S is W."""
    m, n, r, iters = 830, 520, 50, 15
    V, W, H = problem(m, n, r, seed=11)
    _, Wn, Hn = run_engine(V, W, H, iters, 0.0)
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wm, Hm = eng.get_factors()
    assert np.array_equal(Hn, Hm)
    assert rel(Wn, Wm) < 1e-6            # (W S through the smoothing kernel: (1 - 0 + 0) w + 0 * sum)


def test_downloads_uploads_and_constant_basis_vectors_between_ns64_iterations(monkeypatch):
    """A run stepped one iteration at a time with a download after every step ends where the run in one piece ends; set_factors drops every pending state; iterations
    with W held constant (H only) on the fused H step, error terms included, match the generic sequence's."""
    m, n, r, iters, theta = 520, 310, 36, 8, 0.4
    V, W, H = problem(m, n, r, seed=5)
    _, W1, H1 = run_engine(V, W, H, iters, theta)
    eng = na.Engine(m, n, r, "nsnmf", theta=theta)
    eng.upload(V); eng.set_factors(W, H)
    for k in range(1, iters + 1):
        eng.iterate(1, first_iteration=k, error_every=10, last_iteration=iters)
        Wk, Hk = eng.get_factors()
    assert rel(Wk, W1) < 2e-6 and rel(Hk, H1) < 2e-6
    eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    W2, H2 = eng.get_factors()
    assert np.array_equal(W2, W1) and np.array_equal(H2, H1)
    # W held constant behind three full iterations (a pending column scale in force): H-only steps with the error term on the last one
    def held(e):
        e.upload(V); e.set_factors(W, H)
        e.iterate(3, first_iteration=1, error_every=10, last_iteration=0)
        e.iterate(5, first_iteration=4, error_every=4, last_iteration=8, constant_w=True)
        return e.get_factors() + (e.frobenius,)
    Wf, Hf, ff = held(na.Engine(m, n, r, "nsnmf", theta=theta))
    monkeypatch.setenv("NMFAMD_NO_FUSED_MU", "1")
    Wc, Hc, fc = held(na.Engine(m, n, r, "nsnmf", theta=theta))
    assert rel(Wf, Wc) < 2e-5 and rel(Hf, Hc) < 2e-5 and ff == pytest.approx(fc, rel=1e-5)


def test_ns64_through_compute_matches_the_oracle_with_the_threshold_bookkeeping():
    m, n, r, iters, theta = 700, 530, 40, 30, 0.4
    V, W0, H0 = problem(m, n, r, seed=9)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W0, H0))
    ref = oracle.run("nsnmf", V64, W64, H64, iters, theta=theta)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    s = na.Summary()
    assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
    try:
        assert na.compute(V, W, H, algorithm=na.NmfAlgorithm.nsNMF, iterations=iters, parameters={"theta": theta}, summary=s) == na.ResultType.Success
    finally:
        na.finalize()
    assert rel(W, W64) < 2e-4 and rel(H, H64) < 2e-4
    assert s.record(0).frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    assert s.record(0).numIterations == iters


def test_ns64_full_size_properties():
    """Config 2's shape with nsNMF (the oracle does not finish this in seconds): bit-identical repeats, the error decreases, W S comes back with unit-norm columns of W
    behind it (column sums of squares of W S within the bound S allows), nothing negative, nothing non-finite."""
    m, n, r, theta = 10000, 5000, 64, 0.5
    V, W, H = problem(m, n, r, seed=2)
    eng, Wa, Ha = run_engine(V, W, H, 20, theta, error_every=10)
    f20 = eng.frobenius
    eng2, Wb, Hb = run_engine(V, W, H, 20, theta, error_every=10)
    assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb) and f20 == eng2.frobenius
    eng3, _, _ = run_engine(V, W, H, 10, theta, error_every=10)
    assert f20 < eng3.frobenius
    assert np.isfinite(Wa).all() and np.isfinite(Ha).all() and (Wa >= 0).all() and (Ha >= 0).all()
    assert eng.geometry()["fused_launches"] == 4
