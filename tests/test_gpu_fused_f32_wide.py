"""The fused fp32 iteration at padded ranks 128 ... 512 (round 6: Engine::iterate_fused32w; csrc/kernels_wide.hip k_panel_update_wide_f32<.., FX>, k_gram_reduce_x3):
eight launches per iteration -- Gram slices, reduction + split image + pending scale, product, update, twice -- instead of the generic sequence's fourteen (no pack,
smoothing or normalisation launches).  Multiplicative update and nsNMF (ref source/nmf/AlgorithmMultiplicativeFrobenius.h:165-248,
source/nmf/AlgorithmNonSmoothNMF.h:174-218; the column scale of kernel::normalizeColumns, ref KernelNormalizeColumns.cu:37-58, carried as a pending factor).

Tolerance: the fp32 engine tests' 2e-4 on the factors against the fp64 oracle, 1e-5 on the reported error; against the generic sequence of the same library
(NMFAMD_NO_FUSED_MU=1) the two differ by fp32 rounding only."""
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


def run_engine(V, W, H, alg, iters, error_every=10, **kw):
    m, n = V.shape
    eng = na.Engine(m, n, W.shape[1], alg, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=error_every, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    return eng, Wg, Hg


@pytest.mark.parametrize("alg,m,n,r,kw", [
    ("mu", 900, 700, 128, {}),                        # padded rank 128, exactly
    ("mu", 1500, 610, 100, {}),                       # 128 with zero columns behind r
    ("nsnmf", 777, 333, 158, dict(theta=0.5)),        # 256 without bf16 operands
    ("nsnmf", 640, 900, 300, dict(theta=0.3)),        # 384
    ("mu", 410, 1300, 500, {}),                       # 512
    ("nsnmf", 70, 45, 65, dict(theta=0.8)),           # one row block and a half
    ("mu", 33000, 260, 100, {}),                      # panels of 32 768 rows and more (the generic sequence switches to the 128-row update there)
    ("nsnmf", 300, 32800, 140, dict(theta=0.6)),      # ... on the H side, rank 256
])
def test_fused_wide_iteration_against_the_oracle_and_the_generic_sequence(alg, m, n, r, kw, monkeypatch):
    iters = 12
    V, W, H = problem(m, n, r, seed=3 * r + m)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, **kw)
    eng, Wg, Hg = run_engine(V, W, H, alg, iters, error_every=4, **kw)
    g = eng.geometry()
    # (eight launches; seven or six where the Gram slices of a side ride its product launch as passenger workgroups)
    assert g["fused_launches"] == 8 - (g["gram_ride_slices_h"] > 0) - (g["gram_ride_slices_w"] > 0) and g["product_kernel"] == 2 and g["padded_rank"] == 128 * ((r + 127) // 128)
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4, (rel(Wg, W64), rel(Hg, H64))
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    monkeypatch.setenv("NMFAMD_NO_FUSED_MU", "1")
    gen, Wn, Hn = run_engine(V, W, H, alg, iters, error_every=4, **kw)
    assert gen.geometry()["fused_launches"] == 0
    assert rel(Wg, Wn) < 2e-5 and rel(Hg, Hn) < 2e-5, (rel(Wg, Wn), rel(Hg, Hn))
    assert eng.frobenius == pytest.approx(gen.frobenius, rel=1e-5)


def test_downloads_uploads_and_constant_basis_vectors_between_fused_wide_iterations():
    """get_factors folds the pending column scale into the panel (and drops the split image of the unnormalised panel); set_factors drops it; a run stepped one
    iteration at a time with a download after every step ends where the run in one piece ends; H-only iterations (W held) run the generic H step on the normalised W."""
    m, n, r, iters = 520, 310, 140, 8
    V, W, H = problem(m, n, r, seed=5)
    _, W1, H1 = run_engine(V, W, H, "nsnmf", iters, theta=0.4)
    eng = na.Engine(m, n, r, "nsnmf", theta=0.4)
    eng.upload(V); eng.set_factors(W, H)
    for k in range(1, iters + 1):
        eng.iterate(1, first_iteration=k, error_every=10, last_iteration=iters)
        Wk, Hk = eng.get_factors()
    assert rel(Wk, W1) < 2e-6 and rel(Hk, H1) < 2e-6
    eng.iterate(2, first_iteration=1, error_every=10, last_iteration=0)
    eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    W2, H2 = eng.get_factors()
    assert np.array_equal(W2, W1) and np.array_equal(H2, H1)
    eng.iterate(3, first_iteration=1, error_every=10, last_iteration=3, constant_w=True)
    W3, H3 = eng.get_factors()
    assert rel(W3, W2) < 1e-6 and np.isfinite(H3).all() and eng.frobenius > 0


def test_repeated_fused_wide_runs_are_bit_identical():
    m, n, r, iters = 800, 500, 200, 6
    V, W, H = problem(m, n, r, seed=9)
    _, Wa, Ha = run_engine(V, W, H, "mu", iters)
    for _ in range(2):
        _, Wb, Hb = run_engine(V, W, H, "mu", iters)
        assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)


@pytest.mark.parametrize("ride", ["0", "1"])
@pytest.mark.parametrize("alg,m,n,r,kw", [("mu", 1500, 610, 100, {}), ("nsnmf", 2100, 900, 158, dict(theta=0.5)), ("mu", 410, 1300, 500, {})])
def test_gram_slices_as_passengers_and_as_a_launch_give_the_same_bits(alg, m, n, r, kw, ride, monkeypatch, diag_build):
    """The Gram slices of the wide fp32 iteration run as passenger workgroups of the product launch where its first round leaves CUs free (csrc/gram_wide.h), as a
    launch of their own otherwise.  The slice count differs between the two forms, so G differs by rounding; both forms against the oracle at the fp32 tolerance, and
    the form the measurement build is forced into is the one reported."""
    iters = 10
    V, W, H = problem(m, n, r, seed=r + m)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, **kw)
    monkeypatch.setenv("NMFAMD_F32W_RIDE", ride)
    eng, Wg, Hg = run_engine(V, W, H, alg, iters, error_every=5, **kw)
    g = eng.geometry()
    if ride == "0":
        assert g["fused_launches"] == 8 and g["gram_ride_slices_h"] == 0 and g["gram_ride_slices_w"] == 0
    else:
        assert g["gram_ride_slices_h"] > 0 and g["gram_ride_slices_w"] > 0 and g["fused_launches"] == 6
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4, (rel(Wg, W64), rel(Hg, H64))
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
