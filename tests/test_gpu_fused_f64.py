"""The four-launch double-precision iteration (round 6: Engine::iterate_fused64, csrc/kernels_f64.hip gram_ride_f64 / PanelFusedF64) -- the path the reference's own
callers take (ref example/main.cpp:78-130: NmfDescription<double>, nsNMF, 4096 x 165, r = 158; the R binding is double by nature).

Reference semantics: source/nmf/AlgorithmNonSmoothNMF.h:174-218, source/nmf/AlgorithmMultiplicativeFrobenius.h:165-248,
source/nmf/KernelNormalizeColumns.cu:37-58 (the column scale carried as a pending factor).  Tolerance: 1e-9 relative on the factors and on the reported error
against the fp64 oracle (the tolerance of every fp64 engine test).
"""
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
    return F(rng.random((m, n))), F(1.0 - rng.random((m, r))), F(1.0 - rng.random((r, n)))


def run_engine(V, W, H, alg, iters, error_every=10, **kw):
    m, n = V.shape
    eng = na.Engine(m, n, W.shape[1], alg, dtype=np.float64, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=error_every, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    return eng, Wg, Hg


def test_reference_example_shape_200_iterations_against_the_fp64_oracle():
    """The reference example's exact shape and algorithm (ref example/main.cpp:30-32,78-130): 4096 x 165, r = 158, nsNMF theta = 0.5, double, 200 iterations."""
    m, n, r, iters = 4096, 165, 158, 200
    V, W, H = problem(m, n, r, seed=158)
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    ref = oracle.run("nsnmf", V, Wo, Ho, iters, theta=0.5)
    eng, Wg, Hg = run_engine(V, W, H, "nsnmf", iters, theta=0.5)
    g = eng.geometry()
    assert g["padded_rank"] == 192 and g["fused_launches"] == 4 and g["product_kernel"] == 3
    assert rel(Wg, Wo) < 1e-9 and rel(Hg, Ho) < 1e-9, (rel(Wg, Wo), rel(Hg, Ho))
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-9)
    assert eng.rmsd == pytest.approx(ref["frobenius"] / np.sqrt(m * n), rel=1e-9)


@pytest.mark.parametrize("alg,m,n,r,kw", [
    ("mu", 200, 150, 7, {}),                          # padded rank 64, 32-column product form
    ("mu", 1000, 700, 64, {}),                        # padded rank 64, full width
    ("nsnmf", 640, 333, 40, dict(theta=0.3)),         # padded rank 64, smoothing
    ("mu", 700, 330, 129, {}),                        # 192
    ("nsnmf", 700, 610, 158, dict(theta=0.5)),        # 192, the example's rank
    ("nsnmf", 513, 257, 300, dict(theta=0.7)),        # 320: five super-block columns, fifteen super-blocks
    ("mu", 300, 1300, 500, {}),                       # 512: the widest panel the kernels cover
    ("nsnmf", 37, 5, 3, dict(theta=0.9)),             # tiny: slices with no K-steps at all
    ("mu", 3, 90, 2, {}),
])
def test_fused_iteration_against_the_oracle_and_the_generic_sequence(alg, m, n, r, kw, monkeypatch):
    """Against the fp64 oracle (1e-9, error terms on iterations 4, 8, 12 and the last), and against the generic launch sequence of the same library
    (NMFAMD_NO_FUSED_MU=1: Gram + reduction, smoothing and normalisation as launches of their own) -- the two differ by rounding only."""
    iters = 13
    V, W, H = problem(m, n, r, seed=7 * r + m)
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    ref = oracle.run(alg, V, Wo, Ho, iters, **kw)
    eng, Wg, Hg = run_engine(V, W, H, alg, iters, error_every=4, **kw)
    assert eng.geometry()["fused_launches"] == 4
    assert eng.geometry()["gram_ride_slices_h"] >= 1 and eng.geometry()["gram_ride_slices_w"] >= 1
    assert rel(Wg, Wo) < 1e-9 and rel(Hg, Ho) < 1e-9, (rel(Wg, Wo), rel(Hg, Ho))
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-9)
    monkeypatch.setenv("NMFAMD_NO_FUSED_MU", "1")
    gen, Wn, Hn = run_engine(V, W, H, alg, iters, error_every=4, **kw)
    assert gen.geometry()["fused_launches"] == 0
    assert rel(Wg, Wn) < 1e-11 and rel(Hg, Hn) < 1e-11, (rel(Wg, Wn), rel(Hg, Hn))
    assert eng.frobenius == pytest.approx(gen.frobenius, rel=1e-11)


def test_factors_can_be_read_and_set_between_fused_iterations():
    """get_factors folds the pending column scale into the panel (the engine goes on from a normalised W); set_factors drops it; a run stepped one iteration at a
    time with a download after every step ends where the run in one piece ends, and the columns of W are unit vectors at every step."""
    m, n, r, iters = 520, 310, 70, 9
    V, W, H = problem(m, n, r, seed=3)
    _, W1, H1 = run_engine(V, W, H, "mu", iters)
    eng = na.Engine(m, n, r, "mu", dtype=np.float64)
    eng.upload(V); eng.set_factors(W, H)
    for k in range(1, iters + 1):
        eng.iterate(1, first_iteration=k, error_every=10, last_iteration=iters)
        Wk, Hk = eng.get_factors()
        np.testing.assert_allclose(np.linalg.norm(Wk, axis=0), 1.0, rtol=1e-12)
    assert rel(Wk, W1) < 1e-12 and rel(Hk, H1) < 1e-12
    # set_factors in the middle of a run: the pending scale of the old W must not leak into the new one
    eng.iterate(2, first_iteration=1, error_every=10, last_iteration=0)
    eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    W2, H2 = eng.get_factors()
    assert np.array_equal(W2, W1) and np.array_equal(H2, H1)


def test_a_zero_column_of_w_keeps_scale_one_and_constant_basis_vectors_take_the_generic_path():
    """kernel::normalizeColumns' guard (ref KernelNormalizeColumns.cu:54-58: sum > 0): a column of W that is all zero stays zero, its pending scale is 1;
    useConstantBasisVectors (W fixed) runs the generic H step on the materialised W and agrees with the oracle."""
    m, n, r, iters = 260, 140, 9, 6
    V, W, H = problem(m, n, r, seed=11)
    W[:, 4] = 0.0
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    oracle.run("mu", V, Wo, Ho, iters)
    _, Wg, Hg = run_engine(V, W, H, "mu", iters)
    assert np.all(Wg[:, 4] == 0.0) and np.isfinite(Wg).all() and np.isfinite(Hg).all()
    assert rel(Wg, Wo) < 1e-9 and rel(Hg, Ho) < 1e-9
    # three fused iterations, then H-only iterations with W held: the pending scale is folded in first
    eng = na.Engine(m, n, r, "nsnmf", dtype=np.float64, theta=0.4)
    V2, W2, H2 = problem(m, n, r, seed=12)
    eng.upload(V2); eng.set_factors(W2, H2)
    eng.iterate(3, first_iteration=1, error_every=10, last_iteration=0)
    Wa, Ha = eng.get_factors()          # (W S: nsNMF returns the smoothed basis, ref AlgorithmNonSmoothNMF.h:221-225)
    Wr, Hr = W2.copy(order="F"), H2.copy(order="F")
    oracle.run("nsnmf", V2, Wr, Hr, 3, theta=0.4)
    assert rel(Wa, Wr) < 1e-9 and rel(Ha, Hr) < 1e-9
    eng.iterate(4, first_iteration=1, error_every=10, last_iteration=4, constant_w=True)
    Wb, Hb = eng.get_factors()
    assert rel(Wb, Wa) < 1e-14
    assert eng.frobenius > 0 and np.isfinite(Hb).all()


def test_repeated_runs_are_bit_identical():
    """The passengers' partial blocks are added in slice order by whoever arrives last: the same bits every run."""
    m, n, r, iters = 900, 480, 158, 7
    V, W, H = problem(m, n, r, seed=21)
    _, Wa, Ha = run_engine(V, W, H, "nsnmf", iters, theta=0.5)
    for _ in range(3):
        _, Wb, Hb = run_engine(V, W, H, "nsnmf", iters, theta=0.5)
        assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)


def test_compute_in_double_with_the_next_launch_enqueued_ahead():
    """nmfgpu::compute's loop enqueues the next iteration's first launch before it waits for an error value (abi.cpp; round 6: also for the double-precision fused
    iteration).  Runs that end on their last iteration, on the threshold mid-way, and right behind an error iteration return the factors of an engine stepped one
    iteration at a time; the reported error is the oracle's."""
    m, n, r = 700, 260, 90
    V, W, H = problem(m, n, r, seed=41)
    assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
    na.set_verbosity(na.Verbosity.Nothing)
    for iters, threshold in ((37, 0.0), (40, 0.0), (200, 1e9)):      # (1e9: the threshold test passes at the second error iteration -- the launch enqueued ahead there is wasted)
        Wc, Hc = W.copy(order="F"), H.copy(order="F")
        s = na.Summary()
        assert na.compute(V, Wc, Hc, algorithm=na.NmfAlgorithm.nsNMF, iterations=iters, summary=s, parameters={"theta": 0.5},
                          threshold=threshold) == na.ResultType.Success
        done = s.record(0).numIterations
        assert done == (iters if threshold == 0.0 else 20)
        eng = na.Engine(m, n, r, "nsnmf", dtype=np.float64, theta=0.5)
        eng.upload(V); eng.set_factors(W, H)
        for k in range(1, done + 1):
            eng.iterate(1, first_iteration=k, error_every=10, last_iteration=done)
        We, He = eng.get_factors()
        assert rel(Wc, We) < 1e-12 and rel(Hc, He) < 1e-12, (iters, rel(Wc, We), rel(Hc, He))
        assert s.record(0).frobenius == pytest.approx(eng.frobenius, rel=1e-12)


def test_config2_shape_in_double_full_size_against_the_fp64_oracle():
    """BASELINE configs[1]'s shape (10 000 x 5 000, r = 64) in double precision -- `bench.py --workload c2-f64` -- at full size: 12 iterations of the four-launch
    iteration (237 / 240 product workgroups + 16 Gram passengers per launch, 6 / 3 K slices) against the fp64 oracle, 1e-9 on factors and error."""
    m, n, r, iters = 10000, 5000, 64, 12
    V, W, H = problem(m, n, r, seed=2)
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    ref = oracle.run("mu", V, Wo, Ho, iters)
    eng, Wg, Hg = run_engine(V, W, H, "mu", iters)
    g = eng.geometry()
    assert g["fused_launches"] == 4 and g["padded_rank"] == 64 and 8 <= g["gram_ride_slices_h"] <= 16
    assert rel(Wg, Wo) < 1e-9 and rel(Hg, Ho) < 1e-9, (rel(Wg, Wo), rel(Hg, Ho))
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-9)
    np.testing.assert_allclose(np.linalg.norm(Wg, axis=0), 1.0, rtol=1e-12)
