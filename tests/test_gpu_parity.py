"""GPU parity tests: the HIP path (through the C-ABI of libnmfgpu64.so) against the CPU oracle.

Tolerances (fp32 unless noted; the reference's own summation order inside cuBLAS is unknowable,
SURVEY.md section 8c, so floating-point parity is defined against the fp64 oracle):
  * one product:            |gpu - fp64| <= 4e-7 * sum|a*b| per element (k-ordered fmaf chain bound)
                            and BIT-EXACT against the oracle's model of the kernel's summation order
  * factors after k iterations: relative Frobenius difference of W and of H <= 2e-4 (k <= 100)
  * reported error:          relative 1e-5
  * fp64 path:               1e-9 relative
"""
import numpy as np
import pytest

import nmfgpu_amd as na
from oracle import oracle

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(a)


def problem(m, n, r, dtype, seed=1):
    rng = np.random.default_rng(seed)
    V = F(rng.random((m, n)).astype(dtype))
    W = F((1.0 - rng.random((m, r))).astype(dtype))
    H = F((1.0 - rng.random((r, n))).astype(dtype))
    return V, W, H


def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300)


@pytest.fixture(scope="module", autouse=True)
def _library_is_native():
    assert na.device_count() >= 1, "GPU tests need a HIP device"
    assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
    na.set_verbosity(na.Verbosity.Nothing)
    yield
    na.finalize()


# ------------------------------------------------------------------ single kernels

@pytest.mark.parametrize("X,Y,r", [(128, 64, 64), (500, 200, 8), (200, 500, 8), (1000, 777, 64), (130, 2049, 33), (640, 4100, 64)])
def test_factor_product_bit_exact_and_close(X, Y, r):
    rng = np.random.default_rng(X + Y + r)
    A = F(rng.random((X, Y)).astype(np.float32))
    Fm = F(rng.random((r, Y)).astype(np.float32))
    out, slabs = na.op_factor_product(A, Fm)
    want64 = Fm.astype(np.float64) @ A.astype(np.float64).T
    bound = 4e-7 * (np.abs(Fm).astype(np.float64) @ np.abs(A).astype(np.float64).T)
    assert (np.abs(out - want64) <= bound + 1e-30).all()
    model = oracle.emulate_factor_product(A, Fm, slabs)
    assert np.array_equal(out, model), f"not bit-exact (slabs={slabs}, max diff {np.abs(out - model).max()})"


def test_factor_product_asymmetric_identity_layout_check():
    """A = I-like selector with an asymmetric factor: catches row/column swaps in the MFMA C/D map."""
    X, Y, r = 256, 256, 64
    A = F(np.eye(X, Y, dtype=np.float32))
    Fm = F((np.arange(r)[:, None] * 1000 + np.arange(Y)[None, :]).astype(np.float32))
    out, _ = na.op_factor_product(A, Fm)
    assert np.array_equal(out, Fm[:, :X])


def test_factor_product_valu_and_fp64_paths():
    rng = np.random.default_rng(3)
    A = F(rng.random((300, 210)).astype(np.float32)); Fm = F(rng.random((20, 210)).astype(np.float32))
    out, _ = na.op_factor_product(A, Fm, use_valu=True)
    np.testing.assert_allclose(out, Fm.astype(np.float64) @ A.astype(np.float64).T, rtol=2e-5)
    A64, F64 = F(A.astype(np.float64)), F(Fm.astype(np.float64))
    out64, _ = na.op_factor_product(A64, F64)
    np.testing.assert_allclose(out64, F64 @ A64.T, rtol=1e-12)


@pytest.mark.parametrize("X,Y,r", [(128, 64, 64), (500, 200, 8), (1000, 777, 64), (130, 2049, 33), (2600, 4100, 64), (300, 501, 100), (10000, 1203, 64),
                                   (129, 1, 64), (5, 7, 3), (300, 20, 256), (64, 3, 130)])
def test_factor_product_fp64_mfma(X, Y, r):
    """k_factor_product_f64 (v_mfma_f64_16x16x4_f64): operand lane maps, the interleaved row / column order of the four
    tiles, K-steps of four y with ragged ends, the in-workgroup piece sum and the split-K slabs -- against numpy in
    double (1e-13 relative to the absolute-value product: summation order only) and against the VALU kernel."""
    rng = np.random.default_rng(X * 7 + Y)
    A = F(rng.random((X, Y)) - 0.25); Fm = F(rng.random((r, Y)) - 0.4)
    out, slabs = na.op_factor_product(A, Fm)
    want = Fm @ A.T
    bound = 1e-13 * (np.abs(Fm) @ np.abs(A).T) + 1e-300
    assert (np.abs(out - want) <= bound).all()
    assert slabs >= 1
    ref, one = na.op_factor_product(A, Fm, use_valu=True)
    assert one == 1
    assert (np.abs(out - ref) <= 2 * bound).all()


def test_factor_product_fp64_identity_layout_check():
    """F = shifted identity picks single entries of A: any row / column permutation inside a tile shows up exactly."""
    X, Y, r = 384, 256, 64
    A = F(np.arange(X * Y, dtype=np.float64).reshape(X, Y) / 7.0)
    Fm = F(np.zeros((r, Y)))
    for c in range(r):
        Fm[c, (5 * c + 3) % Y] = 1.0
    out, _ = na.op_factor_product(A, Fm)
    want = np.stack([A[:, (5 * c + 3) % Y] for c in range(r)])
    assert np.array_equal(out, want)


@pytest.mark.parametrize("r,length", [(8, 500), (64, 1000), (100, 333), (128, 777), (200, 1001), (256, 5000), (300, 640), (500, 333)])
def test_gram(r, length):
    rng = np.random.default_rng(r)
    P = F(rng.random((r, length)).astype(np.float32))
    G = na.op_gram(P)
    np.testing.assert_allclose(G, P.astype(np.float64) @ P.astype(np.float64).T, rtol=3e-6)
    assert np.array_equal(G, G.T)


@pytest.mark.parametrize("r,length", [(8, 500), (64, 1000), (64, 3), (100, 333), (158, 4096), (256, 5000), (300, 641), (500, 333)])
def test_gram_fp64_mfma(r, length):
    rng = np.random.default_rng(r + 1)
    P = F(rng.random((r, length)) - 0.3)
    G = na.op_gram(P)
    want = P @ P.T
    assert (np.abs(G - want) <= 1e-13 * (np.abs(P) @ np.abs(P).T) + 1e-300).all()
    assert np.array_equal(G, G.T)


@pytest.mark.parametrize("r", [3, 8, 64])
def test_inverse_small(r):
    rng = np.random.default_rng(r)
    B = rng.random((4 * r, r))
    A = F((B.T @ B).astype(np.float32))
    inv = na.op_inverse(A, offdiag=-0.01, diag=0.05)
    Areg = A.astype(np.float64) - 0.01 + np.eye(r) * (0.05 + 0.01)
    np.testing.assert_allclose(inv.astype(np.float64) @ Areg, np.eye(r), atol=5e-4)


# ------------------------------------------------------------------ whole iterations, engine level

@pytest.mark.parametrize("m,n,r,iters", [(500, 200, 8, 60), (300, 700, 64, 30), (257, 131, 5, 40)])
def test_mu_engine_matches_oracle(m, n, r, iters):
    V, W, H = problem(m, n, r, np.float32)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, iters)
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    assert eng.rmsd == pytest.approx(ref["rmsd"], rel=1e-5)
    assert (Wg >= 0).all() and (Hg >= 0).all()
    np.testing.assert_allclose(np.linalg.norm(Wg.astype(np.float64), axis=0), 1.0, rtol=1e-5)


def test_mu_generic_kernel_path_matches_oracle(monkeypatch):
    """The rank-generic kernels (kernels.hip / kernels_fast.hip) instead of the fused rank-64 MU path."""
    monkeypatch.setenv("NMFAMD_NO_FUSED_MU", "1")
    m, n, r, iters = 500, 200, 8, 60
    V, W, H = problem(m, n, r, np.float32)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, iters)
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)


def test_mu_large_rank_uses_chunked_product():
    """r = 100 -> padded rank 128: two 64-row chunks of the factor product, generic update kernels."""
    m, n, r, iters = 260, 300, 100, 10
    V, W, H = problem(m, n, r, np.float32, seed=21)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, iters)
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)


@pytest.mark.parametrize("alg,r,kw", [("mu", 128, {}), ("mu", 200, {}), ("nsnmf", 256, dict(theta=0.5)), ("mu", 300, {}), ("mu", 500, {}), ("acls", 400, dict(lambda_w=1.0, lambda_h=1.0)),
                                      ("acls", 129, dict(lambda_w=0.01, lambda_h=0.01)), ("gdcls", 256, dict(lam=0.01))])
def test_wide_panels_take_the_mfma_update(alg, r, kw):
    """Padded ranks 128 ... 512: k_panel_update_wide_f32 (r x r product on the MFMA pipe, LDS-staged panel rows),
    multiplicative and least-squares forms, with error terms and the column-norm partial sums."""
    m, n, iters = 700, 610, 10
    V, W, H = problem(m, n, r, np.float32, seed=23)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, **kw)
    eng = na.Engine(m, n, r, alg, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    tol = 2e-4 if alg in ("mu", "nsnmf") else 2e-3
    assert rel(Wg, W64) < tol and rel(Hg, H64) < tol
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-4)


def test_mu_fused_path_interleaved_with_downloads():
    """get_factors() in the middle of a run folds the pending column scale into W and the run continues."""
    m, n, r = 640, 2300, 64   # 18 x-tiles on the H side: the Gram reduction rides in the product launch
    V, W, H = problem(m, n, r, np.float32, seed=13)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    oracle.run("mu", V64, W64, H64, 7)
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(3, first_iteration=1, error_every=0)
    W3, _ = eng.get_factors()
    np.testing.assert_allclose(np.linalg.norm(W3.astype(np.float64), axis=0), 1.0, rtol=1e-5)
    eng.iterate(4, first_iteration=4, error_every=0)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 1e-4 and rel(Hg, H64) < 1e-4


def test_mu_single_iteration_vs_fp32_oracle_tight():
    V, W, H = problem(384, 256, 64, np.float32, seed=9)
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    ref = oracle.run("mu", V, Wo, Ho, 1)
    eng = na.Engine(384, 256, 64, "mu")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(1, first_iteration=1, error_every=10, last_iteration=1)
    Wg, Hg = eng.get_factors()
    np.testing.assert_allclose(Hg, Ho, rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(Wg, Wo, rtol=3e-5, atol=1e-7)
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=2e-5)


@pytest.mark.parametrize("alg,kw", [
    ("nsnmf", dict(theta=0.5)),
    ("gdcls", dict(lam=0.01)),
    ("als", dict()),
    ("acls", dict(lambda_w=0.01, lambda_h=0.01)),
    ("ahcls", dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01)),
])
def test_sibling_algorithms_match_oracle(alg, kw):
    m, n, r, iters = 320, 200, 8, 20
    V, W, H = problem(m, n, r, np.float32, seed=4)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, **kw)
    eng = na.Engine(m, n, r, alg, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    tol = 2e-3 if alg in ("als", "acls", "ahcls", "gdcls") else 2e-4   # LS solves amplify fp32 rounding by cond(W^T W)
    assert rel(Wg, W64) < tol and rel(Hg, H64) < tol
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=10 * tol)


def test_fp64_engine_matches_fp64_oracle():
    m, n, r, iters = 200, 150, 7, 30
    V, W, H = problem(m, n, r, np.float64)
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    ref = oracle.run("mu", V, Wo, Ho, iters)
    eng = na.Engine(m, n, r, "mu", dtype=np.float64)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, Wo) < 1e-9 and rel(Hg, Ho) < 1e-9
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-9)


@pytest.mark.parametrize("alg,r,kw,tol", [("mu", 158, {}, 1e-9), ("nsnmf", 158, dict(theta=0.5), 1e-9), ("mu", 129, {}, 1e-9), ("nsnmf", 300, dict(theta=0.3), 1e-9),
                                          ("als", 190, {}, 1e-6), ("gdcls", 158, dict(lam=0.01), 1e-6), ("mu", 200, {}, 1e-9)])
def test_fp64_panels_padded_to_64_columns(alg, r, kw, tol):
    """Round 5: fp64 panels are padded to a multiple of 64 columns above 64 (fp32: 128) -- the reference example's r = 158 runs at 192 instead of 256.  Padded
    ranks 192 and 320 take the 64 x 64 super-blocks of the fp64 Gram kernel and three / five 16-column tiles per wave of the wide update; every algorithm family
    against the fp64 oracle (the least-squares ones amplify rounding by the condition of the normal matrix)."""
    m, n, iters = 700, 330, 12
    V, W, H = problem(m, n, r, np.float64, seed=31 + r)
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    ref = oracle.run(alg, V, Wo, Ho, iters, **kw)
    eng = na.Engine(m, n, r, alg, dtype=np.float64, **kw)
    assert eng.geometry()["padded_rank"] == 64 * ((r + 63) // 64)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=4, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, Wo) < tol and rel(Hg, Ho) < tol, (rel(Wg, Wo), rel(Hg, Ho))
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=max(tol, 1e-9))
    # fp32 keeps its 128-column padding (the 128-column forms of the split-operand product)
    assert na.Engine(m, n, r, alg, **kw).geometry()["padded_rank"] == 128 * ((r + 127) // 128)


@pytest.mark.parametrize("fmt,base", [(1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1)])
def test_sparse_upload_equals_dense_path_bit_exact(fmt, base):
    """Integer-valued V: the densified matrix must be identical, hence identical factors."""
    import scipy.sparse as sp
    rng = np.random.default_rng(fmt * 10 + base)
    m, n, r = 150, 90, 6
    D = ((rng.random((m, n)) < 0.2) * rng.integers(1, 6, size=(m, n))).astype(np.float32)
    _, W, H = problem(m, n, r, np.float32)
    dense = na.Engine(m, n, r, "mu"); dense.upload(F(D)); dense.set_factors(W, H)
    dense.iterate(10, last_iteration=10)
    sparse = na.Engine(m, n, r, "mu")
    if fmt == 1:
        s = sp.csr_matrix(D); sparse.upload_sparse(1, s.data, s.indptr + base, s.indices + base, base)
    elif fmt == 2:
        s = sp.csc_matrix(D); sparse.upload_sparse(2, s.data, s.indptr + base, s.indices + base, base)
    else:
        s = sp.coo_matrix(D); sparse.upload_sparse(3, s.data, s.row + base, s.col + base, base)
    sparse.set_factors(W, H)
    sparse.iterate(10, last_iteration=10)
    Wd, Hd = dense.get_factors(); Ws, Hs = sparse.get_factors()
    assert np.array_equal(Wd, Ws) and np.array_equal(Hd, Hs)
    assert dense.frobenius == sparse.frobenius
    Vdev = sparse.debug_read(6, 256 * 128)  # V panel, ld = 256
    assert np.array_equal(Vdev.reshape(128, 256).T[:m, :n], D)


def test_sparse_upload_duplicate_coordinates_add():
    """Duplicate coordinates in COO / CSR input: the densifying upload ADDS them (as the sparse-compute path does) -- no race between
    the duplicates' threads.  Two duplicates add exactly in either order, so the result is the dense matrix of the sums, bit for bit."""
    rng = np.random.default_rng(9)
    m, n, r = 120, 70, 5
    rows = rng.integers(0, m, 900); cols = rng.integers(0, n, 900); vals = rng.integers(1, 6, 900).astype(np.float32)
    # every coordinate at most twice: drop third and later occurrences
    seen, keep = {}, []
    for k, rc in enumerate(zip(rows, cols)):
        seen[rc] = seen.get(rc, 0) + 1
        keep.append(seen[rc] <= 2)
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    assert max(seen.values()) >= 2
    D = np.zeros((m, n), dtype=np.float32)
    np.add.at(D, (rows, cols), vals)
    _, W, H = problem(m, n, r, np.float32)
    dense = na.Engine(m, n, r, "mu"); dense.upload(F(D)); dense.set_factors(W, H); dense.iterate(5, last_iteration=5)
    for base in (0, 1):
        coo = na.Engine(m, n, r, "mu")
        coo.upload_sparse(3, vals, (rows + base).astype(np.int32), (cols + base).astype(np.int32), base)
        assert np.array_equal(coo.debug_read(6, 128 * 128).reshape(128, 128).T[:m, :n], D)
        coo.set_factors(W, H); coo.iterate(5, last_iteration=5)
        assert np.array_equal(coo.get_factors()[0], dense.get_factors()[0]) and coo.frobenius == dense.frobenius
    # CSR with duplicates inside a row (stable sort by row keeps them)
    order = np.argsort(rows, kind="stable")
    ptr = np.zeros(m + 1, dtype=np.int32); np.add.at(ptr, rows + 1, 1); ptr = np.cumsum(ptr).astype(np.int32)
    csr = na.Engine(m, n, r, "mu")
    csr.upload_sparse(1, vals[order], ptr, cols[order].astype(np.int32), 0)
    assert np.array_equal(csr.debug_read(6, 128 * 128).reshape(128, 128).T[:m, :n], D)


def test_determinism_bitwise_repeatable():
    V, W, H = problem(400, 300, 64, np.float32, seed=12)
    outs = []
    for _ in range(2):
        eng = na.Engine(400, 300, 64, "mu"); eng.upload(V); eng.set_factors(W, H)
        eng.iterate(20, last_iteration=20)
        outs.append(eng.get_factors() + (eng.frobenius,))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


# ------------------------------------------------------------------ through nmfgpu::compute (the drop-in boundary)

def test_compute_config1_plumbing_and_summary():
    """BASELINE config 1: dense 500 x 200, r = 8, MU Frobenius, through nmfgpu_compute_single."""
    V, W, H = problem(500, 200, 8, np.float32)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, 100)
    s = na.Summary()
    out = []
    res = na.compute(V, W, H, algorithm=na.NmfAlgorithm.Multiplicative, iterations=100, seed=7, summary=s, description_out=out)
    assert res == na.ResultType.Success
    assert rel(W, W64) < 2e-4 and rel(H, H64) < 2e-4
    assert s.record_count() == 1 and s.best_run() == 0
    rec = s.record(0)
    assert rec.numIterations == 100
    assert rec.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    assert rec.rmsd == pytest.approx(ref["rmsd"], rel=1e-5)
    # the library writes the first draw of mt19937(seed) back into the caller's struct
    assert out[0].seed == int(oracle.seed_stream(7, 1)[0])
    # reference example's own check (example/main.cpp:133-146): direct residual next to the reported one
    direct = oracle.direct_frobenius(V, W, H)
    assert direct == pytest.approx(rec.frobenius, rel=2e-2)


def test_compute_threshold_stop_matches_oracle_iteration_count():
    V, W, H = problem(200, 120, 4, np.float32, seed=5)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, 2000, threshold_value=1e-2)
    s = na.Summary()
    res = na.compute(V, W, H, iterations=2000, threshold=1e-2, summary=s)
    assert res == na.ResultType.Success
    assert abs(int(s.record(0).numIterations) - ref["iterations"]) <= 10
    assert s.record(0).numIterations % 10 == 0


@pytest.mark.parametrize("m,n,r,thr", [(900, 700, 64, 1.0), (900, 700, 64, 0.0), (640, 500, 40, 2.0)])
def test_compute_with_a_launch_enqueued_ahead_leaves_the_resident_engine_s_bits(m, n, r, thr):
    """Round 5: on error iterations nmfgpu::compute enqueues the W^T V launch of the NEXT iteration before it waits for the error value (Engine::begin_next_iteration;
    the launch writes scratch only, and is wasted when the threshold ends the run there).  The factors it returns must be the bits of a resident engine stepped
    the same number of iterations one by one -- whether the run ends on the threshold, mid-way, or on its last iteration."""
    V, W0, H0 = problem(m, n, r, np.float32, seed=19)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    s = na.Summary()
    total = 300 if thr > 0.0 else 60
    assert na.compute(V, W, H, iterations=total, threshold=thr, summary=s) == na.ResultType.Success
    done = int(s.record(0).numIterations)
    assert (done < total) == (thr > 0.0), (done, thr)
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W0, H0)
    for k in range(1, done + 1):
        eng.iterate(1, first_iteration=k, error_every=10, last_iteration=done)
    We, He = eng.get_factors()
    assert np.array_equal(W, We) and np.array_equal(H, He)
    assert s.record(0).frobenius == eng.frobenius


def test_compute_error_behaviour():
    V, W, H = problem(60, 40, 5, np.float32)
    assert na.compute(V, W, H, algorithm=na.NmfAlgorithm.nsNMF, iterations=5) == na.ResultType.ErrorInvalidArgument  # theta missing
    assert na.compute(V, W, H, algorithm=na.NmfAlgorithm.AHCLS, iterations=5, parameters={"lambdaW": 0.1, "lambdaH": 0.1, "alphaW": 0.1}) == na.ResultType.ErrorInvalidArgument
    Wbig = F(np.ones((60, 41), dtype=np.float32)); Hbig = F(np.ones((41, 40), dtype=np.float32))
    assert na.compute(V, Wbig, Hbig, iterations=5) == na.ResultType.ErrorInvalidArgument  # features > columns
    calls = {"n": 0}

    def interrupt():
        calls["n"] += 1
        return calls["n"] > 3
    assert na.compute(V, W, H, iterations=50, interrupt=interrupt) == na.ResultType.ErrorUserInterrupt
    assert calls["n"] == 4


def test_compute_multiple_runs_random_init_keeps_best():
    V, _, _ = problem(120, 80, 4, np.float32)
    W = F(np.zeros((120, 4), dtype=np.float32)); H = F(np.zeros((4, 80), dtype=np.float32))
    s = na.Summary()
    res = na.compute(V, W, H, init=na.NmfInitializationMethod.AllRandomValues, iterations=30, runs=4, seed=3, summary=s)
    assert res == na.ResultType.Success
    assert 1 <= s.record_count() <= 4
    frobs = [s.record(i).frobenius for i in range(s.record_count())]
    assert s.best_run() == oracle.summary_best_run(frobs)
    assert frobs == sorted(frobs, reverse=True)  # only improving runs are stored
    # the stored factors are the best run's: the direct residual is close to the best reported error
    assert oracle.direct_frobenius(V, W, H) == pytest.approx(min(frobs), rel=5e-2)
    assert (W > 0).any() and (H > 0).any()


@pytest.mark.parametrize("alg,params", [
    (na.NmfAlgorithm.nsNMF, {"theta": 0.5}),
    (na.NmfAlgorithm.GDCLS, {"lambda": 0.01}),
    (na.NmfAlgorithm.AHCLS, {"lambdaW": 0.01, "lambdaH": 0.01, "alphaW": 0.01, "alphaH": 0.01}),
])
def test_compute_dispatches_sibling_algorithms_double(alg, params):
    """The reference example's configuration in miniature: double precision, parameters by name."""
    V, W, H = problem(160, 66, 6, np.float64, seed=8)
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    name = {na.NmfAlgorithm.nsNMF: "nsnmf", na.NmfAlgorithm.GDCLS: "gdcls", na.NmfAlgorithm.AHCLS: "ahcls"}[alg]
    kw = {"theta": params.get("theta", 0.0), "lam": params.get("lambda", 0.0), "lambda_w": params.get("lambdaW", 0.0),
          "lambda_h": params.get("lambdaH", 0.0), "alpha_w": params.get("alphaW", 0.0), "alpha_h": params.get("alphaH", 0.0)}
    ref = oracle.run(name, V, Wo, Ho, 20, **kw)
    s = na.Summary()
    assert na.compute(V, W, H, algorithm=alg, iterations=20, parameters=params, summary=s) == na.ResultType.Success
    assert rel(W, Wo) < 1e-6 and rel(H, Ho) < 1e-6
    assert s.record(0).frobenius == pytest.approx(ref["frobenius"], rel=1e-6)


def test_kmeans_and_host_init_methods_run():
    V, _, _ = problem(90, 150, 5, np.float32, seed=2)
    C_ = F(np.zeros((90, 5), dtype=np.float32))
    res, memb = na.compute_kmeans(V, C_, iterations=50, seed=1)
    assert res == na.ResultType.Success and memb.max() < 5
    # every centroid is the mean of its members
    for c in range(5):
        if (memb == c).any():
            np.testing.assert_allclose(C_[:, c], V[:, memb == c].mean(axis=1), rtol=1e-4)
    for init in (na.NmfInitializationMethod.MeanColumns, na.NmfInitializationMethod.KMeansAndRandomValues,
                 na.NmfInitializationMethod.KMeansAndNonNegativeWTV, na.NmfInitializationMethod.EInNMF):
        W = F(np.zeros((90, 5), dtype=np.float32)); H = F(np.zeros((5, 150), dtype=np.float32))
        s = na.Summary()
        assert na.compute(V, W, H, init=init, iterations=20, seed=4, summary=s) == na.ResultType.Success
        assert np.isfinite(s.record(0).frobenius) and s.record(0).frobenius < np.linalg.norm(V)


# ------------------------------------------------------------------ BASELINE.json full sizes

def _config2(dtype=np.float32):
    V = F(np.random.RandomState(1).random_sample((5000, 10000)).astype(dtype).T)
    W = F((1.0 - np.random.RandomState(2).random_sample((64, 10000))).astype(dtype).T)
    H = F((1.0 - np.random.RandomState(3).random_sample((5000, 64))).astype(dtype).T)
    return V, W, H


def test_config2_full_size_factor_product_properties():
    """10 000 x 5 000, r = 64: sampled entries against fp64 dot products, the sum identity
    sum_{c,x} OUT = sum_y (sum_c F)(sum_x A), and linearity in F."""
    V, W, H = _config2()
    out, slabs = na.op_factor_product(V, H)            # (V H^T)^T, 64 x 10000
    rng = np.random.default_rng(0)
    xs = rng.integers(0, 10000, 200); cs = rng.integers(0, 64, 200)
    want = np.einsum("ij,ij->i", V[xs, :].astype(np.float64), H[cs, :].astype(np.float64))
    np.testing.assert_allclose(out[cs, xs], want, rtol=2e-6)
    total = float(out.astype(np.float64).sum())
    ident = float(H.astype(np.float64).sum(axis=0) @ V.astype(np.float64).sum(axis=0))
    assert total == pytest.approx(ident, rel=1e-6)
    H2 = F(np.roll(H, 7, axis=1) * np.float32(0.5))
    out2, _ = na.op_factor_product(V, H2)
    out12, _ = na.op_factor_product(V, F(H + H2))
    np.testing.assert_allclose(out12, out + out2, rtol=5e-6)
    # a 200-row stripe bit-exact against the summation-order model (the full model is 6.4 GFLOP of fmaf)
    stripe = slice(4000, 4200)
    model = oracle.emulate_factor_product(F(V[stripe, :]), H, slabs)
    assert np.array_equal(out[:, stripe], model)


def test_config2_full_size_mu_matches_oracle_and_invariants():
    V, W, H = _config2()
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    iters = 10
    ref = oracle.run("mu", V64, W64, H64, iters)
    eng = na.Engine(10000, 5000, 64, "mu")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters - 1, first_iteration=1, error_every=0)
    Wprev, _ = eng.get_factors()
    eng.iterate(1, first_iteration=iters, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 1e-4 and rel(Hg, H64) < 1e-4
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    assert (Wg >= 0).all() and (Hg >= 0).all()
    np.testing.assert_allclose(np.linalg.norm(Wg.astype(np.float64), axis=0), 1.0, rtol=1e-5)
    # the reported value is ||V - W_{k-1} H_k||_F, checked directly at full size
    assert oracle.direct_frobenius(V, Wprev, Hg) == pytest.approx(eng.frobenius, rel=2e-5)


@pytest.mark.parametrize("alg,kw,tol", [
    ("ahcls", dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01), 2e-4),     # (measured 4e-5 / 2e-5, tools/ls_family_drift.py)
    ("gdcls", dict(lam=0.01), 2e-4),
    ("acls", dict(lambda_w=0.01, lambda_h=0.01), 2e-4),
    ("nsnmf", dict(theta=0.5), 2e-4),
])
def test_config5_full_size_algorithm_dispatch(alg, kw, tol):
    """BASELINE config 5: AHCLS / GDCLS (and nsNMF) at 10 000 x 5 000, r = 64, fp32 vs the fp64 oracle."""
    V, W, H = _config2()
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    iters = 5
    ref = oracle.run(alg, V64, W64, H64, iters, **kw)
    eng = na.Engine(10000, 5000, 64, alg, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < tol and rel(Hg, H64) < tol
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-4)


def test_factor_product_160_row_tiles_bit_exact(monkeypatch):
    """The 160-row x-tile form of the product kernel (NMFAMD_FP_TILE=160, off by default)."""
    monkeypatch.setenv("NMFAMD_FP_TILE", "160")
    rng = np.random.default_rng(160)
    for X, Y, r in [(1000, 777, 64), (330, 2049, 33)]:
        A = F(rng.random((X, Y)).astype(np.float32)); Fm = F(rng.random((r, Y)).astype(np.float32))
        out, slabs = na.op_factor_product(A, Fm)
        assert np.array_equal(out, oracle.emulate_factor_product(A, Fm, slabs))
    V, W, H = problem(700, 450, 64, np.float32, seed=31)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    oracle.run("mu", V64, W64, H64, 10)
    eng = na.Engine(700, 450, 64, "mu"); eng.upload(V); eng.set_factors(W, H)
    eng.iterate(10, last_iteration=10)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 1e-4 and rel(Hg, H64) < 1e-4


# ------------------------------------------------------------------ sparse-V compute path (extension)

def _sparse_problem(m, n, r, density, dtype, seed=17):
    rng = np.random.default_rng(seed)
    D = ((rng.random((m, n)) < density) * rng.integers(1, 6, size=(m, n))).astype(dtype)
    W = F((1.0 - rng.random((m, r))).astype(dtype)); H = F((1.0 - rng.random((r, n))).astype(dtype))
    return F(D), W, H


@pytest.mark.parametrize("m,n,r,dtype,tol", [(300, 260, 8, np.float32, 2e-4), (280, 200, 100, np.float32, 2e-4), (150, 120, 6, np.float64, 1e-9)])
def test_sparse_compute_frobenius_mu_equals_dense_oracle(m, n, r, dtype, tol):
    """CSR SpMM path: same factorisation as the reference's densify-then-dense route (fp64 oracle on the dense matrix)."""
    import scipy.sparse as sp
    D, W, H = _sparse_problem(m, n, r, 0.15, dtype)
    D64, W64, H64 = (F(x.astype(np.float64)) for x in (D, W, H))
    ref = oracle.run("mu", D64, W64, H64, 20)
    s = sp.csr_matrix(D)
    eng = na.Engine(m, n, r, "mu", dtype=dtype, sparse_compute=True)
    eng.upload_sparse(1, s.data, s.indptr, s.indices, 0)
    eng.set_factors(W, H)
    eng.iterate(20, last_iteration=20)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < tol and rel(Hg, H64) < tol
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=max(10 * tol, 1e-5))


def test_sparse_compute_formats_and_index_bases_bit_identical():
    """CSR / CSC / COO with base 0 and 1, and a dense upload of the same matrix: identical factors (bit-exact indexing)."""
    import scipy.sparse as sp
    m, n, r = 200, 150, 8
    D, W, H = _sparse_problem(m, n, r, 0.1, np.float32)
    results = []
    for fmt in (1, 2, 3, 0):
        for base in ((0, 1) if fmt else (0,)):
            eng = na.Engine(m, n, r, "mu", sparse_compute=True)
            if fmt == 1:
                s = sp.csr_matrix(D); eng.upload_sparse(1, s.data, s.indptr + base, s.indices + base, base)
            elif fmt == 2:
                s = sp.csc_matrix(D); eng.upload_sparse(2, s.data, s.indptr + base, s.indices + base, base)
            elif fmt == 3:
                s = sp.coo_matrix(D)
                perm = np.random.default_rng(1).permutation(s.nnz)   # unordered triplets
                eng.upload_sparse(3, s.data[perm], (s.row + base)[perm], (s.col + base)[perm], base)
            else:
                eng.upload(D)
            eng.set_factors(W, H)
            eng.iterate(10, last_iteration=10)
            results.append(eng.get_factors() + (eng.frobenius,))
    for Wg, Hg, f in results[1:]:
        assert np.array_equal(Wg, results[0][0]) and np.array_equal(Hg, results[0][1]) and f == results[0][2]


@pytest.mark.parametrize("m,n,r,dtype,tol", [(220, 180, 8, np.float32, 5e-4), (160, 140, 70, np.float32, 5e-4), (120, 100, 5, np.float64, 1e-9)])
def test_kl_divergence_mu_matches_literature_oracle(m, n, r, dtype, tol):
    import scipy.sparse as sp
    D, W, H = _sparse_problem(m, n, r, 0.2, dtype, seed=23)
    D64, W64, H64 = (F(x.astype(np.float64)) for x in (D, W, H))
    ref = oracle.run_kl(D64, W64, H64, 20)
    s = sp.csr_matrix(D)
    eng = na.Engine(m, n, r, "mu", dtype=dtype, divergence="kl")
    eng.upload_sparse(1, s.data, s.indptr, s.indices, 0)
    eng.set_factors(W, H)
    eng.iterate(20, last_iteration=20)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < tol and rel(Hg, H64) < tol
    assert eng.kl_divergence == pytest.approx(ref["kl"], rel=max(20 * tol, 1e-6))
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=max(10 * tol, 1e-6))
    assert (Wg >= 0).all() and (Hg >= 0).all()


def test_compute_accepts_extension_parameters():
    import scipy.sparse as sp
    m, n, r = 120, 90, 5
    D, W, H = _sparse_problem(m, n, r, 0.2, np.float32, seed=29)
    s = sp.csr_matrix(D)
    idx = s.indices.astype(np.int32); ptr = s.indptr.astype(np.int32); val = s.data.astype(np.float32)
    desc = na.api.sparse_description(na.StorageFormat.CSR, m, n, val, ptr, idx)
    Wa, Ha = W.copy(order="F"), H.copy(order="F")
    assert na.compute(desc, Wa, Ha, iterations=10, parameters={"sparseCompute": 1.0}) == na.ResultType.Success
    Wb, Hb = W.copy(order="F"), H.copy(order="F")
    assert na.compute(desc, Wb, Hb, iterations=10) == na.ResultType.Success      # reference route: densify
    assert rel(Wa, Wb) < 1e-5 and rel(Ha, Hb) < 1e-5
    Wk, Hk = W.copy(order="F"), H.copy(order="F")
    sm = na.Summary()
    assert na.compute(desc, Wk, Hk, iterations=10, parameters={"divergence": 1.0}, summary=sm) == na.ResultType.Success
    assert np.isfinite(sm.record(0).frobenius) and not np.allclose(Wk, Wa)
    assert na.compute(desc, Wk, Hk, algorithm=na.NmfAlgorithm.ALS, iterations=2, parameters={"sparseCompute": 1.0}) == na.ResultType.ErrorInvalidArgument


def test_sparse_kl_medium_size_properties():
    """20 000 x 5 000 at 1 % density, r = 128 (BASELINE config 3 scaled by 1/20): the divergence does not increase,
    factors stay non-negative, W keeps unit columns."""
    import scipy.sparse as sp
    m, n, r = 20000, 5000, 128
    rng = np.random.default_rng(3)
    s = sp.random(m, n, density=0.01, format="csr", random_state=3, data_rvs=lambda k: rng.integers(1, 6, size=k).astype(np.float32)).astype(np.float32)
    W = F((1.0 - rng.random((m, r))).astype(np.float32)); H = F((1.0 - rng.random((r, n))).astype(np.float32))
    eng = na.Engine(m, n, r, "mu", divergence="kl")
    eng.upload_sparse(1, s.data, s.indptr, s.indices, 0)
    eng.set_factors(W, H)
    kls = []
    for k in range(4):
        eng.iterate(10, first_iteration=10 * k + 1, error_every=10)
        kls.append(eng.kl_divergence)
    assert all(np.isfinite(kls)) and all(b <= a * (1 + 1e-5) for a, b in zip(kls, kls[1:])), kls
    Wg, Hg = eng.get_factors()
    assert (Wg >= 0).all() and (Hg >= 0).all()
    np.testing.assert_allclose(np.linalg.norm(Wg.astype(np.float64), axis=0), 1.0, rtol=1e-4)


# ------------------------------------------------------------------ ragged / edge shapes

@pytest.mark.parametrize("m,n,r", [(1, 1, 1), (2, 3, 1), (5, 64, 3), (64, 5, 4), (129, 127, 64), (127, 129, 63), (300, 65, 65),
                                   (1025, 33, 2), (33, 1025, 7), (640, 641, 32), (257, 2051, 16), (2051, 257, 128)])
def test_mu_edge_shapes_match_oracle(m, n, r):
    """Sizes that are not multiples of any tile (and the smallest possible ones): padding rows/columns stay inert."""
    V, W, H = problem(m, n, r, np.float32, seed=m * 7 + n)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, 12)
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(12, first_iteration=1, error_every=10, last_iteration=12)
    Wg, Hg = eng.get_factors()
    assert np.isfinite(Wg).all() and np.isfinite(Hg).all()
    assert rel(Wg, W64) < 3e-4 and rel(Hg, H64) < 3e-4
    if np.isfinite(ref["frobenius"]) and ref["frobenius"] > 1e-3:
        assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-4)


def test_zero_columns_and_rows_stay_zero():
    """A zero row of V drives the matching row of W to zero; a zero column of W must stay zero (sum > 0 guard, no NaN)."""
    m, n, r = 200, 150, 6
    V, W, H = problem(m, n, r, np.float32, seed=77)
    V[17, :] = 0.0
    W[:, 2] = 0.0
    V, W = F(V), F(W)
    eng = na.Engine(m, n, r, "mu"); eng.upload(V); eng.set_factors(W, H)
    eng.iterate(15, last_iteration=15)
    Wg, Hg = eng.get_factors()
    assert np.isfinite(Wg).all() and np.isfinite(Hg).all() and np.isfinite(eng.frobenius)
    assert (Wg[:, 2] == 0).all() and (Wg[17, :] == 0).all()
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    oracle.run("mu", V64, W64, H64, 15)
    assert rel(Wg, W64) < 3e-4 and rel(Hg, H64) < 3e-4


# ------------------------------------------------------------------ bf16 operand mode (extension)

def _round_bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


@pytest.mark.parametrize("X,Y,r", [(128, 64, 64), (500, 200, 8), (1000, 777, 64), (130, 2049, 33), (2600, 4100, 64),
                                   (300, 500, 100), (1000, 3000, 128), (257, 1111, 256), (40000, 700, 256), (640, 4100, 300),
                                   (300, 20, 256), (130, 5, 200), (129, 1, 256), (64, 33, 128), (5, 7, 3)])
def test_factor_product_bf16_is_exact_product_of_rounded_operands(X, Y, r):
    """Fragment order, lane maps and the K-step bookkeeping: the result must be the fp32-accumulated
    product of the bf16-ROUNDED operands (products of bf16 values are exact in fp32)."""
    rng = np.random.default_rng(X * 3 + Y)
    A = F(rng.random((X, Y)).astype(np.float32)); Fm = F((rng.random((r, Y)) - 0.3).astype(np.float32))
    out = na.op_factor_product_bf16(A, Fm)
    Ab, Fb = _round_bf16(A).astype(np.float64), _round_bf16(Fm).astype(np.float64)
    want = Fb @ Ab.T
    # fp32 accumulation in K order; the 256-column kernel keeps ONE chain per output element over its whole K slice (no K
    # pieces inside the workgroup), so the largest of 10^7 elements sits a little above the 4e-7 of the split chains
    bound = (8e-7 if r > 128 else 4e-7) * (np.abs(Fb) @ np.abs(Ab).T) + 1e-30
    assert (np.abs(out - want) <= bound).all()
    # and it is NOT the fp32 product: the rounding of the operands is visible
    exact = Fm.astype(np.float64) @ A.astype(np.float64).T
    assert np.abs(out - exact).max() > 10 * np.abs(out - want).max()


# ------------------------------------------------------------------ fp32 product by exact 3 x bf16 operand splitting

@pytest.mark.parametrize("X,Y,r", [(128, 64, 64), (500, 200, 8), (1000, 777, 64), (130, 2049, 33), (2600, 4100, 64), (10000, 1203, 64),
                                   (300, 500, 100), (257, 1111, 256), (129, 1, 64), (5, 7, 3), (300, 20, 256), (64, 33, 128), (640, 4100, 300),
                                   (128, 192, 64), (100, 208, 64)])       # (12 K-steps on four wave pieces: the shortest reduction range that takes odd pieces; 13: whole turns)
def test_factor_product_split_operands_has_fp32_accuracy(X, Y, r):
    """The default fp32 product (kernels_x3.hip): six bf16 MFMAs per block on exactly split operands.  Same bound as the
    native fp32 MFMA kernel above -- |gpu - fp64| <= 4e-7 * sum|a*b| per element -- and an rms error within 2x of it."""
    rng = np.random.default_rng(X * 5 + Y + r)
    A = F((rng.random((X, Y)) - 0.25).astype(np.float32)); Fm = F((rng.random((r, Y)) - 0.3).astype(np.float32))
    out = na.op_factor_product_x3(A, Fm)
    want = Fm.astype(np.float64) @ A.astype(np.float64).T
    bound = 4e-7 * (np.abs(Fm).astype(np.float64) @ np.abs(A).astype(np.float64).T) + 1e-30
    assert (np.abs(out - want) <= bound).all()
    native, _ = na.op_factor_product(A, Fm)
    rms = lambda o: np.sqrt(((o - want) ** 2).mean())
    assert rms(out) <= 2.0 * rms(native) + 1e-12


def test_factor_product_split_is_exact_where_fp32_is_exact():
    """hi + mid + lo reproduces all 24 significand bits of either operand: against a one-hot power-of-two partner the
    result is a scaled copy of the operand, bit for bit (a dropped or mis-ordered plane, or a truncating cut, shows here)."""
    rng = np.random.default_rng(7)
    X, Y, r = 384, 96, 64
    A = F((rng.standard_normal((X, Y)) * np.exp(4 * rng.standard_normal((X, Y)))).astype(np.float32))
    Fm = np.zeros((r, Y), dtype=np.float32)
    cols = rng.integers(0, Y, r); scale = np.float32(2.0) ** rng.integers(-20, 20, r)
    Fm[np.arange(r), cols] = scale
    out = na.op_factor_product_x3(A, F(Fm))
    assert np.array_equal(out, (A[:, cols] * scale[None, :]).T)
    # and the factor side: one-hot A selects factor entries
    Fr = F((rng.standard_normal((r, Y)) * np.exp(4 * rng.standard_normal((r, Y)))).astype(np.float32))
    A1 = np.zeros((X, Y), dtype=np.float32)
    ys = rng.integers(0, Y, X); sc = np.float32(2.0) ** rng.integers(-20, 20, X)
    A1[np.arange(X), ys] = sc
    out = na.op_factor_product_x3(F(A1), Fr)
    assert np.array_equal(out, Fr[:, ys] * sc[None, :])
    # small integers: every partial sum is exact in fp32, whatever the order
    Ai = F(rng.integers(0, 128, (X, 512)).astype(np.float32)); Fi = F(rng.integers(0, 128, (r, 512)).astype(np.float32))
    assert np.array_equal(na.op_factor_product_x3(Ai, Fi), (Fi.astype(np.int64) @ Ai.astype(np.int64).T).astype(np.float32))


@pytest.mark.parametrize("alg,r,kw", [("mu", 64, {}), ("mu", 20, {}), ("als", 64, {}), ("gdcls", 50, dict(lam=0.01)),
                                      ("nsnmf", 64, dict(theta=0.5)), ("mu", 130, {})])
def test_native_fp32_mfma_path_still_agrees(alg, r, kw):
    """precision = fp32_mfma keeps the fp32 MFMA instructions (the default before the split-operand product); both
    paths sit within the fp32 tolerance of the fp64 oracle (2e-4) and within 1e-4 of each other after 20 iterations (the
    least-squares algorithms amplify the last-bit differences of the products through the inverse)."""
    m, n = 1300, 900
    V, W, H = problem(m, n, r, np.float32, seed=47)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, 20, **kw)
    got = {}
    for prec in ("native", "fp32_mfma"):
        eng = na.Engine(m, n, r, alg, precision=prec, **kw)
        assert eng.geometry()["product_kernel"] == ({"native": 2 if r > 32 else 0, "fp32_mfma": 0}[prec])
        eng.upload(V); eng.set_factors(W, H)
        eng.iterate(20, last_iteration=20)
        got[prec] = eng.get_factors() + (eng.frobenius,)
        assert rel(got[prec][0], W64) < 2e-4 and rel(got[prec][1], H64) < 2e-4
        assert got[prec][2] == pytest.approx(ref["frobenius"], rel=1e-5)
    assert rel(got["native"][0], got["fp32_mfma"][0]) < 1e-4 and rel(got["native"][1], got["fp32_mfma"][1]) < 1e-4


@pytest.mark.parametrize("alg,r,kw", [("mu", 64, {}), ("als", 40, {}), ("nsnmf", 130, dict(theta=0.4))])
def test_one_resident_image_gives_identical_factors(alg, r, kw, monkeypatch):
    """One resident image of V (chosen automatically when one image fits the 256 MiB memory-side cache and two do not, or
    when two would not fit in HBM; NMFAMD_ONE_IMAGE forces it): W^T V reads the image of V along its reduction index.
    Same arithmetic in the same order => bit-identical factors and errors."""
    m, n = 1500, 1100
    V, W, H = problem(m, n, r, np.float32, seed=53)
    out = {}
    for one in (False, True):
        monkeypatch.setenv("NMFAMD_ONE_IMAGE", "1" if one else "0")
        eng = na.Engine(m, n, r, alg, **kw)
        assert eng.geometry()["resident_images"] == (1 if one else 2)
        eng.upload(V); eng.set_factors(W, H)
        eng.iterate(12, last_iteration=12)
        out[one] = eng.get_factors() + (eng.frobenius,)
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])
    assert out[False][2] == out[True][2]


@pytest.mark.parametrize("X,Y,r", [(300, 500, 64), (1000, 777, 40), (130, 2049, 33), (257, 1111, 256), (129, 1, 64), (5, 7, 3), (128, 192, 64), (100, 208, 64)])
def test_factor_product_split_y_tiled_is_bit_identical(X, Y, r):
    import ctypes as C
    from nmfgpu_amd._lib import library
    rng = np.random.default_rng(X + 3 * Y)
    A = F((rng.random((X, Y)) - 0.2).astype(np.float32)); Fm = F(rng.random((r, Y)).astype(np.float32))
    out = np.zeros((r, X), dtype=np.float32, order="F")
    st = library().nmfamd_op_factor_product_x3_ytiled(C.c_void_p(A.ctypes.data), C.c_long(X), X, Y, C.c_void_p(Fm.ctypes.data), C.c_long(r), r,
                                                      C.c_void_p(out.ctypes.data), C.c_long(r), 0, None)
    assert st == 0
    assert np.array_equal(out, na.op_factor_product_x3(A, Fm))


def test_bf16_operand_mode_tracks_fp32_within_stated_tolerance():
    """precision = bf16: operands of the two big products carry 8 significant bits; factors after 20 iterations
    agree with the fp64 oracle to 2e-2 relative (fp32 mode: 2e-4), the reported error to 1e-3."""
    m, n, r = 900, 700, 64
    V, W, H = problem(m, n, r, np.float32, seed=41)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, 20)
    eng = na.Engine(m, n, r, "mu", precision="bf16")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(20, last_iteration=20)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 2e-2 and rel(Hg, H64) < 2e-2
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-3)
    assert (Wg >= 0).all() and (Hg >= 0).all()
    np.testing.assert_allclose(np.linalg.norm(Wg.astype(np.float64), axis=0), 1.0, rtol=1e-5)
    with pytest.raises(na.EngineError):
        na.Engine(m, n, r, "mu", precision="bf16", sparse_compute=True)   # dense resident V only
    with pytest.raises(na.EngineError):
        na.Engine(m, n, r, "mu", precision="bf16", dtype=np.float64)      # fp32 engine only


@pytest.mark.parametrize("alg,r,kw", [("mu", 100, {}), ("nsnmf", 256, dict(theta=0.5)), ("nsnmf", 40, dict(theta=0.3)),
                                      ("gdcls", 64, dict(lam=0.01)), ("acls", 130, dict(lambda_w=0.01, lambda_h=0.01))])
def test_bf16_operand_mode_every_algorithm_and_wide_panels(alg, r, kw):
    """bf16 operands are a property of the two big products, so every algorithm and every padded rank
    (64, k * 128) takes them; same tolerance statement as above (2e-2 on the factors after 10 iterations)."""
    m, n = 1100, 800
    V, W, H = problem(m, n, r, np.float32, seed=43)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, 10, **kw)
    eng = na.Engine(m, n, r, alg, precision="bf16", **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(10, last_iteration=10)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 2e-2 and rel(Hg, H64) < 2e-2
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=2e-3)


def test_config4_shard_size_nsnmf_bf16_properties():
    """BASELINE configs[3], one GPU's share: 50 000 x 6 250 column shard, r = 256, nsNMF theta = 0.5, bf16 operands.
    Too big for the oracle in seconds, so size-independent properties: the reported error (trace formula, terms from three
    different kernels) agrees with the residual evaluated directly on a row sample; W S has the column sums the smoothing
    implies; unit-norm W columns; non-negativity; the error decreases."""
    m, n, r, theta = 50000, 6250, 256, 0.5
    rs = np.random.RandomState(1)
    V = np.empty((m, n), dtype=np.float32, order="F")
    for j0 in range(0, n, 625):
        V[:, j0:j0 + 625] = rs.random_sample((625, m)).astype(np.float32).T
    W = F((1.0 - np.random.RandomState(2).random_sample((r, m))).astype(np.float32).T)
    H = F((1.0 - np.random.RandomState(3).random_sample((n, r))).astype(np.float32).T)
    eng = na.Engine(m, n, r, "nsnmf", theta=theta, precision="bf16")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(10, first_iteration=1, error_every=10)
    f10 = eng.frobenius
    eng.iterate(10, first_iteration=11, error_every=10, last_iteration=20)
    f20, rmsd20 = eng.frobenius, eng.rmsd
    assert np.isfinite(f20) and f20 < f10
    # the error terms refer to (W_{k-1}, H_k): one more H step with W held gives the pair the formula describes,
    # so instead compare with the direct residual of the CURRENT pair on a row sample, to 1 % (same order of magnitude
    # as one iteration's progress at iteration 20, plus the bf16 operand rounding of the reported terms)
    WS, Hg = eng.get_factors()          # nsNMF returns W S (AlgorithmNonSmoothNMF.h:221-225)
    assert (WS >= 0).all() and (Hg >= 0).all() and np.isfinite(WS).all() and np.isfinite(Hg).all()
    rows = np.random.default_rng(7).choice(m, 400, replace=False)
    R = V[rows, :].astype(np.float64) - WS[rows, :].astype(np.float64) @ (np.eye(r) @ Hg.astype(np.float64))
    # V ~ (W S) H: the model of nsNMF is V = W S H and get_factors hands back W S
    sample_rmsd = np.sqrt((R * R).mean())
    assert sample_rmsd == pytest.approx(rmsd20, rel=2e-2)
    # W = (W S) S^-1 has unit-norm columns; S^-1 = (I - (theta / r) / ((1 - theta) + theta) 1 1^T) / (1 - theta)
    a, b = 1.0 - theta, theta / r
    Wn = (WS.astype(np.float64) - (b / (a + b * r)) * WS.astype(np.float64).sum(axis=1, keepdims=True)) / a
    np.testing.assert_allclose(np.linalg.norm(Wn, axis=0), 1.0, rtol=1e-4)


# ------------------------------------------------------------------ constant basis vectors (semi-supervised projection)

@pytest.mark.parametrize("alg,r,kw", [("mu", 8, {}), ("mu", 64, {}), ("mu", 100, {}), ("nsnmf", 8, dict(theta=0.4)), ("gdcls", 8, dict(lam=0.01)),
                                      ("als", 8, {}), ("acls", 8, dict(lambda_w=0.01, lambda_h=0.01)),
                                      ("ahcls", 8, dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01))])
def test_constant_basis_vectors_fit_h_only(alg, r, kw):
    """useConstantBasisVectors (ref AlgorithmMultiplicativeFrobenius.h:144-146,218-228 and the siblings): W is never
    written, H is fitted to it -- the mode the R binding uses to project new data onto a trained basis."""
    m, n, iters = 300, 180, 20
    V, W, H = problem(m, n, r, np.float32, seed=31)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    W0 = W64.copy()
    ref = oracle.run(alg, V64, W64, H64, iters, const_w=True, **kw)
    eng = na.Engine(m, n, r, alg, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters, constant_w=True)
    Wg, Hg = eng.get_factors()
    tol = 2e-3 if alg in ("als", "acls", "ahcls", "gdcls") else 2e-4
    assert rel(Hg, H64) < tol
    if alg == "nsnmf":
        assert rel(Wg, W64) < tol                      # both sides return W S
    else:
        np.testing.assert_array_equal(W64, W0)         # the oracle left W alone ...
        np.testing.assert_allclose(Wg, W, rtol=1e-6)   # ... and so did the engine (panel round trip only)
    if alg in ("mu", "nsnmf"):
        assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-4)


def test_compute_with_constant_basis_vectors_through_the_boundary():
    V, W, H = problem(240, 150, 6, np.float32, seed=33)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    W_before = W.copy()
    ref = oracle.run("mu", V64, W64, H64, 50, const_w=True)
    s = na.Summary()
    assert na.compute(V, W, H, iterations=50, constant_basis_vectors=True, summary=s) == na.ResultType.Success
    np.testing.assert_allclose(W, W_before, rtol=1e-6)
    assert rel(H, H64) < 2e-4
    assert s.record(0).frobenius == pytest.approx(ref["frobenius"], rel=1e-4)


# ------------------------------------------------------------------ device introspection

def test_device_introspection_and_selection():
    """getNumberOfGpu / getInformationForGpuIndex / chooseGpu (ref Interface.cpp:152-202)."""
    count = na.get_number_of_gpu()
    assert count >= 1 and count == na.device_count()
    res, info = na.get_information_for_gpu_index(0)
    assert res == na.ResultType.Success
    assert len(info.name) > 0
    assert 0 < info.freeMemory <= info.totalMemory and info.totalMemory > (100 << 30)      # an MI355X carries 288 GB
    res, _ = na.get_information_for_gpu_index(count)                                         # one past the end
    assert res == na.ResultType.ErrorDeviceSelection
    assert na.choose_gpu(0) == na.ResultType.Success                                          # fixture has initialised the library
    assert na.choose_gpu(count) == na.ResultType.ErrorDeviceSelection
    # the library still computes on the selected device afterwards
    V, W, H = problem(64, 40, 4, np.float32, seed=3)
    assert na.compute(V, W, H, iterations=5) == na.ResultType.Success


@pytest.mark.parametrize("alg,r,kw", [("mu", 100, {}), ("nsnmf", 158, dict(theta=0.5)), ("mu", 300, {}), ("acls", 400, dict(lambda_w=1.0, lambda_h=1.0)),
                                      ("gdcls", 130, dict(lam=0.05))])
def test_double_precision_wide_panels(alg, r, kw):
    """fp64 at padded ranks 128 ... 512: chunked fp64 MFMA product (one launch, grid.z = chunks) and k_panel_update_wide_f64.
    nsNMF at r = 158 in double is the configuration of the reference's example program (ref example/main.cpp:30-32,130)."""
    m, n, iters = 700, 610, 10
    V, W, H = problem(m, n, r, np.float64, seed=29)
    Wo, Ho = W.copy(order="F"), H.copy(order="F")
    ref = oracle.run(alg, V, Wo, Ho, iters, **kw)
    eng = na.Engine(m, n, r, alg, dtype=np.float64, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    tol = 1e-9 if alg in ("mu", "nsnmf") else 1e-6
    assert rel(Wg, Wo) < tol and rel(Hg, Ho) < tol
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=tol)


# ------------------------------------------------------------------ seeded sweep over shapes / algorithms / number types

def _sweep_case(seed):
    rng = np.random.default_rng(1000 + seed)
    alg = ["mu", "mu", "nsnmf", "gdcls", "acls", "ahcls"][seed % 6]
    dtype = np.float64 if seed % 3 == 0 else np.float32
    m = int(rng.integers(1, 700)); n = int(rng.integers(1, 700))
    r = int(rng.integers(1, 150 if alg in ("mu", "nsnmf") else 24))
    kw = {"mu": {}, "nsnmf": dict(theta=float(rng.uniform(0.1, 0.9))), "gdcls": dict(lam=0.3), "acls": dict(lambda_w=0.5, lambda_h=0.5),
          "ahcls": dict(lambda_w=0.5, lambda_h=0.5, alpha_w=0.3, alpha_h=0.3)}[alg]
    precision = "bf16" if (dtype == np.float32 and seed % 5 == 4 and alg in ("mu", "nsnmf")) else "native"
    return alg, dtype, m, n, r, kw, precision


@pytest.mark.parametrize("seed", range(36))
def test_seeded_sweep_matches_oracle(seed):
    """Ragged shapes (down to a single row / column), ranks on both sides of every padding boundary, all number types:
    five iterations against the oracle.  Catches indexing slips the hand-picked shapes do not."""
    alg, dtype, m, n, r, kw, precision = _sweep_case(seed)
    V, W, H = problem(m, n, r, dtype, seed=seed)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, 5, **kw)
    eng = na.Engine(m, n, r, alg, dtype=dtype, precision=precision, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(5, first_iteration=1, error_every=10, last_iteration=5)
    Wg, Hg = eng.get_factors()
    if precision == "bf16":
        tol = 2e-2
    elif dtype == np.float64:
        tol = 1e-9 if alg in ("mu", "nsnmf") else 1e-6
    else:
        tol = 2e-4 if alg in ("mu", "nsnmf") else 3e-3
    assert np.isfinite(Wg).all() and np.isfinite(Hg).all()
    assert rel(Wg, W64) < tol and rel(Hg, H64) < tol, (alg, dtype, m, n, r, precision)
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=max(10 * tol, 1e-6), abs=1e-9)


@pytest.mark.parametrize("seed", range(16))
def test_seeded_sweep_sparse_paths(seed):
    """Sparse compute (CSR + CSC images, SpMM / SDDMM) over ragged shapes, densities, the three sparse formats, both index
    bases, both objectives and both number types: against the oracle on the dense equivalent (Frobenius) / the literature
    formula (KL)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(500 + seed)
    dtype = np.float64 if seed % 4 == 0 else np.float32
    m = int(rng.integers(2, 500)); n = int(rng.integers(2, 500)); r = int(rng.integers(1, 200))
    density = float(rng.uniform(0.02, 0.4))
    kl = seed % 2 == 1
    D, W, H = _sparse_problem(m, n, r, density, dtype, seed=seed)
    D64, W64, H64 = (F(x.astype(np.float64)) for x in (D, W, H))
    ref = oracle.run_kl(D64, W64, H64, 6) if kl else oracle.run("mu", D64, W64, H64, 6)
    fmt, base = 1 + seed % 3, (seed // 3) % 2
    if fmt == 1:
        s = sp.csr_matrix(D); vals, a, b = s.data, s.indptr + base, s.indices + base
    elif fmt == 2:
        s = sp.csc_matrix(D); vals, a, b = s.data, s.indptr + base, s.indices + base
    else:
        s = sp.coo_matrix(D); vals, a, b = s.data, s.row + base, s.col + base
    eng = na.Engine(m, n, r, "mu", dtype=dtype, divergence="kl" if kl else "frobenius", sparse_compute=True)
    eng.upload_sparse(fmt, vals, a, b, base)
    eng.set_factors(W, H)
    eng.iterate(6, last_iteration=6)
    Wg, Hg = eng.get_factors()
    tol = 1e-9 if dtype == np.float64 else 3e-4
    assert rel(Wg, W64) < tol and rel(Hg, H64) < tol, (m, n, r, density, fmt, base, kl)
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=max(10 * tol, 1e-6), abs=1e-9)
    if kl:
        assert eng.kl_divergence == pytest.approx(ref["kl"], rel=max(30 * tol, 1e-6), abs=1e-6)


@pytest.mark.parametrize("dtype,tol", [(np.float32, 3e-4), (np.float64, 1e-9)])
def test_rank_beyond_the_mfma_update_kernels(dtype, tol):
    """r = 600 -> padded rank 640: past the 512-column limit of the MFMA update kernels, the generic update runs
    (chunked MFMA products, generic panel update / Gram)."""
    m, n, r, iters = 700, 660, 600, 4
    V, W, H = problem(m, n, r, dtype, seed=37)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, iters)
    eng = na.Engine(m, n, r, "mu", dtype=dtype)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < tol and rel(Hg, H64) < tol
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=max(10 * tol, 1e-6))


def test_two_threads_compute_concurrently():
    """The library context is per thread (ref Interface.cpp:51: thread_local): two threads, each with its own
    initialize() / compute() / finalize(), factorise different problems at the same time on the one GPU."""
    import threading
    results = {}

    def worker(tag, m, n, r, alg, seed):
        try:
            V, W, H = problem(m, n, r, np.float32, seed=seed)
            V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
            ref = oracle.run("mu" if alg == na.NmfAlgorithm.Multiplicative else "als", V64, W64, H64, 60)
            assert na.initialize() == na.ResultType.Success          # a fresh thread has no context yet
            s = na.Summary()
            res = na.compute(V, W, H, algorithm=alg, iterations=60, summary=s)
            na.finalize()
            tol = 2e-4 if alg == na.NmfAlgorithm.Multiplicative else 3e-3
            results[tag] = (res == na.ResultType.Success and rel(W, W64) < tol and rel(H, H64) < tol
                            and abs(s.record(0).frobenius - ref["frobenius"]) <= 10 * tol * ref["frobenius"])
        except Exception as exc:      # noqa: BLE001
            results[tag] = repr(exc)

    threads = [threading.Thread(target=worker, args=("a", 900, 700, 64, na.NmfAlgorithm.Multiplicative, 51)),
               threading.Thread(target=worker, args=("b", 640, 500, 12, na.NmfAlgorithm.ALS, 52)),
               threading.Thread(target=worker, args=("c", 300, 1100, 100, na.NmfAlgorithm.Multiplicative, 53))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert results == {"a": True, "b": True, "c": True}, results


def test_engines_release_their_device_memory():
    """Create / run / destroy engines of every flavour repeatedly: the free device memory comes back."""
    import gc
    def free_bytes():
        res, info = na.get_information_for_gpu_index(0)
        assert res == na.ResultType.Success
        return info.freeMemory
    V, W, H = problem(1500, 1100, 40, np.float32, seed=61)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    import scipy.sparse as sp
    s = sp.csr_matrix(V * (V > 0.8))
    def cycle():
        for kw in (dict(algorithm="mu"), dict(algorithm="ahcls", lambda_w=0.1, lambda_h=0.1, alpha_w=0.1, alpha_h=0.1), dict(algorithm="nsnmf", theta=0.5),
                   dict(algorithm="mu", precision="bf16"), dict(algorithm="mu", dtype=np.float64)):
            dt = kw.get("dtype", np.float32)
            eng = na.Engine(1500, 1100, 40, **kw)
            eng.upload(V64 if dt == np.float64 else V); eng.set_factors(W64 if dt == np.float64 else W, H64 if dt == np.float64 else H)
            eng.iterate(3, last_iteration=3)
            assert np.isfinite(eng.frobenius)
            eng.close()
        eng = na.Engine(1500, 1100, 40, "mu", divergence="kl")
        eng.upload_sparse(1, s.data.astype(np.float32), s.indptr, s.indices, 0); eng.set_factors(W, H)
        eng.iterate(3, last_iteration=3)
        eng.close()
    cycle(); gc.collect()
    before = free_bytes()
    for _ in range(5):
        cycle()
    gc.collect()
    after = free_bytes()
    assert before - after < (64 << 20), (before, after)      # allocator granularity, not a leak of 100-MB images


def test_long_horizon_fused_mu_stays_on_the_oracle_trajectory():
    """1 000 iterations of the fused rank-64 path (lazy column scale, passenger Gram reductions, split-K slabs): the error
    trajectory follows the fp64 oracle's to 1e-4 at every evaluation and never increases; no drift accumulates in the
    unnormalised W the engine carries between iterations."""
    m, n, r = 900, 700, 64
    V, W, H = problem(m, n, r, np.float32, seed=71)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, 1000)
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    errs = []
    for k in range(10):
        eng.iterate(100, first_iteration=100 * k + 1, error_every=10)
        errs.append(eng.frobenius)
    hist = np.array(ref["history"])[:, 0]
    np.testing.assert_allclose(errs, hist[9::10], rtol=1e-4)
    assert all(b <= a * (1 + 1e-6) for a, b in zip(errs, errs[1:]))
    Wg, Hg = eng.get_factors()
    # factors themselves: the landscape is flat near a stationary point, so compare the reconstruction, not the factors
    R_gpu = Wg.astype(np.float64) @ Hg.astype(np.float64)
    R_ref = W64 @ H64
    assert np.linalg.norm(R_gpu - R_ref) / np.linalg.norm(R_ref) < 2e-3
    np.testing.assert_allclose(np.linalg.norm(Wg.astype(np.float64), axis=0), 1.0, rtol=1e-5)


@pytest.mark.parametrize("m,n", [(4000, 200), (1500, 900)])
def test_gram_k_slices_and_long_slab_lists_agree_with_one_slice_and_the_oracle(m, n, monkeypatch, diag_build):
    """Round 4 (column shards): W^T W rides in the W^T V launch cut into K slices -- 10 passenger workgroups per slice write UNSCALED partial matrices, the H update
    adds them in order and scales them by the pending column scale, which comes from the W update's per-workgroup sums of squares (gram_image.h, k_mu64_update32).
    The engine picks the slice count from the shard's shape; here every count is forced (NMFAMD_GRAM_KSPLIT, read by the measurement build only: fixture diag_build) at shapes the fp64 oracle covers: same factors and
    reported errors as the one-slice form up to summation order, all within the fp32 tolerance of the oracle.  4 000 x 200 also gives W^T V ten K slices: the H update's
    13-slab batches (k_mu64_update32<false, 13, QS>)."""
    r, iters = 64, 40
    V, W0, H0 = problem(m, n, r, np.float32, seed=77)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W0, H0))
    ref = oracle.run("mu", V64, W64, H64, iters)
    got = {}
    # (round 5: one slice comes in two forms -- ten workgroups with a tile each, key 1, or the tiles' K ranges dealt to all sixteen passengers, the default, key 16)
    for ks in (1, 16, 2, 4, 8):
        monkeypatch.setenv("NMFAMD_GRAM_KSPLIT", str(1 if ks == 16 else ks))
        monkeypatch.setenv("NMFAMD_GRAM_SPREAD", "1" if ks == 16 else "0")
        eng = na.Engine(m, n, r, "mu")
        eng.upload(V); eng.set_factors(W0, H0)
        eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
        Wg, Hg = eng.get_factors()
        got[ks] = (Wg, Hg, eng.frobenius, eng.geometry()["slabs_h"])
        assert eng.geometry()["gram_k_slices"] == ks
        eng.close()
        assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4, (ks, rel(Wg, W64), rel(Hg, H64))
        assert got[ks][2] == pytest.approx(ref["frobenius"], rel=1e-5)
    if (m, n) == (4000, 200):
        assert got[1][3] > 8                                  # more slabs than one batch of eight
    for ks in (16, 2, 4, 8):
        assert rel(got[ks][0], got[1][0]) < 5e-6 and rel(got[ks][1], got[1][1]) < 5e-6


def test_column_split_product_gives_the_bits_of_the_whole_panel_form(monkeypatch, diag_build):
    """Round 4 (narrow column shards): V H^T with a short reduction range runs as 128 x 32 workgroups, two per x-tile (FactorProductPlan::col_split, kernels_x3.hip NBW = 1),
    when the plan has one K slice.  Every output element sees the same MFMAs in the same order as in the 128 x 64 form: same bits, both against the fp64 oracle."""
    m, n, r, iters = 4000, 200, 64, 30
    V, W0, H0 = problem(m, n, r, np.float32, seed=78)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W0, H0))
    ref = oracle.run("mu", V64, W64, H64, iters)
    got = {}
    for cs in ("0", "1"):
        monkeypatch.setenv("NMFAMD_X3_COLSPLIT", cs)
        eng = na.Engine(m, n, r, "mu")
        eng.upload(V); eng.set_factors(W0, H0)
        eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
        got[cs] = (*eng.get_factors(), eng.frobenius, eng.geometry()["slabs_w"])
        eng.close()
        assert got[cs][3] == 1
        assert rel(got[cs][0], W64) < 2e-4 and rel(got[cs][1], H64) < 2e-4
        assert got[cs][2] == pytest.approx(ref["frobenius"], rel=1e-5)
    assert np.array_equal(got["0"][0], got["1"][0]) and np.array_equal(got["0"][1], got["1"][1]) and got["0"][2] == got["1"][2]
