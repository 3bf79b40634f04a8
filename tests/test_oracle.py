"""CPU tests of the oracle itself: reference-derived golden vectors first, then the
known-answer properties SURVEY.md section 8c lists (the reference ships no fixtures)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_host_vectors.json")


def F(a):
    return np.asfortranarray(a)


def problem(m, n, r, dtype, seed=1):
    rng = np.random.default_rng(seed)
    V = F(rng.random((m, n)).astype(dtype))
    W = F((1.0 - rng.random((m, r))).astype(dtype))  # (0, 1]
    H = F((1.0 - rng.random((r, n))).astype(dtype))
    return V, W, H


# ---------------------------------------------------------------- reference-derived pins

def test_seed_stream_matches_reference_golden():
    g = json.load(open(GOLDEN))
    for seed, values in g["seed_stream"].items():
        got = oracle.seed_stream(int(seed), len(values))
        assert [int(x) for x in got] == values, f"seed {seed}"


def test_summary_best_run_matches_reference_golden():
    g = json.load(open(GOLDEN))
    for case in g["summary"]:
        assert oracle.summary_best_run(case["frobenius"]) == case["bestRun"], case


def test_reference_host_units_agree_live():
    """When the reference tree is present, compare against its code directly (not just fixtures)."""
    ref = oracle.ref_lib()
    if ref is None:
        pytest.skip("reference tree not on this box")
    import ctypes as C
    buf = np.zeros(40, dtype=np.uint32)
    for seed in (9, 77, 123456789):
        ref.ref_seed_stream(C.c_uint32(seed), 40, buf.ctypes.data_as(C.c_void_p))
        assert np.array_equal(buf, oracle.seed_stream(seed, 40))


# ---------------------------------------------------------------- products and kernels

@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-12)])
def test_products_against_numpy(dtype, tol):
    rng = np.random.default_rng(0)
    A = F(rng.random((37, 5)).astype(dtype)); B = F(rng.random((37, 11)).astype(dtype))
    np.testing.assert_allclose(oracle.gemm_tn(A, B), A.T.astype(np.float64) @ B, rtol=tol)
    Hh = F(rng.random((5, 11)).astype(dtype)); Vv = F(rng.random((37, 11)).astype(dtype))
    np.testing.assert_allclose(oracle.gemm_nt(Vv, Hh), Vv.astype(np.float64) @ Hh.T, rtol=tol)
    np.testing.assert_allclose(oracle.gemm_nn(A, Hh), A.astype(np.float64) @ Hh, rtol=tol)


def _product(name, a, b, shape, dims, threads=2):
    """The C entry point itself: leading dimensions from the strides (views allowed), team size pinned."""
    import ctypes as C
    ld = lambda x: x.strides[1] // 4 if x.shape[1] > 1 else max(x.shape[0], 1)
    out = np.zeros(shape, dtype=np.float32, order="F")
    oracle.lib().oracle_set_num_threads(threads)
    getattr(oracle.lib(), name)(*dims, a.ctypes.data_as(C.c_void_p), ld(a), b.ctypes.data_as(C.c_void_p), ld(b), out.ctypes.data_as(C.c_void_p), ld(out))
    return out


# shapes chosen for the packed fp32 products (oracle/sgemm_avx2.h): panel tails (rows not a multiple of 16, columns not a multiple
# of 6), reduction ranges on both sides of the 256 / 128 chunk, the short-and-wide case that is cut along the reduction range
@pytest.mark.parametrize("m,n,r", [(1, 1, 1), (17, 5, 3), (300, 77, 7), (517, 1300, 64), (130, 2049, 33), (2000, 260, 65)])
def test_packed_fp32_products_against_numpy(m, n, r):
    rng = np.random.default_rng(m * 7 + n * 3 + r)
    V = F(rng.random((m, n)).astype(np.float32)); W = F(rng.random((m, r)).astype(np.float32)); H = F(rng.random((r, n)).astype(np.float32))
    d = lambda a: a.astype(np.float64)
    np.testing.assert_allclose(oracle.gemm_tn(W, V), d(W).T @ d(V), rtol=5e-6)
    np.testing.assert_allclose(oracle.gemm_tn(W, W), d(W).T @ d(W), rtol=5e-6)
    np.testing.assert_allclose(oracle.gemm_nt(V, H), d(V) @ d(H).T, rtol=5e-6)
    np.testing.assert_allclose(oracle.gemm_nt(H, H), d(H) @ d(H).T, rtol=5e-6)
    # a view with a leading dimension larger than its row count
    big = F(rng.random((m + 5, n)).astype(np.float32)); Vs = big[2:m + 2, :]
    np.testing.assert_allclose(_product("oracle_gemm_nt_f32", Vs, H, (m, r), (m, n, r)), d(Vs) @ d(H).T, rtol=5e-6)
    np.testing.assert_allclose(_product("oracle_gemm_tn_f32", W, Vs, (r, n), (m, r, n)), d(W).T @ d(Vs), rtol=5e-6)


def test_packed_fp32_products_do_not_depend_on_the_thread_count():
    rng = np.random.default_rng(4)
    V = F(rng.random((900, 1500)).astype(np.float32)); W = F(rng.random((900, 20)).astype(np.float32)); H = F(rng.random((20, 1500)).astype(np.float32))
    got = []
    for threads in (1, 3):       # oracle.gemm_* size the team themselves; here it is pinned
        got.append((_product("oracle_gemm_tn_f32", W, V, (20, 1500), (900, 20, 1500), threads), _product("oracle_gemm_nt_f32", V, H, (900, 20), (900, 1500, 20), threads),
                    _product("oracle_gemm_nt_f32", H, H, (20, 20), (20, 1500, 20), threads)))
    for a, b in zip(*got):
        assert np.array_equal(a, b)


def test_multiply_divide_order_and_eps():
    X = F(np.array([[3.0, 0.0], [1e-30, 2.0]], dtype=np.float32))
    N = F(np.array([[2.0, 5.0], [1e-30, 0.0]], dtype=np.float32))
    D = F(np.array([[4.0, 0.0], [0.0, 1.0]], dtype=np.float32))
    eps = np.float32(np.finfo(np.float32).eps)
    want = (X * N) / (D + eps)  # product first, then the quotient, eps only in the denominator
    got = oracle.multiply_divide(X.copy(order="F"), N, D)
    assert np.array_equal(got, want)
    assert got[0, 1] == 0.0 and got[1, 1] == 0.0


def test_normalize_columns_guard_and_no_h_rescale():
    A = F(np.array([[3.0, 0.0], [4.0, 0.0]], dtype=np.float64))
    oracle.normalize_columns(A)
    np.testing.assert_allclose(A[:, 0], [0.6, 0.8])
    assert np.array_equal(A[:, 1], [0.0, 0.0])  # sum == 0: column left alone


def test_trace_multiplication_both_forms():
    rng = np.random.default_rng(3)
    A = F(rng.random((6, 4))); B = F(rng.random((6, 4)))
    np.testing.assert_allclose(oracle.trace_multiplication(True, A, B), np.diag(A.T @ B))
    S = F(rng.random((4, 4))); Tm = F(rng.random((4, 4)))
    np.testing.assert_allclose(oracle.trace_multiplication(False, S, Tm), np.diag(S @ Tm))


def test_resolve_frobenius_is_sorted_interleaved_and_unclamped():
    vtv = np.sort(np.array([4.0, 1.0, 9.0], dtype=np.float64))
    got = oracle.resolve_frobenius(vtv, [1.0, 0.5, 2.0], [0.25, 1.0])
    assert got == pytest.approx(np.sqrt(14.0 - 2 * 3.5 + 1.25))
    assert np.isnan(oracle.resolve_frobenius(np.array([1.0]), [2.0], [0.0]))  # negative radicand -> NaN


@pytest.mark.parametrize("base", [0, 1])
def test_densify_honours_index_base(base):
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    D = (rng.random((7, 9)) < 0.3) * rng.integers(1, 6, size=(7, 9))
    D = D.astype(np.float32)
    csr = sp.csr_matrix(D); csc = sp.csc_matrix(D); coo = sp.coo_matrix(D)
    got = oracle.densify("csr", 7, 9, csr.data, csr.indptr + base, csr.indices + base, base)
    assert np.array_equal(got, D)
    got = oracle.densify("csc", 7, 9, csc.data, csc.indptr + base, csc.indices + base, base)
    assert np.array_equal(got, D)
    got = oracle.densify("coo", 7, 9, coo.data, coo.row + base, coo.col + base, base)
    assert np.array_equal(got, D)


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-3), (np.float64, 1e-10)])
def test_qr_solves_against_numpy(dtype, tol):
    rng = np.random.default_rng(11)
    r = 9
    B = rng.random((40, r))
    A = (B.T @ B + 0.1 * np.eye(r)).astype(dtype)  # symmetric positive definite normal matrix
    X = rng.random((r, 13)).astype(dtype)
    want = np.linalg.solve(A.astype(np.float64), X.astype(np.float64))
    got = oracle.qr_solve_left(F(A.copy()), F(X.copy()))
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol)
    Y = rng.random((21, r)).astype(dtype)
    want = Y.astype(np.float64) @ np.linalg.inv(A.astype(np.float64))
    got = oracle.qr_solve_right(F(A.copy()), F(Y.copy()))
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol)


# ---------------------------------------------------------------- the MU iteration

def test_reported_error_is_v_minus_wold_hnew():
    """the trace formula evaluates ||V - W_{k-1} H_k||_F."""
    V, W, H = problem(60, 40, 5, np.float64)
    W0 = W.copy(order="F")
    res = oracle.run("mu", V, W, H, 1)
    assert res["iterations"] == 1
    assert res["frobenius"] == pytest.approx(oracle.direct_frobenius(V, W0, H), rel=1e-9)
    assert res["rmsd"] == pytest.approx(res["frobenius"] / np.sqrt(60 * 40))


def test_mu_invariants_and_monotone_history():
    V, W, H = problem(80, 50, 6, np.float64)
    res = oracle.run("mu", V, W, H, 100)
    assert (W >= 0).all() and (H >= 0).all()
    np.testing.assert_allclose(np.linalg.norm(W, axis=0), 1.0, rtol=1e-12)
    frobs = [f for f, _ in res["history"]]
    assert len(frobs) == 10
    assert all(b <= a * (1 + 1e-12) for a, b in zip(frobs, frobs[1:]))


def test_mu_fixed_point_of_exact_factorisation():
    rng = np.random.default_rng(2)
    W = rng.random((30, 4)); W /= np.linalg.norm(W, axis=0)
    H = rng.random((4, 25))
    V = F(W @ H); W = F(W); H = F(H)
    W1, H1 = W.copy(order="F"), H.copy(order="F")
    res = oracle.run("mu", V, W1, H1, 10)
    np.testing.assert_allclose(W1, W, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(H1, H, rtol=1e-9, atol=1e-12)
    assert res["frobenius"] < 1e-6 or np.isnan(res["frobenius"])  # radicand ~ 0, may round negative


def test_mu_rank_one_closed_form():
    V, W, H = problem(20, 15, 1, np.float64)
    w0 = W[:, 0].copy()
    oracle.run("mu", V, W, H, 1)
    eps = np.finfo(np.float64).eps
    h_expected = H_expected = (w0 @ V)  # numerator W^T V
    # h <- h * (w^T v) / ((w^T w) h + eps): computed from the untouched inputs
    _, _, H0 = problem(20, 15, 1, np.float64)
    H_expected = H0[0] * h_expected / ((w0 @ w0) * H0[0] + eps)
    np.testing.assert_allclose(H[0], H_expected, rtol=1e-12)


def test_float_and_double_oracles_agree():
    V, W, H = problem(500, 200, 8, np.float64)  # BASELINE config 1 shape
    V32, W32, H32 = (F(x.astype(np.float32)) for x in (V, W, H))
    r64 = oracle.run("mu", V, W, H, 60)
    r32 = oracle.run("mu", V32, W32, H32, 60)
    assert r32["frobenius"] == pytest.approx(r64["frobenius"], rel=5e-6)
    assert np.linalg.norm(W32 - W) / np.linalg.norm(W) < 5e-4
    assert np.linalg.norm(H32 - H) / np.linalg.norm(H) < 5e-4


def test_threshold_stops_on_an_error_iteration():
    V, W, H = problem(40, 30, 3, np.float64)
    res = oracle.run("mu", V, W, H, 1000, threshold_value=1e-3)
    assert res["iterations"] % 10 == 0 and 20 <= res["iterations"] < 1000
    f = [x for x, _ in res["history"]]
    assert abs(f[-1] - f[-2]) < 1e-3


def test_constant_basis_vectors_leave_w_alone():
    V, W, H = problem(30, 20, 3, np.float64)
    W0 = W.copy(order="F")
    oracle.run("mu", V, W, H, 15, const_w=True)
    assert np.array_equal(W, W0)


# ---------------------------------------------------------------- sibling algorithms

def test_nsnmf_theta_zero_is_mu_and_returns_ws():
    V, W, H = problem(50, 35, 4, np.float64)
    Wm, Hm = W.copy(order="F"), H.copy(order="F")
    a = oracle.run("mu", V, Wm, Hm, 20)
    b = oracle.run("nsnmf", V, W, H, 20, theta=0.0)
    np.testing.assert_allclose(W, Wm, rtol=1e-9); np.testing.assert_allclose(H, Hm, rtol=1e-9)
    assert a["frobenius"] == pytest.approx(b["frobenius"], rel=1e-9)
    # theta > 0: the stored W is W S, whose columns are no longer unit norm
    V, W, H = problem(50, 35, 4, np.float64)
    oracle.run("nsnmf", V, W, H, 20, theta=0.5)
    assert not np.allclose(np.linalg.norm(W, axis=0), 1.0)
    assert (W >= 0).all() and (H >= 0).all()


def test_acls_lambda_zero_is_als():
    V, W, H = problem(45, 30, 4, np.float64)
    Wa, Ha = W.copy(order="F"), H.copy(order="F")
    a = oracle.run("als", V, Wa, Ha, 10)
    b = oracle.run("acls", V, W, H, 10, lambda_w=0.0, lambda_h=0.0)
    np.testing.assert_allclose(W, Wa, rtol=1e-12); np.testing.assert_allclose(H, Ha, rtol=1e-12)
    assert a["frobenius"] == pytest.approx(b["frobenius"], rel=1e-12)


@pytest.mark.parametrize("alg,kw", [
    ("gdcls", dict(lam=0.01)),
    ("als", dict()),
    ("acls", dict(lambda_w=0.01, lambda_h=0.01)),
    ("ahcls", dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01)),
    ("nsnmf", dict(theta=0.5)),
])
def test_sibling_algorithms_reduce_the_error(alg, kw):
    V, W, H = problem(60, 40, 5, np.float64, seed=4)
    res = oracle.run(alg, V, W, H, 40, **kw)
    assert (W >= 0).all() and (H >= 0).all()
    frobs = [f for f, _ in res["history"]]
    assert np.isfinite(frobs).all()
    baseline = np.linalg.norm(V)
    assert frobs[-1] < 0.6 * baseline
    if alg != "nsnmf":
        np.testing.assert_allclose(np.linalg.norm(W, axis=0), 1.0, rtol=1e-9)


def test_ls_first_h_solve_is_the_normal_equations():
    """One ALS iteration: H_1 = max(0, (W0^T W0)^-1 W0^T V)."""
    V, W, H = problem(40, 25, 3, np.float64)
    W0 = W.copy()
    Vc = V.copy()
    # run a single iteration but freeze W so H can be compared cleanly
    oracle.run("als", V, W, H, 1, const_w=True)
    want = np.maximum(np.linalg.solve(W0.T @ W0, W0.T @ Vc), 0.0)
    np.testing.assert_allclose(H, want, rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------- KL extension (literature formula)

def test_kl_update_decreases_the_divergence_and_keeps_invariants():
    rng = np.random.default_rng(6)
    m, n, r = 50, 35, 4
    V = F(rng.random((m, n)) * (rng.random((m, n)) < 0.4))
    W = F(1.0 - rng.random((m, r))); H = F(1.0 - rng.random((r, n)))
    kls = []
    for iters in (10, 20, 40, 80):
        Wc, Hc = W.copy(order="F"), H.copy(order="F")
        res = oracle.run_kl(V, Wc, Hc, iters)
        kls.append(res["kl"])
        assert (Wc >= 0).all() and (Hc >= 0).all()
        np.testing.assert_allclose(np.linalg.norm(Wc, axis=0), 1.0, rtol=1e-12)
    assert all(b <= a + 1e-9 for a, b in zip(kls, kls[1:]))
    # the reported divergence is D(V || W_{k-1} H_k) evaluated directly
    Wp, Hp = W.copy(order="F"), H.copy(order="F")
    oracle.run_kl(V, Wp, Hp, 9)
    Wc, Hc = W.copy(order="F"), H.copy(order="F")
    res = oracle.run_kl(V, Wc, Hc, 10)
    WH = Wp @ Hc
    eps = np.finfo(np.float64).eps
    mask = V > 0
    direct = (V[mask] * np.log(V[mask] / (WH[mask] + eps))).sum() - V.sum() + WH.sum()
    assert res["kl"] == pytest.approx(direct, rel=1e-9)
    assert res["frobenius"] == pytest.approx(np.linalg.norm(V - WH), rel=1e-8)


# ------------------------------------------------------------------ frozen trajectories of the oracle itself

def test_oracle_reproduces_its_frozen_trajectories():
    """tests/golden/oracle_trajectories.json freezes the oracle's own double-precision behaviour (generator committed next
    to it): a change to its blocking / threading / summation order may move results by rounding, not more."""
    import importlib.util
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_trajectories.json")))
    spec = importlib.util.spec_from_file_location("make_oracle_trajectories", os.path.join(os.path.dirname(__file__), "golden", "make_oracle_trajectories.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    for case in gold["cases"]:
        V, W, H = gen.problem(case["seed"])
        res = oracle.run(case["algorithm"], V, W, H, gold["iterations"], **case["parameters"])
        # LS algorithms: explicit solves amplify a changed summation order by cond(W^T W); the multiplicative ones do not
        tol = 1e-11 if case["algorithm"] in ("mu", "nsnmf") else 1e-7
        np.testing.assert_allclose(np.array(res["history"]), np.array(case["history"]), rtol=tol)
        assert W.sum() == pytest.approx(case["w_sum"], rel=tol) and H.sum() == pytest.approx(case["h_sum"], rel=tol)
        assert (W * W).sum() == pytest.approx(case["w_sq"], rel=tol) and (H * H).sum() == pytest.approx(case["h_sq"], rel=tol)
    V, W, H = gen.problem(gold["kl"]["seed"])
    V = np.asfortranarray(np.floor(V * 6.0) * (V > 0.6))
    res = oracle.run_kl(V, W, H, gold["iterations"])
    assert res["kl"] == pytest.approx(gold["kl"]["kl"], rel=1e-11) and res["frobenius"] == pytest.approx(gold["kl"]["frobenius"], rel=1e-11)
    assert W.sum() == pytest.approx(gold["kl"]["w_sum"], rel=1e-11)


def test_kl_over_stored_entries_equals_the_dense_kl_restatement():
    """oracle_kl_run_csr (BASELINE config 3's form) against oracle_kl_run on the densified matrix: zero entries contribute
    nothing, so the two are the same arithmetic up to the order of the inner sums."""
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
        m, n, r = 120, 90, 7
        D = ((rng.random((m, n)) < 0.2) * rng.integers(1, 6, size=(m, n))).astype(dtype)
        D[5, :] = 0; D[:, 11] = 0                                       # an empty row and an empty column
        W = np.asfortranarray((1 - rng.random((m, r))).astype(dtype)); H = np.asfortranarray((1 - rng.random((r, n))).astype(dtype))
        Wa, Ha = W.copy(order="F"), H.copy(order="F")
        a = oracle.run_kl(np.asfortranarray(D), Wa, Ha, 20)
        s = sp.csr_matrix(D)
        Wb, Hb = W.copy(order="F"), H.copy(order="F")
        b = oracle.run_kl_csr(m, n, s.data.astype(dtype), s.indptr, s.indices, Wb, Hb, 20)
        assert np.abs(Wa - Wb).max() <= tol and np.abs(Ha - Hb).max() <= tol * max(1.0, np.abs(Ha).max())
        assert b["kl"] == pytest.approx(a["kl"], rel=max(tol, 1e-12)) and b["frobenius"] == pytest.approx(a["frobenius"], rel=max(tol, 1e-12))
