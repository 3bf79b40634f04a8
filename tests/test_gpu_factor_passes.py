"""Padded rank 256 with bf16 product operands (BASELINE config 4): the passes between the update of a factor panel and the next
product that streams it (nmfgpu_amd/csrc/kernels_tri.hip) against numpy restatements of what the reference computes there:
kernel::normalizeColumns (source/kernels/KernelNormalizeColumns.cu:37-58), the smoothing product with S
(source/nmf/AlgorithmNonSmoothNMF.h:131-134,175,194) and syrk of the smoothed matrix (:176,196)."""
import numpy as np
import pytest

import nmfgpu_amd as na

pytestmark = pytest.mark.gpu


def _round_bf16(a):
    """fp32 -> bf16, round to nearest even, widened back (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _smooth(P, theta):
    r = P.shape[1]
    S = (1.0 - theta) * np.eye(r) + theta / r * np.ones((r, r))
    return P.astype(np.float64) @ S


@pytest.mark.parametrize("length,r,theta,normalise", [(1000, 256, 0.5, True), (777, 200, 0.4, True), (50, 129, 0.0, False), (4096, 256, 0.7, False),
                                                      (33, 256, 0.5, True)])
def test_factor_passes_against_numpy(length, r, theta, normalise):
    rng = np.random.default_rng(length + r)
    P = rng.random((length, r), dtype=np.float32) + 0.01
    P[:, 3] = 0.0                                         # an all-zero column keeps its values (sum > 0 ? x / sqrt(sum) : x)
    colsq = (P.astype(np.float64) ** 2).sum(axis=0).astype(np.float32) if normalise else None
    out = na.op_factor_passes(P, theta=theta, colsq=colsq)

    want = P.copy()
    if normalise:
        nrm = np.sqrt(colsq)
        for c in range(r):
            if colsq[c] > 0:
                want[:, c] = P[:, c] / nrm[c]             # fp32 division by the fp32 root, as the kernel
    assert np.array_equal(out["panel"], want), "normalised panel differs from the fp32 division"

    sm = _smooth(out["panel"], theta)
    # the operand of the product: the smoothed value rounded ONCE to bf16 (the fp32 smoothing in between may move a value that sits on a
    # rounding boundary to the neighbour: one bf16 ulp = 2^-8 relative)
    pack = out["pack"].astype(np.float64)
    assert np.all(np.abs(pack - sm) <= 2.0 ** -8 * np.abs(sm) + 1e-30)
    exact = _round_bf16(sm.astype(np.float32))
    assert np.mean(pack == exact) > 0.99

    G = out["panel"].astype(np.float64).T @ out["panel"].astype(np.float64)
    scale = np.sqrt(np.outer(np.diag(G), np.diag(G))) + 1e-30
    assert np.max(np.abs(out["gram"] - G) / np.maximum(scale, 1e-30)) < 4e-7      # fp32-level: the split-operand product's bound
    assert np.array_equal(out["gram"], out["gram"].T), "Gram matrix must be exactly symmetric"
    Gs = sm.T @ sm
    assert np.max(np.abs(out["gram_smoothed"] - Gs) / np.max(np.abs(Gs))) < 1e-6
    assert np.array_equal(out["gram_smoothed"], out["gram_smoothed"].T)


def test_theta_zero_is_the_identity():
    rng = np.random.default_rng(5)
    P = rng.random((300, 256), dtype=np.float32)
    out = na.op_factor_passes(P, theta=0.0)
    assert np.array_equal(out["panel"], P)
    assert np.array_equal(out["pack"], _round_bf16(P))
    assert np.array_equal(out["gram"], out["gram_smoothed"])


def test_rejected_shapes():
    with pytest.raises(na.EngineError):
        na.op_factor_passes(np.ones((64, 64), np.float32))          # padded rank 64: other kernels serve it


# ------------------------------------------------------------------ the update that leaves the next product's operands behind
@pytest.mark.parametrize("length,r,theta,transform,scaled,frag_theta,den", [(1000, 256, 0.5, False, True, 0.0, False), (33, 200, 0.0, False, False, 0.0, False),
                                                                            (777, 256, 0.4, True, True, 0.0, False), (33024, 256, 0.5, False, True, 0.0, False),
                                                                            (33024, 256, 0.3, True, False, 0.0, False), (4096, 129, 0.6, True, True, 0.5, True),
                                                                            (6250, 256, 0.5, True, True, 0.5, True), (500, 256, 0.0, True, True, 0.0, True)])
def test_update_with_pending_scale_fragments_and_output_side_smoothing(length, r, theta, transform, scaled, frag_theta, den):
    """PanelTriExtras (csrc/kernels.h): the old values carry a pending column scale (kernel::normalizeColumns as a factor,
    KernelNormalizeColumns.cu:37-58), the numerator rows get scale + smoothing (S D (W^T V) = (W D S)^T V, AlgorithmNonSmoothNMF.h:174-178),
    the new rows are written unnormalised with their bf16 fragments (the H update smooths them on the way: S H, :194); 33 024 rows without
    a numerator transform take the 128-row kernel (k_panel_update_rows_mu), the others the 32-row one (k_panel_update_wide_f32)."""
    rng = np.random.default_rng(length + r)
    P = rng.random((length, r), dtype=np.float32) + 0.01
    num = rng.random((length, r), dtype=np.float32) * 3.0
    A = rng.random((r, 40), dtype=np.float32)
    Q = (A @ A.T).astype(np.float32)
    P[:, 5] = 0.0                                                     # a zero column stays zero and keeps scale 1
    ocs = (0.5 + 8.0 * rng.random(r, dtype=np.float32)) if scaled else None
    ncs = (0.5 + 8.0 * rng.random(r, dtype=np.float32)) if transform else None
    if ncs is not None:
        ncs[7] = 0.0                                                  # no norm: factor 1
    out = na.op_tri_update(P, num, Q, old_colsq=ocs, transform_num=transform, num_colsq=ncs, theta=theta, frag_theta=frag_theta, transform_den=den)

    d = np.float64
    factor = lambda s: np.where(s > 0, np.float32(1.0) / np.sqrt(np.where(s > 0, s, np.float32(1.0))), np.float32(1.0)).astype(d)
    old = P.astype(d) * (factor(ocs) if scaled else 1.0)
    nm = num.astype(d)
    if transform:
        x = nm * factor(ncs)
        nm = (1.0 - theta) * x + (theta / r) * x.sum(axis=1, keepdims=True)
    Qd = Q.astype(d)
    if den:          # S D Q D S around the product (the H step: Q = W^T W as the Gram reduction leaves it, D its pending scale, AlgorithmNonSmoothNMF.h:176-178)
        Sm = (1.0 - theta) * np.eye(r) + (theta / r) * np.ones((r, r))
        Dm = np.diag(factor(ncs)) if ncs is not None else np.eye(r)
        Qd = Sm @ Dm @ Qd @ Dm @ Sm
    want = old * nm / (old @ Qd + np.finfo(np.float32).eps)
    assert _rel(out["panel"], want) < 2e-6
    assert np.max(np.abs(out["panel"] - want) / (np.abs(want) + 1e-30)) < 2e-5
    if frag_theta == 0.0:
        assert np.array_equal(out["pack"], _round_bf16(out["panel"])), "fragments are the bf16 rounding of the rows the kernel wrote"
    else:
        sm = _smooth(out["panel"], frag_theta)
        assert np.all(np.abs(out["pack"].astype(d) - sm) <= 2.0 ** -8 * np.abs(sm) + 1e-30)
        assert np.mean(out["pack"] == _round_bf16(sm.astype(np.float32))) > 0.99
    sq = (out["panel"].astype(d) ** 2).sum(axis=0)
    scale = np.where(sq > 0, 1.0 / np.sqrt(np.where(sq > 0, sq, 1.0)), 1.0)
    np.testing.assert_allclose(out["scale"], scale, rtol=2e-6)
    assert out["scale"][5] == 1.0
    G = (out["pack"].astype(d) * scale).T @ (out["pack"].astype(d) * scale)
    assert np.max(np.abs(out["gram"] - G)) < 3e-6 * max(1.0, np.max(np.abs(G)))   # fp32 accumulation over `length` rows
    assert np.array_equal(out["gram"], out["gram"].T)
    # the reduction that leaves the unscaled matrix, its split image (three bf16 planes: exact to ~2^-24 relative) and the diagonal
    Graw = out["pack"].astype(d).T @ out["pack"].astype(d)
    assert np.max(np.abs(out["gram_raw"] - Graw)) < 3e-6 * np.max(np.abs(Graw))
    assert np.array_equal(out["gram_raw"], out["gram_raw"].T)
    assert np.array_equal(out["diag"], np.diag(out["gram_raw"]))
    assert np.max(np.abs(out["gram_image"].astype(d) - out["gram_raw"].astype(d))) <= 2.0 ** -22 * np.max(np.abs(Graw)) + 3e-6 * np.max(np.abs(Graw))


# ------------------------------------------------------------------ long panels: 64 rows per workgroup in the wide update kernel
from oracle import oracle  # noqa: E402


def _F(a):
    return np.asfortranarray(a)


def _rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300)


@pytest.mark.parametrize("m,n,r,alg,kw", [(33000, 150, 130, "mu", {}), (150, 33000, 130, "mu", {}), (33000, 140, 256, "nsnmf", dict(theta=0.5)),
                                           (33000, 150, 100, "mu", {})])
def test_long_panels_match_the_oracle(m, n, r, alg, kw):
    """Panels of >= 32 768 rows at padded ranks 128 / 256 take k_panel_update_wide64_mu (kernels_wide.hip): the W update when
    m is long, the H update (with the per-column error terms) when n is.  Same tolerances as every fp32 engine test."""
    rng = np.random.default_rng(m + n + r)
    V = _F(rng.random((m, n)).astype(np.float32))
    W = _F((1.0 - rng.random((m, r))).astype(np.float32))
    H = _F((1.0 - rng.random((r, n))).astype(np.float32))
    iters = 10
    V64, W64, H64 = (_F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, **kw)          # (updates W64 / H64 in place)
    eng = na.Engine(m, n, r, alg, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert _rel(Wg, W64) < 2e-4 and _rel(Hg, H64) < 2e-4
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-4)
    if alg == "mu":
        np.testing.assert_allclose(np.linalg.norm(Wg.astype(np.float64), axis=0), 1.0, rtol=1e-5)


@pytest.mark.parametrize("m,n", [(33000, 140), (500, 420), (150, 33000)])
def test_rank256_bf16_path_tracks_fp32(m, n):
    """nsNMF at r = 256 with bf16 product operands (kernels_tri.hip: pending column scale, scale and smoothing around the H update's products, fragments
    written by the update kernels, Gram matrices from the fragments; long W: the 128-row update kernel, long H: the 32-row one with its transforms)
    against the fp64 oracle within the bf16 mode's stated 2e-2, error terms included."""
    r, theta, iters = 256, 0.5, 10
    rng = np.random.default_rng(m)
    V = _F(rng.random((m, n)).astype(np.float32))
    W = _F((1.0 - rng.random((m, r))).astype(np.float32))
    H = _F((1.0 - rng.random((r, n))).astype(np.float32))
    V64, W64, H64 = (_F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("nsnmf", V64, W64, H64, iters, theta=theta)
    eng = na.Engine(m, n, r, "nsnmf", theta=theta, precision="bf16")
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert _rel(Wg, W64) < 2e-2 and _rel(Hg, H64) < 2e-2
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=2e-3)
    # the same run twice: bit-identical (no atomics, fixed summation orders)
    eng2 = na.Engine(m, n, r, "nsnmf", theta=theta, precision="bf16")
    eng2.upload(V); eng2.set_factors(W, H)
    eng2.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    W2, H2 = eng2.get_factors()
    assert np.array_equal(Wg, W2) and np.array_equal(Hg, H2)


@pytest.mark.parametrize("m,n", [(33000, 140), (700, 1900)])
def test_rank256_gram_matrices_as_passengers_of_the_products(m, n, monkeypatch, diag_build):
    """Round 4: at padded rank 256 the Gram matrix of the operand a product multiplies V with (W^T W from W's fragments, (S H)(S H)^T from the smoothed H's) rides in
    that product's launch as 32 passenger workgroups -- 16 K slices x 2 halves of the tiles; the last workgroup of a half to arrive adds the slices in slice order and
    writes the matrix, its diagonal and its split image (tri_gram_tile.h).  NMFAMD_TRI_RIDE = 0 / h / w (measurement build: fixture diag_build) keeps both / one of them on their own launches
    (k_gram_tri_bf16 + k_gram_tri_reduce_image): same factors and reported errors up to the summation order of the K slices, every form within the bf16 mode's 2e-2 of
    the fp64 oracle, and the riding form bit-identical when the run is repeated (the order of the sum does not depend on who arrives last)."""
    r, theta, iters = 256, 0.5, 20
    rng = np.random.default_rng(5 + m)
    V = _F(rng.random((m, n)).astype(np.float32))
    W = _F((1.0 - rng.random((m, r))).astype(np.float32))
    H = _F((1.0 - rng.random((r, n))).astype(np.float32))
    V64, W64, H64 = (_F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("nsnmf", V64, W64, H64, iters, theta=theta)
    got = {}
    for ride in ("0", "h", "w", None, None):
        if ride is None:
            monkeypatch.delenv("NMFAMD_TRI_RIDE", raising=False)
        else:
            monkeypatch.setenv("NMFAMD_TRI_RIDE", ride)
        eng = na.Engine(m, n, r, "nsnmf", theta=theta, precision="bf16")
        eng.upload(V); eng.set_factors(W, H)
        eng.iterate(10, first_iteration=1, error_every=10)
        f10 = eng.frobenius
        eng.iterate(10, first_iteration=11, error_every=10, last_iteration=iters)
        Wg, Hg = eng.get_factors()
        key = ride if ride is not None else ("both" if "both" not in got else "both again")
        got[key] = (Wg, Hg, f10, eng.frobenius)
        eng.close()
        assert _rel(Wg, W64) < 2e-2 and _rel(Hg, H64) < 2e-2, key
        assert got[key][3] == pytest.approx(ref["frobenius"], rel=2e-3)
    # (the forms differ in the order of fp32 sums -- K slices of the Gram matrices, and 8 instead of 9 K slices of W^T V when W^T W rides -- and a last-bit difference
    #  flips the bf16 rounding of an operand here and there: 2e-5 after 20 iterations, against the mode's 2e-2)
    for key in ("h", "w", "both"):
        assert _rel(got[key][0], got["0"][0]) < 2e-4 and _rel(got[key][1], got["0"][1]) < 2e-4, key
        assert got[key][2] == pytest.approx(got["0"][2], rel=1e-5) and got[key][3] == pytest.approx(got["0"][3], rel=1e-5)
    assert np.array_equal(got["both"][0], got["both again"][0]) and np.array_equal(got["both"][1], got["both again"][1]) and got["both"][3] == got["both again"][3]


def test_rank256_bf16_path_factors_read_in_the_middle_of_a_run():
    """get_factors() folds the pending column scale into W (and drops the fragments and the Gram image made from the unscaled panel); the run
    continues from the normalised W.  Same trajectory as the uninterrupted run up to the bf16 mode's rounding (bf16 of W D instead of bf16 of W)."""
    m, n, r, theta = 33000, 150, 256, 0.5
    rng = np.random.default_rng(77)
    V = _F(rng.random((m, n)).astype(np.float32))
    W = _F((1.0 - rng.random((m, r))).astype(np.float32))
    H = _F((1.0 - rng.random((r, n))).astype(np.float32))
    out = []
    for stops in ((10,), (3, 4, 3)):
        eng = na.Engine(m, n, r, "nsnmf", theta=theta, precision="bf16")
        eng.upload(V); eng.set_factors(W, H)
        it = 1
        for k in stops:
            eng.iterate(k, first_iteration=it, error_every=10, last_iteration=10)
            it += k
            Wg, Hg = eng.get_factors()
            np.testing.assert_allclose(np.linalg.norm(_smooth_inverse(Wg, theta), axis=0), 1.0, rtol=1e-4)      # W S is returned; W itself has unit columns
        out.append((Wg, Hg, eng.frobenius))
        eng.close()
    assert _rel(out[1][0], out[0][0]) < 2e-3 and _rel(out[1][1], out[0][1]) < 2e-3
    assert out[1][2] == pytest.approx(out[0][2], rel=1e-3)


def _smooth_inverse(WS, theta):
    """W from the W S that get_factors returns for nsNMF (S = (1 - theta) I + theta / r 1 1^T, inverted in fp64)."""
    r = WS.shape[1]
    S = (1.0 - theta) * np.eye(r) + theta / r * np.ones((r, r))
    return WS.astype(np.float64) @ np.linalg.inv(S)


@pytest.mark.parametrize("alg,r,kw,const_w", [("mu", 200, {}, False), ("nsnmf", 256, dict(theta=0.3), True), ("mu", 256, {}, True)])
def test_rank256_bf16_path_other_entries(alg, r, kw, const_w):
    """The same fused passes serve the multiplicative update (theta = 0: S = I) and constant basis vectors (W never written, the
    passes over W run once) -- against the fp64 oracle within the bf16 mode's tolerance."""
    m, n, iters = 420, 390, 10
    rng = np.random.default_rng(r + int(const_w))
    V = _F(rng.random((m, n)).astype(np.float32))
    W = _F((1.0 - rng.random((m, r))).astype(np.float32))
    H = _F((1.0 - rng.random((r, n))).astype(np.float32))
    V64, W64, H64 = (_F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, const_w=const_w, **kw)
    eng = na.Engine(m, n, r, alg, precision="bf16", **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters, constant_w=const_w)
    Wg, Hg = eng.get_factors()
    assert _rel(Hg, H64) < 2e-2 and _rel(Wg, W64) < 2e-2
    assert eng.frobenius == pytest.approx(ref["frobenius"], rel=2e-3)
