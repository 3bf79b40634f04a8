"""Round-2 hygiene items, on the GPU: layout choice around the memory-side-cache window, the edges of the split-operand
product (documented divergences from fp32 pinned at kernel level; fallback to the native fp32 MFMA kernel at engine level),
panels longer than grid.y allows, and the round-1 update path (partial Gram matrices) kept green behind its switch."""
import numpy as np
import pytest

import nmfgpu_amd as na
from oracle import oracle

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(a)


def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300)


def problem(m, n, r, dtype=np.float32, seed=1):
    rng = np.random.default_rng(seed)
    return (F(rng.random((m, n)).astype(dtype)), F((1.0 - rng.random((m, r))).astype(dtype)), F((1.0 - rng.random((r, n))).astype(dtype)))


@pytest.fixture(scope="module", autouse=True)
def _library_is_native():
    assert na.device_count() >= 1, "GPU tests need a HIP device"
    assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
    na.set_verbosity(na.Verbosity.Nothing)
    yield
    na.finalize()


def test_one_image_window_follows_the_memory_side_cache(monkeypatch):
    """One resident image of V when it fits the memory-side cache and two do not: 0.6 .. 1.12 x the cache (256 MiB on gfx950,
    a per-architecture table that NMFAMD_MALL_MB overrides), decided once in Engine::allocate."""
    def images(m, n):
        e = na.Engine(m, n, 64, "mu")
        try:
            return e.geometry()["resident_images"]
        finally:
            e.close()
    assert images(7400, 5000) == 2          # 152 MB image: below the window
    assert images(8000, 5000) == 1          # 165 MB: inside (config 2's 207 MB too)
    assert images(10000, 5000) == 1
    assert images(15000, 5000) == 2         # 309 MB: above
    monkeypatch.setenv("NMFAMD_MALL_MB", "128")
    assert images(5000, 5000) == 1          # 105 MB against a 128 MiB cache
    assert images(10000, 5000) == 2
    monkeypatch.setenv("NMFAMD_MALL_MB", "0")
    assert images(10000, 5000) == 2         # unknown cache: no window


def test_split_product_edges_are_the_documented_ones():
    """kernels_x3.hip at its edges: exact wherever every operand is 0 or within 2^-100 .. 2^126; outside
    that the result is NOT the fp32 product -- which is why the engine does not run this kernel on such a V (next test)."""
    rng = np.random.default_rng(0)
    X, Y, r = 256, 320, 64
    A = F(rng.random((X, Y)).astype(np.float32)); Fm = F(rng.random((r, Y)).astype(np.float32) * np.float32(1e-3))
    base = na.op_factor_product_x3(A, Fm)
    exact = Fm.astype(np.float64) @ A.astype(np.float64).T
    assert np.abs(base - exact).max() <= 4e-7 * (np.abs(Fm).astype(np.float64) @ np.abs(A).astype(np.float64).T).max()
    # inside the range, at both ends: still fp32-accurate
    for scale in (np.float32(2.0 ** 120), np.float32(2.0 ** -90)):
        out = na.op_factor_product_x3(F(A * scale), Fm)
        np.testing.assert_allclose(out.astype(np.float64) / float(scale), exact, rtol=2e-6)
    # |v| = 3.4e38 rounds to infinity in the first bf16 cut: fp32 would give 3.4e35-sized finite sums, the split does not
    A2 = A.copy(order="F"); A2[7, 5] = np.float32(3.4e38)
    assert not np.isfinite(na.op_factor_product_x3(A2, Fm)[:, 7]).all()
    assert np.isfinite((Fm.astype(np.float32) @ A2.T)[:, 7]).all()
    # an infinity becomes NaN (inf - inf in the residual) where fp32 keeps the infinity
    A3 = A.copy(order="F"); A3[9, 11] = np.inf
    assert np.isnan(na.op_factor_product_x3(A3, Fm)[:, 9]).any()
    # below 2^-100 the third term falls into the flushed bf16 subnormals: low bits are lost, the value keeps ~2^-8 .. 2^-16
    tiny = np.float32(1.2345678 * 2.0 ** -118)
    A4 = F(np.full((X, Y), tiny, dtype=np.float32))
    out = na.op_factor_product_x3(A4, Fm).astype(np.float64)
    want = Fm.astype(np.float64) @ A4.astype(np.float64).T
    assert np.abs(out / want - 1).max() < 2.0 ** -7


def test_engine_leaves_the_split_product_when_v_is_outside_its_range():
    """Values the exact three-way split does not cover (here: a few entries of 1e-37; infinities and |v| > 2^126 take the same
    route) send the engine -- and nmfgpu::compute -- to the native fp32 MFMA kernel; the result is the fp32 factorisation."""
    m, n, r = 640, 520, 64
    V, W, H = problem(m, n, r, seed=4)
    V[5, 7] = np.float32(1e-37); V[100, 300] = np.float32(3e-35)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, 20)
    eng = na.Engine(m, n, r, "mu")
    assert eng.geometry()["product_kernel"] == 2
    eng.upload(V)
    assert eng.geometry()["product_kernel"] == 0             # recreated on the native fp32 MFMA instructions
    eng.set_factors(W, H)
    eng.iterate(20, last_iteration=20)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4 and eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    Wc, Hc = W.copy(order="F"), H.copy(order="F")
    s = na.Summary()
    assert na.compute(V, Wc, Hc, iterations=20, summary=s) == na.ResultType.Success
    assert rel(Wc, W64) < 2e-4 and rel(Hc, H64) < 2e-4 and s.record(0).frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    # through the team as well (every rank switches, whichever shard holds the odd value)
    Wt, Ht = W.copy(order="F"), H.copy(order="F")
    assert na.compute(V, Wt, Ht, iterations=20, parameters={"numGpus": 2}) == na.ResultType.Success
    assert rel(Wt, W64) < 2e-4 and rel(Ht, H64) < 2e-4
    # ordinary small values stay on the split product
    V2 = F(V * np.float32(1e-20)); V2[5, 7] = 0
    eng2 = na.Engine(m, n, r, "mu"); eng2.upload(V2)
    assert eng2.geometry()["product_kernel"] == 2


def test_panels_longer_than_grid_y_allows():
    """2.2 M rows: the transposes between host layout and panel layout run on a one-dimensional grid (grid.y ends at 65 535
    blocks of 32 rows = 2.09 M rows)."""
    m, n, r = 2_200_000, 48, 3
    rng = np.random.default_rng(2)
    V = F(rng.random((m, n), dtype=np.float32)); W = F(1.0 - rng.random((m, r), dtype=np.float32)); H = F(1.0 - rng.random((r, n), dtype=np.float32))
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    W0, H0 = eng.get_factors()
    assert np.array_equal(W0, W) and np.array_equal(H0, H)          # host -> panel -> host is a pure permutation
    eng.iterate(2, last_iteration=2)
    W2, H2 = eng.get_factors()
    # one MU iteration in fp64 numpy from the same start, for the H step (cheap at n = 48)
    G = W.astype(np.float64).T @ W.astype(np.float64)
    H1 = H * ((W.astype(np.float64).T @ V.astype(np.float64)) / (G @ H + np.finfo(np.float32).eps))
    assert np.isfinite(W2).all() and np.isfinite(H2).all() and (W2 >= 0).all()
    np.testing.assert_allclose(np.linalg.norm(W2.astype(np.float64), axis=0), 1.0, rtol=1e-4)
    eng3 = na.Engine(m, n, r, "mu"); eng3.upload(V); eng3.set_factors(W, H); eng3.iterate(1, error_every=0)
    _, Hone = eng3.get_factors()
    assert rel(Hone, H1) < 1e-5


def test_round1_update_path_with_partial_gram_matrices_still_agrees(monkeypatch):
    """NMFAMD_GRAM_PARTIALS=1 keeps the 64-column update kernel and the partial-Gram passengers (the path the native fp32 MFMA
    product still uses)."""
    monkeypatch.setenv("NMFAMD_GRAM_PARTIALS", "1")
    V, W, H = problem(900, 700, 64, seed=6)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run("mu", V64, W64, H64, 30)
    eng = na.Engine(900, 700, 64, "mu"); eng.upload(V); eng.set_factors(W, H)
    eng.iterate(30, last_iteration=30)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4 and eng.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)


def test_geometry_getter_never_writes_past_the_callers_struct():
    """nmfamd_geometry grew at its end (one_pass); nmfamd_engine_geometry_sized writes min(struct_size, sizeof) bytes."""
    import ctypes as C
    from nmfgpu_amd._lib import library
    eng = na.Engine(300, 200, 8, "mu")
    lib = library()
    buf = (C.c_ubyte * 96)(*([0xAB] * 96))
    assert lib.nmfamd_engine_geometry_sized(eng._h, buf, C.c_ulong(56)) == 0          # a caller compiled against round 2's 56-byte struct
    assert all(b == 0xAB for b in bytes(buf)[56:]) and bytes(buf)[:4] == (300).to_bytes(4, "little")
    assert lib.nmfamd_engine_geometry_sized(eng._h, buf, C.c_ulong(4)) != 0
    g = eng.geometry()
    assert g["m"] == 300 and g["n"] == 200 and g["padded_rank"] == 64
    eng.close()
