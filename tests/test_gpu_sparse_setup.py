"""The CSR and CSC images of a sparse V built on the device (round 6: csrc/kernels_sparse_setup.hip, Engine::upload_sparse_device) against the host path they
replace (NMFAMD_SPARSE_SETUP=host; rounds 1 - 5: stable counting sorts on the host, 456 ms at config 3) -- bit for bit: pointers, indices, values, the blocked
boundary pointers of the KL gather, and the factors and error values that follow.

Reference: the device-side conversions of source/common/Matrix.h:145-232 (cuSPARSE csr2dense / csc2dense / xcoo2csr honouring IndexBase); duplicates stay separate
entries whose contributions add (as the densifying path adds them), entries outside the matrix are dropped."""
import numpy as np
import pytest

import nmfgpu_amd as na

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(a)


def triplets(m, n, density, seed, duplicates=0):
    rng = np.random.default_rng(seed)
    nnz = max(1, int(m * n * density))
    lin = rng.choice(m * n, size=nnz, replace=False)
    rows, cols = (lin // n).astype(np.int32), (lin % n).astype(np.int32)
    if duplicates:
        d = rng.integers(0, nnz, size=duplicates)
        rows, cols = np.concatenate([rows, rows[d]]), np.concatenate([cols, cols[d]])
    vals = rng.integers(1, 6, size=len(rows)).astype(np.float32) + rng.random(len(rows)).astype(np.float32)
    return rows, cols, vals


def as_format(fmt, rows, cols, vals, m, n, base, order):
    """fmt 1 CSR / 2 CSC / 3 COO arrays of the same entries.  order: "sorted" (ascending inner index), "unsorted" (entries of an outer index in random order),
    COO: "sorted" = row-major order, "unsorted" = random order."""
    rng = np.random.default_rng(7)
    if fmt == 3:
        perm = np.lexsort((cols, rows)) if order == "sorted" else rng.permutation(len(vals))
        return vals[perm], (rows[perm] + base).astype(np.int32), (cols[perm] + base).astype(np.int32)
    outer, inner, count = (rows, cols, m) if fmt == 1 else (cols, rows, n)
    tie = inner if order == "sorted" else rng.random(len(vals))
    perm = np.lexsort((tie, outer))
    ptr = np.zeros(count + 1, dtype=np.int64)
    np.add.at(ptr, outer + 1, 1)
    ptr = np.cumsum(ptr)
    return vals[perm], (ptr + base).astype(np.int32), (inner[perm] + base).astype(np.int32)


def images(eng, nnz, m, n):
    g = eng.geometry()
    out = {"csr_val": eng.debug_read(12, nnz), "csc_val": eng.debug_read(13, nnz)}
    for name, which, count in (("csr_ptr", 14, m + 1), ("csr_idx", 15, nnz), ("csc_ptr", 16, n + 1), ("csc_idx", 17, nnz)):
        out[name] = eng.debug_read(which, count).view(np.int32)
    if g["kl_blocks_w"] > 1:
        out["csr_bptr"] = eng.debug_read(18, m * (g["kl_blocks_w"] + 1)).view(np.int32)
    if g["kl_blocks_h"] > 1:
        out["csc_bptr"] = eng.debug_read(19, n * (g["kl_blocks_h"] + 1)).view(np.int32)
    return out


def build(fmt, arrays, base, m, n, r, W, H, host, monkeypatch, kl=False, iters=6):
    if host:
        monkeypatch.setenv("NMFAMD_SPARSE_SETUP", "host")
    else:
        monkeypatch.delenv("NMFAMD_SPARSE_SETUP", raising=False)
    eng = na.Engine(m, n, r, "mu", sparse_compute=True, **({"divergence": "kl"} if kl else {}))
    eng.upload_sparse(fmt, *arrays, base)
    assert eng.geometry()["sparse_setup"] == (0 if host else 1)
    eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=3, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    return eng, Wg, Hg


@pytest.mark.parametrize("fmt", [1, 2, 3])
@pytest.mark.parametrize("base", [0, 1])
@pytest.mark.parametrize("order", ["sorted", "unsorted"])
def test_device_built_images_are_the_host_built_images(fmt, base, order, monkeypatch):
    """Every input format, both index bases, entries in and out of order, duplicate coordinates: the same images, the same factors, the same error."""
    m, n, r = 517, 389, 12
    rows, cols, vals = triplets(m, n, 0.07, seed=10 * fmt + base, duplicates=40)
    arrays = as_format(fmt, rows, cols, vals, m, n, base, order)
    rng = np.random.default_rng(3)
    W, H = F((1.0 - rng.random((m, r))).astype(np.float32)), F((1.0 - rng.random((r, n))).astype(np.float32))
    dev, Wd, Hd = build(fmt, arrays, base, m, n, r, W, H, False, monkeypatch)
    host, Wh, Hh = build(fmt, arrays, base, m, n, r, W, H, True, monkeypatch)
    a, b = images(dev, len(vals), m, n), images(host, len(vals), m, n)
    for key in b:
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(Wd, Wh) and np.array_equal(Hd, Hh)
    assert dev.frobenius == host.frobenius
    # ... and the image is what it should be: CSR rows ascend in their column indices, CSC columns in their row indices, every entry once
    ptr, idx = a["csr_ptr"], a["csr_idx"]
    assert ptr[0] == 0 and ptr[-1] == len(vals) and all(np.all(np.diff(idx[ptr[i]:ptr[i + 1]]) >= 0) for i in range(m))
    ptr, idx = a["csc_ptr"], a["csc_idx"]
    assert ptr[0] == 0 and ptr[-1] == len(vals) and all(np.all(np.diff(idx[ptr[j]:ptr[j + 1]]) >= 0) for j in range(n))
    assert np.isclose(a["csr_val"].astype(np.float64).sum(), vals.astype(np.float64).sum(), rtol=1e-6)


def test_kl_update_with_blocked_gather_from_device_built_images(monkeypatch):
    """The KL update with its L2-sized blocks (forced small): boundary pointers from the device kernel = the host walk's; factors and divergence bit for bit."""
    monkeypatch.setenv("NMFAMD_KL_BLOCK_KB", "512")
    m, n, r = 30000, 14000, 20                  # 7.7 MB of W, 3.6 MB of H at padded rank 64: both half-steps are blocked (above 3 MiB of factor)
    rows, cols, vals = triplets(m, n, 0.002, seed=5)
    arrays = as_format(1, rows, cols, vals, m, n, 0, "sorted")
    rng = np.random.default_rng(4)
    W, H = F((1.0 - rng.random((m, r))).astype(np.float32)), F((1.0 - rng.random((r, n))).astype(np.float32))
    dev, Wd, Hd = build(1, arrays, 0, m, n, r, W, H, False, monkeypatch, kl=True)
    host, Wh, Hh = build(1, arrays, 0, m, n, r, W, H, True, monkeypatch, kl=True)
    assert dev.geometry()["kl_blocks_h"] > 1 and dev.geometry()["kl_blocks_w"] > 1
    a, b = images(dev, len(vals), m, n), images(host, len(vals), m, n)
    assert set(a) == set(b) and "csc_bptr" in a and "csr_bptr" in a
    for key in b:
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(Wd, Wh) and np.array_equal(Hd, Hh)
    assert dev.kl_divergence == host.kl_divergence and dev.frobenius == host.frobenius


def test_inputs_the_device_path_hands_back(monkeypatch):
    """An entry outside the matrix (the host path drops it), a column longer than the LDS sort takes, an empty matrix: the host path builds the images
    (geometry()["sparse_setup"] == 0) and the result is the one of the well-formed input."""
    monkeypatch.delenv("NMFAMD_SPARSE_SETUP", raising=False)
    m, n, r = 300, 200, 6
    rows, cols, vals = triplets(m, n, 0.1, seed=2)
    rng = np.random.default_rng(9)
    W, H = F((1.0 - rng.random((m, r))).astype(np.float32)), F((1.0 - rng.random((r, n))).astype(np.float32))
    good = na.Engine(m, n, r, "mu", sparse_compute=True)
    good.upload_sparse(3, vals, rows, cols, 0)
    assert good.geometry()["sparse_setup"] == 1
    good.set_factors(W, H); good.iterate(4, last_iteration=4)
    bad = na.Engine(m, n, r, "mu", sparse_compute=True)
    bad.upload_sparse(3, np.concatenate([vals, [9.0]]).astype(np.float32), np.concatenate([rows, [m + 5]]).astype(np.int32), np.concatenate([cols, [1]]).astype(np.int32), 0)
    assert bad.geometry()["sparse_setup"] == 0
    bad.set_factors(W, H); bad.iterate(4, last_iteration=4)
    for x, y in zip(good.get_factors(), bad.get_factors()):
        assert np.array_equal(x, y)
    # a dense column of 9 000 entries: longer than 8 192
    m2 = 9000
    rows2 = np.concatenate([np.arange(m2), rng.integers(0, m2, 500)]).astype(np.int32)
    cols2 = np.concatenate([np.full(m2, 3), rng.integers(0, n, 500)]).astype(np.int32)
    vals2 = (1.0 + rng.random(len(rows2))).astype(np.float32)
    long_col = na.Engine(m2, n, r, "mu", sparse_compute=True)
    long_col.upload_sparse(3, vals2, rows2, cols2, 0)
    assert long_col.geometry()["sparse_setup"] == 0
    W2 = F((1.0 - rng.random((m2, r))).astype(np.float32))
    long_col.set_factors(W2, H); long_col.iterate(2, last_iteration=2)
    assert np.isfinite(long_col.frobenius)


def test_kl_iteration_with_the_pending_column_scale_equals_normalising_every_iteration(monkeypatch):
    """Round 6: the KL iteration leaves W unnormalised with its column scale as a pending factor (no pass over the panel per iteration); NMFAMD_NO_FUSED_MU=1 keeps
    the normalising pass.  Same arithmetic up to where the factor 1 / sqrt(sum) is applied: factors, divergence and Frobenius error agree to fp32 rounding, the
    columns of the downloaded W are unit vectors, and stepping one iteration at a time with a download after each step (which folds the scale in) ends in the same place."""
    monkeypatch.delenv("NMFAMD_SPARSE_SETUP", raising=False)
    m, n, r, iters = 3000, 900, 70, 12
    rows, cols, vals = triplets(m, n, 0.02, seed=77)
    arrays = as_format(1, rows, cols, vals, m, n, 0, "sorted")
    rng = np.random.default_rng(8)
    W, H = F((1.0 - rng.random((m, r))).astype(np.float32)), F((1.0 - rng.random((r, n))).astype(np.float32))

    def run(stepwise=False):
        eng = na.Engine(m, n, r, "mu", divergence="kl")
        eng.upload_sparse(1, *arrays, 0)
        eng.set_factors(W, H)
        if stepwise:
            for k in range(1, iters + 1):
                eng.iterate(1, first_iteration=k, error_every=4, last_iteration=iters)
                Wk, _ = eng.get_factors()
                np.testing.assert_allclose(np.linalg.norm(Wk, axis=0), 1.0, rtol=1e-5)
        else:
            eng.iterate(iters, first_iteration=1, error_every=4, last_iteration=iters)
        return eng.get_factors() + (eng.kl_divergence, eng.frobenius)

    Wp, Hp, klp, fp = run()
    Ws, Hs, kls, fs = run(stepwise=True)
    monkeypatch.setenv("NMFAMD_NO_FUSED_MU", "1")
    Wn, Hn, kln, fn = run()
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))
    assert rel(Wp, Wn) < 2e-5 and rel(Hp, Hn) < 2e-5 and rel(Ws, Wn) < 2e-5 and rel(Hs, Hn) < 2e-5
    assert klp == pytest.approx(kln, rel=1e-5) and fp == pytest.approx(fn, rel=1e-5) and kls == pytest.approx(kln, rel=1e-5)
    np.testing.assert_allclose(np.linalg.norm(Wp, axis=0), 1.0, rtol=1e-5)
