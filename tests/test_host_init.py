"""Host-side initialisers (k-means, KMeans*, EIn-NMF, MeanColumns): the engine's C++ against the
oracle's thread-by-thread restatement of the reference kernels.  No GPU, no context: these entry
points (include/nmfgpu_amd.h, nmfamd_host_*) only touch host memory.

Bit-exact is the bar: both sides keep the reference's summation orders (32 lane partials with fma,
xor butterfly; members in ascending column order; Hillis-Steele scan in double)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from nmfgpu_amd import engine as eng
from nmfgpu_amd.api import NmfInitializationMethod as Init
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blobs(m, n, k, dtype, seed):
    rs = np.random.RandomState(seed)
    centres = rs.random_sample((m, k)) * 4
    X = centres[:, rs.randint(0, k, n)] + 0.3 * rs.random_sample((m, n))
    return np.asfortranarray(X.astype(dtype))


# ------------------------------------------------------------------ the oracle on its own

def test_oracle_kmeans_fixed_point_properties():
    X = blobs(37, 400, 6, np.float64, 0)
    C_, memb, passes = oracle.kmeans(X, 6, seed=3, iterations=100, threshold=0.0)
    assert passes < 100 and memb.max() < 6
    # converged (threshold 0): every centre is the mean of its members, every column sits with its nearest centre
    for c in range(6):
        if (memb == c).any():
            np.testing.assert_allclose(C_[:, c], X[:, memb == c].mean(axis=1), rtol=1e-12)
    d = ((X[:, :, None] - C_[:, None, :]) ** 2).sum(axis=0)
    assert (d.argmin(axis=1) == memb).all()


def test_oracle_kmeans_forgy_start_and_single_pass():
    X = blobs(8, 50, 3, np.float32, 1)
    # one pass: centres are the means of the columns assigned against the Forgy start, which is k distinct columns
    C_, memb, passes = oracle.kmeans(X, 3, seed=7, iterations=1, threshold=0.0)
    assert passes == 1
    # the final assignment pass (kMeans.cu:262-270) runs against the updated centres
    d = ((X.astype(np.float64)[:, :, None] - C_.astype(np.float64)[:, None, :]) ** 2).sum(axis=0)
    assert (d.argmin(axis=1) == memb).mean() > 0.95   # float ties aside


def test_oracle_kmeans_first_minimum_wins_and_empty_cluster_keeps_its_centre():
    # all columns identical: every centre starts as the same vector, distance ties -> cluster 0 takes all,
    # the others stay empty and keep their (identical) Forgy centres (kMeans.cu:66-71, :89-93)
    X = np.asfortranarray(np.tile(np.arange(1, 6, dtype=np.float32)[:, None], (1, 20)))
    C_, memb, passes = oracle.kmeans(X, 4, seed=0, iterations=10, threshold=0.0)
    assert (memb == 0).all() and passes == 2   # pass 0 changes everything, pass 1 changes nothing
    assert (C_ == X[:, :4]).all()


def test_oracle_einnmf_closed_form_small():
    rs = np.random.RandomState(2)
    V = np.asfortranarray(rs.random_sample((10, 7)))
    W = np.asfortranarray(rs.random_sample((10, 5)))
    H = oracle.einnmf_h(V, W)
    d = ((W[:, :, None] - V[:, None, :]) ** 2).sum(axis=0)                   # r x n
    expect = 1.0 / (d * np.cumsum(1.0 / (d + 1e-9), axis=0) + 1e-9)
    np.testing.assert_allclose(H, expect, rtol=1e-12)


# ------------------------------------------------------------------ engine host code against the oracle

@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m,n,k", [(1, 40, 3), (31, 64, 5), (33, 257, 8), (90, 150, 5), (200, 1000, 33), (1030, 300, 17)])
def test_host_kmeans_matches_oracle_bit_for_bit(dtype, m, n, k):
    X = blobs(m, n, k, dtype, seed=m + n)
    for seed, iters, thr in ((1, 100, 0.005), (9, 3, 0.0)):
        C_o, memb_o, it_o = oracle.kmeans(X, k, seed=seed, iterations=iters, threshold=thr)
        C_e, memb_e, it_e = eng.host_kmeans(X, k, seed=seed, iterations=iters, threshold=thr)
        assert it_e == it_o
        assert (memb_e == memb_o).all()
        assert (C_e == C_o).all()


def test_host_kmeans_honours_leading_dimension():
    import ctypes as C
    from nmfgpu_amd._lib import library
    big = np.asfortranarray(np.random.RandomState(0).random_sample((50, 80)).astype(np.float32))
    X = np.asfortranarray(big[:37, :])          # the same data, packed
    C_o, memb_o, _ = oracle.kmeans(X, 4, seed=2)
    clusters = np.full((41, 4), -1.0, dtype=np.float32, order="F")   # ldc 41 > rows
    memb = np.zeros(80, dtype=np.uint32)
    st = library().nmfamd_host_kmeans_f32(C.c_void_p(big.ctypes.data), C.c_long(50), 37, 80, C.c_void_p(clusters.ctypes.data), C.c_long(41), 4,
                                          C.c_void_p(memb.ctypes.data), C.c_uint(2), C.c_uint(100), C.c_double(0.005), None)
    assert st == 0
    assert (memb == memb_o).all() and (clusters[:37] == C_o).all() and (clusters[37:] == -1.0).all()


def test_host_kmeans_result_does_not_depend_on_the_thread_count():
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from nmfgpu_amd import engine as eng\n"
            "rs = np.random.RandomState(5); X = np.asfortranarray(rs.random_sample((300, 4000)).astype(np.float32))\n"
            "C_, memb, it = eng.host_kmeans(X, 24, seed=4, iterations=8, threshold=0.0)\n"
            "import hashlib; print(hashlib.sha1(C_.tobytes() + memb.tobytes()).hexdigest(), it)\n") % ROOT
    outs = []
    for threads in ("1", "3", "8"):
        env = dict(os.environ, NMFAMD_HOST_THREADS=threads)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip())
    assert outs[0] == outs[1] == outs[2], outs


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("r", [5, 32, 40, 70])
def test_einnmf_initialiser_matches_oracle_bit_for_bit(dtype, r):
    X = blobs(45, 200, r, dtype, seed=r)
    W, H = eng.host_init(X, r, Init.EInNMF, seed=6)
    C_o, _, _ = oracle.kmeans(X, r, seed=6, iterations=100, threshold=0.005)      # KMeansStrategy.cpp:52-56
    assert (W == C_o).all()
    assert (H == oracle.einnmf_h(X, C_o)).all()
    assert np.isfinite(H).all() and (H >= 0).all()


@pytest.mark.parametrize("dtype,tol", [(np.float32, 1e-5), (np.float64, 1e-13)])
def test_kmeans_wtv_initialisers(dtype, tol):
    X = blobs(60, 120, 7, dtype, seed=3) - dtype(1.0)        # mixed signs so the clip and |.| differ
    C_o, _, _ = oracle.kmeans(X, 7, seed=11)
    wtv = C_o.astype(np.float64).T @ X.astype(np.float64)
    W, H = eng.host_init(X, 7, Init.KMeansAndNonNegativeWTV, seed=11)
    assert (W == C_o).all()
    np.testing.assert_allclose(H, np.maximum(wtv, 0), rtol=tol, atol=tol * np.abs(wtv).max())
    W, H = eng.host_init(X, 7, Init.KMeansAndAbsoluteWTV, seed=11)
    np.testing.assert_allclose(H, np.abs(wtv), rtol=tol, atol=tol * np.abs(wtv).max())
    W, H = eng.host_init(X, 7, Init.KMeansAndRandomValues, seed=11)
    assert (W == C_o).all() and (H > 0).all() and (H <= 1).all()


def test_mean_columns_initialiser_averages_five_columns():
    X = blobs(20, 30, 4, np.float64, seed=8)
    W, H = eng.host_init(X, 6, Init.MeanColumns, seed=2)
    # every W column is the mean of five (not necessarily distinct) data columns: 5*w is a sum of columns,
    # so it lies in the columns' convex hull scaled by 5 -> check bounds and that it is reproducible
    assert (W >= X.min(axis=1, keepdims=True) - 1e-12).all() and (W <= X.max(axis=1, keepdims=True) + 1e-12).all()
    W2, _ = eng.host_init(X, 6, Init.MeanColumns, seed=2)
    assert (W == W2).all() and (H > 0).all()


def test_host_entry_points_reject_bad_arguments():
    X = blobs(10, 12, 3, np.float32, seed=0)
    with pytest.raises(eng.EngineError):
        eng.host_kmeans(X, 12)            # clusters must be fewer than columns (Interface.cpp:366-370)
    with pytest.raises(eng.EngineError):
        eng.host_init(X, 3, Init.AllRandomValues)   # not a host-side method


# ------------------------------------------------------------------ sanitizers (CPU build only: the GPU pool offers none)

@pytest.mark.parametrize("flags", ["-fsanitize=address,undefined", "-fsanitize=thread"])
def test_host_initialisers_under_sanitizers(tmp_path, flags):
    """host_init.cpp alone (no HIP) with AddressSanitizer + UBSan, and with ThreadSanitizer, on odd shapes and several threads:
    k-means and every initialisation method it serves.  Function multiversioning is switched off for the build (gcc cannot
    combine it with -fsanitize)."""
    exe = tmp_path / "hi_san"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-DNMFAMD_CLONES=", *flags.split(), "-fno-omit-frame-pointer", "-pthread", "-ffp-contract=off",
           os.path.join(ROOT, "tests", "cpp", "host_init_sanitize.cpp"), os.path.join(ROOT, "nmfgpu_amd", "csrc", "host_init.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, env={**os.environ, "NMFAMD_HOST_THREADS": "4"}, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "runtime error" not in run.stderr and "Sanitizer" not in run.stderr, run.stderr
    assert run.stdout.strip().endswith("ok")


# ------------------------------------------------------------------ NNDSVD (not in the reference; BASELINE north_star names it)

def _decaying(m, n, k, dtype, seed, noise=0.01):
    """Non-negative matrix with a geometrically decaying spectrum (k directions) plus a little noise: separated singular values, so the truncated SVD's vectors are
    well defined and the comparison with numpy's full SVD is meaningful."""
    rng = np.random.default_rng(seed)
    A, B, s = rng.random((m, k)), rng.random((k, n)), 0.8 ** np.arange(k)
    return np.asfortranarray(((A * s) @ B + noise * rng.random((m, n))).astype(dtype))


@pytest.mark.parametrize("m,n,r,dtype,tol", [(300, 200, 8, np.float64, 1e-9), (500, 200, 8, np.float32, 1e-5), (257, 131, 20, np.float64, 1e-6),
                                             (96, 64, 64, np.float64, 1e-9), (700, 900, 30, np.float32, 2e-4)])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_nndsvd_matches_the_numpy_restatement(m, n, r, dtype, tol, variant):
    """NNDSVD / NNDSVDa / NNDSVDar on the host (block subspace iteration + one-sided Jacobi, in double) against oracle.nndsvd (numpy's full SVD): same W0, H0 -- the
    method is invariant under the sign of a singular pair, so no sign convention is needed -- including the seeded fill of the "ar" form, entry by entry.
    Tolerances: the subspace iteration stops when the Ritz VALUES stand still at 1e-14; the vectors of the last, closely spaced pairs are then good to the square
    root of that (1e-8 measured at 257 x 131, r = 20) -- more than a start value needs; where the block spans the whole space (96 x 64, r = 64) the SVD is exact."""
    V = _decaying(m, n, r + 5, dtype, seed=m + n)
    W, H = eng.host_init(V, r, eng.NNDSVD + variant, seed=11)
    Wo, Ho = oracle.nndsvd(V, r, variant, seed=11)
    assert (W >= 0).all() and (H >= 0).all()
    assert np.abs(W - Wo).max() <= tol * np.abs(Wo).max() and np.abs(H - Ho).max() <= tol * np.abs(Ho).max()
    if variant == 0:
        assert ((W == 0) == (Wo == 0)).mean() > 0.999           # the zero pattern of the plain form
    else:
        assert (W > 0).all() and (H > 0).all()                  # "a" / "ar": no zeros left (multiplicative updates cannot leave a zero)


def test_nndsvd_is_a_head_start_and_reproducible():
    """What the method is for: ||V - W0 H0|| is far below a random start's, on a matrix with structure; and two calls give the same bits (fixed start block, fixed
    summation orders whatever the thread count)."""
    V = _decaying(400, 300, 12, np.float64, seed=3)
    W, H = eng.host_init(V, 10, eng.NNDSVD_A, seed=1)
    W2, H2 = eng.host_init(V, 10, eng.NNDSVD_A, seed=1)
    assert np.array_equal(W, W2) and np.array_equal(H, H2)
    rng = np.random.default_rng(0)
    Wr, Hr = rng.random(W.shape), rng.random(H.shape)
    scale = (V * (Wr @ Hr)).sum() / ((Wr @ Hr) ** 2).sum()       # (the best scalar for the random pair)
    W0, H0 = eng.host_init(V, 10, eng.NNDSVD, seed=1)             # (the plain form: the "a" form's fill with mean(V) is there for the updates, not for the fit)
    assert np.linalg.norm(V - W0 @ H0) < 0.5 * np.linalg.norm(V - scale * (Wr @ Hr))
    env = dict(os.environ, NMFAMD_HOST_THREADS="1")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from nmfgpu_amd import engine as eng; from tests.test_host_init import _decaying; "
            "W, H = eng.host_init(_decaying(400, 300, 12, np.float64, seed=3), 10, eng.NNDSVD_A, seed=1); print(float(W.sum()).hex(), float(H.sum()).hex())" % ROOT)
    one = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-1000:]
    assert one.stdout.split() == [float(W.sum()).hex(), float(H.sum()).hex()]


def test_nndsvd_rank_deficient_and_rejected_arguments():
    rng = np.random.default_rng(5)
    V = np.asfortranarray(rng.random((60, 4)) @ rng.random((4, 50)))            # rank 4, r = 6: the last two pairs have singular value ~ 0
    W, H = eng.host_init(V, 6, eng.NNDSVD, seed=0)
    assert np.isfinite(W).all() and np.isfinite(H).all() and (W >= 0).all() and (H >= 0).all()
    assert np.linalg.norm(V - W[:, :4] @ H[:4, :]) < 0.6 * np.linalg.norm(V)
    with pytest.raises(eng.EngineError):
        eng.host_init(V, 51, eng.NNDSVD, seed=0)                                   # more features than min(m, n)
