"""The one-pass form of the rank-64 multiplicative update (kernels_onepass.hip; MEASUREMENT BUILD only since round 6 -- a recorded dead end, 1.8 x slower
than the two-pass iteration: NMFAMD_ONE_PASS=1 in libnmfgpu64_diag.so) against the fp64 oracle and against the default two-pass iteration on the same inputs.

The H update is column-separable once W^T W is known (reference: source/nmf/AlgorithmMultiplicativeFrobenius.h:176-191,
source/nmf/KernelMultiplyDivide.cu:29-43), so a column panel of V is fetched once for W^T V, the update of its H columns and
V H^T (:240-241).  Same arithmetic as the two-pass path (six-term split products, the reference's update formula); different
summation order (rows cut per workgroup), hence agreement to fp32 rounding, not bit for bit.
"""
import numpy as np
import pytest

import nmfgpu_amd as na
from oracle import oracle

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(a)


def problem(m, n, r, seed=1):
    rng = np.random.default_rng(seed)
    V = F(rng.random((m, n)).astype(np.float32))
    W = F((1.0 - rng.random((m, r))).astype(np.float32))
    H = F((1.0 - rng.random((r, n))).astype(np.float32))
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


def run(V, W, H, r, iters, one_pass, monkeypatch):
    if one_pass:
        monkeypatch.setenv("NMFAMD_ONE_PASS", "1")
    else:
        monkeypatch.delenv("NMFAMD_ONE_PASS", raising=False)
    m, n = V.shape
    eng = na.Engine(m, n, r, "mu")
    try:
        mode = eng.geometry()["one_pass"]
        eng.upload(V); eng.set_factors(W, H)
        eng.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
        frob = eng.frobenius
        Wg, Hg = eng.get_factors()
        assert eng.geometry()["one_pass"] == mode, "a one-pass launch gave up"
        return Wg, Hg, frob, mode
    finally:
        eng.close()


# shapes: ragged rows (not a multiple of 16 / 128), ragged columns (not a multiple of 32), fewer panels than XCDs (groups with no
# panel at all), fewer panels than the pipeline is deep, ranks below the padded 64, the largest row count the cut takes
@pytest.mark.parametrize("m,n,r,iters", [(1000, 333, 10, 20), (517, 40, 64, 10), (2000, 1100, 33, 20), (4100, 2049, 64, 12), (10240, 300, 64, 10)])
def test_one_pass_matches_the_oracle(m, n, r, iters, monkeypatch, diag_build):
    V, W, H = problem(m, n, r, seed=m + n + r)
    ref = oracle.run("mu", F(V.astype(np.float64)), Wo := F(W.astype(np.float64)), Ho := F(H.astype(np.float64)), iters)
    Wg, Hg, frob, mode = run(V, W, H, r, iters, True, monkeypatch)
    assert mode == 1, "the one-pass iteration was not selected"
    assert rel(Wg, Wo) < 2e-4 and rel(Hg, Ho) < 2e-4
    assert frob == pytest.approx(ref["frobenius"], rel=1e-5)
    assert (Wg >= 0).all() and (Hg >= 0).all()


def test_one_pass_equals_two_pass_at_config2(monkeypatch, diag_build):
    """BASELINE configs[1] (10 000 x 5 000, r = 64): both forms of the iteration from the same start, 20 iterations."""
    V, W, H = problem(10000, 5000, 64, seed=1)
    W1, H1, f1, mode1 = run(V, W, H, 64, 20, True, monkeypatch)
    W2, H2, f2, mode2 = run(V, W, H, 64, 20, False, monkeypatch)
    assert (mode1, mode2) == (1, 0)
    assert rel(W1, W2) < 1e-5 and rel(H1, H2) < 1e-5
    assert f1 == pytest.approx(f2, rel=1e-6)


def test_one_pass_is_opt_in_and_deterministic(monkeypatch, diag_build):
    V, W, H = problem(3000, 700, 20, seed=5)
    a = run(V, W, H, 20, 10, True, monkeypatch)
    b = run(V, W, H, 20, 10, True, monkeypatch)
    assert a[3] == 1 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]      # no atomics on data: same bits
    c = run(V, W, H, 20, 10, False, monkeypatch)
    assert c[3] == 0                                                                                         # default: two passes
    # rows beyond what the cut takes (32 workgroups x 320 rows per XCD): two-pass even when asked
    monkeypatch.setenv("NMFAMD_ONE_PASS", "1")
    eng = na.Engine(10300, 64, 8, "mu")
    try:
        assert eng.geometry()["one_pass"] == 0
    finally:
        eng.close()
