"""The column-sharded iteration INSIDE the library: nmfgpu::compute with Parameter "numGpus", rank
threads, reduce-scatter by row blocks of W -> row-block update -> all-reduce of the column norms -> all-gather, and the
same iteration through the native sharded API with an RCCL communicator (RCCL's C API, no torch).

The box has ONE GPU: ranks of a "numGpus" team share it and talk through the in-process peer-read transport (RCCL refuses
two ranks on one device); RCCL itself is exercised with a one-rank clique, where every collective is issued for real."""
import numpy as np
import pytest

import nmfgpu_amd as na
from oracle import oracle

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(a)


def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300)


def problem(m, n, r, dtype, seed=1):
    rng = np.random.default_rng(seed)
    return (F(rng.random((m, n)).astype(dtype)), F((1.0 - rng.random((m, r))).astype(dtype)), F((1.0 - rng.random((r, n))).astype(dtype)))


@pytest.fixture(scope="module", autouse=True)
def _library_is_native():
    assert na.device_count() >= 1, "GPU tests need a HIP device"
    assert na.initialize() in (na.ResultType.Success, na.ResultType.ErrorAlreadyInitialized)
    na.set_verbosity(na.Verbosity.Nothing)
    yield
    na.finalize()


@pytest.mark.parametrize("alg,r,params,dtype,tol", [
    (na.NmfAlgorithm.Multiplicative, 16, {}, np.float32, 2e-4),
    (na.NmfAlgorithm.Multiplicative, 64, {}, np.float32, 2e-4),
    (na.NmfAlgorithm.Multiplicative, 100, {}, np.float32, 2e-4),
    (na.NmfAlgorithm.nsNMF, 40, {"theta": 0.4}, np.float32, 2e-4),
    (na.NmfAlgorithm.Multiplicative, 20, {}, np.float64, 1e-9),
])
@pytest.mark.parametrize("ranks,mode", [(2, 0), (2, 1), (3, 0)])
def test_compute_with_numgpus_matches_oracle_and_single_gpu(alg, r, params, dtype, tol, ranks, mode):
    m, n, iters = 700, 530, 30                       # 530 columns on 3 ranks: ragged shards
    V, W0, H0 = problem(m, n, r, dtype, seed=r + ranks)
    name = "mu" if alg == na.NmfAlgorithm.Multiplicative else "nsnmf"
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W0, H0))
    ref = oracle.run(name, V64, W64, H64, iters, theta=params.get("theta", 0.0))
    W1, H1 = W0.copy(order="F"), H0.copy(order="F")
    s1 = na.Summary()
    assert na.compute(V, W1, H1, algorithm=alg, iterations=iters, parameters=params, summary=s1) == na.ResultType.Success
    Wn, Hn = W0.copy(order="F"), H0.copy(order="F")
    sn = na.Summary()
    p = dict(params, numGpus=ranks, shardMode=mode)
    assert na.compute(V, Wn, Hn, algorithm=alg, iterations=iters, parameters=p, summary=sn) == na.ResultType.Success
    assert rel(Wn, W64) < tol and rel(Hn, H64) < tol
    assert rel(Wn, W1) < tol and rel(Hn, H1) < tol
    assert sn.record(0).frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    assert sn.record(0).frobenius == pytest.approx(s1.record(0).frobenius, rel=1e-6)
    assert sn.record(0).rmsd == pytest.approx(s1.record(0).rmsd, rel=1e-6)
    assert sn.record(0).numIterations == iters


@pytest.mark.parametrize("ranks", [2, 3, 5])
def test_direct_exchange_gives_the_bits_of_the_reduced_exchange(ranks, monkeypatch, diag_build):
    """Round 4's small-message form of the W step (replicated mode at padded rank 64: the W update reads the ranks' exchange panels in place and adds them
    in rank order in its prologue, two alternating buffers, one rendezvous per iteration) against round 3's form of the same step (the panels summed by the
    transport's reduction kernel in rank order, the sum copied back, the W update on the one reduced panel; NMFAMD_SHARD_NO_DIRECT=1 in the measurement build): the same values are
    added in the same order, so the factors and the reported error must agree BIT FOR BIT -- on ragged shards, across an odd rank count, error iterations included."""
    m, n, r, iters = 700, 530, 48, 31
    V, W0, H0 = problem(m, n, r, np.float32, seed=40 + ranks)
    out = []
    for no_direct in ("0", "1"):
        monkeypatch.setenv("NMFAMD_SHARD_NO_DIRECT", no_direct)
        W, H = W0.copy(order="F"), H0.copy(order="F")
        s = na.Summary()
        assert na.compute(V, W, H, iterations=iters, parameters={"numGpus": ranks, "shardMode": 1}, summary=s) == na.ResultType.Success
        out.append((W, H, s.record(0).frobenius))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    # ... and both are the single-GPU run up to the summation order of the column shards
    W1, H1 = W0.copy(order="F"), H0.copy(order="F")
    assert na.compute(V, W1, H1, iterations=iters) == na.ResultType.Success
    assert rel(out[0][0], W1) < 2e-5 and rel(out[0][1], H1) < 2e-5


def test_compute_with_numgpus_random_init_draws_one_stream_for_all_shards():
    """AllRandomValues on N ranks: every rank fills its columns of H from the position those columns have in the ONE stream,
    so the start -- and with it the whole run -- is the single-GPU run's up to rounding."""
    m, n, r = 500, 410, 12
    V, _, _ = problem(m, n, r, np.float32, seed=5)
    out = []
    for ranks in (1, 2):
        W = F(np.zeros((m, r), dtype=np.float32)); H = F(np.zeros((r, n), dtype=np.float32))
        p = {"numGpus": ranks} if ranks > 1 else {}
        for iters in (0, 25):
            s = na.Summary()
            assert na.compute(V, W, H, init=na.NmfInitializationMethod.AllRandomValues, iterations=iters, seed=9, parameters=p, summary=s) == na.ResultType.Success
            out.append((W.copy(), H.copy(), s.record(0).frobenius))
    (W0a, H0a, _), (Wa, Ha, fa), (W0b, H0b, _), (Wb, Hb, fb) = out
    assert np.array_equal(W0a, W0b) and np.array_equal(H0a, H0b)          # identical start values, bit for bit
    assert rel(Wb, Wa) < 2e-4 and rel(Hb, Ha) < 2e-4 and fb == pytest.approx(fa, rel=1e-5)


def test_compute_with_numgpus_kmeans_init_threshold_stop_and_runs():
    """Host-side initialiser once for the whole matrix, several runs, threshold stop: same bookkeeping as on one GPU."""
    m, n, r = 300, 260, 6
    V, _, _ = problem(m, n, r, np.float32, seed=8)
    res = []
    for ranks in (1, 2):
        W = F(np.zeros((m, r), dtype=np.float32)); H = F(np.zeros((r, n), dtype=np.float32))
        s = na.Summary()
        p = {"numGpus": ranks} if ranks > 1 else {}
        assert na.compute(V, W, H, init=na.NmfInitializationMethod.KMeansAndNonNegativeWTV, iterations=200, runs=3, seed=3, threshold=0.05,
                          parameters=p, summary=s) == na.ResultType.Success
        res.append((W, H, [s.record(i) for i in range(s.record_count())], s.best_run()))
    (W1, H1, rec1, b1), (W2, H2, rec2, b2) = res
    assert len(rec1) == len(rec2) and b1 == b2
    for a, b in zip(rec1, rec2):
        assert a.numIterations == b.numIterations and a.frobenius == pytest.approx(b.frobenius, rel=1e-5)
    assert rel(W2, W1) < 5e-4 and rel(H2, H1) < 5e-4


@pytest.mark.parametrize("ranks,mode", [(2, 1), (3, 0)])
def test_compute_with_numgpus_sparse_input_formats_shard_by_column_blocks(ranks, mode):
    """sparse V shards by column blocks.  CSR / CSC / COO x index base 0 / 1 (COO unordered, one duplicated coordinate
    pair whose values add) on N ranks: every format gives the SAME bits as the dense upload of the same matrix on the same ranks
    (each rank densifies its own block: exact indexing, reference Matrix.h:145-232), and the run agrees with the single-GPU one."""
    import scipy.sparse as sp
    m, n, r, iters = 300, 250, 12, 15
    rng = np.random.default_rng(17)
    D = F((rng.random((m, n)) * (rng.random((m, n)) < 0.15)).astype(np.float32))
    W0 = F((1.0 - rng.random((m, r))).astype(np.float32)); H0 = F((1.0 - rng.random((r, n))).astype(np.float32))
    p = {"numGpus": ranks, "shardMode": mode}

    def run(V, params):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        s = na.Summary()
        assert na.compute(V, W, H, iterations=iters, parameters=params, summary=s) == na.ResultType.Success
        return W, H, s.record(0).frobenius
    Wd, Hd, fd = run(D, p)
    W1, H1, f1 = run(D, {})
    assert rel(Wd, W1) < 2e-4 and rel(Hd, H1) < 2e-4 and fd == pytest.approx(f1, rel=1e-6)
    keep = []
    for fmt in (na.StorageFormat.CSR, na.StorageFormat.CSC, na.StorageFormat.COO):
        for base in (0, 1):
            if fmt == na.StorageFormat.CSR:
                sm = sp.csr_matrix(D); vals, a, b = sm.data, sm.indptr + base, sm.indices + base
            elif fmt == na.StorageFormat.CSC:
                sm = sp.csc_matrix(D); vals, a, b = sm.data, sm.indptr + base, sm.indices + base
            else:
                sm = sp.coo_matrix(D)
                perm = rng.permutation(sm.nnz)
                vals, a, b = sm.data[perm].copy(), (sm.row + base)[perm], (sm.col + base)[perm]
                # split one entry into two triplets with the same coordinates: 0.75 v + 0.25 v (exact in fp32)
                vals = np.concatenate([vals, vals[:1] * np.float32(0.25)]); vals[0] *= np.float32(0.75)
                a = np.concatenate([a, a[:1]]); b = np.concatenate([b, b[:1]])
            vals = np.ascontiguousarray(vals, np.float32); a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
            keep.append((vals, a, b))
            desc = na.api.sparse_description(fmt, m, n, vals, a, b, na.IndexBase.One if base else na.IndexBase.Zero)
            Ws, Hs, fs = run(desc, p)
            assert np.array_equal(Ws, Wd) and np.array_equal(Hs, Hd) and fs == fd, (fmt, base)


@pytest.mark.parametrize("ranks,mode,r", [(2, 1, 12), (3, 0, 12), (2, 0, 100)])
def test_compute_with_numgpus_sparse_compute_shards_the_csc_and_csr_images(ranks, mode, r):
    """Parameter "sparseCompute" on N ranks (sparse V shards by column blocks of the CSC mirror): every rank keeps the CSC / CSR images of
    its column block resident and runs the two products as SpMMs over them; same run as the one-GPU sparse-compute engine and as the dense oracle."""
    import scipy.sparse as sp
    m, n, iters = 400, 330, 20
    rng = np.random.default_rng(5 + r)
    D = F((rng.random((m, n)) * (rng.random((m, n)) < 0.2)).astype(np.float32))
    W0 = F((1.0 - rng.random((m, r))).astype(np.float32)); H0 = F((1.0 - rng.random((r, n))).astype(np.float32))
    D64, W64, H64 = (F(x.astype(np.float64)) for x in (D, W0, H0))
    ref = oracle.run("mu", D64, W64, H64, iters)
    sm = sp.csr_matrix(D)
    vals = np.ascontiguousarray(sm.data, np.float32); ptr = np.ascontiguousarray(sm.indptr, np.int32); idx = np.ascontiguousarray(sm.indices, np.int32)
    desc = na.api.sparse_description(na.StorageFormat.CSR, m, n, vals, ptr, idx)
    out = []
    for params in ({"sparseCompute": 1}, {"sparseCompute": 1, "numGpus": ranks, "shardMode": mode}):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        s = na.Summary()
        assert na.compute(desc, W, H, iterations=iters, parameters=params, summary=s) == na.ResultType.Success
        out.append((W, H, s.record(0).frobenius))
    (W1, H1, f1), (Wn, Hn, fn) = out
    assert rel(Wn, W64) < 5e-4 and rel(Hn, H64) < 5e-4
    assert rel(Wn, W1) < 2e-4 and rel(Hn, H1) < 2e-4
    assert fn == pytest.approx(ref["frobenius"], rel=1e-4) and fn == pytest.approx(f1, rel=1e-5)


@pytest.mark.parametrize("ranks,r,dtype,tol", [(2, 8, np.float32, 5e-4), (3, 70, np.float32, 5e-4), (2, 5, np.float64, 1e-9)])
def test_compute_with_numgpus_kl_divergence_update(ranks, r, dtype, tol):
    """The KL-divergence update (extension; Lee & Seung 2001) on N column shards: H half-step local, the W half-step's numerator, the row sums of H and
    the per-row error terms summed over the ranks in ONE all-reduce, W update replicated.  Against the literature oracle and the one-GPU engine."""
    import scipy.sparse as sp
    m, n, iters = 260, 230, 20
    rng = np.random.default_rng(31 + r)
    D = F((rng.random((m, n)) * (rng.random((m, n)) < 0.25)).astype(dtype))
    W0 = F((1.0 - rng.random((m, r))).astype(dtype)); H0 = F((1.0 - rng.random((r, n))).astype(dtype))
    D64, W64, H64 = (F(x.astype(np.float64)) for x in (D, W0, H0))
    ref = oracle.run_kl(D64, W64, H64, iters)
    sm = sp.csc_matrix(D)
    vals = np.ascontiguousarray(sm.data, dtype); ptr = np.ascontiguousarray(sm.indptr, np.int32); idx = np.ascontiguousarray(sm.indices, np.int32)
    desc = na.api.sparse_description(na.StorageFormat.CSC, m, n, vals, ptr, idx)
    out = []
    for params in ({"divergence": 1}, {"divergence": 1, "numGpus": ranks}):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        s = na.Summary()
        assert na.compute(desc, W, H, iterations=iters, parameters=params, summary=s) == na.ResultType.Success
        out.append((W, H, s.record(0).frobenius, s.record(0).rmsd))
    (W1, H1, f1, r1), (Wn, Hn, fn, rn) = out
    assert rel(Wn, W64) < tol and rel(Hn, H64) < tol
    assert rel(Wn, W1) < tol and rel(Hn, H1) < tol
    assert fn == pytest.approx(ref["frobenius"], rel=1e-4 if dtype == np.float32 else 1e-9)
    assert fn == pytest.approx(f1, rel=1e-5) and rn == pytest.approx(r1, rel=1e-5)


def test_compute_with_numgpus_kl_divergence_update_with_a_blocked_gather(monkeypatch):
    """The same with the W half-step's gather of H cut into L2-sized blocks on every rank (13 000 columns per rank x 64 padded features > 3 MiB;
    1 MiB blocks asked for): the blocks' partial numerators are summed in block order before they join the sum over the ranks."""
    import scipy.sparse as sp
    monkeypatch.setenv("NMFAMD_KL_BLOCK_KB", "1024")
    m, n, r, iters = 200, 26000, 8, 10
    rng = np.random.default_rng(3)
    D = F((rng.random((m, n)) * (rng.random((m, n)) < 0.02)).astype(np.float32))
    W0 = F((1.0 - rng.random((m, r))).astype(np.float32)); H0 = F((1.0 - rng.random((r, n))).astype(np.float32))
    D64, W64, H64 = (F(x.astype(np.float64)) for x in (D, W0, H0))
    ref = oracle.run_kl(D64, W64, H64, iters)
    sm = sp.csr_matrix(D)
    vals = np.ascontiguousarray(sm.data, np.float32); ptr = np.ascontiguousarray(sm.indptr, np.int32); idx = np.ascontiguousarray(sm.indices, np.int32)
    desc = na.api.sparse_description(na.StorageFormat.CSR, m, n, vals, ptr, idx)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    s = na.Summary()
    assert na.compute(desc, W, H, iterations=iters, parameters={"divergence": 1, "numGpus": 2}, summary=s) == na.ResultType.Success
    assert rel(W, W64) < 5e-4 and rel(H, H64) < 5e-4
    assert s.record(0).frobenius == pytest.approx(ref["frobenius"], rel=1e-4)


@pytest.mark.parametrize("alg,params,r", [(na.NmfAlgorithm.Multiplicative, {}, 24), (na.NmfAlgorithm.nsNMF, {"theta": 0.4}, 64),
                                          (na.NmfAlgorithm.AHCLS, {"lambdaW": 0.01, "lambdaH": 0.01, "alphaW": 0.01, "alphaH": 0.01}, 16),
                                          (na.NmfAlgorithm.GDCLS, {"lambda": 0.01}, 16)])
@pytest.mark.parametrize("ranks", [2, 3])
def test_compute_with_numgpus_constant_basis_vectors(alg, params, r, ranks):
    """useConstantBasisVectors (ref AlgorithmMultiplicativeFrobenius.h:144-146,218-228 and siblings) on N ranks: W is given and stays, so every rank
    fits its own columns of H alone; the error is the root of the shards' summed squares.  Same H and error as the one-GPU run (other K split of the
    product per shard: fp32 rounding), W untouched."""
    m, n, iters = 500, 410, 20
    V, W0, H0 = problem(m, n, r, np.float32, seed=r + ranks)
    out = []
    for extra in ({}, {"numGpus": ranks}):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        s = na.Summary()
        assert na.compute(V, W, H, algorithm=alg, iterations=iters, constant_basis_vectors=True, parameters=dict(params, **extra), summary=s) == na.ResultType.Success
        out.append((W, H, s.record(0).frobenius, s.record(0).rmsd))
    (W1, H1, f1, r1), (Wn, Hn, fn, rn) = out
    assert rel(Hn, H1) < 1e-4
    assert rel(Wn, W1) < 1e-6                       # (what compute returns for W in this mode is the same on one and on N ranks)
    assert fn == pytest.approx(f1, rel=1e-4) and rn == pytest.approx(r1, rel=1e-4)


def test_compute_with_numgpus_rejects_what_does_not_shard():
    V, W, H = problem(60, 50, 4, np.float32)
    assert na.compute(V, W, H, iterations=2, constant_basis_vectors=True, parameters={"numGpus": 2, "divergence": 1}) == na.ResultType.ErrorInvalidArgument
    assert na.compute(V, W, H, iterations=2, parameters={"numGpus": 17}) == na.ResultType.ErrorInvalidArgument
    assert na.compute(V, W, H, iterations=2, algorithm=na.NmfAlgorithm.nsNMF, parameters={"numGpus": 2, "divergence": 1, "theta": 0.5}) == na.ResultType.ErrorInvalidArgument
    assert na.compute(V, W, H, iterations=2, parameters={"numGpus": 1}) == na.ResultType.Success


@pytest.mark.parametrize("alg,r,kw,mode", [("mu", 64, {}, na.SHARD_ROW_BLOCKS), ("mu", 64, {}, na.SHARD_REPLICATED), ("mu", 20, {}, na.SHARD_ROW_BLOCKS),
                                           ("nsnmf", 130, dict(theta=0.5), na.SHARD_ROW_BLOCKS)])
def test_native_sharded_run_over_a_one_rank_rccl_clique(alg, r, kw, mode):
    """RCCL through its C API (librccl.so loaded by the library): ncclReduceScatter / ncclAllReduce / ncclAllGather are all
    issued on the engine's stream; with one rank they are identities, so the result is the single-GPU factorisation."""
    assert na.RcclComm.available()
    m, n, iters = 640, 500, 30
    V, W, H = problem(m, n, r, np.float32, seed=r)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, **kw)
    comm = na.RcclComm(na.RcclComm.unique_id(), 1, 0)
    eng = na.Engine(m, n, r, alg, row_blocks=1, **kw)
    eng.upload(V); eng.set_factors(W, H)
    run = na.ShardedRun(eng, comm, m, n, mode)
    run.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = eng.get_factors()
    assert rel(Wg, W64) < 2e-4 and rel(Hg, H64) < 2e-4
    assert run.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)
    assert run.rmsd == pytest.approx(ref["rmsd"], rel=1e-5)
    run.close(); eng.close(); comm.close()


LS_PARAMS = {"gdcls": (na.NmfAlgorithm.GDCLS, {"lambda": 0.01}), "als": (na.NmfAlgorithm.ALS, {}),
             "acls": (na.NmfAlgorithm.ACLS, {"lambdaW": 0.01, "lambdaH": 0.01}),
             "ahcls": (na.NmfAlgorithm.AHCLS, {"lambdaW": 0.01, "lambdaH": 0.01, "alphaW": 0.01, "alphaH": 0.01})}
LS_ORACLE_KW = {"gdcls": dict(lam=0.01), "als": {}, "acls": dict(lambda_w=0.01, lambda_h=0.01), "ahcls": dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01)}


@pytest.mark.parametrize("name", ["gdcls", "als", "acls", "ahcls"])
@pytest.mark.parametrize("ranks,r,dtype,tol", [(2, 12, np.float32, 1e-3), (3, 64, np.float32, 1e-3), (3, 10, np.float64, 1e-8)])
def test_compute_with_numgpus_least_squares_family(name, ranks, r, dtype, tol):
    """GDCLS / ACLS / AHCLS / ALS shard like the multiplicative update -- their H step is local (an r x r solve per rank
    from the replicated W), their W step needs the same two sums over the shards (V H^T, H H^T); the r x r inverse and the row
    update then run replicated (reference: AlgorithmGradientDescentConstrainedLeastSquares.h:175-272,
    AlgorithmAlternatingHoyerConstrainedLeastSquares.h:183-296).  Against the fp64 oracle and the single-engine run."""
    alg, params = LS_PARAMS[name]
    m, n, iters = 600, 431, 12                       # ragged shards
    V, W0, H0 = problem(m, n, r, dtype, seed=r + ranks)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W0, H0))
    ref = oracle.run(name, V64, W64, H64, iters, **LS_ORACLE_KW[name])
    W1, H1 = W0.copy(order="F"), H0.copy(order="F")
    s1 = na.Summary()
    assert na.compute(V, W1, H1, algorithm=alg, iterations=iters, parameters=params, summary=s1) == na.ResultType.Success
    Wn, Hn = W0.copy(order="F"), H0.copy(order="F")
    sn = na.Summary()
    assert na.compute(V, Wn, Hn, algorithm=alg, iterations=iters, parameters=dict(params, numGpus=ranks), summary=sn) == na.ResultType.Success
    assert rel(Wn, W64) < tol and rel(Hn, H64) < tol, (rel(Wn, W64), rel(Hn, H64))
    assert rel(Wn, W1) < tol and rel(Hn, H1) < tol
    assert sn.record(0).frobenius == pytest.approx(ref["frobenius"], rel=max(1e-5, tol / 10))
    assert sn.record(0).frobenius == pytest.approx(s1.record(0).frobenius, rel=max(1e-5, tol / 10))


@pytest.mark.parametrize("m", [2048, 2000])
def test_eight_ranks_rank256_against_the_oracle(m):
    """BASELINE configs[3]'s cut (8 column shards, nsNMF theta = 0.5, r = 256) at a size the fp64 oracle covers, fp32 and bf16 operands,
    both shard modes; the ranks share the box's one device (in-process transport).  m = 2000: the row-block form pads the rows to 8 x 256 = 2 048, i.e. 128
    K-steps of bf16 fragments travel in the all-gather where the product reads 125 (the fragment buffer is allocated for what travels)."""
    n, r, iters = 8 * 160, 256, 6
    V, W0, H0 = problem(m, n, r, np.float32, seed=8)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W0, H0))
    ref = oracle.run("nsnmf", V64, W64, H64, iters, theta=0.5)
    for prec, tol in ((0, 2e-4), (1, 2e-2)):
        for mode in (0, 1):
            Wn, Hn = W0.copy(order="F"), H0.copy(order="F")
            sn = na.Summary()
            p = {"theta": 0.5, "numGpus": 8, "shardMode": mode, "precision": prec}
            assert na.compute(V, Wn, Hn, algorithm=na.NmfAlgorithm.nsNMF, iterations=iters, parameters=p, summary=sn) == na.ResultType.Success
            assert rel(Wn, W64) < tol and rel(Hn, H64) < tol, (prec, mode, rel(Wn, W64), rel(Hn, H64))
            assert sn.record(0).frobenius == pytest.approx(ref["frobenius"], rel=5 * tol)


def test_config4_as_stated_eight_shards_on_one_device():
    """BASELINE configs[3] as it is stated -- 8 column shards, a collective every iteration, nsNMF theta = 0.5, r = 256, bf16 operands,
    50 000 rows -- with 1 024 columns per shard (the full 6 250 per shard is 10 GB of host matrix per run and eight bf16 images on one
    device; the per-shard kernels at full size are covered by test_config4_shard_size_nsnmf_bf16_properties).  Eight rank threads share
    the device.  Properties of the gathered result, agreement of the two shard modes, and agreement with the single-engine run."""
    m, n, r, iters = 50000, 8 * 1024, 256, 4
    rng = np.random.default_rng(4)
    V = np.empty((m, n), dtype=np.float32, order="F")
    for j0 in range(0, n, 1024):
        V[:, j0:j0 + 1024] = rng.random((m, 1024), dtype=np.float32)
    W0 = F((1.0 - rng.random((m, r))).astype(np.float32)); H0 = F((1.0 - rng.random((r, n))).astype(np.float32))
    base = {"theta": 0.5, "precision": 1}
    out = {}
    for key, extra in (("single", {}), ("rows", {"numGpus": 8, "shardMode": 0}), ("repl", {"numGpus": 8, "shardMode": 1})):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        s = na.Summary()
        assert na.compute(V, W, H, algorithm=na.NmfAlgorithm.nsNMF, iterations=iters, parameters=dict(base, **extra), summary=s) == na.ResultType.Success, key
        out[key] = (W, H, s.record(0).frobenius)
    for key in ("rows", "repl"):
        W, H, f = out[key]
        assert np.isfinite(W).all() and np.isfinite(H).all() and (W >= 0).all() and (H >= 0).all()
        # nsNMF returns W S (AlgorithmNonSmoothNMF.h:221-225): S = (1 - theta) I + theta / r 11^T; the columns of W itself are unit vectors
        theta = 0.5
        Wn = (W.astype(np.float64) - (theta / r) * W.astype(np.float64).sum(axis=1, keepdims=True) / ((1 - theta) + theta)) / (1 - theta)
        np.testing.assert_allclose(np.linalg.norm(Wn, axis=0), 1.0, rtol=2e-3)
        assert rel(W, out["single"][0]) < 2e-2 and rel(H, out["single"][1]) < 2e-2, (key, rel(W, out["single"][0]), rel(H, out["single"][1]))
        assert f == pytest.approx(out["single"][2], rel=2e-2)
    assert rel(out["rows"][0], out["repl"][0]) < 1e-3 and rel(out["rows"][1], out["repl"][1]) < 1e-3
    # eight rank threads with a stream each on one device: the same bits every time (round 3 found one or two ranks per run going wrong by a few
    # per cent -- packed fp32 instructions with a scalar source misbehaving beside other kernels' waves, docs/DESIGN_r05.md section 11; tests/test_gpu_shared_device.py)
    for _ in range(2):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        assert na.compute(V, W, H, algorithm=na.NmfAlgorithm.nsNMF, iterations=iters, parameters=dict(base, numGpus=8, shardMode=1)) == na.ResultType.Success
        assert np.array_equal(W, out["repl"][0]) and np.array_equal(H, out["repl"][1])


def test_team_of_rank_threads_through_the_c_abi_runs_twice_on_one_communicator():
    """The in-process team below the boundary (include/nmfgpu_amd.h: nmfamd_local_group_*, nmfamd_comm_create_local, nmfamd_sharded_*), as bench.py --gpus N drives it:
    three rank threads on one device, rank-64 multiplicative update (the direct exchange), TWO sharded runs after each other on the same communicators (the exchange
    buffers are reused), both W-step modes.  Every rank holds the same bits of W, the replicated mode repeats bit for bit, and the result is the single-engine run's."""
    import threading
    import torch
    m, n, r, world, iters = 900, 610, 40, 3, 25
    V, W0, H0 = problem(m, n, r, np.float32, seed=91)
    group = na.LocalGroup(world)
    gate = threading.Barrier(world)
    results, errors = {}, []

    def rank_thread(g):
        eng = comm = run = None
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            comm = na.LocalComm(group, g)
            c0, nc = na.shard_columns(n, world, g)
            eng = na.Engine(m, nc, r, "mu", row_blocks=world, stream=stream.cuda_stream)
            eng.upload(F(V[:, c0:c0 + nc]))
            for name, mode in (("repl", na.SHARD_REPLICATED), ("repl again", na.SHARD_REPLICATED), ("rows", na.SHARD_ROW_BLOCKS)):
                eng.set_factors(W0, F(H0[:, c0:c0 + nc]))
                gate.wait()
                run = na.ShardedRun(eng, comm, m, n, mode)
                run.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
                Wg, Hg = eng.get_factors()
                results[(name, g)] = (Wg, Hg, run.frobenius)
                run.close(); run = None
                gate.wait()
        except BaseException as e:          # noqa: BLE001
            errors.append((g, e)); group.abort(); gate.abort()
        finally:
            for obj in (run, eng, comm):
                if obj is not None:
                    try:
                        obj.close()
                    except Exception:       # noqa: BLE001
                        pass

    threads = [threading.Thread(target=rank_thread, args=(g,), daemon=True) for g in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)][:2] or errors[:1]
    eng1 = na.Engine(m, n, r, "mu")
    eng1.upload(V); eng1.set_factors(W0, H0)
    eng1.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    W1, H1 = eng1.get_factors(); f1 = eng1.frobenius
    eng1.close()
    for name in ("repl", "repl again", "rows"):
        Wg = results[(name, 0)][0]
        Hg = np.concatenate([results[(name, g)][1] for g in range(world)], axis=1)
        for g in range(1, world):
            assert np.array_equal(results[(name, g)][0], Wg) and results[(name, g)][2] == results[(name, 0)][2]
        assert rel(Wg, W1) < 2e-5 and rel(Hg, H1) < 2e-5 and results[(name, 0)][2] == pytest.approx(f1, rel=1e-6)
    assert np.array_equal(results[("repl", 0)][0], results[("repl again", 0)][0])
    for g in range(world):
        assert np.array_equal(results[("repl", g)][1], results[("repl again", g)][1])


def test_row_block_bf16_run_leaves_whole_fp32_rows_without_a_collective_at_download():
    """row-block mode at padded rank 256 with bf16 operands exchanges bf16 fragments between W updates; the other ranks' fp32 rows in a rank's panel are
    stale until gathered.  nmfamd_sharded_iterate gathers them inside the batch that ends on last_iteration (a collective call anyway), so that afterwards ONE rank
    alone can download its factors -- also after the sharded run is closed -- and every rank holds the same W.  m = 2 000: rows padded to 2 x 1 024 (the fragment
    all-gather moves 128 K-steps where the product reads 125)."""
    import threading
    import torch
    m, n, r, world, iters = 2000, 2 * 192, 256, 2, 5
    V, W0, H0 = problem(m, n, r, np.float32, seed=256)
    group = na.LocalGroup(world)
    gate = threading.Barrier(world)
    results, errors = {}, []

    def rank_thread(g):
        eng = comm = run = None
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            comm = na.LocalComm(group, g)
            c0, nc = na.shard_columns(n, world, g)
            eng = na.Engine(m, nc, r, "nsnmf", theta=0.5, precision="bf16", row_blocks=world, stream=stream.cuda_stream)
            eng.upload(F(V[:, c0:c0 + nc]))
            eng.set_factors(W0, F(H0[:, c0:c0 + nc]))
            gate.wait()
            run = na.ShardedRun(eng, comm, m, n, na.SHARD_ROW_BLOCKS)
            run.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
            eng.synchronize()
            gate.wait()
            run.close(); run = None          # (the gather hook is gone with the run)
            gate.wait()
            if g == 0:
                results[g] = eng.get_factors()          # rank 0 alone: no peer takes part
            gate.wait()
            if g != 0:
                results[g] = eng.get_factors()
        except BaseException as e:          # noqa: BLE001
            errors.append((g, e)); group.abort(); gate.abort()
        finally:
            for obj in (run, eng, comm):
                if obj is not None:
                    try:
                        obj.close()
                    except Exception:       # noqa: BLE001
                        pass

    threads = [threading.Thread(target=rank_thread, args=(g,), daemon=True) for g in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)][:2] or errors[:1]
    assert np.array_equal(results[0][0], results[1][0])
    eng1 = na.Engine(m, n, r, "nsnmf", theta=0.5, precision="bf16")
    eng1.upload(V); eng1.set_factors(W0, H0)
    eng1.iterate(iters, first_iteration=1, error_every=10, last_iteration=iters)
    W1, H1 = eng1.get_factors()
    eng1.close()
    Hg = np.concatenate([results[g][1] for g in range(world)], axis=1)
    assert rel(results[0][0], W1) < 2e-2 and rel(Hg, H1) < 2e-2


def _join_group(world, env_report=None):
    """`world` rank threads on device 0 join one in-process group; returns (errors, selftest report of the group)."""
    import threading
    import torch
    group = na.LocalGroup(world)
    errors, comms = [], [None] * world

    def rank_thread(g):
        try:
            torch.cuda.set_device(0)
            comms[g] = na.LocalComm(group, g)
        except BaseException as e:          # noqa: BLE001
            errors.append((g, e))

    threads = [threading.Thread(target=rank_thread, args=(g,), daemon=True) for g in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    report = group.selftest_report()
    for c in comms:
        if c is not None:
            c.close()
    return errors, report


@pytest.mark.parametrize("world", [2, 5])
def test_peer_transport_selftest_passes_on_a_shared_device(world):
    """Set-up self-test of the in-process transport: pattern A then pattern B at the SAME addresses, both exchange slots and the collectives' buffers,
    read through the iteration's own kernels (k_mu64_update32 with PeerSlabs, k_sum_peers, k_local_sum, k_local_gather), every word checked; a one-off of about a millisecond."""
    import re
    errors, report = _join_group(world)
    assert not errors, errors[:1]
    assert f"{world} ranks" in report and "passed" in report, report
    # (the first group of a fresh process pays for loading the kernels' code objects -- ~100 ms once; the second group is the steady figure)
    errors, report = _join_group(world)
    assert not errors and "passed" in report
    ms = float(re.search(r"\(([\d.]+) ms\)", report).group(1))
    print(report)
    # (measured: 2.0 - 2.1 ms for two ranks; 4.3 - 11.3 ms for five rank threads on ONE device, where every rank's hipDeviceSynchronize waits for the other ranks' streams
    #  too -- and once 62.7 ms on a box whose host threads were slow to come round: the bound says "a one-off of milliseconds", not a figure of merit)
    assert ms < (20.0 if world == 2 else 250.0), report


def test_peer_transport_selftest_names_the_pair_when_an_owner_skips_its_second_write(tmp_path):
    """Fault injection (measurement build only: NMFAMD_SELFTEST_FAULT = rank): the owner skips its pattern-B writes, so every reader finds pattern A at the addresses it
    read before -- what a stale line would look like.  The set-up fails on EVERY rank and the failure text names reader and owner; nmfgpu::compute with numGpus then
    returns ErrorExternalLibrary instead of iterating on a transport that did not deliver."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "nmfgpu_amd", "lib", "libnmfgpu64_diag.so")
    if not os.path.exists(diag):
        pytest.skip("measurement build (python -m nmfgpu_amd.build --diag) not present")
    code = r'''
import sys, threading
sys.path.insert(0, %r)
import numpy as np, torch
import nmfgpu_amd as na
world = 3
group = na.LocalGroup(world)
errors = []
def rank_thread(g):
    try:
        torch.cuda.set_device(0)
        na.LocalComm(group, g)
    except BaseException as e:
        errors.append((g, str(e)))
ts = [threading.Thread(target=rank_thread, args=(g,)) for g in range(world)]
[t.start() for t in ts]; [t.join(120) for t in ts]
print("ERRORS", len(errors)); print("TEXT", errors[0][1] if errors else ""); print("REPORT", group.selftest_report())
na.initialize(); na.set_verbosity(na.Verbosity.Nothing)
rng = np.random.default_rng(0)
V = np.asfortranarray(rng.random((300, 200)).astype(np.float32))
W = np.asfortranarray(rng.random((300, 64)).astype(np.float32)); H = np.asfortranarray(rng.random((64, 200)).astype(np.float32))
res = na.compute(V, W, H, algorithm=na.NmfAlgorithm.Multiplicative, iterations=5, parameters={"numGpus": 2, "shardMode": 1})
print("COMPUTE", res.name)
''' % root
    env = dict(os.environ, NMFAMD_LIBRARY=diag, NMFAMD_SELFTEST_FAULT="1", NMFAMD_COMM="p2p")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stdout
    assert "ERRORS 3" in text, text                                   # every rank's set-up failed
    assert "owner rank 1" in text and "reader rank" in text and "pattern B" in text, text
    assert "FAILED" in text
    assert "COMPUTE ErrorExternalLibrary" in text, text
