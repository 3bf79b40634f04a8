"""Kernels of several engines on one device at once: identical problems must give identical bits.

Round 3 found that they did not: with waves of OTHER kernels on the same SIMD, packed fp32 instructions that take a scalar register as a
source (v_pk_fma_f32 v[..], s[..], ...: what the SLP vectoriser makes of `uniform * x + uniform`) returned a wrong low half in lanes 48..63 --
the rank-256 H update lost the `den_a * acc` term of its denominator in a few columns of half the rows of a workgroup, H came out ~2x there.
The build keeps such instructions out (nmfgpu_amd/build.py, tests/test_abi.py); this is the run-time side of that rule: engines on their own
streams, driven by their own host threads, whole iterations (every kernel of config 4's iteration beside every other)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_engines(make_engine, count, iterations, reads):
    import torch
    torch.cuda.set_device(0)
    streams = [torch.cuda.Stream() for _ in range(count)]
    engines = [make_engine(s.cuda_stream) for s in streams]
    failures = []

    def work(k, barrier, it):
        try:
            torch.cuda.set_device(0)
            barrier.wait()
            engines[k].iterate(1, first_iteration=it + 1, error_every=1000)
            engines[k].synchronize()
        except Exception as exc:      # noqa: BLE001 -- reported below, in the test's own thread
            failures.append(exc)

    try:
        for it in range(iterations):
            barrier = threading.Barrier(count)
            threads = [threading.Thread(target=work, args=(k, barrier, it)) for k in range(count)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert not failures, failures
            for which, n in reads:
                vals = [e.debug_read(which, n).view(np.uint32) for e in engines]
                for k in range(1, count):
                    bad = np.flatnonzero(vals[k] != vals[0])
                    assert bad.size == 0, f"iteration {it + 1}: engine {k} differs from engine 0 in {bad.size} words of intermediate {which}, first at {bad[0]}"
    finally:
        for e in engines:
            e.close()


def test_rank256_engines_side_by_side_give_identical_bits():
    from nmfgpu_amd.engine import Engine
    m, n, r = 50000, 1024, 256      # config 4's shard shape
    rng = np.random.default_rng(4)
    V = np.asfortranarray(rng.random((m, n), dtype=np.float32))
    W0 = np.asfortranarray((1.0 - rng.random((m, r))).astype(np.float32))
    H0 = np.asfortranarray((1.0 - rng.random((r, n))).astype(np.float32))

    def make(stream):
        e = Engine(m, n, r, algorithm="nsnmf", theta=0.5, precision="bf16", stream=stream)
        e.upload(V); e.set_factors(W0, H0); e.synchronize()
        return e
    # (the old build failed this within two or three iterations, every time)
    _run_engines(make, 8, 6, [(1, 256 * n), (8, 65536), (9, 256), (11, 128 * n), (0, 256 * m)])


def test_rank64_engines_side_by_side_give_identical_bits():
    from nmfgpu_amd.engine import Engine
    m, n, r = 10000, 5000, 64       # config 2
    rng = np.random.default_rng(5)
    V = np.asfortranarray(rng.random((m, n), dtype=np.float32))
    W0 = np.asfortranarray((1.0 - rng.random((m, r))).astype(np.float32))
    H0 = np.asfortranarray((1.0 - rng.random((r, n))).astype(np.float32))

    def make(stream):
        e = Engine(m, n, r, algorithm="mu", stream=stream)
        e.upload(V); e.set_factors(W0, H0); e.synchronize()
        return e
    _run_engines(make, 6, 10, [(1, 64 * n), (0, 64 * m)])


def test_pairs_of_different_engines_side_by_side_agree():
    """Two identical engines of each kind, all kinds at once (the old build: the rank-256 pair disagreed from the second round on)."""
    import scipy.sparse as sp
    import torch
    from nmfgpu_amd.engine import Engine
    rng = np.random.default_rng(9)
    F = np.asfortranarray
    S = sp.random(20000, 4000, density=0.02, format="csr", random_state=3, dtype=np.float32)
    S.data = np.abs(S.data) + 0.1
    kinds = [
        dict(m=50000, n=1024, r=256, kw=dict(algorithm="nsnmf", theta=0.5, precision="bf16")),
        dict(m=10000, n=5000, r=64, kw=dict(algorithm="mu")),
        dict(m=5000, n=2000, r=64, kw=dict(algorithm="ahcls", lambda_w=0.1, lambda_h=0.1, alpha_w=0.5, alpha_h=0.5)),
        dict(m=5000, n=2000, r=64, kw=dict(algorithm="gdcls", lam=0.1)),
        dict(m=8000, n=3000, r=200, kw=dict(algorithm="mu")),
        dict(m=3000, n=2000, r=48, dtype=np.float64, kw=dict(algorithm="mu")),
        dict(m=20000, n=4000, r=64, sparse=S, kw=dict(algorithm="mu", divergence="kl", sparse_compute=True)),
        dict(m=20000, n=4000, r=64, sparse=S, kw=dict(algorithm="mu", sparse_compute=True)),
    ]
    torch.cuda.set_device(0)
    streams, engines = [], []
    try:
        for kd in kinds:
            m, n, r, dt = kd["m"], kd["n"], kd["r"], kd.get("dtype", np.float32)
            V = None if "sparse" in kd else F(rng.random((m, n)).astype(dt))
            W0, H0 = F((1.0 - rng.random((m, r))).astype(dt)), F((1.0 - rng.random((r, n))).astype(dt))
            for _ in range(2):
                streams.append(torch.cuda.Stream())
                e = Engine(m, n, r, dtype=dt, stream=streams[-1].cuda_stream, **kd["kw"])
                if V is None:
                    e.upload_sparse(1, kd["sparse"].data.astype(dt), kd["sparse"].indptr, kd["sparse"].indices, 0)
                else:
                    e.upload(V)
                e.set_factors(W0, H0); e.synchronize()
                engines.append(e)
        failures = []

        def work(k, barrier, it):
            try:
                torch.cuda.set_device(0)
                barrier.wait()
                engines[k].iterate(2, first_iteration=2 * it + 1, error_every=1000)
                engines[k].synchronize()
            except Exception as exc:      # noqa: BLE001
                failures.append(exc)
        for it in range(5):
            barrier = threading.Barrier(len(engines))
            threads = [threading.Thread(target=work, args=(k, barrier, it)) for k in range(len(engines))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert not failures, failures
            for p, kd in enumerate(kinds):
                rp = (kd["r"] + 63) // 64 * 64
                for which, cnt in ((1, rp * kd["n"]), (0, rp * kd["m"])):
                    a = engines[2 * p].debug_read(which, cnt).view(np.uint8)
                    b = engines[2 * p + 1].debug_read(which, cnt).view(np.uint8)
                    assert np.array_equal(a, b), (it, kd["kw"], "HW"[which], int(np.count_nonzero(a != b)))
    finally:
        for e in engines:
            e.close()
