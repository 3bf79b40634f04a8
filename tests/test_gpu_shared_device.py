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
