"""BASELINE configs[3] at its stated width, once: dense V 50 000 x 50 000, r = 256, nsNMF theta = 0.5, bf16 MFMA operands, EIGHT column shards of
6 250 columns with a collective every iteration -- the whole cut in one run (round 3 ran 8 x 1 024 columns, and one full-width shard alone).

The box has one GPU: the eight ranks are threads that share device 0 and talk through the in-process transport of include/nmfgpu_amd.h
(nmfamd_local_group_*, the transport of nmfgpu::compute's "numGpus").  The matrix is generated and uploaded SHARD BY SHARD (1.25 GB of fp32 at a
time, under a lock): the host never holds the 10 GB matrix; the device holds the eight pairs of bf16 images (8 x 1.25 GB) plus panels and slabs.
Measured limits: none hit on the 288 GB device; host peak ~3 GB.

Too big for the oracle: the size-independent properties of test_config4_shard_size_nsnmf_bf16_properties on the GATHERED result (the reported error
against the direct residual on a row sample of the WHOLE matrix, unit-norm columns of W behind the returned W S, non-negativity, decreasing error),
both W-step modes agreeing to 1e-3, two runs bit-identical, every rank holding the same bits of W.  Reference algorithm:
source/nmf/AlgorithmNonSmoothNMF.h:174-218."""
import threading

import numpy as np
import pytest

import nmfgpu_amd as na

pytestmark = pytest.mark.gpu

M, NSH, WORLD, R, THETA = 50000, 6250, 8, 256, 0.5
SAMPLE = 256


def _shard(g):
    """Columns [g NSH, (g + 1) NSH) of V (U[0,1), seed 1 + g, BASELINE.md section 2 / SURVEY 8d: generated per shard) and of H0."""
    rs = np.random.RandomState(1 + g)
    V = np.empty((M, NSH), dtype=np.float32, order="F")
    for j0 in range(0, NSH, 625):
        V[:, j0:j0 + 625] = rs.random_sample((625, M)).astype(np.float32).T
    H = np.asfortranarray((1.0 - np.random.RandomState(3 + 1000 * g).random_sample((NSH, R))).astype(np.float32).T)
    return V, H


def test_config4_full_width_eight_shards_in_one_run():
    assert na.device_count() >= 1, "GPU tests need a HIP device"
    W0 = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((R, M))).astype(np.float32).T)
    rows = np.sort(np.random.default_rng(7).choice(M, SAMPLE, replace=False))
    group = na.LocalGroup(WORLD)
    gate = threading.Barrier(WORLD)
    upload_lock = threading.Lock()
    v_rows = [None] * WORLD                 # the sampled rows of every shard: what the direct residual needs of V
    results = {}                            # (pass, rank) -> (W S or None, H shard, frobenius, rmsd)
    errors = []
    passes = [("repl", na.SHARD_REPLICATED), ("rows", na.SHARD_ROW_BLOCKS), ("repl again", na.SHARD_REPLICATED)]

    def rank_thread(g):
        eng = comm = run = None
        try:
            import torch
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()            # a stream per rank, as nmfgpu::compute's rank threads have (kept alive to the end of the thread)
            comm = na.LocalComm(group, g)
            eng = na.Engine(M, NSH, R, "nsnmf", theta=THETA, precision="bf16", row_blocks=WORLD, stream=stream.cuda_stream)
            with upload_lock:               # one 1.25 GB shard in host memory at a time
                V, H0 = _shard(g)
                v_rows[g] = V[rows, :].copy()
                eng.upload(V)
                del V
            for name, mode in passes:
                eng.set_factors(W0, H0)
                gate.wait()
                run = na.ShardedRun(eng, comm, M, NSH * WORLD, mode)
                run.iterate(10, first_iteration=1, error_every=10)
                f10 = run.frobenius
                run.iterate(10, first_iteration=11, error_every=10, last_iteration=20)
                f20, rmsd20 = run.frobenius, run.rmsd
                Wg, Hg = eng.get_factors()
                results[(name, g)] = (Wg if g in (0, WORLD - 1) else None, Hg, f10, f20, rmsd20)
                run.close(); run = None
                gate.wait()
        except BaseException as e:          # noqa: BLE001
            errors.append((g, e))
            group.abort(); gate.abort()
        finally:
            for obj in (run, eng, comm):
                if obj is not None:
                    try:
                        obj.close()
                    except Exception:       # noqa: BLE001
                        pass

    threads = [threading.Thread(target=rank_thread, args=(g,), daemon=True) for g in range(WORLD)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=1500)
    real = [(g, e) for g, e in errors if not isinstance(e, threading.BrokenBarrierError)]
    assert not errors, (real or errors)[:2]

    a, b = 1.0 - THETA, THETA / R
    gathered = {}
    for name, _ in passes:
        WS = results[(name, 0)][0]
        H = np.concatenate([results[(name, g)][1] for g in range(WORLD)], axis=1)
        _, _, f10, f20, rmsd20 = results[(name, 0)]
        gathered[name] = (WS, H, f20)
        assert WS.shape == (M, R) and H.shape == (R, NSH * WORLD)
        assert np.isfinite(WS).all() and np.isfinite(H).all() and (WS >= 0).all() and (H >= 0).all()
        assert np.isfinite(f20) and f20 < f10
        # every rank reports the same error (resolved from the same gathered terms) and holds the same bits of W
        for g in range(1, WORLD):
            assert results[(name, g)][3] == f20
        assert np.array_equal(results[(name, WORLD - 1)][0], WS)
        # the reported error (trace formula over all 50 000 columns, terms from three kernels on eight ranks) against the residual evaluated directly on a
        # row sample of the WHOLE matrix: V ~ (W S) H, get_factors hands back W S (AlgorithmNonSmoothNMF.h:221-225)
        Vs = np.concatenate(v_rows, axis=1).astype(np.float64)
        Rs = Vs - WS[rows, :].astype(np.float64) @ H.astype(np.float64)
        assert np.sqrt((Rs * Rs).mean()) == pytest.approx(rmsd20, rel=2e-2)
        # W = (W S) S^-1 has unit-norm columns
        Wn = (WS.astype(np.float64) - (b / (a + b * R)) * WS.astype(np.float64).sum(axis=1, keepdims=True)) / a
        np.testing.assert_allclose(np.linalg.norm(Wn, axis=0), 1.0, rtol=1e-4)

    def rel(x, y):
        return np.linalg.norm(x.astype(np.float64) - y.astype(np.float64)) / np.linalg.norm(y.astype(np.float64))
    # the two W-step modes (one all-reduce + replicated update; reduce-scatter by row blocks + all-gather) agree
    assert rel(gathered["rows"][0], gathered["repl"][0]) < 1e-3 and rel(gathered["rows"][1], gathered["repl"][1]) < 1e-3
    assert gathered["rows"][2] == pytest.approx(gathered["repl"][2], rel=1e-3)
    # and a run gives the same bits when it is repeated
    assert np.array_equal(gathered["repl again"][0], gathered["repl"][0]) and np.array_equal(gathered["repl again"][1], gathered["repl"][1])
