"""World-size-2 gloo test of the column-sharded MU orchestration (nmfgpu_amd/distributed.py).

The ORCHESTRATION under test is the product's (ShardedMU: h_step -> w_products -> all_reduce ->
w_finish, error-term gathering).  The per-rank compute backend here is a test-only stand-in built
on the CPU oracle's primitives (no GPU in this container); the GPU backend (EngineShard) has the
same five methods and is exercised on the GPU box by bench.py / test_gpu_distributed.py.
The check: two ranks holding column shards produce the same W, H and reported error as the
single-process oracle run on the whole matrix.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleShard:
    """Test-only backend: the sharded MU steps restated with oracle primitives (float64)."""

    def __init__(self, V_local, W, H_local, theta=None, row_blocks=1):
        import torch
        from oracle import oracle
        self.o = oracle
        self.torch = torch
        self.V = np.asfortranarray(V_local)
        self.W = np.asfortranarray(W.copy())
        self.H = np.asfortranarray(H_local.copy())
        m, r = self.W.shape
        self.m, self.r = m, r
        # the engine's geometry in miniature: panel layout [row][rank], rows padded to whole 128-row tiles per row block
        self.padded_rank = r
        self.padded_m = m if row_blocks == 1 else -(-m // (128 * row_blocks)) * 128 * row_blocks
        self.exchange = torch.zeros(r * self.padded_m + r * r, dtype=torch.float64)
        self.panel = self.exchange[: r * self.padded_m]
        self.hht = self.exchange[r * self.padded_m:]
        self.w_panel = torch.zeros(r * self.padded_m, dtype=torch.float64)
        self.colsq = torch.zeros(r, dtype=torch.float64)
        self.eps = np.finfo(np.float64).eps
        self.vtv = oracle.vtv_sorted(self.V)
        self.psN = None
        self.psR = None
        # nsNMF: S = (1 - theta) I + (theta / r) 1 1^T  (AlgorithmNonSmoothNMF.h:131-134); None = plain MU
        self.S = None if theta is None else np.asfortranarray((1.0 - theta) * np.eye(r) + (theta / r) * np.ones((r, r)))

    def h_step(self, compute_error):
        o = self.o
        Ws = self.W if self.S is None else o.gemm_nn(self.W, self.S)
        self.G = o.gemm_tn(Ws, Ws)
        RN = o.gemm_tn(Ws, self.V)
        RN2 = o.gemm_nn(self.G, self.H)
        o.multiply_divide(self.H, RN, RN2)
        if compute_error:
            self.psN = o.trace_multiplication(True, self.H, RN)

    def w_products(self):
        o = self.o
        Hs = self.H if self.S is None else o.gemm_nn(self.S, self.H)
        MR = o.gemm_nt(self.V, Hs)                # m x r
        HHt = o.gemm_nt(Hs, Hs)                   # r x r
        ex = self.exchange.numpy()
        ex[: self.r * self.m] = np.ascontiguousarray(MR).ravel()     # panel layout [x][c]
        ex[self.r * self.padded_m:] = HHt.ravel(order="F")

    def w_finish(self, compute_error):
        o = self.o
        ex = self.exchange.numpy()
        MR = np.asfortranarray(ex[: self.r * self.m].reshape(self.m, self.r))
        HHt = np.asfortranarray(ex[self.r * self.padded_m:].reshape(self.r, self.r, order="F"))
        if compute_error:
            wtw = self.G if self.S is None else o.gemm_tn(self.W, self.W)     # nsNMF: unsmoothed W^T W (:201-202)
            self.psR = o.trace_multiplication(False, HHt, wtw)
        MR2 = o.gemm_nn(self.W, HHt)
        o.multiply_divide(self.W, MR, MR2)
        o.normalize_columns(self.W)

    # ---- row-block form (what sharded.cpp / Engine::w_update_rows do on the GPU) ----
    def w_update_rows(self, block, row0, rows, compute_error):
        o = self.o
        HHt = np.asfortranarray(self.hht.numpy().reshape(self.r, self.r, order="F"))
        if compute_error:
            wtw = self.G if self.S is None else o.gemm_tn(self.W, self.W)
            self.psR = o.trace_multiplication(False, HHt, wtw)
        lo, hi = min(row0, self.m), min(row0 + rows, self.m)
        self._rows = (lo, hi)
        sq = np.zeros(self.r)
        if hi > lo:
            Wb = np.asfortranarray(self.W[lo:hi, :])
            MR = np.asfortranarray(block.numpy().reshape(rows, self.r)[: hi - lo, :])
            MR2 = o.gemm_nn(Wb, HHt)
            o.multiply_divide(Wb, MR, MR2)
            self._Wb = Wb
            sq = (Wb * Wb).sum(axis=0)
        self.colsq.copy_(self.torch.from_numpy(sq))
        return self.colsq

    def w_normalize_rows(self, row0, rows):
        lo, hi = self._rows
        out = self.w_panel.numpy().reshape(self.padded_m, self.r)
        if hi > lo:
            s = self.colsq.numpy()
            nrm = np.where(s > 0, np.sqrt(s), 1.0)                    # sum > 0 ? x / sqrt(sum) : x   (KernelNormalizeColumns.cu:37-58)
            out[lo:hi, :] = self._Wb / nrm

    def w_rows_replaced(self):
        self.W = np.asfortranarray(self.w_panel.numpy().reshape(self.padded_m, self.r)[: self.m, :].copy())

    def error_terms(self, which):
        return [self.vtv, self.psN, self.psR][which]

    def resolve(self, vtv_sorted, htwtv, hhtwtw):
        return self.o.resolve_frobenius(np.ascontiguousarray(vtv_sorted), htwtv, hhtwtw)

    def factors(self):
        return (self.W if self.S is None else self.o.gemm_nn(self.W, self.S)), self.H


def _worker(rank, world, port, m, n, r, iters, out_dir, theta, mode="replicated"):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nmfgpu_amd.distributed import ShardedMU
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # the box's hostname may not resolve: gloo would probe interfaces
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    V = rng.random((m, n)); W = 1.0 - rng.random((m, r)); H = 1.0 - rng.random((r, n))
    per = n // world
    cols = slice(rank * per, (rank + 1) * per)
    backend = OracleShard(V[:, cols], W, H[:, cols], theta, row_blocks=world if mode == "row_blocks" else 1)
    drv = ShardedMU(backend, total_columns=n, rows=m, mode=mode)
    drv.run(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = backend.factors()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=Wg, H=Hg, frob=drv.frobenius, rmsd=drv.rmsd)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["replicated", "row_blocks"])
@pytest.mark.parametrize("theta", [None, 0.5], ids=["mu", "nsnmf"])
def test_sharded_mu_two_ranks_equals_single_process(tmp_path, theta, mode):
    """mode row_blocks: reduce-scatter of the panel by row blocks of W -> row-block update -> all-reduce of the column sums of
    squares -> all-gather; m = 160 on two ranks: rank 0 owns rows 0..127, rank 1 rows 128..159 plus padding."""
    import torch.multiprocessing as mp
    from oracle import oracle
    m, n, r, iters, world = (160 if mode == "row_blocks" else 60), 48, 5, 25, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, m, n, r, iters, str(tmp_path), theta, mode), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    V = np.asfortranarray(rng.random((m, n))); W = np.asfortranarray(1.0 - rng.random((m, r))); H = np.asfortranarray(1.0 - rng.random((r, n)))
    ref = oracle.run("mu", V, W, H, iters) if theta is None else oracle.run("nsnmf", V, W, H, iters, theta=theta)
    outs = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    per = n // world
    for k, o in enumerate(outs):
        # column sums are re-associated across ranks: agreement to rounding, not bit-exact
        np.testing.assert_allclose(o["W"], W, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(o["H"], H[:, k * per:(k + 1) * per], rtol=1e-10, atol=1e-13)
        assert float(o["frob"]) == pytest.approx(ref["frobenius"], rel=1e-10)
        assert float(o["rmsd"]) == pytest.approx(ref["rmsd"], rel=1e-10)
    # replicas of W stay bit-identical across ranks (same reduced sums, same update)
    assert np.array_equal(outs[0]["W"], outs[1]["W"])


def test_sharded_driver_single_rank_without_process_group():
    """world = 1: no collective is issued and the driver degenerates to the plain iteration."""
    sys.path.insert(0, ROOT)
    from nmfgpu_amd.distributed import ShardedMU
    from oracle import oracle
    rng = np.random.default_rng(8)
    m, n, r = 30, 20, 3
    V = rng.random((m, n)); W = 1.0 - rng.random((m, r)); H = 1.0 - rng.random((r, n))
    backend = OracleShard(V, W, H)
    drv = ShardedMU(backend, total_columns=n, rows=m)
    drv.run(10, last_iteration=10)
    Wr, Hr = np.asfortranarray(W.copy()), np.asfortranarray(H.copy())
    ref = oracle.run("mu", np.asfortranarray(V), Wr, Hr, 10)
    np.testing.assert_allclose(backend.W, Wr, rtol=1e-12)
    assert drv.frobenius == pytest.approx(ref["frobenius"], rel=1e-12)
