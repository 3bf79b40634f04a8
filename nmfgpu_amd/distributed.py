"""Column-sharded multiplicative update (MU, nsNMF) across the GPUs of one node (one process per GPU).

Rank g holds V(:, J_g), H(:, J_g) and a full replica of W (the reference is
single-GPU, source/nmf/SingleGpuDispatcher.h:36, so this layer has no counterpart there).
Per iteration:

    h_step        local     H(:, J_g) <- H .* (W^T V_g) ./ (W^T W H + eps)            no communication
    w_products    local     exchange <- [ (V_g H_g^T)^T | H_g H_g^T ]
    all_reduce    RCCL      sum of `exchange` over the ranks (2.56 MB + 16 KB at 10000 x 5000, r = 64)
    w_finish      replicated W <- W .* (V H^T) ./ (W H H^T + eps), column normalisation

nsNMF (AlgorithmNonSmoothNMF.h:174-218) shards the same way: the H step uses the replicated W S, the exchange
carries the sums over the SMOOTHED local columns [ (V_g (S H_g)^T)^T | (S H_g)(S H_g)^T ], the W update is
replicated.  51.2 MB + 256 KB per iteration at 50000 x 50000, r = 256 (BASELINE config 4).

mode = "row_blocks" (SURVEY 8e's second form; what the native loop nmfamd_sharded_* and nmfgpu::compute with
Parameter "numGpus" do inside the library, sharded.cpp) replaces the all-reduce + replicated update by

    reduce_scatter  the m x r panel by row blocks of W (+ all_reduce of the r x r H H^T)
    w_update_rows   every rank updates ITS m / N rows of W, leaves r partial sums of squares
    all_reduce      of those r sums; w_normalize_rows
    all_gather      of the row blocks: every rank holds the same bits of the new W

Every rank applies the identical W update to identical reduced sums, so the replicas stay
bit-identical without a broadcast.  On error iterations the per-column terms of tr(H^T W^T V)
are all-gathered so that the host-side sorted summation (source/nmf/FrobeniusResolver.cpp:29-51)
sees the same vectors as a single-GPU run would.

The class is backend-agnostic on purpose: the product backend is `EngineShard` (HIP engine +
torch CUDA tensors + backend "nccl" = RCCL); the CPU tests drive the same orchestration with a
test-only backend over gloo.
"""
from __future__ import annotations

import numpy as np


class EngineShard:
    """Product backend: one nmfgpu_amd.Engine on this rank's GPU, exchange buffer owned by torch."""

    def __init__(self, V_local: np.ndarray, W: np.ndarray, H_local: np.ndarray, device=None,
                 algorithm: str = "mu", theta: float = 0.0, precision: str = "native", row_blocks: int = 1):
        import torch
        from .engine import Engine
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("EngineShard needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        m, n = V_local.shape
        r = W.shape[1]
        for name, a, shape in (("W", W, (m, r)), ("H_local", H_local, (r, n))):
            if a.dtype != V_local.dtype:
                raise TypeError(f"{name} must share V_local's dtype {V_local.dtype}, got {a.dtype}")
            if a.shape != shape:
                raise ValueError(f"{name} must have shape {shape}, got {a.shape}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if algorithm not in ("mu", "nsnmf"):
            raise ValueError("the sharded iteration covers the multiplicative algorithms: 'mu' and 'nsnmf'")
        self._precision_asked = precision
        self.engine = Engine(m, n, r, algorithm, dtype=V_local.dtype, stream=stream, theta=theta, precision=precision, row_blocks=row_blocks)
        self.engine.upload(V_local)
        # Engine.upload switches THIS rank to the native fp32 MFMA instructions when its shard holds values outside the exact range
        # of the split-operand product; W is replicated, so every rank must then run the same arithmetic (and take the same update
        # path): the ranks agree, and the ones that did not switch by themselves switch now (as nmfgpu::compute's rank team does)
        self._agree_on_value_range(V_local)
        self.engine.set_factors(W, H_local)
        g = self.engine.geometry()
        count = g["exchange_count"]
        tdtype = torch.float32 if V_local.dtype == np.float32 else torch.float64
        self.exchange = torch.zeros(count, dtype=tdtype, device=self.device)
        self.dtype = V_local.dtype
        # row-block form: the exchange buffer is [panel: padded_m rows of padded_rank | H H^T]; the engine's own W panel
        # is wrapped as a tensor so that the all-gather lands in it directly
        self.padded_rank, self.padded_m = g["padded_rank"], g["padded_m"]
        self.row_blocks = row_blocks
        self.panel = self.exchange[: self.padded_rank * self.padded_m]
        self.hht = self.exchange[self.padded_rank * self.padded_m:]
        self.colsq = torch.zeros(self.padded_rank, dtype=tdtype, device=self.device)

        class _Raw:       # the W panel inside the engine (device memory the library owns)
            pass
        raw = _Raw()
        raw.__cuda_array_interface__ = {"shape": (self.padded_rank * self.padded_m,), "typestr": "<f4" if tdtype == torch.float32 else "<f8",
                                        "data": (self.engine.w_panel_ptr(), False), "version": 2}
        self._raw = raw
        self.w_panel = torch.as_tensor(raw, device=self.device)

    def _agree_on_value_range(self, V_local):
        torch = self.torch
        try:
            import torch.distributed as dist
        except Exception:      # noqa: BLE001
            return
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        switched = self.engine._ctor["params"][8] == -1.0 and self._precision_asked == "native"
        on_gpu = dist.get_backend() == "nccl"
        flag = torch.tensor([1 if switched else 0], dtype=torch.int32, device=self.device if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) != 0 and not switched and self._precision_asked == "native":
            self.engine.close()
            self.engine._ctor["params"][8] = -1.0
            self.engine._create()
            self.engine.upload(V_local)

    def w_update_rows(self, block, row0: int, rows: int, compute_error: bool):
        self.engine.w_update_rows(block.data_ptr(), self.hht.data_ptr(), row0, rows, compute_error, self.colsq.data_ptr())
        return self.colsq

    def w_normalize_rows(self, row0: int, rows: int):
        self.engine.w_normalize_rows(row0, rows, self.colsq.data_ptr())

    def w_rows_replaced(self):
        self.engine.w_rows_replaced()

    def h_step(self, compute_error: bool):
        self.engine.h_step(compute_error)

    def w_products(self):
        self.engine.w_products(self.exchange.data_ptr())

    def w_finish(self, compute_error: bool):
        self.engine.w_finish(self.exchange.data_ptr(), compute_error)

    def error_terms(self, which: int) -> np.ndarray:
        return self.engine.error_terms(which)

    def error_terms_async(self):
        """Device tensor [n_local tr(H^T W^T V) terms | r tr(H H^T W^T W) terms] of the last error iteration,
        filled by a device-to-device copy ordered on the engine's (= torch's current) stream: no host wait."""
        torch = self.torch
        n_local, r = self.engine.n, self.engine.r
        t = torch.empty(n_local + r, dtype=self.exchange.dtype, device=self.device)
        self.engine.error_terms_to_device(t.data_ptr(), n_local + r)
        return t, n_local, r

    def resolve(self, vtv_sorted, htwtv, hhtwtw) -> float:
        from .engine import resolve_frobenius
        return resolve_frobenius(vtv_sorted, htwtv, hhtwtw)

    def factors(self):
        return self.engine.get_factors()

    def synchronize(self):
        self.engine.synchronize()


class ShardedMU:
    """Drives one backend per rank through the sharded iteration; `dist` is torch.distributed."""

    def __init__(self, backend, total_columns: int, rows: int, group=None, force_collectives: bool = False, mode: str = "replicated"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.backend = backend
        self.group = group
        if mode not in ("replicated", "row_blocks"):
            raise ValueError("mode: 'replicated' or 'row_blocks'")
        self.mode = mode
        self._rs_native = True
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collectives: issue the all-reduce / all-gather even in a one-rank group (they are identities there);
        # lets a single-GPU box exercise the RCCL calls, their stream ordering against the engine's kernels included
        self.collectives = self.world > 1 or (force_collectives and dist.is_initialized())
        if self.world == 1 and mode == "replicated" and hasattr(backend, "engine"):
            backend.engine.set_sole_rank(True)      # (identity collectives included: the buffer reaches w_finish as w_products left it)
        self.total_elements = int(np.uint32(rows) * np.uint32(total_columns))  # the reference multiplies unsigned ints
        self._frobenius = 0.0
        self._rmsd = 0.0
        self._vtv_all = None
        self._pending = None      # (event, pinned host tensor, n_local, r) of an error iteration not yet resolved

    # The error of the most recent error iteration.  With a backend that offers error_terms_async the terms
    # travel (all-gather, device-to-host copy) stream-ordered behind the iteration that produced them and are
    # summed on the host only when somebody looks -- the iteration loop never waits for the GPU.
    @property
    def frobenius(self) -> float:
        self._resolve_pending()
        return self._frobenius

    @property
    def rmsd(self) -> float:
        self._resolve_pending()
        return self._rmsd

    def _resolve_pending(self):
        if self._pending is None:
            return
        ev, pinned, n_local, r = self._pending
        self._pending = None
        ev.synchronize()
        arr = pinned.numpy()
        per = n_local + r
        htwtv = np.concatenate([arr[k * per:k * per + n_local] for k in range(self.world)])
        hhtwtw = arr[n_local:per].copy()          # identical on every rank (reduced H H^T, replicated W^T W)
        self._frobenius = self.backend.resolve(self._vtv_all, htwtv, hhtwtw)
        self._rmsd = self._frobenius / np.sqrt(float(self.total_elements))

    def _launch_error_gather(self):
        torch, dist, b = self.torch, self.dist, self.backend
        loc, n_local, r = b.error_terms_async()
        if self.collectives:
            out = torch.empty(self.world * loc.numel(), dtype=loc.dtype, device=loc.device)
            dist.all_gather(list(out.chunk(self.world)), loc, group=self.group)
        else:
            out = loc
        # two pinned landing buffers, allocated once (pinning memory is slow and synchronises the device)
        if getattr(self, "_pinned", None) is None or self._pinned[0].numel() != out.numel():
            self._pinned = [torch.empty(out.numel(), dtype=out.dtype, pin_memory=True) for _ in range(2)]
            self._pin_turn = 0
        pinned = self._pinned[self._pin_turn]
        self._pin_turn ^= 1
        pinned.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (ev, pinned, n_local, r)

    def _all_gather_host(self, local: np.ndarray) -> np.ndarray:
        """Gathers equally sized host vectors of every rank (error iterations only)."""
        if not self.collectives:
            return local
        torch, dist = self.torch, self.dist
        t = torch.from_numpy(np.ascontiguousarray(local))
        dev = getattr(self.backend, "device", None)
        if dev is not None:
            t = t.to(dev)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t, group=self.group)
        return np.concatenate([o.cpu().numpy() for o in outs])

    def _reduce_scatter(self, out, panel):
        """out <- this rank's block of the sum of `panel` over the ranks.  gloo has no reduce_scatter: one reduce per block."""
        dist = self.dist
        if self._rs_native:
            try:
                dist.reduce_scatter_tensor(out, panel, op=dist.ReduceOp.SUM, group=self.group)
                return
            except (RuntimeError, NotImplementedError):
                self._rs_native = False
        rank = dist.get_rank(self.group)
        for p, chunk in enumerate(panel.chunk(self.world)):
            dist.reduce(chunk, dst=dist.get_global_rank(self.group, p) if self.group is not None else p, op=dist.ReduceOp.SUM, group=self.group)
            if p == rank:
                out.copy_(chunk)

    def _iterate_row_blocks(self, compute_error: bool):
        b, dist = self.backend, self.dist
        rank = dist.get_rank(self.group) if self.collectives and dist.is_initialized() else 0
        rows = b.padded_m // self.world
        if rows * self.world != b.padded_m or rows % 128 != 0:
            raise ValueError("row_blocks: the backend's padded row count must be a multiple of 128 * world (EngineShard(row_blocks=world))")
        row0 = rank * rows
        block = b.panel[row0 * b.padded_rank:(row0 + rows) * b.padded_rank]
        if self.collectives:
            mine = self.torch.empty_like(block)
            self._reduce_scatter(mine, b.panel)
            dist.all_reduce(b.hht, op=dist.ReduceOp.SUM, group=self.group)
            block = mine
        colsq = b.w_update_rows(block, row0, rows, compute_error)
        if self.collectives:
            dist.all_reduce(colsq, op=dist.ReduceOp.SUM, group=self.group)
        b.w_normalize_rows(row0, rows)
        if self.collectives:
            mine = b.w_panel[row0 * b.padded_rank:(row0 + rows) * b.padded_rank].clone()
            dist.all_gather(list(b.w_panel.chunk(self.world)), mine, group=self.group)
        b.w_rows_replaced()

    def iterate(self, compute_error: bool = False):
        b = self.backend
        b.h_step(compute_error)
        b.w_products()
        if self.mode == "row_blocks":
            self._iterate_row_blocks(compute_error)
        else:
            if self.collectives:
                self.dist.all_reduce(b.exchange, op=self.dist.ReduceOp.SUM, group=self.group)
            b.w_finish(compute_error)
        if compute_error:
            if self._vtv_all is None:      # once per factorisation (V does not change)
                self._vtv_all = np.sort(self._all_gather_host(b.error_terms(0)))
            if hasattr(b, "error_terms_async"):
                self._resolve_pending()    # the previous error iteration's terms arrived long ago
                self._launch_error_gather()
            else:
                htwtv = self._all_gather_host(b.error_terms(1))
                hhtwtw = b.error_terms(2)
                self._frobenius = b.resolve(self._vtv_all, htwtv, hhtwtw)
                self._rmsd = self._frobenius / np.sqrt(float(self.total_elements))

    def run(self, count: int, first_iteration: int = 1, error_every: int = 10, last_iteration: int = 0):
        for k in range(count):
            it = first_iteration + k
            err = (error_every > 0 and it % error_every == 0) or (last_iteration > 0 and it == last_iteration)
            self.iterate(err)
