"""ctypes wrapper of include/nmfgpu_amd.h: the device-resident engine and the single-kernel ops."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from ._lib import library

ALGORITHMS = {"mu": 0, "gdcls": 1, "als": 2, "acls": 3, "ahcls": 4, "nsnmf": 5}
_STATUS = {0: "ok", 1: "invalid argument", 2: "out of device memory", 3: "out of host memory", 4: "HIP error", 5: "no HIP device",
           6: "values outside the exact range of the split-operand product"}


class EngineError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        super().__init__(f"{what}: {_STATUS.get(status, status)}{(' (' + detail + ')') if detail else ''}")
        self.status = status


class _Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("lam", "lambdaW", "lambdaH", "alphaW", "alphaH", "theta", "divergence", "sparse_compute", "precision")]


class _Geometry(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("r", C.c_int), ("padded_rank", C.c_int),
                ("padded_m", C.c_long), ("padded_n", C.c_long), ("slabs_h", C.c_int), ("slabs_w", C.c_int),
                ("exchange_count", C.c_long), ("product_kernel", C.c_int), ("resident_images", C.c_int), ("one_pass", C.c_int),
                ("kl_blocks_w", C.c_int), ("kl_blocks_h", C.c_int), ("gram_k_slices", C.c_int), ("w_col_split", C.c_int),
                ("fused_launches", C.c_int), ("sparse_setup", C.c_int), ("gram_ride_slices_h", C.c_int), ("gram_ride_slices_w", C.c_int)]


def device_count() -> int:
    fn = library().nmfamd_device_count
    fn.restype = C.c_int
    return int(fn())


def _f(a: np.ndarray) -> np.ndarray:
    if a.ndim != 2 or not a.flags.f_contiguous:
        raise ValueError("matrices must be 2-D Fortran-ordered arrays")
    return a


def _ld(a: np.ndarray) -> int:
    return a.strides[1] // a.itemsize if a.shape[1] > 1 else max(a.shape[0], 1)


class Engine:
    """One factorisation resident on the current HIP device (see include/nmfgpu_amd.h)."""

    def __init__(self, m: int, n: int, r: int, algorithm: str = "mu", dtype=np.float32, stream: int = 0,
                 lam=0.0, lambda_w=0.0, lambda_h=0.0, alpha_w=0.0, alpha_h=0.0, theta=0.0, divergence: str = "frobenius",
                 sparse_compute: bool = False, precision: str = "native", row_blocks: int = 1):
        self._lib = library()
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("float32 or float64")
        self.m, self.n, self.r = m, n, r
        self._ctor = dict(algorithm=algorithm, stream=stream, row_blocks=row_blocks,
                          params=[lam, lambda_w, lambda_h, alpha_w, alpha_h, theta, {"frobenius": 0.0, "kl": 1.0}[divergence], float(sparse_compute),
                                  {"native": 0.0, "bf16": 1.0, "fp32_mfma": -1.0}[precision]])
        self._h = None
        self._create()
        self._lib.nmfamd_engine_frobenius.restype = C.c_double
        self._lib.nmfamd_engine_rmsd.restype = C.c_double
        self._lib.nmfamd_engine_kl_divergence.restype = C.c_double
        self._lib.nmfamd_engine_last_error.restype = C.c_char_p
        self._lib.nmfamd_engine_error_terms.restype = C.c_long

    def _create(self):
        c = self._ctor
        p = _Params(*c["params"])
        h = C.c_void_p()
        # row_blocks > 1: the padded row count is a multiple of 128 * row_blocks (row-block form of the sharded W step)
        st = self._lib.nmfamd_engine_create_blocks(self.m, self.n, self.r, ALGORITHMS[c["algorithm"]], C.byref(p), self.dtype.itemsize,
                                                   C.c_void_p(c["stream"]), int(c["row_blocks"]), C.byref(h))
        if st != 0:
            raise EngineError(st, "nmfamd_engine_create")
        self._h = h

    def _check(self, st: int, what: str):
        if st != 0:
            raise EngineError(st, what, (self._lib.nmfamd_engine_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nmfamd_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, V: np.ndarray):
        V = _f(V)
        if V.dtype != self.dtype:
            raise TypeError(f"V must be {self.dtype}, got {V.dtype}")
        if V.shape != (self.m, self.n):
            raise ValueError(f"V must have shape {(self.m, self.n)}, got {V.shape}")
        self._upload(lambda: self._lib.nmfamd_engine_upload_dense(self._h, C.c_void_p(V.ctypes.data), C.c_long(_ld(V))), "upload_dense")

    def _upload(self, call, what: str):
        st = call()
        if st == 6 and self._ctor["params"][8] == 0.0:
            # NMFAMD_VALUE_RANGE: infinities, NaN, |v| > 2^126 or 0 < |v| < 2^-100 in V -- the split-operand product is not the
            # fp32 product there; recreate the engine on the native fp32 MFMA instructions, as nmfgpu::compute does.
            # The handle changes: anything created from the old one (a ShardedRun, w_panel_ptr()) is void -- upload V before
            # creating those.  Column-sharded callers must make this switch on every rank together (EngineShard does).
            self.close()
            self._ctor["params"][8] = -1.0
            self._create()
            st = call()
        self._check(st, what)

    def upload_sparse(self, fmt: int, values: np.ndarray, a: np.ndarray, b: np.ndarray, base: int = 0):
        values = np.ascontiguousarray(values, dtype=self.dtype)
        a = np.ascontiguousarray(a, dtype=np.int32); b = np.ascontiguousarray(b, dtype=np.int32)
        self._upload(lambda: self._lib.nmfamd_engine_upload_sparse(self._h, fmt, C.c_void_p(values.ctypes.data), C.c_void_p(a.ctypes.data),
                                                                   C.c_void_p(b.ctypes.data), C.c_long(len(values)), base), "upload_sparse")

    def set_factors(self, W: Optional[np.ndarray], H: Optional[np.ndarray]):
        # the C side sees raw pointers and leading dimensions only: a float64 W on a float32 engine, or a wrong shape,
        # would be reinterpreted (or read out of bounds) silently
        for name, a, shape in (("W", W, (self.m, self.r)), ("H", H, (self.r, self.n))):
            if a is None:
                continue
            if not isinstance(a, np.ndarray) or a.dtype != self.dtype:
                raise TypeError(f"{name} must be a {self.dtype} ndarray, got {getattr(a, 'dtype', type(a))}")
            if a.shape != shape:
                raise ValueError(f"{name} must have shape {shape}, got {a.shape}")
        wp = C.c_void_p(_f(W).ctypes.data) if W is not None else None
        hp = C.c_void_p(_f(H).ctypes.data) if H is not None else None
        self._check(self._lib.nmfamd_engine_set_factors(self._h, wp, C.c_long(_ld(W) if W is not None else 0),
                                                        hp, C.c_long(_ld(H) if H is not None else 0)), "set_factors")

    def get_factors(self):
        W = np.zeros((self.m, self.r), dtype=self.dtype, order="F")
        H = np.zeros((self.r, self.n), dtype=self.dtype, order="F")
        self._check(self._lib.nmfamd_engine_get_factors(self._h, C.c_void_p(W.ctypes.data), C.c_long(self.m),
                                                        C.c_void_p(H.ctypes.data), C.c_long(self.r)), "get_factors")
        return W, H

    def randomize(self, seed: int, w: bool = True, h: bool = True):
        self._check(self._lib.nmfamd_engine_randomize(self._h, C.c_uint(seed), int(w), int(h)), "randomize")

    def iterate(self, count: int, first_iteration: int = 1, error_every: int = 10, last_iteration: int = 0, constant_w: bool = False):
        self._check(self._lib.nmfamd_engine_iterate(self._h, count, first_iteration, error_every, last_iteration, int(constant_w)), "iterate")

    def synchronize(self):
        self._check(self._lib.nmfamd_engine_synchronize(self._h), "synchronize")

    @property
    def frobenius(self) -> float:
        return float(self._lib.nmfamd_engine_frobenius(self._h))

    @property
    def kl_divergence(self) -> float:
        return float(self._lib.nmfamd_engine_kl_divergence(self._h))

    @property
    def rmsd(self) -> float:
        return float(self._lib.nmfamd_engine_rmsd(self._h))

    def kernel_timing(self, every: int):
        """every = k > 0: time the factor-product launches of every k-th iteration; 0: off."""
        self._check(self._lib.nmfamd_engine_kernel_timing(self._h, int(every)), "kernel_timing")

    def kernel_timing_read(self):
        ms = C.c_double(0); cnt = C.c_long(0)
        self._check(self._lib.nmfamd_engine_kernel_timing_read(self._h, C.byref(ms), C.byref(cnt)), "kernel_timing_read")
        return ms.value, cnt.value

    def kernel_timing_read2(self):
        """(total ms, launches, ms an empty event pair reports on the idle stream)."""
        ms = C.c_double(0); cnt = C.c_long(0); ov = C.c_double(0)
        self._check(self._lib.nmfamd_engine_kernel_timing_read2(self._h, C.byref(ms), C.byref(cnt), C.byref(ov)), "kernel_timing_read2")
        return ms.value, cnt.value, ov.value

    def kernel_timing_read3(self):
        """(total_ms, launches, idle_pair_ms, (ms_h, ms_w), (launches_h, launches_w)): the product launches split into the H side (W^T V) and the W side (V H^T)."""
        ms, cnt, ov = C.c_double(0), C.c_long(0), C.c_double(0)
        km, kc = (C.c_double * 2)(), (C.c_long * 2)()
        self._check(self._lib.nmfamd_engine_kernel_timing_read3(self._h, C.byref(ms), C.byref(cnt), C.byref(ov), km, kc), "kernel_timing_read3")
        return ms.value, cnt.value, ov.value, (km[0], km[1]), (kc[0], kc[1])

    def geometry(self) -> dict:
        # (the sized getter: a library older or newer than this mirror never writes past the struct, include/nmfgpu_amd.h)
        g = _Geometry()
        self._check(self._lib.nmfamd_engine_geometry_sized(self._h, C.byref(g), C.c_ulong(C.sizeof(g))), "geometry")
        return {k: getattr(g, k) for k, _ in _Geometry._fields_}

    # ---- column-sharded form ----
    def h_step(self, compute_error: bool = False):
        self._check(self._lib.nmfamd_engine_h_step(self._h, int(compute_error)), "h_step")

    def set_sole_rank(self, sole: bool):
        """A team of one rank: the exchange buffer goes from w_products to w_finish unchanged (nmfamd_engine_set_sole_rank)."""
        self._check(self._lib.nmfamd_engine_set_sole_rank(self._h, int(bool(sole))), "set_sole_rank")

    def w_products(self, exchange_ptr: int):
        self._check(self._lib.nmfamd_engine_w_products(self._h, C.c_void_p(exchange_ptr)), "w_products")

    def w_finish(self, exchange_ptr: int, compute_error: bool = False):
        self._check(self._lib.nmfamd_engine_w_finish(self._h, C.c_void_p(exchange_ptr), int(compute_error)), "w_finish")

    # ---- row-block form of the sharded W step (engine created with row_blocks = world) ----
    def w_update_rows(self, num_rows_ptr: int, hht_ptr: int, row0: int, rows: int, compute_error: bool, colsq_ptr: int):
        self._check(self._lib.nmfamd_engine_w_update_rows(self._h, C.c_void_p(num_rows_ptr), C.c_void_p(hht_ptr), C.c_long(row0), C.c_long(rows),
                                                          int(compute_error), C.c_void_p(colsq_ptr)), "w_update_rows")

    def w_normalize_rows(self, row0: int, rows: int, colsq_ptr: int):
        self._check(self._lib.nmfamd_engine_w_normalize_rows(self._h, C.c_long(row0), C.c_long(rows), C.c_void_p(colsq_ptr)), "w_normalize_rows")

    def w_rows_replaced(self):
        self._check(self._lib.nmfamd_engine_w_rows_replaced(self._h), "w_rows_replaced")

    def w_panel_ptr(self) -> int:
        fn = self._lib.nmfamd_engine_w_panel
        fn.restype = C.c_void_p
        return int(fn(self._h))

    def error_terms(self, which: int) -> np.ndarray:
        cap = max(self.n, self.r)
        out = np.zeros(cap, dtype=self.dtype)
        cnt = self._lib.nmfamd_engine_error_terms(self._h, which, C.c_void_p(out.ctypes.data), C.c_long(cap))
        if cnt < 0:
            raise EngineError(1, "error_terms")
        return out[:cnt]

    def error_terms_to_device(self, dst_ptr: int, capacity: int) -> int:
        """[n_local tr(H^T W^T V) terms | r tr(H H^T W^T W) terms] of the last error iteration -> device buffer (async)."""
        fn = self._lib.nmfamd_engine_error_terms_to_device
        fn.restype = C.c_long
        cnt = fn(self._h, C.c_void_p(dst_ptr), C.c_long(capacity))
        if cnt < 0:
            raise EngineError(1, "error_terms_to_device")
        return int(cnt)

    def debug_read(self, which: int, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=self.dtype)
        self._check(self._lib.nmfamd_engine_debug_read(self._h, which, C.c_void_p(out.ctypes.data), C.c_long(count)), "debug_read")
        return out


class RcclComm:
    """One rank of an RCCL clique, created through RCCL's C API inside libnmfgpu64.so (no torch).  Rank 0 calls
    RcclComm.unique_id() and hands the 128 bytes to the other ranks; every rank then constructs its communicator with
    its HIP device current (blocks until all ranks have arrived)."""

    def __init__(self, unique_id: bytes, world: int, rank: int):
        self._lib = library()
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id has 128 bytes")
        h = C.c_void_p()
        st = self._lib.nmfamd_comm_create_rccl(C.c_char_p(unique_id), int(world), int(rank), C.byref(h))
        if st != 0:
            raise EngineError(st, "nmfamd_comm_create_rccl")
        self._h = h
        self.world, self.rank = int(world), int(rank)

    @staticmethod
    def available() -> bool:
        return bool(library().nmfamd_comm_rccl_available())

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        st = library().nmfamd_comm_unique_id(buf)
        if st != 0:
            raise EngineError(st, "nmfamd_comm_unique_id")
        return buf.raw

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nmfamd_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalGroup:
    """Shared state of the rank THREADS of one process (include/nmfgpu_amd.h, nmfamd_local_group_*): create one, hand it to every rank
    thread, each constructs its LocalComm with its own HIP device current."""

    def __init__(self, world: int):
        self._lib = library()
        h = C.c_void_p()
        st = self._lib.nmfamd_local_group_create(int(world), C.byref(h))
        if st != 0:
            raise EngineError(st, "nmfamd_local_group_create")
        self._h, self.world = h, int(world)

    def abort(self):
        if getattr(self, "_h", None):
            self._lib.nmfamd_local_group_abort(self._h)

    def selftest_report(self) -> str:
        """One line about the transport's set-up self-test (empty before the ranks have joined, or when NMFAMD_SELFTEST=0 skipped it)."""
        if not getattr(self, "_h", None):
            return ""
        self._lib.nmfamd_local_group_selftest.restype = C.c_char_p
        return (self._lib.nmfamd_local_group_selftest(self._h) or b"").decode()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nmfamd_local_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalComm:
    """One rank of the in-process transport: the ranks' kernels read each other's buffers where they lie (same device, or peer-mapped devices over xGMI).
    The constructor blocks until every rank of the group has joined."""

    def __init__(self, group: LocalGroup, rank: int):
        self._lib = library()
        self.group = group                       # keep it alive
        h = C.c_void_p()
        st = self._lib.nmfamd_comm_create_local(group._h, int(rank), C.byref(h))
        if st != 0:
            self._lib.nmfamd_local_group_last_error.restype = C.c_char_p
            why = (self._lib.nmfamd_local_group_last_error(group._h) or b"").decode()
            raise EngineError(st, "nmfamd_comm_create_local", why or "a rank of the group failed or the group was aborted")
        self._h = h
        self.world, self.rank = group.world, int(rank)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nmfamd_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SHARD_ROW_BLOCKS, SHARD_REPLICATED = 0, 1


class ShardedRun:
    """The column-sharded iteration driven natively (include/nmfgpu_amd.h, nmfamd_sharded_*): `engine` holds this rank's
    columns [total_columns * rank / world, total_columns * (rank + 1) / world) and was created with row_blocks = world."""

    def __init__(self, engine: Engine, comm: RcclComm, rows: int, total_columns: int, mode: int = SHARD_ROW_BLOCKS):
        self._lib = library()
        self.engine, self.comm = engine, comm          # keep both alive
        h = C.c_void_p()
        st = self._lib.nmfamd_sharded_create(engine._h, comm._h, int(mode), C.c_long(rows), C.c_long(total_columns), C.byref(h))
        if st != 0:
            raise EngineError(st, "nmfamd_sharded_create")
        self._h = h
        self._lib.nmfamd_sharded_frobenius.restype = C.c_double
        self._lib.nmfamd_sharded_rmsd.restype = C.c_double
        self._lib.nmfamd_sharded_last_error.restype = C.c_char_p

    def iterate(self, count: int, first_iteration: int = 1, error_every: int = 10, last_iteration: int = 0):
        st = self._lib.nmfamd_sharded_iterate(self._h, int(count), int(first_iteration), int(error_every), int(last_iteration))
        if st != 0:
            raise EngineError(st, "nmfamd_sharded_iterate", (self._lib.nmfamd_sharded_last_error(self._h) or b"").decode())

    def gather_w(self):
        """Row-block mode with the bf16 fragment exchange: gathers the fp32 rows of every rank's block (a collective: every rank calls it)."""
        st = self._lib.nmfamd_sharded_gather_w(self._h)
        if st != 0:
            raise EngineError(st, "nmfamd_sharded_gather_w", (self._lib.nmfamd_sharded_last_error(self._h) or b"").decode())

    @property
    def frobenius(self) -> float:
        return float(self._lib.nmfamd_sharded_frobenius(self._h))

    @property
    def rmsd(self) -> float:
        return float(self._lib.nmfamd_sharded_rmsd(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nmfamd_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_columns(total: int, world: int, rank: int):
    """(first, count) of the columns rank `rank` holds when `total` columns are dealt to `world` ranks (sharded.h)."""
    a, b = (total * rank) // world, (total * (rank + 1)) // world
    return a, b - a


def resolve_frobenius(vtv_sorted: np.ndarray, htwtv: np.ndarray, hhtwtw: np.ndarray) -> float:
    lib = library()
    sfx = "f32" if vtv_sorted.dtype == np.float32 else "f64"
    fn = getattr(lib, f"nmfamd_resolve_frobenius_{sfx}")
    fn.restype = C.c_double
    a = np.ascontiguousarray(vtv_sorted); b = np.array(htwtv, dtype=a.dtype); c = np.array(hhtwtw, dtype=a.dtype)
    return float(fn(C.c_void_p(a.ctypes.data), C.c_long(len(a)), C.c_void_p(b.ctypes.data), C.c_long(len(b)),
                    C.c_void_p(c.ctypes.data), C.c_long(len(c))))


def op_factor_product(A: np.ndarray, F: np.ndarray, use_valu: bool = False):
    """OUT (r x X) = F (r x Y) A^T for a host X x Y matrix A.  Returns (OUT, slabs)."""
    A = _f(A); F = _f(F)
    X, Y = A.shape
    r = F.shape[0]
    assert F.shape[1] == Y and A.dtype == F.dtype
    out = np.zeros((r, X), dtype=A.dtype, order="F")
    lib = library()
    if A.dtype == np.float32:
        slabs = C.c_int(0)
        st = lib.nmfamd_op_factor_product_f32(C.c_void_p(A.ctypes.data), C.c_long(_ld(A)), X, Y, C.c_void_p(F.ctypes.data), C.c_long(_ld(F)), r,
                                              C.c_void_p(out.ctypes.data), C.c_long(r), int(use_valu), C.byref(slabs))
        if st != 0:
            raise EngineError(st, "nmfamd_op_factor_product_f32")
        return out, slabs.value
    slabs = C.c_int(0)
    st = lib.nmfamd_op_factor_product_f64(C.c_void_p(A.ctypes.data), C.c_long(_ld(A)), X, Y, C.c_void_p(F.ctypes.data), C.c_long(_ld(F)), r,
                                          C.c_void_p(out.ctypes.data), C.c_long(r), int(use_valu), C.byref(slabs))
    if st != 0:
        raise EngineError(st, "nmfamd_op_factor_product_f64")
    return out, slabs.value


def op_factor_product_bf16(A: np.ndarray, F: np.ndarray) -> np.ndarray:
    """OUT (r x X) = F A^T with both operands rounded to bf16 and fp32 accumulation (any r)."""
    A = _f(A); F = _f(F)
    X, Y = A.shape
    r = F.shape[0]
    out = np.zeros((r, X), dtype=np.float32, order="F")
    st = library().nmfamd_op_factor_product_bf16(C.c_void_p(A.ctypes.data), C.c_long(_ld(A)), X, Y, C.c_void_p(F.ctypes.data), C.c_long(_ld(F)), r,
                                                 C.c_void_p(out.ctypes.data), C.c_long(r))
    if st != 0:
        raise EngineError(st, "nmfamd_op_factor_product_bf16")
    return out


def op_factor_product_x3(A: np.ndarray, F: np.ndarray, reps: int = 0):
    """OUT (r x X) = F A^T at fp32 accuracy on the bf16 matrix pipe: both operands split exactly into three bf16
    terms, six cross products, fp32 accumulation (any r).  Returns OUT, or (OUT, microseconds per launch) if reps > 0."""
    A = _f(A); F = _f(F)
    X, Y = A.shape
    r = F.shape[0]
    out = np.zeros((r, X), dtype=np.float32, order="F")
    us = C.c_double(0.0)
    st = library().nmfamd_op_factor_product_x3(C.c_void_p(A.ctypes.data), C.c_long(_ld(A)), X, Y, C.c_void_p(F.ctypes.data), C.c_long(_ld(F)), r,
                                               C.c_void_p(out.ctypes.data), C.c_long(r), int(reps), C.byref(us))
    if st != 0:
        raise EngineError(st, "nmfamd_op_factor_product_x3")
    return (out, us.value) if reps > 0 else out


def op_gram(P: np.ndarray) -> np.ndarray:
    P = _f(P)
    r, length = P.shape
    G = np.zeros((r, r), dtype=P.dtype, order="F")
    fn = library().nmfamd_op_gram_f32 if P.dtype == np.float32 else library().nmfamd_op_gram_f64
    st = fn(C.c_void_p(P.ctypes.data), C.c_long(_ld(P)), r, length, C.c_void_p(G.ctypes.data), C.c_long(r))
    if st != 0:
        raise EngineError(st, "nmfamd_op_gram")
    return G


def op_inverse(A: np.ndarray, offdiag: float = 0.0, diag: float = 0.0) -> np.ndarray:
    A = _f(A)
    r = A.shape[0]
    out = np.zeros((r, r), dtype=np.float32, order="F")
    st = library().nmfamd_op_inverse_f32(C.c_void_p(A.ctypes.data), C.c_long(_ld(A)), r, C.c_float(offdiag), C.c_float(diag),
                                         C.c_void_p(out.ctypes.data), C.c_long(r))
    if st != 0:
        raise EngineError(st, "nmfamd_op_inverse_f32")
    return out


def op_factor_passes(P: np.ndarray, theta: float = 0.0, colsq: Optional[np.ndarray] = None, reps: int = 0):
    """The passes between the update of a factor panel and the next product at padded rank 256 with bf16 product operands
    (kernels_tri.hip).  P: (len, r) panel rows, 129 <= r <= 256.  Returns a dict: `panel` (after the optional column normalisation
    by the r sums of squares `colsq`), `pack` (the smoothed panel as the bf16 product operand, widened to fp32), `gram` (P^T P of
    the returned panel), `gram_smoothed`, and with reps > 0 `us_finish`, `us_gram` (microseconds per launch group)."""
    P = np.ascontiguousarray(P, dtype=np.float32)
    length, r = P.shape
    out = {"panel": np.zeros_like(P), "pack": np.zeros_like(P), "gram": np.zeros((r, r), np.float32), "gram_smoothed": np.zeros((r, r), np.float32)}
    sq = None if colsq is None else np.ascontiguousarray(colsq, dtype=np.float32)
    if sq is not None and sq.shape != (r,):
        raise ValueError("colsq must hold r values")
    t0, t1 = C.c_double(0.0), C.c_double(0.0)
    st = library().nmfamd_op_factor_passes_f32(C.c_void_p(P.ctypes.data), C.c_long(r), r, length, C.c_void_p(sq.ctypes.data) if sq is not None else None,
                                               C.c_float(theta), C.c_void_p(out["panel"].ctypes.data), C.c_void_p(out["pack"].ctypes.data),
                                               C.c_void_p(out["gram"].ctypes.data), C.c_void_p(out["gram_smoothed"].ctypes.data), int(reps), C.byref(t0), C.byref(t1))
    if st != 0:
        raise EngineError(st, "nmfamd_op_factor_passes_f32")
    if reps > 0:
        out["us_finish"], out["us_gram"] = t0.value, t1.value
    return out


def op_tri_update(P: np.ndarray, num: np.ndarray, Q: np.ndarray, *, old_colsq: Optional[np.ndarray] = None, transform_num: bool = False,
                  num_colsq: Optional[np.ndarray] = None, theta: float = 0.0, frag_theta: float = 0.0, transform_den: bool = False):
    """One multiplicative update of a (len, r) panel the way the rank-256 bf16 path runs it (nmfamd_op_tri_update_f32): pending column scale
    on the old values (given as the sums of squares it comes from), optional scale + smoothing of the numerator rows, the new rows
    unnormalised (`panel`), their bf16 rounding as the product operand (`pack`; smoothed by frag_theta first), the new pending scale
    (`scale`) and the scaled Gram matrix of the rounded rows (`gram`); `gram_raw`, `gram_image` and `diag`: the unscaled matrix, the same matrix read back
    from the split image the reduction writes, its diagonal.  transform_den: the denominator is S D Q D S old instead of old Q (D from num_colsq)."""
    P = np.ascontiguousarray(P, dtype=np.float32); num = np.ascontiguousarray(num, dtype=np.float32); Q = np.ascontiguousarray(Q, dtype=np.float32)
    length, r = P.shape
    if num.shape != P.shape or Q.shape != (r, r):
        raise ValueError("shapes: P, num (len, r); Q (r, r)")
    osc = None if old_colsq is None else np.ascontiguousarray(old_colsq, dtype=np.float32)
    nsc = None if num_colsq is None else np.ascontiguousarray(num_colsq, dtype=np.float32)
    out = {"panel": np.zeros_like(P), "pack": np.zeros_like(P), "scale": np.zeros(r, np.float32), "gram": np.zeros((r, r), np.float32),
           "gram_raw": np.zeros((r, r), np.float32), "gram_image": np.zeros((r, r), np.float32), "diag": np.zeros(r, np.float32)}
    st = library().nmfamd_op_tri_update_f32(C.c_void_p(P.ctypes.data), C.c_void_p(num.ctypes.data), C.c_void_p(Q.ctypes.data), r, length,
                                            C.c_void_p(osc.ctypes.data) if osc is not None else None, int(bool(transform_num)),
                                            C.c_void_p(nsc.ctypes.data) if nsc is not None else None, C.c_float(theta), C.c_float(frag_theta),
                                            int(bool(transform_den)),
                                            C.c_void_p(out["panel"].ctypes.data), C.c_void_p(out["pack"].ctypes.data), C.c_void_p(out["scale"].ctypes.data),
                                            C.c_void_p(out["gram"].ctypes.data), C.c_void_p(out["gram_raw"].ctypes.data), C.c_void_p(out["gram_image"].ctypes.data),
                                            C.c_void_p(out["diag"].ctypes.data))
    if st != 0:
        raise EngineError(st, "nmfamd_op_tri_update_f32")
    return out


def host_kmeans(data: np.ndarray, k: int, *, seed: int = 0, iterations: int = 100, threshold: float = 0.005):
    """The host-side Lloyd k-means behind computeKMeans and the KMeans*/EInNMF initialisers, without a
    device or context (nmfamd_host_kmeans_*).  Returns (clusters m x k, membership, passes)."""
    lib = library()
    data = _f(data)
    m, n = data.shape
    clusters = np.zeros((m, k), dtype=data.dtype, order="F")
    membership = np.zeros(n, dtype=np.uint32)
    it = C.c_uint(0)
    fn = getattr(lib, "nmfamd_host_kmeans_f32" if data.dtype == np.float32 else "nmfamd_host_kmeans_f64")
    st = fn(C.c_void_p(data.ctypes.data), C.c_long(_ld(data)), m, n, C.c_void_p(clusters.ctypes.data), C.c_long(m), k,
            C.c_void_p(membership.ctypes.data), C.c_uint(seed), C.c_uint(iterations), C.c_double(threshold), C.byref(it))
    if st != 0:
        raise EngineError(st, "host_kmeans")
    return clusters, membership, int(it.value)


NNDSVD, NNDSVD_A, NNDSVD_AR = 100, 101, 102      # `method` values of host_init for the SVD-based start (Parameter "nndsvd" = 0 / 1 / 2 of nmfgpu::compute)


def host_init(V: np.ndarray, r: int, method: int, *, seed: int = 0, want_h: bool = True):
    """W (m x r) and H (r x n) of the MeanColumns / KMeans* / EInNMF initialisers, or of NNDSVD / NNDSVDa / NNDSVDar (method 100 / 101 / 102) (nmfamd_host_init_*)."""
    lib = library()
    V = _f(V)
    m, n = V.shape
    W = np.zeros((m, r), dtype=V.dtype, order="F")
    H = np.zeros((r, n), dtype=V.dtype, order="F") if want_h else None
    fn = getattr(lib, "nmfamd_host_init_f32" if V.dtype == np.float32 else "nmfamd_host_init_f64")
    st = fn(C.c_void_p(V.ctypes.data), C.c_long(_ld(V)), m, n, r, int(method), C.c_uint(seed), C.c_void_p(W.ctypes.data),
            C.c_void_p(H.ctypes.data) if want_h else None)
    if st != 0:
        raise EngineError(st, "host_init")
    return W, H
