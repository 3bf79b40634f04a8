"""Python mirror of include/nmfgpu.h: the same enums, packed descriptors and entry points.

Structures use ``_pack_ = 4`` like the header's ``#pragma pack(push, 4)`` (ref include/nmfgpu.h:47)
and are passed to the library's ``extern "C"`` layer (ref include/nmfgpu.h:329-349) unchanged, so
this module is exactly what an FFI user of the reference (e.g. the R binding) does.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Optional, Sequence

import numpy as np

from ._lib import library


class ResultType(enum.IntEnum):  # ref include/nmfgpu.h:52-77
    Success = 0
    ErrorAlreadyInitialized = 1
    ErrorNotInitialized = 2
    ErrorInvalidArgument = 3
    ErrorNotEnoughHostMemory = 4
    ErrorNotEnoughDeviceMemory = 5
    ErrorExternalLibrary = 6
    ErrorUserInterrupt = 7
    ErrorDeviceSelection = 8


class NmfInitializationMethod(enum.IntEnum):  # ref :80-100
    CopyExisting = 0
    AllRandomValues = 1
    MeanColumns = 2
    KMeansAndRandomValues = 3
    KMeansAndAbsoluteWTV = 4
    KMeansAndNonNegativeWTV = 5
    EInNMF = 6


class NmfThresholdType(enum.IntEnum):  # ref :102-105
    Frobenius = 0
    RMSD = 1


class NmfAlgorithm(enum.IntEnum):  # ref :107-114
    Multiplicative = 0
    GDCLS = 1
    ALS = 2
    ACLS = 3
    AHCLS = 4
    nsNMF = 5


class Verbosity(enum.IntEnum):  # ref :117-126
    Nothing = 0
    Summary = 1
    Informative = 2
    Debugging = 3


class IndexBase(enum.IntEnum):  # ref :129-134
    Zero = 0
    One = 1


class StorageFormat(enum.IntEnum):  # ref :177-186
    Dense = 0
    CSR = 1
    CSC = 2
    COO = 3


class NmfError(RuntimeError):
    def __init__(self, result: ResultType, what: str):
        super().__init__(f"{what}: {result.name}")
        self.result = result


class ExecutionRecord(C.Structure):  # ref :138-147
    _pack_ = 4
    _fields_ = [("frobenius", C.c_double), ("rmsd", C.c_double), ("elapsedTime", C.c_double),
                ("sparsityW", C.c_double), ("sparsityH", C.c_double), ("numIterations", C.c_uint)]


class _Dense(C.Structure):
    _pack_ = 4
    _fields_ = [("values", C.c_void_p), ("leadingDimension", C.c_uint)]


class _Sparse(C.Structure):  # csr / csc / coo share one shape (ref :205-231)
    _pack_ = 4
    _fields_ = [("values", C.c_void_p), ("ptrA", C.c_void_p), ("ptrB", C.c_void_p), ("nnz", C.c_uint), ("base", C.c_int)]


class _Storage(C.Union):
    _pack_ = 4
    _fields_ = [("dense", _Dense), ("sparse", _Sparse)]


class MatrixDescription(C.Structure):  # ref :188-232 (same layout for float and double)
    _pack_ = 4
    _anonymous_ = ("u",)
    _fields_ = [("rows", C.c_uint), ("columns", C.c_uint), ("format", C.c_int), ("u", _Storage)]


class Parameter(C.Structure):  # ref :234-237
    _pack_ = 4
    _fields_ = [("name", C.c_char_p), ("value", C.c_double)]


UserInterruptCallback = C.CFUNCTYPE(C.c_bool)  # ref :136


class NmfDescription(C.Structure):  # ref :239-273
    _pack_ = 4
    _fields_ = [("algorithm", C.c_int), ("useConstantBasisVectors", C.c_bool),
                ("inputMatrix", MatrixDescription), ("inputLabels", C.c_void_p),
                ("outputMatrixW", MatrixDescription), ("outputMatrixH", MatrixDescription),
                ("features", C.c_uint), ("initMethod", C.c_int), ("numIterations", C.c_uint),
                ("numRuns", C.c_uint), ("seed", C.c_uint), ("thresholdType", C.c_int),
                ("thresholdValue", C.c_double), ("callbackUserInterrupt", UserInterruptCallback),
                ("parameters", C.POINTER(Parameter)), ("numParameters", C.c_uint)]


class GpuInformation(C.Structure):  # ref :288-292
    _pack_ = 4
    _fields_ = [("name", C.c_char * 256), ("totalMemory", C.c_size_t), ("freeMemory", C.c_size_t)]


class KMeansDescription(C.Structure):  # ref :301-310
    _pack_ = 4
    _fields_ = [("inputMatrix", MatrixDescription), ("outputMatrixClusters", MatrixDescription),
                ("outputMemberships", C.c_void_p), ("numClusters", C.c_uint), ("numIterations", C.c_uint),
                ("seed", C.c_uint), ("thresholdValue", C.c_double)]


def _dtype_suffix(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "single"
    if dtype == np.float64:
        return "double"
    raise TypeError("nmfgpu supports float32 and float64 matrices")


def dense_description(a: np.ndarray) -> MatrixDescription:
    """Describes a Fortran-ordered 2-D numpy array (the array must outlive the call)."""
    if a.ndim != 2 or not a.flags.f_contiguous:
        raise ValueError("dense matrices must be 2-D and column-major (np.asfortranarray)")
    d = MatrixDescription()
    d.rows, d.columns, d.format = a.shape[0], a.shape[1], StorageFormat.Dense
    d.dense.values = a.ctypes.data
    d.dense.leadingDimension = max(a.strides[1] // a.itemsize, a.shape[0]) if a.shape[1] > 1 else a.shape[0]
    return d


def sparse_description(fmt: StorageFormat, rows: int, cols: int, values: np.ndarray, a: np.ndarray, b: np.ndarray,
                       base: IndexBase = IndexBase.Zero) -> MatrixDescription:
    """CSR: a=rowPtr b=columnIndices; CSC: a=columnPtr b=rowIndices; COO: a=rowIndices b=columnIndices (int32)."""
    for arr in (a, b):
        if arr.dtype != np.int32 or not arr.flags.c_contiguous:
            raise ValueError("index arrays must be contiguous int32")
    d = MatrixDescription()
    d.rows, d.columns, d.format = rows, cols, int(fmt)
    d.sparse.values = values.ctypes.data
    d.sparse.ptrA = a.ctypes.data
    d.sparse.ptrB = b.ctypes.data
    d.sparse.nnz = len(values)
    d.sparse.base = int(base)
    return d


# ---- entry points (the extern "C" layer, ref include/nmfgpu.h:329-349) ----------------------

def _call(name, *args, restype=C.c_int):
    fn = getattr(library(), name)
    fn.restype = restype
    return fn(*args)


def initialize() -> ResultType:
    return ResultType(_call("nmfgpu_initialize"))


def finalize() -> ResultType:
    return ResultType(_call("nmfgpu_finalize"))


def version() -> int:
    return int(_call("nmfgpu_version"))


def set_verbosity(v: Verbosity) -> None:
    _call("nmfgpu_set_verbosity", C.c_int(int(v)), restype=None)


def choose_gpu(index: int) -> ResultType:
    return ResultType(_call("nmfgpu_choose_gpu", C.c_uint(index)))


def get_number_of_gpu() -> int:
    return int(_call("nmfgpu_get_number_of_gpu", restype=C.c_uint))


def get_information_for_gpu_index(index: int):
    info = GpuInformation()
    res = ResultType(_call("nmfgpu_get_information_for_gpu_index", C.c_uint(index), C.byref(info)))
    return res, info


class Summary:
    """nmfgpu::ISummary through its vtable (destroy, bestRun, record, recordCount; ref :149-175)."""

    _PROTOS = (C.CFUNCTYPE(None, C.c_void_p), C.CFUNCTYPE(C.c_uint, C.c_void_p),
               C.CFUNCTYPE(None, C.c_void_p, C.c_uint, C.POINTER(ExecutionRecord)), C.CFUNCTYPE(C.c_uint, C.c_void_p))

    def __init__(self):
        ptr = C.c_void_p()
        res = ResultType(_call("nmfgpu_create_summary", C.byref(ptr)))
        if res != ResultType.Success:
            raise NmfError(res, "nmfgpu_create_summary")
        self._ptr = ptr
        vtable = C.cast(C.cast(ptr, C.POINTER(C.c_void_p))[0], C.POINTER(C.c_void_p))
        self._fns = [proto(vtable[i]) for i, proto in enumerate(self._PROTOS)]

    @property
    def pointer(self):
        return self._ptr

    def best_run(self) -> int:
        return int(self._fns[1](self._ptr))

    def record_count(self) -> int:
        return int(self._fns[3](self._ptr))

    def record(self, index: int) -> ExecutionRecord:
        rec = ExecutionRecord()
        self._fns[2](self._ptr, index, C.byref(rec))
        return rec

    def destroy(self):
        if self._ptr:
            self._fns[0](self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def compute(V, W: np.ndarray, H: np.ndarray, *, algorithm: NmfAlgorithm = NmfAlgorithm.Multiplicative,
            init: NmfInitializationMethod = NmfInitializationMethod.CopyExisting, iterations: int = 100, runs: int = 1,
            seed: int = 0, threshold_type: NmfThresholdType = NmfThresholdType.Frobenius, threshold: float = 0.0,
            constant_basis_vectors: bool = False, parameters: Optional[dict] = None, summary: Optional[Summary] = None,
            interrupt=None, description_out: Optional[list] = None) -> ResultType:
    """nmfgpu::compute (ref include/nmfgpu.h:298-299).  V: Fortran ndarray or a MatrixDescription
    (sparse); W (m x r) and H (r x n) are Fortran arrays that provide the start values for
    CopyExisting and receive the result."""
    if W.dtype != H.dtype:
        raise TypeError("W and H must share a dtype")
    d = NmfDescription()
    d.algorithm = int(algorithm)
    d.useConstantBasisVectors = constant_basis_vectors
    if isinstance(V, MatrixDescription):
        d.inputMatrix = V
    else:
        if V.dtype != W.dtype:
            raise TypeError("V, W and H must share a dtype")
        d.inputMatrix = dense_description(V)
    d.outputMatrixW = dense_description(W)
    d.outputMatrixH = dense_description(H)
    d.features = W.shape[1]
    d.initMethod = int(init)
    d.numIterations, d.numRuns, d.seed = iterations, runs, seed
    d.thresholdType, d.thresholdValue = int(threshold_type), threshold
    cb = UserInterruptCallback(interrupt) if interrupt is not None else UserInterruptCallback()
    d.callbackUserInterrupt = cb
    params = parameters or {}
    arr = (Parameter * max(len(params), 1))()
    keep = []
    for i, (k, v) in enumerate(params.items()):
        name = k.encode()
        keep.append(name)
        arr[i].name, arr[i].value = name, float(v)
    d.parameters = C.cast(arr, C.POINTER(Parameter))
    d.numParameters = len(params)
    res = ResultType(_call(f"nmfgpu_compute_{_dtype_suffix(W.dtype)}", C.byref(d), summary.pointer if summary else None))
    if description_out is not None:
        description_out.append(d)
    return res


def compute_kmeans(V: np.ndarray, clusters: np.ndarray, *, iterations: int = 100, seed: int = 0, threshold: float = 0.005):
    """nmfgpu::computeKMeans (ref include/nmfgpu.h:326-327).  Returns (ResultType, memberships)."""
    d = KMeansDescription()
    d.inputMatrix = dense_description(V)
    d.outputMatrixClusters = dense_description(clusters)
    memb = np.zeros(V.shape[1], dtype=np.uint32)
    d.outputMemberships = memb.ctypes.data
    d.numClusters, d.numIterations, d.seed, d.thresholdValue = clusters.shape[1], iterations, seed, threshold
    res = ResultType(_call(f"nmfgpu_compute_kmeans_{_dtype_suffix(V.dtype)}", C.byref(d)))
    return res, memb
