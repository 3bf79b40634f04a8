"""nmfgpu_amd -- host-side Python mirror of the nmfgpu interface on top of libnmfgpu64.so.

The compute path is the C-ABI shared library built from nmfgpu_amd/csrc (hand-written gfx950
HIP kernels).  There is no Python, PyTorch or CPU fallback: importing the package works
anywhere (so that build and ABI checks can run without a GPU), but every compute entry point
raises / returns an error status when the library or a HIP device is missing.
"""
from ._lib import library, library_path, LibraryMissing  # noqa: F401
from .api import (  # noqa: F401
    ResultType, NmfInitializationMethod, NmfThresholdType, NmfAlgorithm, Verbosity, IndexBase, StorageFormat,
    ExecutionRecord, MatrixDescription, NmfDescription, Parameter, GpuInformation, KMeansDescription,
    initialize, finalize, version, choose_gpu, get_number_of_gpu, get_information_for_gpu_index,
    set_verbosity, compute, compute_kmeans, Summary, NmfError,
)
from .engine import Engine, EngineError, op_factor_product, op_factor_product_bf16, op_factor_product_x3, op_gram, op_inverse, op_factor_passes, op_tri_update, device_count, host_kmeans, host_init, RcclComm, LocalGroup, LocalComm, ShardedRun, shard_columns, SHARD_ROW_BLOCKS, SHARD_REPLICATED  # noqa: F401
