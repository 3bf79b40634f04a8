"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE -- see oracle/nmf_oracle.c).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg import this
module.  The shipped engine (``nmfgpu_amd``) never does.

All matrices are numpy arrays in Fortran (column-major) order, like the reference's
DeviceMatrix; ``ld`` is taken from the array's strides.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libnmf_oracle.so")
_KM_PATH = os.path.join(_HERE, "_build", "libkmeans_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libnmfgpu_refhost.so")

ALGORITHMS = {"mu": 0, "gdcls": 1, "als": 2, "acls": 3, "ahcls": 4, "nsnmf": 5}


def build(force: bool = False) -> str:
    """Compile the oracle (and, if /root/reference is present, the reference host units)."""
    stale = force or not os.path.exists(_LIB_PATH) or not os.path.exists(_KM_PATH) or (
        os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                           for f in ("nmf_oracle.c", "nmf_oracle_impl.h", "sgemm_avx2.h"))) or (
        os.path.getmtime(_KM_PATH) < os.path.getmtime(os.path.join(_HERE, "kmeans_oracle.cpp")))
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference/source/nmf") and not os.path.exists(_REF_PATH):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return _LIB_PATH


_lib = None


def cpu_share() -> int:
    """Threads this process may use: its affinity mask, capped at 16 (a GPU box's CPU share per GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("NMF_ORACLE_MAX_THREADS", "16"))))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        os.environ.setdefault("OMP_NUM_THREADS", str(cpu_share()))
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = C.c_int
        _lib.oracle_summary_best_run.restype = C.c_uint
        for sfx in ("f32", "f64"):
            getattr(_lib, f"oracle_resolve_frobenius_{sfx}").restype = C.c_double
            getattr(_lib, f"oracle_direct_frobenius_{sfx}").restype = C.c_double
            getattr(_lib, f"oracle_run_{sfx}").restype = C.c_int
    return _lib


_km = None


def kmeans_lib() -> C.CDLL:
    global _km
    if _km is None:
        build()
        _km = C.CDLL(_KM_PATH)
        _km.oracle_kmeans_f32.restype = C.c_uint
        _km.oracle_kmeans_f64.restype = C.c_uint
    return _km


def ref_lib():
    """The reference's own Algorithm.cpp / Summary.cpp (None when not built, e.g. on the GPU box)."""
    if not os.path.exists(_REF_PATH):
        if not os.path.isdir("/root/reference/source/nmf"):
            return None
        build()
    r = C.CDLL(_REF_PATH)
    r.ref_summary_best_run.restype = C.c_uint
    return r


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _f(a: np.ndarray) -> np.ndarray:
    """Column-major, writable, contiguous-within-column view requirements."""
    assert a.ndim == 2 and a.flags.f_contiguous, "oracle matrices must be Fortran-ordered 2-D arrays"
    return a


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _ld(a: np.ndarray) -> int:
    return a.strides[1] // a.itemsize if a.shape[1] > 1 else max(a.shape[0], 1)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def set_threads_for(work: float) -> int:
    """One thread for small problems (fork/join would dominate), the CPU share otherwise."""
    n = 1 if work < 2e7 else cpu_share()
    lib().oracle_set_num_threads(n)
    return n


def seed_stream(seed: int, count: int) -> np.ndarray:
    out = np.zeros(count, dtype=np.uint32)
    lib().oracle_seed_stream(C.c_uint32(seed), C.c_int(count), _ptr(out))
    return out


def summary_best_run(frobenius) -> int:
    f = np.ascontiguousarray(frobenius, dtype=np.float64)
    return int(lib().oracle_summary_best_run(_ptr(f), C.c_int(len(f))))


def gemm_tn(A, B):
    """A^T B."""
    s = _sfx(A.dtype); _f(A); _f(B)
    out = np.zeros((A.shape[1], B.shape[1]), dtype=A.dtype, order="F")
    getattr(lib(), f"oracle_gemm_tn_{s}")(A.shape[0], A.shape[1], B.shape[1], _ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(out), _ld(out))
    return out


def gemm_nt(A, B):
    """A B^T."""
    s = _sfx(A.dtype); _f(A); _f(B)
    out = np.zeros((A.shape[0], B.shape[0]), dtype=A.dtype, order="F")
    getattr(lib(), f"oracle_gemm_nt_{s}")(A.shape[0], A.shape[1], B.shape[0], _ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(out), _ld(out))
    return out


def gemm_nn(A, B):
    s = _sfx(A.dtype); _f(A); _f(B)
    out = np.zeros((A.shape[0], B.shape[1]), dtype=A.dtype, order="F")
    getattr(lib(), f"oracle_gemm_nn_{s}")(A.shape[0], B.shape[1], A.shape[1], _ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(out), _ld(out))
    return out


def multiply_divide(X, Num, Den):
    """In place: X = X * Num / (Den + eps(T))."""
    s = _sfx(X.dtype)
    getattr(lib(), f"oracle_multiply_divide_{s}")(X.shape[0], X.shape[1], _ptr(_f(X)), _ld(X), _ptr(_f(Num)), _ld(Num), _ptr(_f(Den)), _ld(Den))
    return X


def normalize_columns(A):
    s = _sfx(A.dtype)
    getattr(lib(), f"oracle_normalize_columns_{s}")(A.shape[0], A.shape[1], _ptr(_f(A)), _ld(A))
    return A


def trace_multiplication(transpose_a: bool, A, B):
    s = _sfx(A.dtype)
    ps = np.zeros(B.shape[1], dtype=A.dtype)
    getattr(lib(), f"oracle_trace_multiplication_{s}")(int(transpose_a), B.shape[1], B.shape[0], _ptr(_f(A)), _ld(A), _ptr(_f(B)), _ld(B), _ptr(ps))
    return ps


def vtv_sorted(V):
    s = _sfx(V.dtype)
    ps = np.zeros(V.shape[1], dtype=V.dtype)
    getattr(lib(), f"oracle_vtv_sorted_{s}")(V.shape[0], V.shape[1], _ptr(_f(V)), _ld(V), _ptr(ps))
    return ps


def resolve_frobenius(vtv_sorted_, htwtv, hhtwtw) -> float:
    s = _sfx(vtv_sorted_.dtype)
    a = np.ascontiguousarray(vtv_sorted_); b = np.array(htwtv, dtype=a.dtype); c = np.array(hhtwtw, dtype=a.dtype)
    return float(getattr(lib(), f"oracle_resolve_frobenius_{s}")(_ptr(a), len(a), _ptr(b), len(b), _ptr(c), len(c)))


def direct_frobenius(V, W, H) -> float:
    s = _sfx(V.dtype)
    return float(getattr(lib(), f"oracle_direct_frobenius_{s}")(V.shape[0], V.shape[1], W.shape[1], _ptr(_f(V)), _ld(V), _ptr(_f(W)), _ld(W), _ptr(_f(H)), _ld(H)))


def densify(fmt: str, rows: int, cols: int, values, a, b, base: int = 0, dtype=None):
    """fmt 'csr': a=rowPtr b=colIdx; 'csc': a=colPtr b=rowIdx; 'coo': a=rowIdx b=colIdx."""
    values = np.ascontiguousarray(values if dtype is None else np.asarray(values, dtype=dtype))
    s = _sfx(values.dtype)
    a = np.ascontiguousarray(a, dtype=np.int32); b = np.ascontiguousarray(b, dtype=np.int32)
    out = np.zeros((rows, cols), dtype=values.dtype, order="F")
    if fmt == "coo":
        getattr(lib(), f"oracle_densify_coo_{s}")(rows, cols, _ptr(values), _ptr(a), _ptr(b), len(values), base, _ptr(out), _ld(out))
    else:
        getattr(lib(), f"oracle_densify_{fmt}_{s}")(rows, cols, _ptr(values), _ptr(a), _ptr(b), base, _ptr(out), _ld(out))
    return out


def qr_solve_left(A, X):
    """X <- A^-1 X by Householder QR (A is overwritten with its factorisation)."""
    s = _sfx(A.dtype)
    getattr(lib(), f"oracle_qr_solve_left_{s}")(A.shape[0], _ptr(_f(A)), _ld(A), X.shape[1], _ptr(_f(X)), _ld(X))
    return X


def qr_solve_right(A, X):
    """X <- X Q R^-T with A = Q R (for symmetric A this is X A^-1)."""
    s = _sfx(A.dtype)
    getattr(lib(), f"oracle_qr_solve_right_{s}")(A.shape[0], _ptr(_f(A)), _ld(A), X.shape[0], _ptr(_f(X)), _ld(X))
    return X


def run(algorithm: str, V, W, H, num_iterations: int, *, threshold_type: int = 0, threshold_value: float = 0.0,
        const_w: bool = False, lam: float = 0.0, lambda_w: float = 0.0, lambda_h: float = 0.0,
        alpha_w: float = 0.0, alpha_h: float = 0.0, theta: float = 0.0):
    """One CopyExisting run of the reference's iteration loop.  W and H are updated in place.

    Returns dict(iterations, frobenius, rmsd, history=[(frob, rmsd), ...])."""
    s = _sfx(V.dtype)
    assert W.dtype == V.dtype and H.dtype == V.dtype
    m, n = V.shape
    r = W.shape[1]
    assert W.shape == (m, r) and H.shape == (r, n)
    params = (C.c_double * 6)(lam, lambda_w, lambda_h, alpha_w, alpha_h, theta)
    set_threads_for(float(m) * n * r)
    frob = C.c_double(0.0); rmsd = C.c_double(0.0)
    cap = num_iterations // 10 + 2
    hist = np.zeros((cap, 2), dtype=np.float64)
    hl = C.c_int(0)
    it = getattr(lib(), f"oracle_run_{s}")(
        ALGORITHMS[algorithm], m, n, r, _ptr(_f(V)), _ld(V), _ptr(_f(W)), _ld(W), _ptr(_f(H)), _ld(H),
        num_iterations, threshold_type, C.c_double(threshold_value), int(const_w), params,
        C.byref(frob), C.byref(rmsd), _ptr(hist), cap, C.byref(hl))
    if it < 0:
        raise ValueError("unknown algorithm")
    return {"iterations": int(it), "frobenius": frob.value, "rmsd": rmsd.value,
            "history": [tuple(x) for x in hist[: hl.value]]}


def emulate_factor_product(A, F, splits: int):
    """Bit-exact fp32 model of the engine's MFMA factor product: OUT (r x X) = F (r x Y) A^T."""
    assert A.dtype == np.float32 and F.dtype == np.float32
    X, Y = A.shape
    r = F.shape[0]
    out = np.zeros((r, X), dtype=np.float32, order="F")
    set_threads_for(float(X) * Y * r)
    lib().oracle_emulate_factor_product_f32(X, Y, r, _ptr(_f(A)), _ld(A), _ptr(_f(F)), _ld(F), splits, _ptr(out), _ld(out))
    return out


def run_kl(V, W, H, num_iterations: int):
    """KL-divergence multiplicative update (extension, literature formula; see oracle_kl_run).
    W and H are updated in place.  Returns dict(frobenius, rmsd, kl)."""
    s = _sfx(V.dtype)
    m, n = V.shape
    r = W.shape[1]
    set_threads_for(float(m) * n * r)
    fn = getattr(lib(), f"oracle_kl_run_{s}")
    fn.restype = C.c_int
    frob = C.c_double(0); rmsd = C.c_double(0); kl = C.c_double(0)
    fn(m, n, r, _ptr(_f(V)), _ld(V), _ptr(_f(W)), _ld(W), _ptr(_f(H)), _ld(H), num_iterations, C.byref(frob), C.byref(rmsd), C.byref(kl))
    return {"frobenius": frob.value, "rmsd": rmsd.value, "kl": kl.value}


def run_kl_csr(rows: int, cols: int, values, row_ptr, col_idx, W, H, num_iterations: int):
    """The KL-divergence update over the stored entries of a 0-based CSR matrix (oracle_kl_run_csr): the form BASELINE
    config 3 needs.  W and H are updated in place.  Returns dict(frobenius, rmsd, kl)."""
    values = np.ascontiguousarray(values)
    s = _sfx(values.dtype)
    assert W.dtype == values.dtype and H.dtype == values.dtype
    r = W.shape[1]
    assert W.shape == (rows, r) and H.shape == (r, cols)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32); col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    assert len(row_ptr) == rows + 1 and row_ptr[0] == 0 and row_ptr[-1] == len(values) == len(col_idx)
    set_threads_for(float(len(values)) * r * 8)
    fn = getattr(lib(), f"oracle_kl_run_csr_{s}")
    fn.restype = C.c_int
    frob = C.c_double(0); rmsd = C.c_double(0); kl = C.c_double(0)
    it = fn(rows, cols, r, _ptr(row_ptr), _ptr(col_idx), _ptr(values), _ptr(_f(W)), _ld(W), _ptr(_f(H)), _ld(H), num_iterations,
            C.byref(frob), C.byref(rmsd), C.byref(kl))
    if it < 0:
        raise ValueError("rank above 1024")
    return {"frobenius": frob.value, "rmsd": rmsd.value, "kl": kl.value}


def kmeans(data, k: int, *, seed: int = 0, iterations: int = 100, threshold: float = 0.005):
    """Lloyd k-means with a Forgy start (source/kmeans/kMeans.cu:126-278).  Returns (clusters m x k, membership, passes)."""
    data = _f(data)
    m, n = data.shape
    clusters = np.zeros((m, k), dtype=data.dtype, order="F")
    membership = np.zeros(n, dtype=np.uint32)
    fn = getattr(kmeans_lib(), f"oracle_kmeans_{_sfx(data.dtype)}")
    it = fn(_ptr(data), C.c_long(_ld(data)), C.c_uint(m), C.c_uint(n), _ptr(clusters), C.c_long(m), C.c_uint(k),
            _ptr(membership), C.c_uint(seed), C.c_uint(iterations), C.c_double(threshold))
    return clusters, membership, int(it)


def einnmf_h(V, W):
    """EIn-NMF membership degrees of every column of V against the centres W (source/init/EInNMF.cu:44-119)."""
    V = _f(V); W = _f(W)
    m, n = V.shape
    r = W.shape[1]
    H = np.zeros((r, n), dtype=V.dtype, order="F")
    getattr(kmeans_lib(), f"oracle_einnmf_h_{_sfx(V.dtype)}")(_ptr(V), C.c_long(_ld(V)), _ptr(W), C.c_long(_ld(W)), C.c_uint(m),
                                                              C.c_uint(r), C.c_uint(n), _ptr(H), C.c_long(r))
    return H


# ---- NNDSVD (not in the reference: BASELINE north_star names it; Boutsidis & Gallopoulos, Pattern Recognition 41 (2008) 1350-1362) ----------------------------

def _uniform01(seed: int, index: int) -> float:
    """The counter-based (0, 1] generator of the engine (csrc/host_init.cpp uniform01 / kernels.hip k_fill_uniform), double form."""
    M64 = (1 << 64) - 1
    z = (seed * 0x9E3779B97F4A7C15 + index + 0x632BE59BD9B4E019) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    z = z ^ (z >> 31)
    return ((z >> 11) + 1) * (1.0 / 9007199254740992.0)


def nndsvd(V, r: int, variant: int = 0, seed: int = 0):
    """W0 (m x r), H0 (r x n) of the SVD-based start, computed with numpy's full SVD in double: the leading pair by absolute values, pair j > 0 by the
    larger of its positive and negative sections (section II of the paper); variant 1 (NNDSVDa) fills the zeros with mean(V), variant 2 (NNDSVDar) with
    mean(V) / 100 x U(0, 1] from the engine's counter generator (stream seed for W, seed + 0x9e3779b9 for H, index = the entry's position column-major
    in W (m x r) and in H (r x n))."""
    V64 = np.asarray(V, dtype=np.float64)
    m, n = V64.shape
    U, S, Vt = np.linalg.svd(V64, full_matrices=False)
    W = np.zeros((m, r)); H = np.zeros((r, n))
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0]); H[0, :] = np.sqrt(S[0]) * np.abs(Vt[0, :])
    for j in range(1, min(r, len(S))):
        x, y = U[:, j], Vt[j, :]
        xp, xn, yp, yn = np.maximum(x, 0), np.maximum(-x, 0), np.maximum(y, 0), np.maximum(-y, 0)
        nxp, nxn, nyp, nyn = (np.linalg.norm(a) for a in (xp, xn, yp, yn))
        mp, mn = nxp * nyp, nxn * nyn
        if mp > mn:
            u, v, sigma = xp / nxp, yp / nyp, mp
        else:
            if mn == 0.0:
                continue
            u, v, sigma = xn / nxn, yn / nyn, mn
        lbd = np.sqrt(S[j] * sigma)
        W[:, j] = lbd * u; H[j, :] = lbd * v
    tiny = 1e-6 * (1.0 if np.asarray(V).dtype == np.float32 else 1e-4)
    mean = V64.mean()
    if variant == 0:
        W[W < tiny] = 0.0; H[H < tiny] = 0.0
    elif variant == 1:
        W[W < tiny] = mean; H[H < tiny] = mean
    else:
        Wf, Hf = W.reshape(-1, order="F"), H.reshape(-1, order="F")          # (copies: column-major positions)
        for e in np.nonzero(Wf < tiny)[0]:
            Wf[e] = mean * 0.01 * _uniform01(seed, int(e))
        for e in np.nonzero(Hf < tiny)[0]:
            Hf[e] = mean * 0.01 * _uniform01(seed + 0x9e3779b9, int(e))
        W, H = Wf.reshape((m, r), order="F"), Hf.reshape((r, n), order="F")
    dt = np.asarray(V).dtype
    return np.asfortranarray(W.astype(dt)), np.asfortranarray(H.astype(dt))
