"""Build recipe of libnmfgpu64.so (the C-ABI shared library) for gfx950.

    python -m nmfgpu_amd.build            # incremental
    python -m nmfgpu_amd.build --force

hipcc cross-compiles without a GPU.  The library is built IN TREE (nmfgpu_amd/lib/) so that it
travels with the repository snapshot to the GPU box; it is git-ignored.  The file name is the
reference's (source/CMakeLists.txt:78-92) so existing loaders find it.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libnmfgpu64.so")
SOURCES = ["kernels.hip", "kernels_fast.hip", "kernels_mu64.hip", "kernels_sparse.hip", "kernels_bf16.hip", "kernels_wide.hip", "kernels_f64.hip", "kernels_x3.hip", "kernels_onepass.hip", "kernels_tri.hip", "comm.hip", "engine.cpp", "sharded.cpp", "amd_api.cpp", "abi.cpp", "host_init.cpp"]
ARCH = os.environ.get("NMFAMD_OFFLOAD_ARCH", "gfx950")
# translation units without device code or HIP runtime calls: plain C++ (function multiversioning
# is rejected by the device pass of a -x hip compile); no implicit contraction: where the reference's
# nvcc build fuses a multiply-add the source says std::fma)
HOST_ONLY = {"host_init.cpp"}
# per-source extra flags.  kernels_x3.hip: the SLP vectoriser pairs the scalar subtractions of the operand split
# into v2f32 values, which costs a v_mov per element to line the pairs up and re-serialises the chain
EXTRA_FLAGS = {"kernels_x3.hip": ["-fno-slp-vectorize"], "kernels_onepass.hip": ["-fno-slp-vectorize"]}
# experiment switches: NMFAMD_CXXFLAGS="-DNAME=1 ..." is appended to every compile (and forces nothing: use --force)
USER_FLAGS = os.environ.get("NMFAMD_CXXFLAGS", "").split()
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DNMFGPU_EXPORTING",
         "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function", "-Wno-unused-result"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the engine has no CPU fallback and cannot be built without ROCm")
    return exe


def _deps() -> float:
    newest = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            newest = max(newest, os.path.getmtime(os.path.join(root, f)))
    return newest


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """diag: the measurement build (-DNMFAMD_DIAG_BUILD, csrc/tuning.h) -> lib/libnmfgpu64_diag.so; select it with NMFAMD_LIBRARY."""
    global LIB, OBJDIR, USER_FLAGS
    if diag:
        LIB = os.path.join(LIBDIR, "libnmfgpu64_diag.so")
        OBJDIR = os.path.join(LIBDIR, "obj_diag")
        USER_FLAGS = [*USER_FLAGS, "-DNMFAMD_DIAG_BUILD"]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if src in HOST_ONLY:
            cmd = [cc, *[f for f in FLAGS if not f.startswith("--offload-arch")], "-x", "c++", "-pthread", "-ffp-contract=off", "-c", os.path.join(CSRC, src), "-o", obj]
        else:
            cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(src, []), *USER_FLAGS, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    # -z defs: an undefined symbol fails the link here, not at the first call inside a running process
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-z,defs", "-o", LIB + ".tmp", *objs]
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, diag="--diag" in sys.argv))
