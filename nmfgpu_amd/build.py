"""Build recipe of libnmfgpu64.so (the C-ABI shared library) for gfx950.

    python -m nmfgpu_amd.build            # incremental
    python -m nmfgpu_amd.build --force

hipcc cross-compiles without a GPU.  The library is built IN TREE (nmfgpu_amd/lib/) so that it
travels with the repository snapshot to the GPU box; it is git-ignored.  The file name is the
reference's (source/CMakeLists.txt:78-92) so existing loaders find it.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libnmfgpu64.so")
# the measurement build (-DNMFAMD_DIAG_BUILD, csrc/tuning.h): the same library with its A/B switches and stamped kernel variants compiled in; the tests that compare
# kernel FORMS with each other select them there (tests/conftest.py, fixture diag_build); never loaded by the product path
DIAG_LIB = os.path.join(LIBDIR, "libnmfgpu64_diag.so")
SOURCES = ["kernels.hip", "kernels_fast.hip", "kernels_mu64.hip", "kernels_sparse.hip", "kernels_sparse_setup.hip", "kernels_bf16.hip", "kernels_wide.hip", "kernels_f64.hip", "kernels_x3.hip", "kernels_onepass.hip", "kernels_tri.hip", "comm.hip", "engine.cpp", "sharded.cpp", "amd_api.cpp", "abi.cpp", "host_init.cpp"]
ARCH = os.environ.get("NMFAMD_OFFLOAD_ARCH", "gfx950")
# translation units without device code or HIP runtime calls: plain C++ (function multiversioning
# is rejected by the device pass of a -x hip compile); no implicit contraction: where the reference's
# nvcc build fuses a multiply-add the source says std::fma)
HOST_ONLY = {"host_init.cpp"}
# sources of the MEASUREMENT build only: the one-pass iteration (round 3: built, correct, 1.8 x slower than the two-pass iteration -- a recorded dead end that the
# shipped library no longer carries; NMFAMD_ONE_PASS=1 selects it in libnmfgpu64_diag.so, tests/test_gpu_onepass.py runs it there)
DIAG_ONLY = {"kernels_onepass.hip"}
# per-source extra flags.  kernels_x3.hip: the SLP vectoriser pairs the scalar subtractions of the operand split
# into v2f32 values, which costs a v_mov per element to line the pairs up and re-serialises the chain
# Since round 3 the SLP vectoriser is off for EVERY device source (DEVICE_FLAGS): it also pairs `uniform * x + uniform` into
# v_pk_fma_f32 with a scalar-register source, which gave wrong low halves in lanes 48..63 whenever waves of another kernel shared
# the SIMD (docs/DESIGN_r05.md section 11; csrc/split3.h in_vgpr; check_packed_scalar_sources below).
EXTRA_FLAGS = {}
DEVICE_FLAGS = ["-fno-slp-vectorize"]
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


_PACKED_SCALAR = re.compile(r"\bv_pk_\w+_[fb]32\b.*(?<![\w.])s(\d+|\[\d+:\d+\])")


def packed_scalar_sources(obj: str) -> list:
    """Disassembles the gfx950 code object inside a compiled .hip object and returns "kernel: instruction" for every packed 32-bit
    instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_pk_mov_b32) that reads a scalar register -- the form that misbehaved
    when other kernels' waves shared the SIMD (docs/DESIGN_r05.md section 11).  Empty list = clean."""
    llvm = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                               f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--output={co}"])
        text = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], check=True, stdout=subprocess.PIPE, text=True).stdout
    found, kernel = [], "?"
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel = m.group(1)
            continue
        ins = line.split("//")[0].strip()
        if _PACKED_SCALAR.search(ins):
            found.append(f"{kernel}: {ins}")
    return found


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """diag: the measurement build (-DNMFAMD_DIAG_BUILD, csrc/tuning.h) -> lib/libnmfgpu64_diag.so; select it with NMFAMD_LIBRARY."""
    lib, objdir, user_flags = LIB, OBJDIR, USER_FLAGS
    if diag:
        lib = DIAG_LIB
        objdir = os.path.join(LIBDIR, "obj_diag")
        user_flags = [*USER_FLAGS, "-DNMFAMD_DIAG_BUILD"]
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _deps():
        return lib
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    objs = []
    procs = []
    sources = [src for src in SOURCES if diag or src not in DIAG_ONLY]
    for src in sources:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if src in HOST_ONLY:
            cmd = [cc, *[f for f in FLAGS if not f.startswith("--offload-arch")], "-x", "c++", "-pthread", "-ffp-contract=off", "-c", os.path.join(CSRC, src), "-o", obj]
        else:
            cmd = [cc, *FLAGS, *DEVICE_FLAGS, *EXTRA_FLAGS.get(src, []), *user_flags, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
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
    bad = []
    try:
        for src, obj in zip(sources, objs):
            if src not in HOST_ONLY and src.endswith(".hip"):
                bad += [f"{src}: {line}" for line in packed_scalar_sources(obj)]
    except FileNotFoundError as exc:      # (no llvm-objcopy / clang-offload-bundler / llvm-objdump beside hipcc: the flag above still holds, tests/test_abi.py re-checks)
        sys.stderr.write(f"nmfgpu_amd.build: packed-instruction check skipped ({exc})\n")
    if bad:
        raise RuntimeError("packed fp32 instructions with a scalar-register source (csrc/split3.h, in_vgpr):\n" + "\n".join(bad[:40]))
    # -z defs: an undefined symbol fails the link here, not at the first call inside a running process
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-z,defs", "-o", lib + ".tmp", *objs]
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, diag="--diag" in sys.argv))
