"""The C++ face of the drop-in boundary on a real device: a small C++ program (tests/cpp/cxx_api_driver.cpp, written
against include/nmfgpu.h only) is compiled with g++, linked against libnmfgpu64.so and run -- mangled nmfgpu::compute
for NmfDescription<double>, ISummary::create and its vtable, initialize / finalize -- and its factors are compared
with the oracle.  double + nsNMF is what the reference's own example program runs (ref example/main.cpp:78-131)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import nmfgpu_amd as na  # noqa: E402
from oracle import oracle  # noqa: E402

pytestmark = pytest.mark.gpu

ALG = {"mu": 0, "gdcls": 1, "als": 2, "acls": 3, "ahcls": 4, "nsnmf": 5}
KW = {"mu": {}, "nsnmf": dict(theta=0.5), "gdcls": dict(lam=0.01), "ahcls": dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01)}


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cxx") / "cxx_api_driver")
    libdir = os.path.dirname(na.library_path())
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "cxx_api_driver.cpp"),
                           "-o", exe, "-L", libdir, "-lnmfgpu64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.parametrize("alg,m,n,r,iters", [("nsnmf", 500, 300, 8, 40), ("mu", 700, 450, 64, 30), ("ahcls", 300, 200, 5, 20), ("gdcls", 260, 190, 12, 20)])
def test_cxx_caller_double(driver, tmp_path, alg, m, n, r, iters):
    rng = np.random.default_rng(m + r)
    V = np.asfortranarray(rng.random((m, n))); W = np.asfortranarray(1.0 - rng.random((m, r))); H = np.asfortranarray(1.0 - rng.random((r, n)))
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<5i", m, n, r, ALG[alg], iters))
        f.write(V.tobytes(order="F")); f.write(W.tobytes(order="F")); f.write(H.tobytes(order="F"))
    out = subprocess.run([driver, fin, fout], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    raw = np.fromfile(fout, dtype=np.float64)
    frob, rmsd, its, seconds = raw[:4]
    Wg = raw[4:4 + m * r].reshape((m, r), order="F"); Hg = raw[4 + m * r:].reshape((r, n), order="F")
    ref = oracle.run(alg, V, W, H, iters, **KW[alg])
    tol = 1e-6 if alg in ("ahcls", "gdcls") else 1e-9          # LS solves amplify rounding by cond(W^T W)
    assert np.linalg.norm(Wg - W) / np.linalg.norm(W) < tol
    assert np.linalg.norm(Hg - H) / np.linalg.norm(H) < tol
    assert int(its) == iters
    assert frob == pytest.approx(ref["frobenius"], rel=max(tol, 1e-9))
    assert rmsd == pytest.approx(ref["rmsd"], rel=max(tol, 1e-9))
