"""End-to-end wall time of the reference example's configuration (4096 x 165, r = 158, nsNMF, double, 2000 iterations)
through a g++-built C++ caller of nmfgpu::compute (tests/cpp/cxx_api_driver.cpp)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nmfgpu_amd as na
m, n, r, iters = 4096, 165, 158, 2000
rng = np.random.default_rng(1)
V = np.asfortranarray(rng.integers(0, 255, (m, n)) / 255.0); W = np.asfortranarray(rng.integers(1, 255, (m, r)) / 255.0); H = np.asfortranarray(rng.integers(1, 255, (r, n)) / 255.0)
with tempfile.TemporaryDirectory() as td:
    exe = os.path.join(td, "drv"); libdir = os.path.dirname(na.library_path())
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "cxx_api_driver.cpp"), "-o", exe,
                           "-L", libdir, "-lnmfgpu64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<5i", m, n, r, 5, iters)); f.write(V.tobytes(order="F")); f.write(W.tobytes(order="F")); f.write(H.tobytes(order="F"))
    for _ in range(2):
        print(subprocess.run([exe, fin, fout], capture_output=True, text=True).stdout.strip())
