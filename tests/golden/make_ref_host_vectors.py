"""Generates tests/golden/ref_host_vectors.json from the REFERENCE's own code.

Runs the two reference translation units that build without CUDA
(/root/reference/source/nmf/Algorithm.cpp and Summary.cpp, compiled by oracle/Makefile into
oracle/_ref/libnmfgpu_refhost.so) and records

  * the per-run seed stream (IAlgorithm::generateRandomNumber) for a set of seeds, and
  * Summary::insert / bestRun() / recordCount() on a set of frobenius sequences.

The output is data only (inputs and the reference's outputs).  Re-run in the build container:
    python tests/golden/make_ref_host_vectors.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

ref = oracle.ref_lib()
if ref is None:
    raise SystemExit("reference tree not present: cannot regenerate golden vectors")

out = {"generator": "tests/golden/make_ref_host_vectors.py",
       "source": ["source/nmf/Algorithm.cpp:26-31", "source/nmf/Summary.cpp:27-60"],
       "seed_stream": {}, "summary": []}

for seed in (0, 1, 2, 3, 42, 12345, 2**31, 2**32 - 1):
    buf = np.zeros(16, dtype=np.uint32)
    ref.ref_seed_stream(C.c_uint32(seed), 16, buf.ctypes.data_as(C.c_void_p))
    out["seed_stream"][str(seed)] = [int(x) for x in buf]

rng = np.random.default_rng(7)
sequences = [
    [5.0], [5.0, 4.0, 3.0], [3.0, 4.0, 5.0], [4.0, 4.0, 4.0], [5.0, 3.0, 3.0, 2.0, 2.0],
    [2.0, 1.0, 1.5, 0.5, 0.5, 0.25], [1.0, float(np.nextafter(1.0, 0.0)), 1.0],
]
sequences += [list(map(float, rng.integers(1, 6, size=int(k)).astype(float))) for k in (4, 7, 9)]
for seq in sequences:
    f = np.asarray(seq, dtype=np.float64)
    cnt = C.c_uint(0)
    rb = np.zeros(len(f))
    best = ref.ref_summary_best_run(f.ctypes.data_as(C.c_void_p), len(f), C.byref(cnt), rb.ctypes.data_as(C.c_void_p))
    assert np.array_equal(rb, f)
    out["summary"].append({"frobenius": [float(x) for x in f], "bestRun": int(best), "recordCount": int(cnt.value)})

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_host_vectors.json")
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote", path)
