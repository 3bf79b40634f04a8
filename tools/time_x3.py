"""Launch time of the split-operand fp32 product alone, alternating two images of the streamed matrix."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_x3 import x3

rs = np.random.RandomState(0)
for (X, Y, r) in ((10000, 5000, 64), (5000, 10000, 64)):
    A = np.asfortranarray(rs.random_sample((X, Y)).astype(np.float32))
    F = np.asfortranarray(rs.random_sample((r, Y)).astype(np.float32))
    reps = int(os.environ.get("REPS", "40"))
    _, us = x3(A, F, reps=reps)
    print(f"{X}x{Y} r={r}: {us:.1f} us  ({X * Y * 4 / us / 1e6:.2f} TB/s of A)", flush=True)
