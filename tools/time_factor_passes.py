"""Time the factor-side passes of config 4 (kernels_tri.hip) on a 50 000 x 256 and a 6 250 x 256 panel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na  # noqa: E402

for length in (50000, 6250):
    rng = np.random.default_rng(1)
    P = rng.random((length, 256), dtype=np.float32)
    sq = (P.astype(np.float64) ** 2).sum(axis=0).astype(np.float32)
    out = na.op_factor_passes(P, theta=0.5, colsq=sq, reps=20)
    mb = P.nbytes / 1e6
    print(f"len {length}: finish (normalise + smooth + pack, {2.5 * mb:.0f} MB) {out['us_finish']:.1f} us; gram + reduce + smooth ({mb:.0f} MB read) {out['us_gram']:.1f} us", flush=True)
