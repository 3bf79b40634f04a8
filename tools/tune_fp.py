"""Diagnosis of the factor-product kernel on the GPU box (not part of the product or the tests).

    python tools/tune_fp.py            # timing of both BASELINE config-2 shapes + stamp breakdown
    NMFAMD_FP_DEPTH=8 python tools/tune_fp.py --no-stamps
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na  # noqa: E402

lib = na.library()


def run(X, Y, reps=50, stamps=True):
    avg = C.c_double(0)
    cap = 8 * 4096
    buf = np.zeros(cap, dtype=np.uint64)
    cnt = C.c_long(0)
    st = lib.nmfamd_tune_factor_product(X, Y, reps, C.byref(avg), C.c_void_p(buf.ctypes.data) if stamps else None, C.c_long(cap), C.byref(cnt))
    assert st == 0, st
    flops = 2.0 * X * Y * 64
    print(f"X={X} Y={Y} depth={os.environ.get('NMFAMD_FP_DEPTH', '6')}: {avg.value:.2f} us/launch  {flops / avg.value / 1e6:.1f} TFLOP/s (algorithmic)")
    if stamps and cnt.value:
        s = buf[: 8 * cnt.value].reshape(-1, 8).astype(np.int64)
        s = s[s[:, 6] > 0]
        pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
        life = s[:, 3] - s[:, 0]
        real_ns = (s[:, 5] - s[:, 4]) * 10.0
        clock = life / real_ns
        start = (s[:, 4] - s[:, 4].min()) * 10.0
        end = (s[:, 5] - s[:, 4].min()) * 10.0
        q = lambda a: f"min {np.min(a):9.0f} med {np.median(a):9.0f} max {np.max(a):9.0f}"
        print(f"   waves {len(s)}  steps {q(s[:, 6])}")
        print(f"   prologue cycles  {q(pro)}")
        print(f"   main loop cycles {q(loop)}   per step {np.median(loop / s[:, 6]):.1f} (ideal at two waves/SIMD: 1024)")
        print(f"   epilogue cycles  {q(epi)}")
        print(f"   wave lifetime    {q(life)}  = {np.median(real_ns) / 1e3:.2f} us, clock {np.median(clock):.3f} GHz")
        print(f"   wave start (ns after first) {q(start)};  last wave end {np.max(end) / 1e3:.2f} us")
        for x in range(8):
            sel = s[:, 7] == x
            if sel.any():
                print(f"      XCC {x}: waves {sel.sum():4d}  loop med {np.median(loop[sel]):9.0f}  clock {np.median(clock[sel]):.3f}")


if __name__ == "__main__":
    stamps = "--no-stamps" not in sys.argv
    run(10000, 5000, stamps=stamps)   # (V H^T)^T: x = rows of V
    run(5000, 10000, stamps=stamps)   # W^T V:     x = columns of V
