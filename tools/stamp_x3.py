"""Main-loop cycles per K-step of the split product in its two production forms on ONE 16-row-tile image (diagnostic build:
NMFAMD_LIBRARY=.../libnmfgpu64_diag.so, NMFAMD_X3_VARIANT = 10..13 x-tiled V H^T / 30..33 y-tiled W^T V; last digit 0 = no ring refill,
1 = A only, 2 = F only, 3 = the production loop).  Arguments: rows columns of V (default 10000 5000)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmfgpu_amd._lib import library

lib = library()
X, Y = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 5000)
cap = 8 * 4 * 400 * 8
buf = np.zeros(cap, dtype=np.uint64)
waves = C.c_long(0)
st = lib.nmfamd_tune_factor_product_x3(X, Y, C.c_void_p(buf.ctypes.data), C.c_long(cap), C.byref(waves))
assert st == 0, st
s = buf[: 8 * waves.value].reshape(-1, 8).astype(np.float64)
s = s[s[:, 2] > 0]
cyc, ticks, steps = s[:, 0], s[:, 1], s[:, 2]
print(f"variant {os.environ.get('NMFAMD_X3_VARIANT')}: waves {len(s)}, steps/wave {steps.mean():.1f}, cycles/K-step median {np.median(cyc / steps):.0f} "
      f"(p10 {np.percentile(cyc / steps, 10):.0f}, p90 {np.percentile(cyc / steps, 90):.0f}) = {np.median(cyc / steps) / 48:.1f} per MFMA; "
      f"loop time median {np.median(ticks) / 100:.1f} us, clock {np.median(cyc / ticks) * 100:.0f} MHz")
t0 = s[:, 3].min()
for name, col in (("entry", 3), ("loop start", 4), ("loop end", 5), ("tail end", 6), ("exit", 7)):
    v = (s[:, col] - t0) / 100
    print(f"  {name:10s}: median {np.median(v):6.2f} us  min {v.min():6.2f}  max {v.max():6.2f}")
# which workgroups end the launch: exit time by x-tile, by K slice and by XCD (the kernel's placement: block b runs on XCD b % 8; each XCD takes a contiguous
# range of (slice, x-tile) pairs -- kernels_x3.hip, XCD_REMAP)
full = buf[: 8 * waves.value].reshape(-1, 8).astype(np.float64)
nblk = len(full) // 4
variant = int(os.environ.get("NMFAMD_X3_VARIANT", "13"))
y_tiled = variant >= 30
xtiles = ((Y if y_tiled else X) + 127) // 128
live = full[:, 2].reshape(nblk, 4).max(axis=1) > 0
pblocks = int(live.sum())
splits = pblocks // xtiles
ex = (full[:, 7].reshape(nblk, 4).max(axis=1) - t0) / 100
q8, r8 = divmod(pblocks, 8)
rows = []
for b in range(pblocks):
    xcd, idx = b % 8, b // 8
    vb = (xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8) + idx
    rows.append((b, xcd, vb % xtiles, vb // xtiles, ex[b]))
rows = np.array(rows)
print(f"  workgroups {pblocks} = {xtiles} x-tiles x {splits} K slices; exit: median {np.median(rows[:, 4]):.2f} us, p90 {np.percentile(rows[:, 4], 90):.2f}, max {rows[:, 4].max():.2f}")
late = rows[np.argsort(rows[:, 4])[-8:]]
print("  last eight (block, XCD, x-tile, slice, exit us):", [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), round(float(r[4]), 1)) for r in late])
print("  median exit by XCD:", [round(float(np.median(rows[rows[:, 1] == k, 4])), 1) for k in range(8)])
print("  median exit by slice:", [round(float(np.median(rows[rows[:, 3] == k, 4])), 1) for k in range(splits)])
bx = np.array([np.median(rows[rows[:, 2] == k, 4]) for k in range(xtiles)])
print("  x-tiles with the latest median exit:", [(int(k), round(float(bx[k]), 1)) for k in np.argsort(bx)[-5:]])
# the epilogue as the launch sees it: from the moment a workgroup's LAST wave leaves the loop to the workgroup's exit
tl = (full[:, 6].reshape(nblk, 4).max(axis=1) - t0) / 100
fl = (full[:, 6].reshape(nblk, 4).min(axis=1) - t0) / 100
ep = (ex - tl)[:pblocks]
print(f"  epilogue (last wave out of the loop -> exit): median {np.median(ep):.2f} us, p90 {np.percentile(ep, 90):.2f}; first to last wave out of the loop: median {np.median((tl - fl)[:pblocks]):.2f} us")
