"""Per-workgroup timeline of config 4's bf16 product launches (k_factor_product_bf16_r2): which workgroups end a launch, by K slice, tile, XCD, passengers.
Measurement build only:  NMFAMD_LIBRARY=.../libnmfgpu64_diag.so NMFAMD_BF_STAMPS=/tmp/bf.bin python bench.py --workload c4 --steps 60 --warmup 10 --no-cpu-baseline
then  python tools/stamp_bf16.py /tmp/bf.bin [tiles_h splits_h tiles_w splits_w]   (defaults: config 4's shard, 28 x 8 and 224 x 1)"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2, 512, 4, 4).astype(np.int64)
plans = [(28, 8), (224, 1)]
if len(sys.argv) >= 6:
    plans = [(int(sys.argv[2]), int(sys.argv[3])), (int(sys.argv[4]), int(sys.argv[5]))]
for kind, name in enumerate(("W^T V", "V H^T")):
    s = a[kind]
    live = s[:, 0, 0] > 0
    idx = np.nonzero(live)[0]
    if len(idx) == 0:
        print(name, "no stamps"); continue
    t0 = s[live][:, :, 0].min()
    tiles, splits = plans[kind]
    nblk = tiles * splits
    ent = (s[:, :, 0].min(axis=1) - t0) / 100.0
    ls = (np.where(s[:, :, 1] > 0, s[:, :, 1], t0).max(axis=1) - t0) / 100.0
    le = (np.where(s[:, :, 2] > 0, s[:, :, 2], t0).max(axis=1) - t0) / 100.0
    ex = (s[:, :, 3].max(axis=1) - t0) / 100.0
    prod = [b for b in idx if b < nblk]
    pas = [b for b in idx if b >= nblk]
    print(f"{name}: {len(prod)} product workgroups ({tiles} tiles x {splits} K slices), {len(pas)} passengers")
    print(f"  entry max {ent[prod].max():.1f} us; loop start median {np.median(ls[prod]):.1f}; loop end median {np.median(le[prod]):.1f} p90 {np.percentile(le[prod], 90):.1f} max {le[prod].max():.1f}; "
          f"exit median {np.median(ex[prod]):.1f} p90 {np.percentile(ex[prod], 90):.1f} max {ex[prod].max():.1f}; epilogue median {np.median(ex[prod] - le[prod]):.1f}")
    if pas:
        print(f"  passengers: exit min {ex[pas].min():.1f} median {np.median(ex[pas]):.1f} max {ex[pas].max():.1f}")
    q8, r8 = divmod(nblk, 8)
    rows = []
    for b in prod:
        xcd, i = b % 8, b // 8
        vb = (xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8) + i
        rows.append((b, xcd, vb % tiles, vb // tiles, le[b], ex[b]))
    rows = np.array(rows)
    print("  median loop end by XCD:", [round(float(np.median(rows[rows[:, 1] == k, 4])), 1) for k in range(8)])
    print("  median loop end by K slice:", [round(float(np.median(rows[rows[:, 3] == k, 4])), 1) for k in range(splits)])
    late = rows[np.argsort(rows[:, 5])[-8:]]
    print("  last eight (block, XCD, tile, slice, loop end, exit):", [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), round(float(r[4]), 1), round(float(r[5]), 1)) for r in late])
    bt = np.array([np.median(rows[rows[:, 2] == k, 5]) for k in range(tiles)])
    print("  tiles with the latest median exit:", [(int(k), round(float(bt[k]), 1)) for k in np.argsort(bt)[-5:]], " earliest:", [(int(k), round(float(bt[k]), 1)) for k in np.argsort(bt)[:3]])
# several files: do the same workgroups end the launch every time?
if len(sys.argv) > 2 and sys.argv[2].endswith(".bin"):
    import collections
    for kind, name in enumerate(("W^T V", "V H^T")):
        tiles, splits = plans[kind]
        nblk = tiles * splits
        late = collections.Counter()
        exits = []
        for f in sys.argv[1:]:
            s = np.fromfile(f, dtype=np.uint64).reshape(2, 512, 4, 4).astype(np.int64)[kind]
            t0 = s[:nblk, :, 0].min()
            ex = (s[:nblk, :, 3].max(axis=1) - t0) / 100.0
            exits.append(ex)
            for b in np.argsort(ex)[-22:]:
                late[int(b)] += 1
        exits = np.array(exits)
        print(f"{name}: workgroups among the last 22 of a launch in k of {len(sys.argv) - 1} runs:", sorted(collections.Counter(late.values()).items()))
        print("   always late:", sorted(b for b, c in late.items() if c == len(sys.argv) - 1))
        print("   correlation of exit times between runs:", np.round(np.corrcoef(exits)[0, 1:], 2))
