"""Several engines on ONE device, each on its own stream and driven by its own host thread, all given the same problem (config 4's
shard shape: 50 000 x 1 024, r = 256, nsNMF, bf16 operands): after every iteration the factors and the rank-256 intermediates of all
engines are compared bit for bit -- identical inputs and deterministic kernels must give identical bits whatever else runs on the device.
usage: python tools/shared_device_diff.py   (ENGINES=8 ITERS=6 STEP=iterate|h_step SHARE_STREAM=0|1; needs a GPU; test infrastructure only)"""
import os, sys, threading
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmfgpu_amd.engine import Engine

K = int(os.environ.get("ENGINES", "8")); iters = int(os.environ.get("ITERS", "6")); step = os.environ.get("STEP", "iterate")
m, n, r = int(os.environ.get("M", "50000")), int(os.environ.get("N", "1024")), 256
rng = np.random.default_rng(4)
V = np.asfortranarray(rng.random((m, n), dtype=np.float32))
W0 = np.asfortranarray((1.0 - rng.random((m, r))).astype(np.float32)); H0 = np.asfortranarray((1.0 - rng.random((r, n))).astype(np.float32))
torch.cuda.set_device(0)
shared = torch.cuda.Stream() if os.environ.get("SHARE_STREAM") == "1" else None
streams = [shared or torch.cuda.Stream() for _ in range(K)]
engines = []
for k in range(K):
    e = Engine(m, n, r, algorithm="nsnmf", theta=0.5, precision="bf16", stream=streams[k].cuda_stream)
    e.upload(V); e.set_factors(W0, H0); e.synchronize()
    engines.append(e)
names = {1: ("H", 256 * n), 4: ("slabs", 256 * n), 8: ("WtW", 65536), 9: ("colsq", 256), 10: ("Wfrag", 128 * (m // 16) * 16), 11: ("Hfrag", 128 * (n // 16) * 16), 0: ("W", 256 * m)}
# AGGRESSORS=a: a more engines that are not compared, running whole iterations beside the compared ones (AGGRESSOR=c4: the same problem; mu64: a 10 000 x 5 000, r = 64
# fp32 problem -- other kernels altogether), AGG_ITERS of them per round
A = int(os.environ.get("AGGRESSORS", "0")); agg_kind = os.environ.get("AGGRESSOR", "c4"); agg_iters = int(os.environ.get("AGG_ITERS", "1"))
aggressors = []
for k in range(A):
    st = torch.cuda.Stream(); streams.append(st)
    if agg_kind == "c4":
        e = Engine(m, n, r, algorithm="nsnmf", theta=0.5, precision="bf16", stream=st.cuda_stream); e.upload(V); e.set_factors(W0, H0)
    else:
        e = Engine(10000, 5000, 64, algorithm="mu", stream=st.cuda_stream); e.upload(np.asfortranarray(rng.random((10000, 5000), dtype=np.float32))); e.randomize(7)
    e.synchronize(); aggressors.append(e)
def work(k, barrier):
    torch.cuda.set_device(0)
    barrier.wait()
    if k >= K:
        aggressors[k - K].iterate(agg_iters, first_iteration=2, error_every=1000); aggressors[k - K].synchronize()
        return
    if step == "h_step":
        engines[k].h_step(False)
    else:
        engines[k].iterate(1, first_iteration=it + 1, error_every=1000)
    engines[k].synchronize()
bad_total = 0
prev_h = None
for it in range(iters):
    barrier = threading.Barrier(K + A)
    ts = [threading.Thread(target=work, args=(k, barrier)) for k in range(K + A)]
    for t in ts: t.start()
    for t in ts: t.join()
    report = []
    for which, (name, count) in names.items():      # (W last: reading it folds the pending column scale into the panel)
        if step == "h_step" and which == 0:
            continue
        vals = [e.debug_read(which, count).view(np.uint32) for e in engines]
        groups = {}
        if which == 1: next_prev = vals[0]
        for k, v in enumerate(vals):
            groups.setdefault(v.tobytes(), []).append(k)
        if len(groups) > 1:
            ref = max(groups.values(), key=len)
            if which == 1: next_prev = vals[ref[0]]
            for g in groups.values():
                if g is ref: continue
                d = np.flatnonzero(vals[g[0]] != vals[ref[0]])
                where = ""
                if which in (1, 4, 0):      # panels: [row][256]
                    rows, cols = np.unique(d // 256), np.unique(d % 256)
                    where = f" ({rows.size} rows {rows[:6].tolist()}.., {cols.size} columns {cols[:6].tolist()}..)"
                if which == 1 and os.environ.get("VALUES"):
                    a, b = vals[g[0]].view(np.float32), vals[ref[0]].view(np.float32)
                    p0 = prev_h.view(np.float32) if prev_h is not None else b
                    where += " values (row, col, got, expected, previous): " + ", ".join(f"({i // 256}, {i % 256}, {a[i]:.6g}, {b[i]:.6g}, {p0[i]:.6g})" for i in d[:int(os.environ["VALUES"])])
                report.append(f"{name}: engines {g} differ from {ref} in {d.size} words, first at {d[0]}, last at {d[-1]}{where}")
    prev_h = next_prev
    bad_total += len(report)
    print(f"iteration {it + 1}: " + ("all engines identical" if not report else "; ".join(report)), flush=True)
print("DIFFERENCES" if bad_total else "IDENTICAL", flush=True)
