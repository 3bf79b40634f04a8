"""One column shard of configs[1] (10 000 x NC, r = 64, MU) through the native sharded loop on a one-rank RCCL clique (or with the
in-library exchange, NMFAMD_COMM=p2p), for rocprofv3 --kernel-trace --stats: itemises the shard iteration's fixed cost.
usage: shard_trace.py NC MODE(0 row blocks | 1 replicated | fused) [ITERS]"""
# (the switches this tool sets are read by the measurement build only: csrc/tuning.h)
import os as _os
_os.environ.setdefault("NMFAMD_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nmfgpu_amd", "lib", "libnmfgpu64_diag.so"))
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na

M, R = 10000, 64
NC = int(sys.argv[1]); MODE = sys.argv[2]; ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rs = np.random.RandomState(1)
V = np.asfortranarray(rs.random_sample((M, NC)).astype(np.float32))
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((R, M))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((NC, R))).astype(np.float32).T)


def timed(fn, sync, iters=ITERS, warm=40):
    fn(warm, 1); sync()
    t0 = time.perf_counter(); fn(iters, warm + 1); sync()
    return (time.perf_counter() - t0) / iters * 1e6


if MODE == "fused":
    e = na.Engine(M, NC, R, "mu")
    e.upload(V); e.set_factors(W, H)
    us = timed(lambda k, f: e.iterate(k, first_iteration=f, error_every=10), e.synchronize)
    geo = e.geometry()
    e.close()
else:
    # SHARD_TRACE_COMM=rccl: a one-rank RCCL clique (its proxy thread and streams exist beside the loop); default: the in-process transport a team uses
    if os.environ.get("SHARD_TRACE_COMM", "local") == "rccl":
        comm = na.RcclComm(na.RcclComm.unique_id(), 1, 0)
    else:
        group = na.LocalGroup(1)
        comm = na.LocalComm(group, 0)
    e = na.Engine(M, NC, R, "mu", row_blocks=1)
    e.upload(V); e.set_factors(W, H)
    run = na.ShardedRun(e, comm, M, NC, int(MODE))
    us = timed(lambda k, f: run.iterate(k, first_iteration=f, error_every=10), e.synchronize)
    geo = e.geometry()
    run.close(); e.close(); comm.close()
print(f"shard n = {NC} mode {MODE}: {us:.1f} us/iteration  geometry {geo}", flush=True)
