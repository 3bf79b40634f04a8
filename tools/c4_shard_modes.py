"""One config-4 shard (50 000 x 6 250, r = 256, nsNMF theta = 0.5, bf16 operands) through the fused loop and the native sharded loop of a team of one in both
W-step modes (0 = row blocks, 1 = replicated): what the row-block form costs before a byte crosses a link.  usage: c4_shard_modes.py [ITERS]"""
# (the rehearsal switch this tool sets is read by the measurement build only: csrc/tuning.h)
import os as _os
_os.environ.setdefault("NMFAMD_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nmfgpu_amd", "lib", "libnmfgpu64_diag.so"))
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na

M, NC, R = 50000, 6250, 256
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(1)
V = np.empty((M, NC), dtype=np.float32, order="F")
for j0 in range(0, NC, 625):
    V[:, j0:j0 + 625] = rs.random_sample((625, M)).astype(np.float32).T
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((R, M))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((NC, R))).astype(np.float32).T)


def timed(fn, sync, warm=10):
    fn(warm, 1); sync()
    t0 = time.perf_counter(); fn(ITERS, warm + 1); sync()
    return (time.perf_counter() - t0) / ITERS * 1e6


e = na.Engine(M, NC, R, "nsnmf", theta=0.5, precision="bf16", row_blocks=1)
e.upload(V); del V
e.set_factors(W, H)
timed(lambda k, f: e.iterate(k, first_iteration=f, error_every=10), e.synchronize)      # (the first timed loop of a process runs ~13 us per iteration slow: discarded)
e.set_factors(W, H)
print(f"fused loop: {timed(lambda k, f: e.iterate(k, first_iteration=f, error_every=10), e.synchronize):.1f} us/iteration", flush=True)
group = na.LocalGroup(1)
comm = na.LocalComm(group, 0)
for mode in (1, 0):
    e.set_factors(W, H)
    run = na.ShardedRun(e, comm, M, NC, mode)
    us = timed(lambda k, f: run.iterate(k, first_iteration=f, error_every=10), e.synchronize)
    print(f"team of one, mode {mode} ({'replicated' if mode else 'row blocks'}): {us:.1f} us/iteration", flush=True)
    run.close()
e.close()
# what a rank of eight enqueues in the row-block form (an eighth of the rows updated, normalised, re-packed; no link time, no peer reads): timing only
os.environ["NMFAMD_SHARD_REHEARSE"] = "8"
e = na.Engine(M, NC, R, "nsnmf", theta=0.5, precision="bf16", row_blocks=8)
rs = np.random.RandomState(1)
V = np.empty((M, NC), dtype=np.float32, order="F")
for j0 in range(0, NC, 625):
    V[:, j0:j0 + 625] = rs.random_sample((625, M)).astype(np.float32).T
e.upload(V); del V
e.set_factors(W, H)
run = na.ShardedRun(e, comm, M, NC, 0)
us = timed(lambda k, f: run.iterate(k, first_iteration=f, error_every=0), e.synchronize)
print(f"rank-of-8 rehearsal, row blocks (1/8 of the rows in the W step): {us:.1f} us/iteration", flush=True)
run.close(); e.close(); comm.close()
