"""Strong-scaling readiness of configs[1]: iteration time of ONE column shard of the 10 000 x 5 000 problem at the widths an
N-GPU strong-scaling run gives a rank (n = 5000 / N), native sharded loop with its collectives issued for real on a one-rank RCCL
clique (identities: what remains is the fixed cost of the calls), both shard modes; plus the fused single-GPU loop at n = 5000.
Output: a table (us per iteration) and the projection of DESIGN section 6 (modelled xGMI time added)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmfgpu_amd as na

M, N, R = 10000, 5000, 64
rs = np.random.RandomState(1)
V = np.empty((M, N), dtype=np.float32, order="F")
for j0 in range(0, N, 512):
    V[:, j0:j0 + 512] = rs.random_sample((M, min(512, N - j0))).astype(np.float32)
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((R, M))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((N, R))).astype(np.float32).T)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, sync, iters=300, warm=60):
    fn(warm, 1); sync()
    t0 = time.perf_counter(); fn(iters, warm + 1); sync()
    return (time.perf_counter() - t0) / iters * 1e6


eng = na.Engine(M, N, R, "mu", stream=stream)
eng.upload(V); eng.set_factors(W, H)
fused = timed(lambda k, f: eng.iterate(k, first_iteration=f, error_every=10), eng.synchronize)
eng.close()
print(f"fused single-GPU loop, n = {N}: {fused:.1f} us/iteration", flush=True)
rows = []
for world in (1, 2, 4, 8):
    nc = N // world
    res = {}
    for mode in (na.SHARD_REPLICATED, na.SHARD_ROW_BLOCKS):
        comm = na.RcclComm(na.RcclComm.unique_id(), 1, 0)
        e = na.Engine(M, nc, R, "mu", stream=stream, row_blocks=1)
        e.upload(np.asfortranarray(V[:, :nc])); e.set_factors(W, np.asfortranarray(H[:, :nc]))
        run = na.ShardedRun(e, comm, M, nc, mode)
        res[mode] = timed(lambda k, f: run.iterate(k, first_iteration=f, error_every=10), e.synchronize)
        run.close(); e.close(); comm.close()
    rows.append((world, nc, res[na.SHARD_REPLICATED], res[na.SHARD_ROW_BLOCKS]))
    print(f"shard of a {world}-GPU run, n = {nc}: replicated-update mode {res[na.SHARD_REPLICATED]:.1f} us, row-block mode {res[na.SHARD_ROW_BLOCKS]:.1f} us", flush=True)
# modelled collective time over xGMI: 7 links x 153 GB/s per GPU, point to point
LINK = 153e9
print("\nprojection (measured shard time on one GPU + modelled exchange; NOT a measurement on N GPUs):")
print("| GPUs | columns per GPU | shard iteration us (replicated / row-block) | modelled exchange us (all-reduce ring / direct RS+AG) | projected it/s (best) |")
print("|---|---|---|---|---|")
S = 4.0 * (M * R + R * R)
for world, nc, a, b in rows:
    if world == 1:
        print(f"| 1 | {nc} | {a:.1f} / {b:.1f} (fused loop {fused:.1f}) | - | {1e6 / fused:.0f} |")
        continue
    ring = 2.0 * (world - 1) / world * S / LINK * 1e6                 # ring all-reduce: one link's worth of bandwidth
    direct = 2.0 * (S / world) / LINK * 1e6 * 1.0                     # direct reduce-scatter + all-gather: every peer link at once
    best = min(a + ring, b + direct)
    print(f"| {world} | {nc} | {a:.1f} / {b:.1f} | {ring:.1f} / {direct:.1f} | {1e6 / best:.0f} |")
