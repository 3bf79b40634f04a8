"""Strong-scaling readiness of configs[1]: iteration time of ONE column shard of the 10 000 x 5 000 problem at the widths an
N-GPU strong-scaling run gives a rank (n = 5000 / N), UNPROFILED, native sharded loop of a team of one:
  * "rank of N" = NMFAMD_SHARD_REHEARSE=1: exactly what a rank of an N-GPU team enqueues per iteration with the in-library direct exchange
    (exchange panel written, published, read back by pointer by the W update, r x r part summed by the small launch) -- what is missing is the
    link time of the peers' panels and the cross-device event wait, which the projection adds as a MODEL;
  * round 3's figures for the same shapes: profiles/r03_shard_sizes_projection.txt (121 / 79.9 / 82.5 / 71.6 us);
  * plus the fused single-GPU loop and the team-of-one loop (which now is the fused loop) at n = 5000.
Output: a table (us per iteration) and the projection of DESIGN section 6."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(nc, mode, **env):
    # (the better of two processes: the placement of the 207 MB image differs from process to process and moves an iteration by up to ~8 us)
    return min(run_once(nc, mode, **env), run_once(nc, mode, **env))


def run_once(nc, mode, **env):
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_trace.py"), str(nc), str(mode), "600"], env=e, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        if "us/iteration" in line:
            return float(line.split(":")[1].split("us/iteration")[0])
    raise RuntimeError(out.stdout + out.stderr)


M, N, R = 10000, 5000, 64
fused = run(N, "fused")
one = run(N, 1)
print(f"fused single-GPU loop, n = {N}: {fused:.1f} us/iteration; sharded loop of a team of one (replicated mode, in-process transport): {one:.1f}; "
      f"the same beside a one-rank RCCL clique: {run(N, 1, SHARD_TRACE_COMM='rccl'):.1f}; row-block mode: {run(N, 0):.1f}", flush=True)
rows = []
for world in (1, 2, 4, 8):
    nc = N // world
    a = run(nc, 1, NMFAMD_SHARD_REHEARSE=1)
    c = run(nc, 0)
    rows.append((world, nc, a, c))
    print(f"shard of a {world}-GPU run, n = {nc}: rank-of-N rehearsal (direct exchange) {a:.1f} us, row-block mode (team of one) {c:.1f} us", flush=True)
# modelled exchange over xGMI: 7 links x 153 GB/s per GPU, point to point.  Direct exchange: every rank reads the N - 1 peers' panels, each over
# its own link, inside the W update: S / link seconds whatever N, + one cross-device event wait (taken as 10 us: NOT measured, no multi-GPU box)
LINK, WAIT = 153e9, 10.0
S = 4.0 * (M * R + R * R)
print("\nprojection (measured shard time on one GPU + modelled exchange; NOT a measurement on N GPUs):")
print("| GPUs | columns per GPU | rank-of-N iteration us (direct exchange) | modelled link us + event wait us | projected it/s | projected speed-up |")
print("|---|---|---|---|---|---|")
for world, nc, a, c in rows:
    if world == 1:
        print(f"| 1 | {nc} | {one:.1f} (fused loop {fused:.1f}) | - | {1e6 / one:.0f} | 1.00 |")
        continue
    link = S / LINK * 1e6
    print(f"| {world} | {nc} | {a:.1f} | {link:.1f} + {WAIT:.0f} | {1e6 / (a + link + WAIT):.0f} | {one / (a + link + WAIT):.2f} |")
