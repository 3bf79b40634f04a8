#!/bin/bash
# usage: tools/ride64_ab.sh [workload]  -- what the Gram passengers of the fp64 product launch cost (measurement build): per-launch times of the two product launches
# with the passengers cut short at stage 1 (return at once), 2 (partial block stored), 3 (counted in), 0 (complete).  Results of stages 1 - 3 are void: timing only.
w=${1:-example}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
export NMFAMD_LIBRARY=$root/nmfgpu_amd/lib/libnmfgpu64_diag.so
for stop in 1 2 3 0; do
  export NMFAMD_RIDE64_STOP=$stop
  echo "== stop=$stop"
  bash tools/profile_bench.sh ride64_$stop --workload $w --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ride64_$stop.txt 2>&1
  python3 tools/trace_by_grid.py gpurun_out/prof_ride64_$stop/*/*kernel_trace.csv 4
done
