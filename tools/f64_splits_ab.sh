#!/bin/bash
# usage: tools/f64_splits_ab.sh [workload]  -- K slices of the fp64 product: at least N K-steps per wave piece (measurement build, NMFAMD_F64_MIN_STEPS)
w=${1:-example}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
export NMFAMD_LIBRARY=$root/nmfgpu_amd/lib/libnmfgpu64_diag.so
for ms in 16 8 4; do
  export NMFAMD_F64_MIN_STEPS=$ms
  echo "== min steps per wave piece = $ms"
  python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step']*1e3,2), 'us per iteration')"
  bash tools/profile_bench.sh f64ms_$ms --workload $w --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/f64ms_$ms.txt 2>&1
  python3 tools/trace_by_grid.py gpurun_out/prof_f64ms_$ms/*/*kernel_trace.csv 4
done
