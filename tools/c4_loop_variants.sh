#!/bin/bash
# usage: tools/c4_loop_variants.sh  -- the config-4 product's SHIPPED loop with parts taken out (measurement build, NMFAMD_BF_VARIANT; results void): per-launch
# times under rocprofv3 of variant 0 (production) 1 (no MFMAs) 2 (streamed operand's path only) 3 (factor fragments' loads only) 4 (no slab stores)
# 5 (MFMAs + LDS reads only) 6 (A loads only: no LDS ring, no barrier) 7 (A and F loads only).  profiles/r06_c4_loop.md is written from this.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
export NMFAMD_LIBRARY=$root/nmfgpu_amd/lib/libnmfgpu64_diag.so
for v in ${VARIANTS:-0 1 2 3 4 5 6 7}; do
  if [ $v = 0 ]; then unset NMFAMD_BF_VARIANT; else export NMFAMD_BF_VARIANT=$v; fi
  echo "== variant $v"
  bash tools/profile_bench.sh c4var_$v --workload c4 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c4var_$v.txt 2>&1
  python3 tools/trace_by_grid.py gpurun_out/prof_c4var_$v/*/*kernel_trace.csv 3
done
