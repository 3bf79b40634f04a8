#!/bin/bash
# usage: tools/ab.sh ROUNDS variantA.so variantB.so ... -- interleaved rounds of the default bench per library build
# ("base" = the in-tree library); prints us/iteration per round and build (guide: perf deltas from interleaved rounds in ONE call)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset NMFAMD_LIBRARY; else export NMFAMD_LIBRARY=$PWD/nmfgpu_amd/lib/variants/$v; fi
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-events ${AB_ARGS} | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step']*1e3,2))"
  done
done
