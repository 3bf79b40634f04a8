#!/bin/bash
# usage: tools/bench_all.sh TAG  -- every workload's bench line, unprofiled (gpurun_out/TAG_bench_<w>.json) and under rocprofv3 --kernel-trace --stats
# (gpurun_out/TAG_bench_<w>_profiled.json + TAG_kernel_stats_bench_<w>.csv); copy what is to be judged into profiles/
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
for w in c2 c5 c5-gdcls c4 c3 c2-f64 example; do
  name=$(echo $w | tr -d -)
  [ $name = c5gdcls ] && name=c5g
  [ $name = c2f64 ] && name=c2f64
  steps=200; [ $w = c4 ] && steps=60
  timeout -k 10 300 python3 bench.py --workload $w --steps $steps --warmup 20 > gpurun_out/${tag}_bench_$name.json 2> gpurun_out/${tag}_bench_$name.err || { echo "FAILED $w"; tail -3 gpurun_out/${tag}_bench_$name.err; exit 1; }
  echo "== $w: $(python3 -c "import json,sys; d=json.loads([l for l in open('gpurun_out/${tag}_bench_$name.json') if l.startswith('{')][0]); print(round(d['value'],1), d['unit'], round(d['ms_per_step']*1e3,1), 'us/step')")"
  bash tools/profile_bench.sh ${tag}_$name --workload $w --steps $((steps / 2)) --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_profile_$name.txt 2>&1 || { echo "profile FAILED $w"; tail -5 gpurun_out/${tag}_profile_$name.txt; exit 1; }
  head -8 gpurun_out/${tag}_profile_$name.txt
  cp gpurun_out/prof_${tag}_${name}_kernel_stats.csv gpurun_out/${tag}_kernel_stats_bench_$name.csv
  cp gpurun_out/prof_${tag}_$name.json gpurun_out/${tag}_bench_${name}_profiled.json
done
