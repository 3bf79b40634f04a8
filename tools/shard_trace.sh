#!/bin/bash
# usage: tools/shard_trace.sh TAG ["NC MODE[:ENV=VAL,...]" ...]
# rocprofv3 kernel stats of the sharded loop at the strong-scaling shard widths of configs[1] (tools/shard_trace.py);
# default set: the fused loop, the team-of-one loop, and the rehearsal of what a rank of an N-GPU team runs (NMFAMD_SHARD_REHEARSE=1)
tag=${1:-r04}; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- "5000 fused" "5000 1" "5000 1:NMFAMD_SHARD_REHEARSE=1" "2500 1:NMFAMD_SHARD_REHEARSE=1" "1250 1:NMFAMD_SHARD_REHEARSE=1" "625 1:NMFAMD_SHARD_REHEARSE=1"
for spec in "$@"; do
  nc=${spec%% *}; rest=${spec#* }; mode=${rest%%:*}; envs=""
  [ "$rest" != "$mode" ] && envs=${rest#*:}
  name=n${nc}_m${mode}$( [ -n "$envs" ] && echo _$(echo $envs | tr -c 'A-Za-z0-9\n' '_') )
  out=$root/gpurun_out/shardtrace_${tag}_$name
  ( for kv in $(echo $envs | tr ',' ' '); do export $kv; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/shard_trace.py $nc $mode > $out.txt 2> $out.err ) || { echo "FAILED $spec"; tail -5 $out.err; exit 1; }
  f=$(ls $out/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== n=$nc mode=$mode $envs: $(grep 'us/iteration' $out.txt)"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    if int(r['Calls']) < 100: continue
    print(f"  {r['Name'].split('(')[0][:70]:72s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.2f}us pct={float(r['Percentage']):6.2f}")
PY
  cp "$f" $root/gpurun_out/shardtrace_${tag}_${name}_kernel_stats.csv
done
