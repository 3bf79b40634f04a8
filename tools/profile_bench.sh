#!/bin/bash
# usage: tools/profile_bench.sh TAG [bench.py arguments ...]
# rocprofv3 --kernel-trace --stats of one bench.py run; leaves gpurun_out/prof_TAG/ (raw), gpurun_out/prof_TAG_kernel_stats.csv
# (the summary that is copied to profiles/) and gpurun_out/prof_TAG.json (the bench line of the profiled run).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py "$@" > $root/gpurun_out/prof_$tag.json 2> $root/gpurun_out/prof_$tag.err
rc=$?
cd $root
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_${tag}_kernel_stats.csv && echo "== $tag rc=$rc" && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'].split('(')[0][:78]:80s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:9.2f}us pct={float(r['Percentage']):6.2f}")
PY
tail -c 600 gpurun_out/prof_$tag.json
exit $rc
