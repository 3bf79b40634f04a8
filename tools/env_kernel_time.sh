#!/bin/bash
# usage: tools/env_kernel_time.sh KERNEL_SUBSTRING "bench args" VAR value1 value2 ...   -- like variant_kernel_time.sh, one run per value of an environment variable
kern=$1; args=$2; var=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  export $var=$v
  rm -rf /tmp/ek_$v; mkdir -p /tmp/ek_$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ek_$v -- python3 $root/bench.py $args > /tmp/ek_$v/out.json 2> /tmp/ek_$v/err.txt)
  f=$(ls /tmp/ek_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" "$kern" "$var=$v" /tmp/ek_$v/out.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
ms = None
for line in open(sys.argv[4]):
    if line.startswith('{'): ms = json.loads(line)['ms_per_step']
for r in rows:
    if sys.argv[2] in r['Name']:
        print(f"{sys.argv[3]:28s} {r['Name'].split('(')[0][:50]:52s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.2f}us   step={ms*1e3 if ms else -1:.1f}us", flush=True)
PY
done
