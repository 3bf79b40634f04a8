#!/bin/bash
# usage: tools/variant_kernel_time.sh KERNEL_SUBSTRING "bench args" variant1 variant2 ...   ("base" = the in-tree library)
# rocprofv3 kernel stats of one bench.py run per library variant; prints the average duration of the kernels whose name contains KERNEL_SUBSTRING
kern=$1; args=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  if [ "$v" = base ]; then unset NMFAMD_LIBRARY; else export NMFAMD_LIBRARY=$root/nmfgpu_amd/lib/variants/$v.so; fi
  rm -rf /tmp/vk_$v; mkdir -p /tmp/vk_$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vk_$v -- python3 $root/bench.py $args > /tmp/vk_$v/out.json 2> /tmp/vk_$v/err.txt)
  f=$(ls /tmp/vk_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" "$kern" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if sys.argv[2] in r['Name']:
        print(f"{sys.argv[3]:8s} {r['Name'].split('(')[0][:60]:62s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.2f}us max={float(r['MaxNs'])/1e3:8.2f}", flush=True)
PY
done
