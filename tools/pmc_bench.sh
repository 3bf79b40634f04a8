#!/bin/bash
# usage: tools/pmc_bench.sh TAG "COUNTER COUNTER ..." [KERNEL-SUBSTRING] [bench.py arguments ...]
# one rocprofv3 --pmc pass over a short bench run (counters only, no tracing), then the per-kernel averages.
# FETCH_SIZE / WRITE_SIZE / TCC_* are derived over many TCC instances: ONE of them per pass ("exceeds the capabilities of
# the hardware" otherwise, and rocprofv3 then hangs in its abort handler -- hence the timeout).
tag=$1; ctrs=$2; kern=${3:-factor_product}; shift; shift; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 170 rocprofv3 --pmc $ctrs --output-format csv -d $root/gpurun_out/pmc_$tag -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events "$@" > $root/gpurun_out/pmc_$tag.log 2>&1
rc=$?
cd $root && echo "== $tag rc=$rc" && python3 tools/pmc_summary.py gpurun_out/pmc_$tag $kern
exit $rc
