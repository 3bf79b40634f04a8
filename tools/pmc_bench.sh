#!/bin/bash
# usage: tools/pmc_bench.sh TAG "COUNTER COUNTER ..." -- one rocprofv3 --pmc pass over a short default bench (counters only, no tracing).
# FETCH_SIZE / WRITE_SIZE / TCC_* are derived over many TCC instances: ONE of them per pass ("exceeds the capabilities of
# the hardware" otherwise, and rocprofv3 then hangs in its abort handler -- hence the timeout).
tag=$1; ctrs=$2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 170 rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT && echo "== $tag rc=$rc" && python tools/pmc_summary.py gpurun_out/pmc_$tag factor_product
exit $rc
