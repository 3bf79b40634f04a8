#!/bin/bash
# usage: tools/insitu.sh TAG [ENV=VALUE ...] -- kernel-trace the default bench under the given environment, print per-kernel stats
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/insitu_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 > $GRAFT_REPO_ROOT/gpurun_out/insitu_$tag.log 2>&1
cd $GRAFT_REPO_ROOT && f=$(ls gpurun_out/insitu_$tag/*/*kernel_trace.csv | head -1) && echo "== $tag $*" && python tools/kstats.py $f | head -4
