#!/bin/bash
# usage: tools/trace_shape.sh TAG M N R ALG PREC ITERS -- rocprofv3 kernel stats of tools/run_shape.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ts_$tag -- python3 $GRAFT_REPO_ROOT/tools/run_shape.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/ts_$tag.log 2>&1
cd $GRAFT_REPO_ROOT && f=$(ls gpurun_out/ts_$tag/*/*kernel_trace.csv | head -1) && echo "== $tag $*" && tail -1 gpurun_out/ts_$tag.log && python tools/kstats.py $f | head -12
