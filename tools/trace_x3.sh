#!/bin/bash
# usage: tools/trace_x3.sh VARIANT -- rocprofv3 kernel durations of the standalone product timing loop
v=$1
export NMFAMD_X3_VARIANT=$v REPS=60
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_x3_$v -- python3 $GRAFT_REPO_ROOT/tools/time_x3.py > $GRAFT_REPO_ROOT/gpurun_out/trace_x3_$v.log 2>&1
cd $GRAFT_REPO_ROOT && f=$(ls gpurun_out/trace_x3_$v/*/*kernel_trace.csv | head -1) && echo "== variant $v" && cat gpurun_out/trace_x3_$v.log | grep "us" && python tools/kstats.py $f | grep factor_product
