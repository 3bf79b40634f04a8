#!/bin/bash
# usage: tools/trace_shape.sh "M N R ALG DTYPE [theta]" [rows]   (on the GPU box) -- one traced run of 300 iterations of the resident engine at a shape, split by
# (kernel, grid): the H-side and W-side launches of one kernel apart (tools/launches_per_iteration.py's child under rocprofv3 --kernel-trace, tools/trace_by_grid.py)
set -e
cd "$(dirname "$0")/.."
R=$PWD
D=$(mktemp -d /tmp/trace_shape_XXXX)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/launches_per_iteration.py --child 300 $1 > /dev/null 2>&1)
python3 tools/trace_by_grid.py $(find $D -name '*kernel_trace.csv' | head -1) ${2:-14}
