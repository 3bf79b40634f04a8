#!/bin/bash
# In-kernel timeline of the split-operand product's two production forms (diag build), all variants, one box: tools/stamp_all.sh TAG
set -e
TAG=${1:-stamps}
OUT=gpurun_out/${TAG}.txt
mkdir -p gpurun_out
: > $OUT
export NMFAMD_LIBRARY=$PWD/nmfgpu_amd/lib/libnmfgpu64_diag.so
for v in 10 11 12 13 30 31 32 33; do
	NMFAMD_X3_VARIANT=$v python3 tools/stamp_x3.py 10000 5000 >> $OUT 2>&1
done
cat $OUT
