#!/bin/bash
# usage (GPU box): bash tools/f32w_ride_ab.sh  -- the wide fp32 iteration with its Gram slices as a launch of their own (NMFAMD_F32W_RIDE=0), as passengers of the
# product launch wherever a CU is free (=1), and as the library decides (unset): launches and us per iteration at five shapes (measurement build)
export NMFAMD_LIBRARY=$PWD/nmfgpu_amd/lib/libnmfgpu64_diag.so
for shape in "10000 5000 128 mu f32" "10000 5000 158 nsnmf f32" "4096 165 158 nsnmf f32" "10000 5000 500 mu f32" "40000 4000 128 mu f32" "2000 1500 300 mu f32"; do
  for ride in 0 1 default; do
    echo "== $shape  NMFAMD_F32W_RIDE=$ride"
    if [ $ride = default ]; then unset NMFAMD_F32W_RIDE; else export NMFAMD_F32W_RIDE=$ride; fi
    timeout -k 10 300 python3 tools/launches_per_iteration.py $shape | head -7
  done
done
