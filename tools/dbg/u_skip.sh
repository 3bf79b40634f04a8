#!/bin/bash
# timing experiment: which part of k_mu64_update costs what (results are WRONG with a skip mask; timing only)
for m in 0 1 2 4 8 15; do
  export NMFAMD_U_SKIP=$m
  bash tools/profile_bench.sh uskip$m --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-events > gpurun_out/uskip$m.txt 2>&1
  echo "== skip mask $m"; grep -E "k_mu64_update|k_factor_product" gpurun_out/uskip$m.txt
done
