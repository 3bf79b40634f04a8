#!/bin/bash
# interleaved A/B of the update tail (UpdateTail, csrc/kernels.h) on the rank-of-N rehearsal of config 2's column shards: us per iteration, tail on / off
for i in 1 2 3; do
  for nc in 625 1250 2500; do
    a=$(NMFAMD_SHARD_REHEARSE=1 python3 tools/shard_trace.py $nc 1 600 | grep -o ": [0-9.]* us/iteration" | grep -o "[0-9.]*")
    b=$(NMFAMD_NO_FUSED_TAIL=1 NMFAMD_SHARD_REHEARSE=1 python3 tools/shard_trace.py $nc 1 600 | grep -o ": [0-9.]* us/iteration" | grep -o "[0-9.]*")
    echo "round $i n = $nc: tail $a us, separate update launch $b us"
  done
done
for nc in 5000 2500; do
  a=$(python3 tools/shard_trace.py $nc fused 600 | grep -o ": [0-9.]* us/iteration" | grep -o "[0-9.]*")
  b=$(NMFAMD_NO_FUSED_TAIL=1 python3 tools/shard_trace.py $nc fused 600 | grep -o ": [0-9.]* us/iteration" | grep -o "[0-9.]*")
  echo "fused loop n = $nc: tails $a us, separate update launches $b us"
done
