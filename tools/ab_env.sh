#!/bin/bash
# usage: tools/ab_env.sh ROUNDS "ENV=VAL ..." "ENV=VAL ..." ...   ("-" = no extra environment)
# interleaved rounds of the default bench, one line per round and setting: us per iteration
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-events ${AB_ARGS} | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$v]', round(d['ms_per_step']*1e3,2), 'frob', d['frobenius_last'])"
  done
done
