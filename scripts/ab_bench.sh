#!/bin/bash
# A/B of one environment switch inside ONE gpurun call (boxes differ by a few per cent): alternates the two settings.
#   scripts/ab_bench.sh <config> <VAR> <valueA> <valueB> [rounds] [extra bench args]
cfg=$1; var=$2; a=$3; b=$4; rounds=${5:-3}; shift 5 2>/dev/null
for i in $(seq $rounds); do
  for val in $a $b; do
    env $var=$val python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $var=$val', round(d['value']), round(d['ms_per_step'],2))"
  done
done
