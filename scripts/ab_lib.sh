#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call: scripts/ab_lib.sh <config> <old .so> [rounds] [extra bench args]
cfg=$1; old=$2; rounds=${3:-3}; shift 3 2>/dev/null
for i in $(seq $rounds); do
  for lib in "$old" ""; do
    env DG_LIB=$lib python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg lib=${lib:-new}', round(d['value']), round(d['ms_per_step'],2))"
  done
done
