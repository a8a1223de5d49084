#!/bin/bash
# A/B of one environment variable on the bench step, alternating runs: tools/gpu/ab_env.sh VAR valueA valueB [runs]
VAR=$1; A=$2; B=$3; RUNS=${4:-3}
for i in $(seq 1 $RUNS); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', round(d['ms_per_step'], 2))"
  done
done
