#!/bin/bash
# A/B of one environment switch on the default bench workload, alternating runs: tools/exp/ab_env.sh VAR A B [runs] [bench args...]
VAR=$1; A=$2; B=$3; RUNS=${4:-2}; shift 4
for r in $(seq 1 $RUNS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 200 --warmup 20 --cpu-rows 0 --extra-steps 0 "$@" 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = l['kernel_ms_per_step']; i = l['kernel_ms_per_step_single_stream']
print('$VAR=$v', 'ms/step %.4f' % l['ms_per_step'], 'imp/s %.0f' % l['value'], 'lanes', l['batches_in_flight'], 'in-region', {a: round(b, 3) for a, b in k.items()}, 'solo', {a: round(b, 3) for a, b in i.items()})
"
  done
done
