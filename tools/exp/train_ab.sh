#!/bin/bash
# alternating untraced training-step timings for values of an environment switch: tools/exp/train_ab.sh VAR v1 v2 [v3] -- [bench flags]
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in "${VALS[@]}"; do
    ms=$(env $VAR=$v python3 $ROOT/bench.py --mode train --steps 60 --warmup 10 --impressions 4096 "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$VAR=$v rep $rep: $ms ms"
  done
done
