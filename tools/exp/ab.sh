for cfg in "0 3" "64 3" "128 3" "32 3" "96 3" "64 2" "128 2" "192 3"; do
  set -- $cfg
  echo "== side CUs $1 lanes $2"
  DIGAT_SIDE_CUS=$1 DIGAT_BENCH_LANES=$2 python bench.py --steps 150 --warmup 10 --extra-steps 0 --cpu-rows 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['kernel_ms_per_step'])"
done
