for cfg in "0 3" "1 3" "2 3" "3 3" "1 2" "2 2" "3 2" "1 4" "2 4"; do
  set -- $cfg
  echo "== dummies $1 lanes $2"
  DIGAT_BENCH_DUMMY_STREAMS=$1 DIGAT_BENCH_LANES=$2 python bench.py --steps 150 --warmup 10 --extra-steps 0 --cpu-rows 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'])"
done
