for pe in 4 8 1000 4 8 1000; do
  echo "== DIGAT_BENCH_PROFILE_EVERY=$pe"
  DIGAT_BENCH_PROFILE_EVERY=$pe DIGAT_BENCH_LANES=3 python bench.py --steps 120 --warmup 10 --extra-steps 0 --cpu-rows 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline'].get('launches'))"
done
