for i in 1 2 3; do
  DIGAT_BENCH_LANES=3 python bench.py --steps 150 --warmup 10 --extra-steps 0 --cpu-rows 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['kernel_ms_per_step']['xattn'], j['kernel_ms_per_step_single_stream']['xattn'], j['roofline_xattn']['isolated_avg_launch_ms'])"
done
