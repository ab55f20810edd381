# A/B template (run through gpurun from the repository root): alternate two settings, two rounds each, print the bench's
# throughput / step time / Eq. 8 and projection solo times.  Edit VAR and the values.
VAR=DIGAT_BENCH_LANES
for rep in 1 2; do
for v in 2 3; do
  echo "== $VAR=$v"
  env $VAR=$v python bench.py --steps 120 --warmup 10 --extra-steps 0 --cpu-rows 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
iso = j['kernel_ms_per_step_single_stream']
print(j['value'], j['ms_per_step'], 'solo proj', iso['proj'], 'xattn', iso['xattn'])"
done
done
