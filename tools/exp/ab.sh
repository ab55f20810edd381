for pe in 1 4 1 4; do
  echo "== DIGAT_BENCH_PROFILE_EVERY=$pe"
  DIGAT_BENCH_PROFILE_EVERY=$pe python bench.py --steps 120 --warmup 10 --extra-steps 0 --cpu-rows 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline'].get('launches'), j['roofline']['avg_launch_ms'], j['roofline_xattn']['avg_launch_ms'])"
done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
DIGAT_BENCH_PROFILE_EVERY=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/reg1 -o t -- python3 bench.py --extra-steps 0 --cpu-rows 0 > gpurun_out/reg1.json 2>/dev/null
python tools/trace_region.py gpurun_out/reg1/t_kernel_trace.csv | head -6
python -c "
import json
j=json.loads(open('gpurun_out/reg1.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['launches'], j['roofline_xattn']['avg_launch_ms'], j['roofline_xattn']['launches'])"
