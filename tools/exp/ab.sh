for rep in 1 2; do
for v in "" ring4; do
  lib=""; [ -n "$v" ] && lib="$PWD/tools/exp/libdigat_$v.so"
  echo "== ${v:-default}"
  DIGAT_HIP_LIB=$lib DIGAT_GEMM_F16X3=1 python tools/kbench.py linear 34304 1200 400 2>&1 | grep bf16x6
  DIGAT_HIP_LIB=$lib python bench.py --steps 120 --warmup 10 --extra-steps 0 --cpu-rows 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['isolated_avg_launch_ms'], j['kernel_ms_per_step_single_stream']['proj'])"
done
done
DIGAT_HIP_LIB=$PWD/tools/exp/libdigat_ring4.so timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_lowprec.py -x -q 2>&1 | tail -2
