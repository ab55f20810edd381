for f in 0 1; do
  echo "== DIGAT_GEMM_F16X3=$f"
  DIGAT_GEMM_F16X3=$f python tools/kbench.py linear 34304 1200 400 2>&1 | grep -E "bf16x6|linear M"
  DIGAT_GEMM_F16X3=$f python tools/kbench.py linear 10240 1200 400 2>&1 | grep -E "bf16x6"
  DIGAT_GEMM_F16X3=$f DIGAT_BENCH_LANES=3 python bench.py --steps 150 --warmup 10 --extra-steps 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['valid'], j['auc_match']['max_abs_metric_diff'], j['auc_match']['max_abs_score_diff'], j['kernel_ms_per_step'], j['kernel_ms_per_step_single_stream']['proj'])"
done
DIGAT_GEMM_F16X3=1 timeout 900 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -5
