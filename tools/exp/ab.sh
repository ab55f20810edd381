for rep in 1 2; do
for o in 0 1; do
  echo "== DIGAT_SPARSE_ONLINE=$o"
  DIGAT_SPARSE_ONLINE=$o python bench.py --steps 120 --warmup 10 --extra-steps 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['valid'], j['auc_match']['max_abs_score_diff'], j['kernel_ms_per_step_single_stream']['xattn'], j['roofline_xattn']['isolated_avg_launch_ms'], j['kernel_ms_per_step']['xattn'])"
done
done
DIGAT_SPARSE_ONLINE=1 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "not staged" 2>&1 | tail -3
