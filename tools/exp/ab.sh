for i in 1 2 3; do
  python bench.py --steps 120 --warmup 10 --extra-steps 0 2>/dev/null | tail -n 1 | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['valid'], j['batches_in_flight'], j['auc_match']['max_abs_score_diff'])"
done
