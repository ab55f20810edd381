# A/B of the sparse Eq. 8 workgroup -> XCD mapping (DIGAT_SPARSE_XCD = workgroups per XCD run; 0 = round-robin)
for rep in 1 2; do
for g in 0 8 4 16 2; do
  echo "== DIGAT_SPARSE_XCD=$g"
  DIGAT_SPARSE_XCD=$g python bench.py --steps 100 --warmup 10 --extra-steps 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
print(j['ms_per_step'], j['kernel_ms_per_step'], 'solo', j.get('kernel_ms_per_step_single_stream'), 'xattn', j['roofline_xattn']['avg_launch_ms'], j['roofline_xattn']['isolated_avg_launch_ms'], 'valid', j.get('valid'))"
done
done
