mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_hip_lowprec.py -m gpu -q -s > gpurun_out/r02n/pytest.log 2>&1
for m in bf16x6 pq-bf16 pq-bf16-x1; do python bench.py --extra-steps 0 --cpu-rows 512 --cpu-seconds 8 --steps 30 --projection $m > gpurun_out/r02n/b_$m.json 2>/dev/null; done
grep -v "^$" gpurun_out/r02n/pytest.log | tail -n 30
