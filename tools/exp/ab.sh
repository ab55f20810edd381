mkdir -p gpurun_out/r02t
timeout 900 python -m pytest tests/test_hip_training.py tests/test_hip_lowprec.py -m gpu -x -q -s > gpurun_out/r02t/pytest.log 2>&1; grep -v "^$" gpurun_out/r02t/pytest.log | tail -n 12
python bench.py --mode train --steps 20 > gpurun_out/r02t/train_fp32.json 2>/dev/null
python bench.py --mode train --steps 20 --train-precision bf16 > gpurun_out/r02t/train_bf16.json 2>/dev/null
python bench.py --workload mind-small-stress --extra-steps 0 --cpu-rows 0 --steps 12 --impressions 4096 > gpurun_out/r02t/stress_fp32.json 2>/dev/null
python bench.py --workload mind-small-stress --extra-steps 0 --cpu-rows 0 --steps 12 --impressions 4096 --projection pq-bf16 > gpurun_out/r02t/stress_pq.json 2>/dev/null
cat gpurun_out/r02t/train_fp32.json gpurun_out/r02t/train_bf16.json | cut -c1-400
