mkdir -p gpurun_out/r02l
timeout 900 python -m pytest tests/test_hip_training.py tests/test_hip_ddp.py -m gpu -x -q > gpurun_out/r02l/pytest.log 2>&1
python tools/train_bench.py > gpurun_out/r02l/train.txt 2>&1
tail -n 25 gpurun_out/r02l/pytest.log; cat gpurun_out/r02l/train.txt | tail -n 3
