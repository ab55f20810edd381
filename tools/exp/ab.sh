for i in 1 2; do python tools/train_bench.py 2>/dev/null | tail -n 1; done
timeout 600 python -m pytest tests/test_hip_training.py tests/test_hip_ddp.py -m gpu -x -q 2>&1 | tail -n 2
