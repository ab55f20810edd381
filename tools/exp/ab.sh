mkdir -p gpurun_out/r02aa
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02aa/pytest.log 2>&1; tail -n 6 gpurun_out/r02aa/pytest.log
