mkdir -p gpurun_out/r02k
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02k/pytest.log 2>&1
python bench.py --extra-steps 0 --cpu-rows 0 --steps 30 > gpurun_out/r02k/b.json 2>/dev/null
python bench.py --extra-steps 0 --cpu-rows 0 --steps 30 > gpurun_out/r02k/b2.json 2>/dev/null
tail -n 5 gpurun_out/r02k/pytest.log
