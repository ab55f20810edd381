mkdir -p gpurun_out/r02p
python bench.py --extra-steps 0 --cpu-rows 512 --cpu-seconds 8 --steps 30 > gpurun_out/r02p/b_off.json 2>/dev/null
for c in 0 2 3; do DIGAT_XATTN_STAGED=1 DIGAT_STAGED_CFG=$c python bench.py --extra-steps 0 --cpu-rows 512 --cpu-seconds 8 --steps 30 > gpurun_out/r02p/b_cfg$c.json 2>/dev/null; done
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "staged" > gpurun_out/r02p/pytest.log 2>&1; tail -n 3 gpurun_out/r02p/pytest.log
