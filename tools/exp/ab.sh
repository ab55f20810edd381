mkdir -p gpurun_out/r02w
for p in 0 40000; do DIGAT_GEMM_DYNLDS=$p python bench.py --extra-steps 0 --cpu-rows 0 --steps 60 > gpurun_out/r02w/b_dyn${p}.json 2>gpurun_out/r02w/err$p.txt; done
