mkdir -p gpurun_out/r02x
for p in 0 1 0 1; do DIGAT_SPARSE_XCD=$p python bench.py --extra-steps 0 --cpu-rows 512 --cpu-seconds 6 --steps 40 > gpurun_out/r02x/b_xcd${p}_$RANDOM.json 2>/dev/null; done
