mkdir -p gpurun_out/r02j
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02j/pmc_a -o t -- python3 $GRAFT_REPO_ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 8 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02j/err_a.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02j/pmc_b -o t -- python3 $GRAFT_REPO_ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 8 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02j/err_b.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02j/pmc_fetch -o t -- python3 $GRAFT_REPO_ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 8 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02j/err_c.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02j/pmc_write -o t -- python3 $GRAFT_REPO_ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 8 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02j/err_d.txt
cd $GRAFT_REPO_ROOT; python tools/pmc_table.py gpurun_out/r02j/pmc_a gpurun_out/r02j/pmc_b gpurun_out/r02j/pmc_fetch gpurun_out/r02j/pmc_write --match xattn_entry > gpurun_out/r02j/table.txt 2>&1
find gpurun_out/r02j -name "*.csv" -size +2M -delete
