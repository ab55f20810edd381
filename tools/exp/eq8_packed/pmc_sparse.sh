#!/bin/bash
# usage: bash pmc_sparse.sh <tag>   (run from repo root on the GPU box)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python3 $ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 8 --warmup 2 --e2e-impressions 0 --impressions 12000"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o t -- $B > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_mfma -o t -- $B > /dev/null 2> $OUT/pmc_mfma.err
cd $ROOT
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_mfma --match xattn_sparse > $OUT/sparse_pmc.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +1M -delete
cat $OUT/sparse_pmc.txt
