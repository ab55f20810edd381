#!/bin/bash
# memory-path counters of the Eq. 8 kernels (development aid): tools/exp/mem_pmc.sh <outdir>
OUT=$(realpath -m $1); shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_all.txt 2>&1
grep -o -E "\b(TA|TCP|TCC|TD|SQ|GRBM|SPI)_[A-Za-z0-9_]+" $OUT/counters_all.txt | sort -u > $OUT/counters.txt
B="python3 $ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 6 --warmup 2 --e2e-impressions 0 --impressions 8000"
i=0
while read -r line; do
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -o t -- $B > /dev/null 2> $OUT/p$i.err
  tail -2 $OUT/p$i.err > $OUT/p$i.tail; rm -f $OUT/p$i.err
done <<'LIST'
TA_BUSY_avr TA_BUSY_max TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TA_BUSY_sum
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES
GRBM_GUI_ACTIVE GRBM_COUNT
LIST
cd $ROOT
python3 tools/pmc_table.py $OUT/p* --match xattn_sparse > $OUT/sparse_mem_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/p* --match gemm_bf16x6s_kernel\<3 > $OUT/gemm_mem_pmc.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*.csv" -size +4M -delete
