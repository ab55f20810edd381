#!/bin/bash
# Collect the round's measurements on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_profiles.sh r02
# Writes gpurun_out/<tag>/{bench.json, trace/, pmc_fetch/, pmc_write/, pmc_sq/ ...}; tools/summarize_profile.py then condenses
# them into profiles/<tag>_final_*.  Counter passes are separate runs (--pmc never together with a trace: the pool refuses it).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python3 $ROOT/bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
B="python3 $ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 30 --e2e-impressions 0"
# the traced run records its event pairs on EVERY step of the timed region, so that the library's averages and the trace's
# (tools/trace_region.py, below) are over the same launches
DIGAT_BENCH_PROFILE_EVERY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B --detail $OUT/bench_traced_detail.json > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_mfma -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_mfma.err
cd $ROOT
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_mfma --match gemm_bf16x6s > $OUT/gemm_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write --match xattn_sparse > $OUT/sparse_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write --match xattn_small_lds > $OUT/news_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write --match gemm_skinny_split > $OUT/skinny_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write --match topic_pool > $OUT/topic_pmc.txt 2>&1
# configs[2] / configs[3]: kernel stats of the stress and MIND-large workloads (single-stream trace, grouped by kernel and grid)
bash tools/exp/solo_trace.sh $OUT/stress --workload mind-small-stress --impressions 4096 > $OUT/stress_solo_kernels.txt 2>&1
bash tools/exp/solo_trace.sh $OUT/large --workload mind-large-default --impressions 4096 > $OUT/large_solo_kernels.txt 2>&1
bash tools/exp/solo_trace.sh $OUT/default_solo > $OUT/default_solo_kernels.txt 2>&1
# the other adjacency regime (full histories in 2-4 categories: 16 entries per node), single-stream kernel table
bash tools/exp/solo_trace.sh $OUT/heavy --workload mind-small-heavy-history --impressions 4096 > $OUT/heavy_solo_kernels.txt 2>&1
python3 tools/summarize_profile.py $OUT $OUT/final > $OUT/summary.txt 2>&1
# per-kernel averages of the traced run's TIMED REGION only (between bench.py's two marker kernels), next to that run's own line
python3 tools/trace_region.py $OUT/trace/t_kernel_trace.csv > $OUT/timed_region_kernels.txt 2>&1
python3 - >> $OUT/timed_region_kernels.txt 2>&1 <<PYEOF
import json
j = json.load(open("$OUT/bench_traced_detail.json"))
r, x = j["roofline"], j["roofline_xattn"]
print()
print("the same run's own line (library events, sampled steps): ms_per_step %.4f; %s avg_launch_ms %.4f over %d launches; "
      "xattn avg_launch_ms %.4f over %d launches" % (j["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["launches"], x["avg_launch_ms"], x["launches"]))
PYEOF
# keep what is merged back small: drop the raw per-dispatch tables
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT
