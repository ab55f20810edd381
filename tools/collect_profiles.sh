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
B="python3 $ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 30 --e2e-impressions 0 --impressions 12000"
# the traced run records its event pairs on EVERY step of the timed region, so that the library's averages and the trace's
# (tools/trace_region.py, below) are over the same launches
DIGAT_BENCH_PROFILE_EVERY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B --detail $OUT/bench_traced_detail.json > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_mfma -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_mfma.err
# the memory path of the Eq. 8 kernels (round 5): texture addresser / data busy, L1 stalled on outstanding misses, L2 hits
rocprofv3 --pmc TA_BUSY_avr TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mem -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_mem.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_mem2 -o t -- $B --steps 8 --warmup 2 > /dev/null 2> $OUT/pmc_mem2.err
cd $ROOT
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_mfma --match gemm_bf16x6s > $OUT/gemm_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mem $OUT/pmc_mem2 --match xattn_sparse > $OUT/sparse_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write --match xattn_small_lds > $OUT/news_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write --match gemm_skinny_split > $OUT/skinny_pmc.txt 2>&1
python3 tools/pmc_table.py $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write --match topic_pool > $OUT/topic_pmc.txt 2>&1
# configs[2] / configs[3]: kernel stats of the stress and MIND-large workloads (single-stream trace, grouped by kernel and grid)
bash tools/exp/solo_trace.sh $OUT/stress --workload mind-small-stress --impressions 4096 > $OUT/stress_solo_kernels.txt 2>&1
bash tools/exp/solo_trace.sh $OUT/large --workload mind-large-default --impressions 4096 > $OUT/large_solo_kernels.txt 2>&1
bash tools/exp/solo_trace.sh $OUT/default_solo > $OUT/default_solo_kernels.txt 2>&1
# the other adjacency regime (full histories in 2-4 categories: 16 entries per node), single-stream kernel table
bash tools/exp/solo_trace.sh $OUT/heavy --workload mind-small-heavy-history --impressions 4096 > $OUT/heavy_solo_kernels.txt 2>&1
# one 1024-row drop-in pass on one caller stream, launch by launch (queue, start, duration, idle time in front)
bash tools/exp/dropin_timeline.sh $OUT/dropin > $OUT/dropin_timeline.txt 2>&1
python3 tools/summarize_profile.py $OUT $OUT/final > $OUT/summary.txt 2>&1
# the matrix-core ceiling of the projection GEMM's tiling (operands resident: no DMA, no split) and the training step's kernel table
[ -x tools/exp/mfma_ceiling ] && ./tools/exp/mfma_ceiling > $OUT/mfma_ceiling.txt 2>&1
python3 $ROOT/bench.py --mode train --steps 60 --warmup 10 --impressions 4096 > $OUT/train_plain.json 2>/dev/null
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -o t -- python3 $ROOT/bench.py --mode train --steps 30 --warmup 5 --impressions 4096 > $OUT/train_bench.json 2> $OUT/train_trace.err )
python3 - > $OUT/train_step_kernels.txt 2>&1 <<PYEOF
import csv, glob, json
f = glob.glob("$OUT/train_trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
line = json.loads(open("$OUT/train_bench.json").read().strip().splitlines()[-1])
plain = json.loads(open("$OUT/train_plain.json").read().strip().splitlines()[-1])
steps = 30 + 5 + 3            # timed + warm-up + the profiled MFMA pass; the clock-based pre-warm adds more: calls per step are quoted per TIMED-RUN step count below
tot_calls = sum(int(r["Calls"]) for r in rows); tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
# the number of steps in the trace = calls of a once-per-step kernel
per_step = max(1, min(int(r["Calls"]) for r in rows if "click_loss_kernel" in r["Name"]))      # once per step
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --steps 30 --warmup 5 --impressions 4096  (pre-warm steps included)")
print("# %d training steps of 64 x 5 rows in the trace: %.3f ms of kernel time and %.0f launches per step; the traced run's own line: %.3f ms per step; UNTRACED run (60 steps): %.3f ms per step" % (per_step, tot_ns / per_step / 1e6, tot_calls / per_step, line["ms_per_step"], plain["ms_per_step"]))
print("%-72s %10s %9s %8s" % ("kernel", "calls/step", "us/step", "avg us"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
    print("%-72s %10.1f %9.1f %8.1f" % (r["Name"][:72], int(r["Calls"]) / per_step, float(r["TotalDurationNs"]) / per_step / 1e3, float(r["AverageNs"]) / 1e3))
PYEOF
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
