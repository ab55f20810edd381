#!/bin/bash
# Copy the condensed results of tools/collect_profiles.sh <tag> (gpurun_out/<tag>/) into profiles/ (tracked): tools/publish_profiles.sh r05
set -e
TAG=${1:?tag}
cd "$(dirname "$0")/.."
S=gpurun_out/$TAG
P=profiles/$TAG
cp $S/bench_detail.json ${P}_final_bench.json               # the full document of the default `python bench.py` run
cp $S/bench.json ${P}_final_bench_line.json                 # ... and the compact line the driver parses
cp $S/bench_traced_detail.json ${P}_traced_bench.json
cp $S/final_kernel_stats.csv ${P}_final_kernel_stats.csv
cp $S/final_pmc.json ${P}_final_pmc.json
cp $S/gemm_pmc.txt ${P}_gemm_pmc.txt
cp $S/sparse_pmc.txt ${P}_sparse_pmc.txt
cp $S/news_pmc.txt ${P}_news_lds_pmc.txt
cp $S/skinny_pmc.txt ${P}_skinny_split_pmc.txt
cp $S/topic_pmc.txt ${P}_topic_pmc.txt
cp $S/timed_region_kernels.txt ${P}_timed_region_kernels.txt
for w in default stress large heavy; do cp $S/${w}_solo_kernels.txt ${P}_${w}_solo_kernels.txt; done
[ -f $S/dropin_timeline.txt ] && cp $S/dropin_timeline.txt ${P}_dropin_timeline.txt
[ -f $S/mfma_ceiling.txt ] && cp $S/mfma_ceiling.txt ${P}_mfma_ceiling.txt
[ -f $S/train_step_kernels.txt ] && cp $S/train_step_kernels.txt ${P}_train_step_kernels.txt
ls -la ${P}_*
