#!/bin/bash
# the training step's kernel table (the same recipe as tools/collect_profiles.sh): tools/exp/train_trace.sh <outdir>
OUT=$(realpath -m $1); shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
python3 $ROOT/bench.py --mode train --steps 60 --warmup 10 --impressions 4096 "$@" > $OUT/train_plain.json 2>/dev/null
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -o t -- python3 $ROOT/bench.py --mode train --steps 30 --warmup 5 --impressions 4096 "$@" > $OUT/train_bench.json 2> $OUT/train_trace.err )
python3 - > $OUT/train_step_kernels.txt 2>&1 <<PYEOF
import csv, glob, json
f = glob.glob("$OUT/train_trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
line = json.loads(open("$OUT/train_bench.json").read().strip().splitlines()[-1])
plain = json.loads(open("$OUT/train_plain.json").read().strip().splitlines()[-1])
tot_calls = sum(int(r["Calls"]) for r in rows); tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
per_step = max(1, min(int(r["Calls"]) for r in rows if "click_loss_kernel" in r["Name"]))      # once per step
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --steps 30 --warmup 5 --impressions 4096  (pre-warm steps included)")
print("# %d training steps of 64 x 5 rows in the trace: %.3f ms of kernel time and %.0f launches per step; the run's own line: %.3f ms per step; untraced run: %.3f ms per step" % (per_step, tot_ns / per_step / 1e6, tot_calls / per_step, line["ms_per_step"], plain["ms_per_step"]))
print("%-72s %10s %9s %8s" % ("kernel", "calls/step", "us/step", "avg us"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:70]:
    print("%-72s %10.1f %9.1f %8.1f" % (r["Name"][:72], int(r["Calls"]) / per_step, float(r["TotalDurationNs"]) / per_step / 1e3, float(r["AverageNs"]) / 1e3))
PYEOF
find $OUT -name "*kernel_trace.csv" -size +8M -delete
head -75 $OUT/train_step_kernels.txt
