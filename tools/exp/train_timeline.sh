#!/bin/bash
# the ordered kernel timeline of ONE training step (start us, duration us, gap to the previous kernel's end, kernel):
#   tools/exp/train_timeline.sh <outdir> [bench.py flags]
OUT=$(realpath -m $1); shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 $ROOT/bench.py --mode train --steps 12 --warmup 3 --impressions 4096 "$@" > $OUT/tl_bench.json 2> $OUT/tl.err )
python3 - > $OUT/train_timeline.txt 2>&1 <<PYEOF
import csv, glob
f = glob.glob("$OUT/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with the optimiser's last multi_tensor_apply launch: cut at the first xattn_score launch after one
adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
cuts = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 20]
a, b = cuts[-3] + 1, cuts[-2] + 1
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%8.1f %7.1f %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].split("(")[0][:90]))
    prev = e
print("# launches %d  busy %.1f us  span %.1f us" % (b - a, busy / 1e3, (prev - t0) / 1e3))
PYEOF
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -3 $OUT/train_timeline.txt
