#!/bin/bash
# timeline of ONE 1024-row drop-in pass (Model.inference on expanded user tensors, one caller stream): every launch with its queue,
# start offset, duration and the idle gap in front of it: tools/exp/dropin_timeline.sh <outdir> [bench args]
OUT=$(realpath -m $1); shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export DIGAT_BENCH_LANES=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --batch 1024 --per-row-users --extra-steps 0 --cpu-rows 0 --e2e-impressions 0 --steps 40 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/trace.err
cd $ROOT
tail -c 600 $OUT/bench.json
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "digat_region_marker" in r["Kernel_Name"]]
seg = rows[idx[0] + 1: idx[1]]
steps = 40
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
busy, end = 0, 0
for r in seg:       # union of the kernels' intervals
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > end:
        busy += e - max(s, end); end = e
print(f"\n{len(seg)} launches in {steps} passes; per pass: span {span/steps/1e3:.1f} us, some kernel running {busy/steps/1e3:.1f} us, "
      f"sum of kernel times {sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/steps/1e3:.1f} us, {len(seg)/steps:.1f} launches")
# one pass from the middle: from a build_user_nodes launch to the next
firsts = [i for i, r in enumerate(seg) if r["Kernel_Name"].startswith("user_row_runs") or "user_rows_same" in r["Kernel_Name"]]
if len(firsts) < 22:
    firsts = [i for i, r in enumerate(seg) if "row_logits" in r["Kernel_Name"]]
a, b = firsts[20], firsts[21]
t0, end = int(seg[a]["Start_Timestamp"]), 0
print(f"{'kernel':52s} {'queue':>5s} {'start us':>9s} {'dur us':>8s} {'idle before':>11s}")
for r in seg[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end) / 1e3 if end else 0.0
    print(f"{r['Kernel_Name'].split('(')[0][:52]:52s} {r.get('Queue_Id', '?'):>5s} {(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} {gap:11.1f}")
    end = max(end, e)
PY
find $OUT -name "*kernel_trace.csv" -size +8M -delete
