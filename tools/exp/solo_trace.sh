#!/bin/bash
# single-stream kernel trace of the default workload, grouped by (kernel, grid): tools/exp/solo_trace.sh <outdir> [bench args]
OUT=$(realpath -m $1); shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
DIGAT_SINGLE_STREAM=1 DIGAT_BENCH_LANES=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --extra-steps 0 --cpu-rows 0 --steps 40 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/trace.err
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# timed region: between the marker kernels
idx = [i for i, r in enumerate(rows) if "digat_region_marker" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "digat_region_marker" in r["Kernel_Name"]]
seg = rows[idx[0] + 1: idx[1]] if len(idx) >= 2 else rows
agg = collections.defaultdict(list)
for r in seg:
    name = r["Kernel_Name"].split("(")[0][:60]
    agg[(name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
steps = 40
tot = sum(sum(v) for v in agg.values())
print(f"timed region: {len(seg)} launches, {tot/steps:.1f} us of kernel time per step")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:60s} grid {k[1]:>9s} wg {k[2]:>5s}  n/step {len(v)/steps:5.2f}  avg {sum(v)/len(v):7.1f} us  per step {sum(v)/steps:7.1f} us")
PY
find $OUT -name "*kernel_trace.csv" -size +8M -delete
