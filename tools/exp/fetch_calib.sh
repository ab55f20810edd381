#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (run on the GPU box from the repo root): prints known bytes / counted bytes per access shape
# and writes gpurun_out/<tag>/fetch_calib.json
TAG=${1:-calib}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o t -- $ROOT/tools/exp/fetch_calib > $OUT/cal_known.txt 2> $OUT/cal_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o t -- $ROOT/tools/exp/fetch_calib > /dev/null 2> $OUT/cal_write.err
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
known = {}
toks = open(os.path.join(out, "cal_known.txt")).read().split()
for i, t in enumerate(toks):
    if t in ("stream_f4", "stream_dma", "rows128_dma", "rows1600_f4", "rows1600_f4g", "store_f4"):
        known[t] = int(toks[i + 1])
def counters(path, name):
    per = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                per.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]))
    return per
fetch, write = counters(os.path.join(out, "cal_fetch"), "FETCH_SIZE"), counters(os.path.join(out, "cal_write"), "WRITE_SIZE")
res = {}
def put(shape, kernel, per, pick):
    v = per.get(kernel)
    if not v:
        return
    v = pick(v)
    res[shape] = {"known_bytes": known[shape], "counter_kb": v, "factor": known[shape] / (v * 1024.0)}
put("stream_f4", "stream_f4", fetch, lambda v: v[-1])
put("stream_dma", "stream_dma", fetch, lambda v: v[-1])
put("rows128_dma", "rows128_dma", fetch, lambda v: v[-1])
put("rows1600_f4", "rows1600_f4", fetch, lambda v: v[-2])          # launches alternate: contiguous rows, gathered rows
put("rows1600_f4g", "rows1600_f4", fetch, lambda v: v[-1])
put("store_f4", "store_f4", write, lambda v: v[-1])
json.dump(res, open(os.path.join(out, "fetch_calib.json"), "w"), indent=1)
for k, v in res.items():
    print(f"{k:14s} known {v['known_bytes'] / 1e6:9.1f} MB  counter {v['counter_kb'] / 1e3:9.1f} MB(KB x 1e-3)  factor {v['factor']:.3f}")
PY
find $OUT -name "*counter_collection.csv" -size +1M -delete
