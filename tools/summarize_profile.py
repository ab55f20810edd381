#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<round>/...) into the small files kept under profiles/.

  python tools/summarize_profile.py gpurun_out/r01 profiles/r01

Writes <out>_kernel_stats.csv (copy of --kernel-trace --stats), <out>_pmc.json (per-kernel FETCH_SIZE /
WRITE_SIZE per launch, raw KB and bytes corrected as MI355X_MICROARCH.md §HBM prescribes: gfx950
FETCH_SIZE counts wide coalesced reads at half their size -> reads = 2 * FETCH_SIZE * 1024,
writes = WRITE_SIZE * 1024) and prints a table.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counters(path, name):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                per[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return per


def main():
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, out + "_kernel_stats.csv")
    fetch, write = counters(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), counters(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    rows_per_step = 1024
    try:      # the launch size the counted run used (bench.py's own line of the same collection)
        rows_per_step = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])["config"]["rows_per_step"]
    except (OSError, ValueError, KeyError, IndexError):
        pass
    # issue-side counters of the same command (pmc_sq / pmc_mfma passes): how busy the vector ALUs were and how long waves sat
    # waiting — the Eq. 8 kernels are bound there, not by bytes (VALU busy = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES: quad-cycles of
    # VALU activity summed over the four SIMDs of a CU x 4 / (4 SIMDs x busy CU cycles))
    sq = {n: counters(os.path.join(src, d), n) for d, names in (("pmc_sq", ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU")),
                                                                ("pmc_mfma", ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")))
          for n in names}

    def mean(name, k):
        v = sq.get(name, {}).get(k, [])
        return sum(v) / len(v) if v else None
    summary = {}
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, []), write.get(k, [])
        if not fk or not wk or "at::native" in k or "rocclr" in k:
            continue
        summary[k] = {
            "launches": len(fk),
            "fetch_kb_mean": sum(fk) / len(fk), "fetch_kb_max": max(fk),
            "write_kb_mean": sum(wk) / len(wk), "write_kb_max": max(wk),
            "hbm_bytes_mean": (2 * sum(fk) / len(fk) + sum(wk) / len(wk)) * 1024,
            "hbm_bytes_max_launch": (2 * max(fk) + max(wk)) * 1024,
        }
        wc, wa, av, bc, mf = (mean(n, k) for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"))
        if wc and wa is not None:
            summary[k]["wait_fraction_of_wave_cycles"] = wa / wc
        if bc and av is not None:
            summary[k]["valu_busy_fraction"] = av / bc
        if bc and mf:
            summary[k]["mfma_busy_fraction"] = mf / (4.0 * bc)
    json.dump({"unit_note": "FETCH_SIZE/WRITE_SIZE in KB per launch; hbm_bytes = (2*FETCH + WRITE)*1024 "
                            "(gfx950 half-count correction on wide reads, MI355X_MICROARCH.md §HBM)",
               "rows_per_step": rows_per_step, "kernels": summary}, open(out + "_pmc.json", "w"), indent=1)
    for k, v in summary.items():
        print(f"{k[:48]:48s} n={v['launches']:4d} fetch(max)={v['fetch_kb_max']/1e3:8.1f} MB  write(max)={v['write_kb_max']/1e3:8.1f} MB"
              f"  hbm(max launch, corrected)={v['hbm_bytes_max_launch']/1e6:8.1f} MB")


if __name__ == "__main__":
    main()
