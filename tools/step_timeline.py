#!/usr/bin/env python3
"""Print the kernel timeline of one bench step from a rocprofv3 --kernel-trace CSV:
  python tools/step_timeline.py gpurun_out/r01h/trace/t_kernel_trace.csv [step]
A step ends with row_logits_kernel; columns: start us, end us, duration us, queue, kernel."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("row_logits")]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) // 2
    a, b = ends[k - 1], ends[k]
    t0 = int(rows[a]["End_Timestamp"])
    busy = {}
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0) + e - s
        print(f"{s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{r['Queue_Id']}  {r['Kernel_Name'].split('(')[0][:48]}")
    print("busy us per queue:", {q: round(v / 1e3, 1) for q, v in busy.items()},
          " step:", round((int(rows[b]["End_Timestamp"]) - t0) / 1e3, 1))


if __name__ == "__main__":
    main()
