"""Per-kernel statistics of bench.py's TIMED REGION cut out of a rocprofv3 kernel trace (csv).

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py ...
    python tools/trace_region.py out/*/*kernel_trace.csv [--region K]

bench.py launches digat_region_marker_kernel at both ends of every timed region (the main workload first, then the extra
workloads); region K (default 0) is the K-th pair.  The averages printed here are what roofline.avg_launch_ms must agree with."""
import csv
import sys
import collections


def main():
    path = sys.argv[1]
    region = int(sys.argv[sys.argv.index("--region") + 1]) if "--region" in sys.argv else 0
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    marks = [s for s, e, n in rows if "digat_region_marker_kernel" in n]
    if len(marks) < 2 * region + 2:
        sys.exit(f"only {len(marks)} markers in the trace")
    lo, hi = marks[2 * region], marks[2 * region + 1]
    stats = collections.defaultdict(list)
    for s, e, n in rows:
        if lo <= s and e <= hi and "digat_region_marker_kernel" not in n:
            stats[n.split("(")[0]].append(e - s)
    total = sum(sum(v) for v in stats.values())
    print(f"timed region {(hi - lo) / 1e6:.3f} ms, {sum(len(v) for v in stats.values())} launches, kernel time {total / 1e6:.3f} ms")
    print(f"{'kernel':70s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
    for n, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n[:70]:70s} {len(v):6d} {sum(v) / len(v) / 1e3:9.2f} {sum(v) / 1e6:9.3f} {100.0 * sum(v) / total:6.2f}")


if __name__ == "__main__":
    main()
