"""Busy / overlap analysis of a rocprofv3 kernel trace (csv): how much of the wall time has 0, 1, 2+ kernels running."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
ev.sort()
# keep the steady-state middle third
t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
lo, hi = t0 + (t1 - t0) * 0.5, t0 + (t1 - t0) * 0.8
pts = []
for s, e, n, q in ev:
    if e < lo or s > hi:
        continue
    pts.append((max(s, lo), 1)); pts.append((min(e, hi), -1))
pts.sort()
depth, last, hist = 0, lo, collections.Counter()
for t, d in pts:
    hist[min(depth, 4)] += t - last
    last = t; depth += d
hist[min(depth, 4)] += hi - last
tot = hi - lo
print("window %.2f ms" % (tot / 1e6))
for k in sorted(hist):
    print("  %d kernels running: %5.1f %%" % (k, 100.0 * hist[k] / tot))
by_q = collections.Counter()
for s, e, n, q in ev:
    if e < lo or s > hi:
        continue
    by_q[q] += min(e, hi) - max(s, lo)
for q, v in by_q.most_common(8):
    print("  queue/stream %s busy %5.1f %%" % (q, 100.0 * v / tot))
