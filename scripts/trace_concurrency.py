#!/usr/bin/env python3
"""GPU occupancy in time from a rocprofv3 kernel_trace.csv: fraction of the traced window with >= 1 kernel
running, mean number of kernels in flight, and the same per phase.  usage: trace_concurrency.py kernel_trace.csv"""
import csv
import sys

ev = []
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
# only look at the last 60 % of the window (the timed steps, after warm-up and workload upload)
lo = t0 + int(0.4 * (t1 - t0))
cur, last, busy, area = 0, ev[0][0], 0, 0
hist = {}
for t, d in ev:
    if t > lo and last >= lo:
        dt = t - last
        if cur > 0:
            busy += dt
        area += cur * dt
        hist[min(cur, 8)] = hist.get(min(cur, 8), 0) + dt
    last = max(t, lo) if t > lo else t
    cur += d
win = t1 - lo
print("window %.1f ms: busy %.1f %%, mean kernels in flight %.2f" % (win / 1e6, 100.0 * busy / win, area / win))
for k in sorted(hist):
    print("  %d%s kernels: %.1f %%" % (k, "+" if k == 8 else "", 100.0 * hist[k] / win))
