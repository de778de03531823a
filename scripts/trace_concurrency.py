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

# the longest stretches with nothing running (position in the window, what ended before and what started after)
ended, started = {}, {}
for r in rows:
    ended[int(r["End_Timestamp"])] = r["Kernel_Name"].split("(")[0][-40:]
    started[int(r["Start_Timestamp"])] = r["Kernel_Name"].split("(")[0][-40:]
gaps = []
cur, last_end = 0, None
for t, d in ev:
    if cur == 0 and d == 1 and last_end is not None and t > lo:
        gaps.append((t - last_end, last_end, t))
    cur += d
    if cur == 0:
        last_end = t
gaps.sort(reverse=True)
idle = sum(g[0] for g in gaps)
print("idle stretches: %d, %.2f ms in all; stretches over 100 us: %.2f ms" % (len(gaps), idle / 1e6, sum(g[0] for g in gaps if g[0] > 100000) / 1e6))
for g, a, b in gaps[:12]:
    print("  %7.1f us at %6.1f ms   after %-40s before %s" % (g / 1e3, (a - lo) / 1e6, ended.get(a, "?"), started.get(b, "?")))
