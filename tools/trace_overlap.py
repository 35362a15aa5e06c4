#!/usr/bin/env python3
"""How busy the GPU is over a rocprofv3 --kernel-trace: the union of the kernels' intervals against the span they
cover, and how many kernels run side by side.  python tools/trace_overlap.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5            # look at the steady second half
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * skip
ev = [e for e in ev if e[0] >= cut]
span = max(e[1] for e in ev) - ev[0][0]
pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
busy = 0
depth_time = {}
cur, last = 0, pts[0][0]
for t, d in pts:
    if cur > 0:
        busy += t - last
    depth_time[cur] = depth_time.get(cur, 0) + (t - last)
    cur += d
    last = t
total_k = sum(e[1] - e[0] for e in ev)
print(f"kernels {len(ev)}  span {span / 1e3:.1f} us  busy (>= 1 kernel) {busy / 1e3:.1f} us = {busy / span:.3f}  sum of durations {total_k / 1e3:.1f} us "
      f"= {total_k / span:.2f} x span")
for k in sorted(depth_time):
    print(f"  {k} kernels in flight: {depth_time[k] / span:.3f} of the span")
by = {}
for s, e, n in ev:
    n = n.split("(")[0][-40:]
    by.setdefault(n, [0, 0])
    by[n][0] += e - s
    by[n][1] += 1
for n, (d, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {n:40s} {c:6d} launches  avg {d / c / 1e3:7.1f} us  {d / span:.3f} of the span")
