#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints, for the last third of the run, per-kernel average durations, the
fraction of wall time with 0/1/2/3+ kernels running, and per-queue busy fractions.  usage: trace_timeline.py <csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 2 // 3
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
print(f"{len(ev)} kernels in {span/1e6:.3f} ms")
d = collections.defaultdict(list)
for s, e, n, q, st in ev: d[n].append(e - s)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:40s} n={len(v):5d} avg {sum(v)/len(v)/1e3:8.2f} us  total {sum(v)/1e6:7.3f} ms ({100*sum(v)/span:.0f}% of span)")
pts = []
for s, e, *_ in ev: pts += [(s, 1), (e, -1)]
pts.sort()
hist = collections.Counter(); cur = 0; last = pts[0][0]
for t, dlt in pts:
    hist[min(cur, 4)] += t - last; last = t; cur += dlt
print("  concurrency:", {k: f"{100*v/span:.0f}%" for k, v in sorted(hist.items())})
q = collections.defaultdict(int)
for s, e, n, qq, st in ev: q[(qq, st)] += e - s
print("  busy per (queue,stream):", {k: f"{100*v/span:.0f}%" for k, v in sorted(q.items())})
