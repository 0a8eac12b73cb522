#!/usr/bin/env python3
"""Prints the kernels of the LAST burst of a rocprofv3 kernel-trace CSV as a timeline: start (us since the burst began),
duration, queue, name.  A burst = the kernels after the last gap of more than GAP us (default 300).  usage: trace_burst.py <csv> [GAP]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 300e3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:28], r.get("Queue_Id", "?")) for r in rows)
start = 0
end_so_far = ev[0][1]
for i in range(1, len(ev)):
    if ev[i][0] - end_so_far > gap:
        start = i
    end_so_far = max(end_so_far, ev[i][1])
burst = ev[start:]
t0 = burst[0][0]
print(f"{len(burst)} kernels, {(max(e[1] for e in burst) - t0) / 1e3:.1f} us")
for s, e, n, q in burst:
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  q{q}  {n}")
