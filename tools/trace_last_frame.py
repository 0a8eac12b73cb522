#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV of tools/prof_lone.py and prints the LAST frame's chain: per kernel its start
(from the frame's first kernel), duration and the gap since the previous kernel ended.  usage: trace_last_frame.py <csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]) for r in rows)
last_begin = max(i for i, e in enumerate(ev) if e[2].startswith("k_begin"))
fr = ev[last_begin:]
t0, prev_end, busy = fr[0][0], fr[0][0], 0
for s, e, n in fr:
    print(f"  {n:28s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f} us")
    busy += e - s; prev_end = max(prev_end, e)
print(f"  chain {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {(prev_end - t0 - busy) / 1e3:.1f} us, {len(fr)} launches")
