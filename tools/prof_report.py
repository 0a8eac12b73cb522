#!/usr/bin/env python3
"""Summarises rocprofv3 csv output (kernel trace and/or pmc counter collection) per kernel."""
import csv, glob, sys, collections, statistics
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in by.items():
        print(f"{k:40s} calls {len(v):4d} avg {statistics.mean(v):9.1f} us  med {statistics.median(v):9.1f}  min {min(v):9.1f} max {max(v):9.1f}")
    b = [r for r in rows if r["Kernel_Name"].startswith("k_bounce")]
    if b:
        print("k_bounce resources:", {k: b[0][k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size")})
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in b]
        print("k_bounce last frame per level (us):", [round(x, 1) for x in durs[-(len(durs) // max(1, len(durs) // 5)):][:8]])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        print(k, {a: b for a, b in c.items()})
