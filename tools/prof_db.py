#!/usr/bin/env python3
"""Prints the per-kernel totals of a rocprofv3 results database (rocpd sqlite): name, calls, total ms, average us, percent."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for name, calls, total, avg, pct in db.execute("select * from top_kernels limit 14"):
    print(f"{name.split('(')[0][:60]:60s} calls {calls:5d} total {total / 1e3:9.3f} ms ({total / 1e3 / frames:7.3f} /frame) avg {avg:9.1f} us {pct:5.1f} %")
