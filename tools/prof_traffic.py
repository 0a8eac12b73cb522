#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from the two PMC passes of tools/prof_traffic.sh.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in kilobytes; on gfx950 FETCH_SIZE counts 128-byte requests as 64
bytes (MI355X_MICROARCH.md, section HBM), so it is doubled.  Writes traffic.json next to the passes."""
import collections, csv, glob, json, sys
d = sys.argv[1]
def per_kernel(sub, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}
fetch, rd = per_kernel("fetch", "FETCH_SIZE"), per_kernel("fetch", "TCC_EA0_RDREQ_sum")
write, wr = per_kernel("write", "WRITE_SIZE"), per_kernel("write", "TCC_EA0_WRREQ_sum")
out = {"method": "rocprofv3 --pmc, separate passes; bytes = FETCH_SIZE[KB]*1024*2 (gfx950 correction) + WRITE_SIZE[KB]*1024; per launch = mean over dispatches", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    fb = fetch.get(k, (0, 0))[0] * 1024 * 2
    wb = write.get(k, (0, 0))[0] * 1024
    out["kernels"][k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "launches_seen": fetch.get(k, (0, 0))[1],
                         "rdreq_per_launch": rd.get(k, (0, 0))[0], "wrreq_per_launch": wr.get(k, (0, 0))[0]}
    print(f"{k:28s} fetch {fb/1e6:9.2f} MB  write {wb/1e6:9.2f} MB per launch  ({fetch.get(k,(0,0))[1]} launches; rdreq {rd.get(k,(0,0))[0]:.0f} wrreq {wr.get(k,(0,0))[0]:.0f})")
q = [v for k, v in out["kernels"].items() if k.startswith("k_trace_quad")]
if q:
    out["k_trace_quad_bytes_per_launch"] = q[0]["fetch_bytes_per_launch"] + q[0]["write_bytes_per_launch"]
json.dump(out, open(f"{d}/traffic.json", "w"), indent=1)
