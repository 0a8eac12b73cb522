#!/usr/bin/env python3
"""Per-kernel summary of the PMC passes of tools/prof_passes.sh (one directory per pass) as markdown:
VALU utilisation, wait fractions, L2 hit rate, HBM read traffic, against the kernel's GPU time."""
import collections, csv, glob, sys
d, frames = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for f in glob.glob(d + "/*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[k] += 1
print(f"PMC summary of `{d}` ({frames} frames; rocprofv3 --pmc, one pass per counter group, MI355X)\n")
print("| kernel | launches/frame | waves/frame | VALU instr/frame (M) | VALU issue util. of the SIMDs | active lanes / VALU instr | wave time in s_waitcnt | issue-stalled | L2 hit rate | HBM/MALL read MB/frame (RDREQ x 128 B) | HBM MB/frame (FETCH_SIZE x 2 + WRITE_SIZE) | GPU ms/frame |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
import json
summary = {}
for k, c in sorted(acc.items()):
    if not k.startswith("k_") or c.get("SQ_WAVES", 0) == 0:
        continue
    wc = c["SQ_WAVE_CYCLES"]
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 2.4 GHz nominal
    ms = c.get("GRBM_GUI_ACTIVE", 0) / 8 / 2.4e6 / frames
    simd_cycles = c.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024
    valu_util = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles if simd_cycles else 0
    req = c.get("TCC_REQ_sum", 0)
    lanes = c.get("SQ_THREAD_CYCLES_VALU", 0) / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else 0.0
    # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE counts a 128-byte request as 64 bytes on gfx950 (MI355X_MICROARCH.md): doubled
    hbm = (c.get("FETCH_SIZE", 0) * 2 + c.get("WRITE_SIZE", 0)) * 1024 / frames if ("FETCH_SIZE" in c or "WRITE_SIZE" in c) else None
    print(f"| `{k}` | {calls[k] / frames:.2f} | {c['SQ_WAVES'] / frames:,.0f} | {c['SQ_INSTS_VALU'] / frames / 1e6:.1f} | {100 * valu_util:.0f} % | {lanes:.0f} | "
          f"{100 * c['SQ_WAIT_ANY'] / wc:.0f} % | {100 * c['SQ_WAIT_INST_ANY'] / wc:.0f} % | {100 * c.get('TCC_HIT_sum', 0) / req if req else 0:.0f} % | "
          f"{c.get('TCC_EA0_RDREQ_sum', 0) * 128 / frames / 1e6:.1f} | {(hbm or 0) / 1e6:.1f} | {ms:.3f} |")
    summary[k] = {"valu_issue_util": valu_util, "active_lanes_per_valu_inst": lanes, "wave_time_waiting": c["SQ_WAIT_ANY"] / wc,
                  "l2_hit_rate": c.get("TCC_HIT_sum", 0) / req if req else None,
                  "valu_instructions_per_frame": c["SQ_INSTS_VALU"] / frames, "gpu_ms_per_frame_serialised": ms,
                  "hbm_bytes_per_frame": hbm, "hbm_fetch_bytes_per_frame": c.get("FETCH_SIZE", 0) * 2 * 1024 / frames,
                  "hbm_write_bytes_per_frame": c.get("WRITE_SIZE", 0) * 1024 / frames}
if len(sys.argv) > 3:  # machine-readable copy (bench.py quotes the dominant kernel's VALU figures from it)
    import os
    json.dump({"source": "rocprofv3 --pmc passes of tools/prof_passes.sh (" + os.path.basename(d.rstrip("/")) + "), summarised by tools/prof_summary.py",
               "source_file": "profiles/" + os.path.basename(sys.argv[3]), "frames": frames, "kernels": summary}, open(sys.argv[3], "w"), indent=1)
print("\nVALU issue utilisation = 4 cycles x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles); HBM/MALL read = TCC_EA0_RDREQ x 128 B "
      "(gfx950 tallies 128-byte requests, MI355X_MICROARCH.md); GPU ms from GRBM_GUI_ACTIVE at 2.4 GHz (PMC runs serialise kernels).")
