#!/usr/bin/env python3
"""Not collected by pytest: a long randomized parity run on the GPU box.
usage: python tests/campaign_fuzz.py FIRST LAST [WIDTH HEIGHT MAXREC]
Every seed: tests/fuzz_scenes.random_scene rendered by re::HipRaytracer (BVH path; the reference kd-tree path every
fourth seed; SSAA every eighth) (odd seeds: BVHs built on the device) against the genuine reference when oracle/_ref is built, else the C restatement: ids,
pixels and ray counts bit for bit.  Prints one line per mismatch and a progress line every 10 seeds; exit code 1 on
any mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from fuzz_scenes import random_scene
from raytracing_amd import host

first, last = int(sys.argv[1]), int(sys.argv[2])
W, H, MAXREC = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (160, 112, 5)
lib = oracle.load_ref() if oracle.have_ref() else oracle.load_oracle()
print("checker:", "genuine reference" if oracle.have_ref() else "C restatement", f"{W}x{H} maxrec {MAXREC}", flush=True)
bad = 0
t0 = time.time()
for seed in range(first, last + 1):
    sd = random_scene(seed)
    os.environ["RE_BVH_DEVICE_MIN_TRIS"] = "1" if seed % 2 else "0"   # odd seeds: every mesh's BVH is built on the device
    for aa in ((0, 1) if seed % 8 == 0 else (0,)):
        ref = sd.build(lib).render(W, H, ids=True, threads=16, max_recursion=MAXREC, aa=aa)
        for use_kd in ((False, True) if seed % 4 == 0 else (False,)):
            hs = host.HostScene(sd)
            hs.set_accel(use_kd)
            got = hs.render(W, H, ids=True, max_recursion=MAXREC, aa=aa)
            ids_ok = np.array_equal(got["shape"], ref["shape"]) and np.array_equal(got["tri"], ref["tri"])
            npx = int((got["pixels"] != ref["pixels"]).sum())
            if not ids_ok or npx or got["rays"] != ref["rays"]:
                bad += 1
                print(f"MISMATCH seed {seed} kd={use_kd} aa={aa}: ids_ok={ids_ok} pixels_differ={npx} rays {got['rays']} vs {ref['rays']}", flush=True)
    if (seed - first) % 10 == 9:
        print(f"  ... seed {seed} done, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"{last - first + 1} seeds, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
