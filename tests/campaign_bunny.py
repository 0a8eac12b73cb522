#!/usr/bin/env python3
"""Not collected by pytest: BASELINE config 3 (Demo 4 bunny) from random viewpoints, GPU against the genuine reference.
usage: python tests/campaign_bunny.py FIRST LAST [WIDTH HEIGHT MAXREC]
Every seed: a camera on a random orbit (radius 0.6 - 9 around the bunny, sometimes below the plane or inside the
mesh's box) looking at a jittered point of the model; ids, pixels, ray counts bit for bit (reference kd-tree path
every fourth seed, SSAA every eighth).  Exit code 1 on any mismatch."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from fuzz_scenes import _Rng
from raytracing_amd import host, scenes

first, last = int(sys.argv[1]), int(sys.argv[2])
W, H, MAXREC = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (480, 270, 3)
lib = oracle.load_ref() if oracle.have_ref() else oracle.load_oracle()
print("checker:", "genuine reference" if oracle.have_ref() else "C restatement", f"{W}x{H} maxrec {MAXREC}", flush=True)
bad, hits, t0 = 0, 0, time.time()
for seed in range(first, last + 1):
    r = _Rng(seed)
    sd, _ = scenes.config(3)
    a, e, rad = r.u(0, 2 * math.pi), r.u(-0.3, 1.4), r.u(0.6, 9.0)
    pos = (rad * math.cos(a) * math.cos(e), -0.5 + rad * math.sin(e), rad * math.sin(a) * math.cos(e))
    sd.look_at(pos, (r.u(-0.5, 0.5), r.u(-1.0, 0.2), r.u(-0.5, 0.5)))
    for aa in ((0, 1) if seed % 8 == 0 else (0,)):
        ref = sd.build(lib).render(W, H, ids=True, threads=16, max_recursion=MAXREC, aa=aa)
        hits += int((ref["tri"] >= 0).sum())
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
        print(f"  ... seed {seed} done, {bad} mismatches, {hits} primary bunny hits so far, {time.time() - t0:.0f} s", flush=True)
print(f"{last - first + 1} viewpoints, {bad} mismatches, {hits} primary rays hit the bunny, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
