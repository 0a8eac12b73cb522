#!/usr/bin/env python3
"""Not collected by pytest: the multi-GPU data path on random scenes, on one GPU.
usage: python tests/campaign_parts.py FIRST LAST
Every seed: a tests/fuzz_scenes scene, a random frame size (ragged against the tiles), a random part count (2..9), a
random batch of 1..16 jittered cameras, a random tile shape; every part renders its tile-compact buffers with
rthip_render_batch_device, rthip_assemble_batch puts the frames together, and every frame must equal the single-part,
single-frame render of its camera bit for bit; the parts' ray counts must add up to the frames'.  (The single-part
renders themselves are pinned against the reference by the other campaigns.)  Exit code 1 on any mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from fuzz_scenes import _Rng, random_scene
from raytracing_amd import device, host

first, last = int(sys.argv[1]), int(sys.argv[2])
ctx = device.Context(0)
bad, t0 = 0, time.time()
for seed in range(first, last + 1):
    r = _Rng(seed * 31 + 7)
    hs = host.HostScene(random_scene(seed))
    ctx.upload(hs.flatten())
    w, h = 40 + r.i(200), 24 + r.i(140)
    parts, nf, mr = 2 + r.i(8), 1 + r.i(16), r.i(5)
    tile_w, tile_h = (32, 8) if r.i(2) else (8 * (1 + r.i(6)), 8 * (1 + r.i(3)))
    aa = 1 if r.i(6) == 0 else 0
    cams = []
    for k in range(nf):
        cam = hs.camera(w, h)
        for a in range(3):
            cam.position[a] += r.u(-0.3, 0.3)
        cams.append(cam)
    full = device.make_params(w, h, mr, aa=aa, tile_w=tile_w, tile_h=tile_h)
    singles, rays = [], 0
    for cam in cams:
        singles.append(ctx.render(cam, full)["pixels"].copy())
        st = ctx.stats()
        rays += st.rays_primary + st.rays_shadow + st.rays_reflect
    pl = device.make_params(w, h, mr, aa=aa, part_count=parts, layout=1, tile_w=tile_w, tile_h=tile_h)
    n = device.part_pixels(pl)
    gathered = torch.zeros(parts * nf * n, dtype=torch.int32, device="cuda")
    part_rays = 0
    for p in range(parts):
        pp = device.make_params(w, h, mr, aa=aa, part_index=p, part_count=parts, layout=1, tile_w=tile_w, tile_h=tile_h)
        ctx.render_batch_device(cams, pp, gathered.data_ptr() + 4 * nf * n * p, sync=True)
        st = ctx.stats()
        part_rays += st.rays_primary + st.rays_shadow + st.rays_reflect
    frames = torch.zeros(nf * w * h, dtype=torch.int32, device="cuda")
    ctx.assemble_batch(pl, nf, gathered.data_ptr(), frames.data_ptr(), sync=True)
    got = frames.cpu().numpy().view(np.uint32).reshape(nf, h, w)
    wrong = [k for k in range(nf) if not np.array_equal(got[k], singles[k])]
    if wrong or part_rays != rays:
        bad += 1
        print(f"MISMATCH seed {seed}: {w}x{h} parts={parts} frames={nf} maxrec={mr} tile={tile_w}x{tile_h} aa={aa}: frames {wrong} differ, rays {part_rays} vs {rays}", flush=True)
    if (seed - first) % 20 == 19:
        print(f"  ... seed {seed} done, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"{last - first + 1} seeds, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
