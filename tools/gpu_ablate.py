#!/usr/bin/env python3
"""Where does config 3's frame time go?  Scene ablations (diagnostic, not a benchmark)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTHIP_PROFILE_LAUNCHES"] = "1"
import numpy as np
from raytracing_amd import device, host, scenes

def variant(kind):
    sd = scenes.SceneDesc(kind)
    sd.camera((0, 0, 5)); sd.position(0, -1, 0)
    black = sd.uniform_material(0, 0.7); white = sd.uniform_material(0xffffff, 0.7)
    if "noplane" not in kind:
        if "uniplane" in kind: sd.plane(white, 0, 1, 0)
        else: sd.plane(sd.interpolated_material(sd.checker(1), black, white), 0, 1, 0)
    bm = sd.uniform_material(0xff2222, 0.5)
    if "nomesh" not in kind:
        v, n = scenes.load_bunny(); sd.mesh(bm, v, n, scenes.NORMAL_VERTEX)
    sun = sd.directional_light(0xffcccc, 1, 1, 1)
    if "colorbg" in kind: sd.color_background(0x3050ff)
    else: sd.skybox(0xff0000, 0x0000ff, sun, seed=301)
    return sd

W, H = 1920, 1080
KINDS = sys.argv[1].split(",") if len(sys.argv) > 1 else ("full", "colorbg", "nomesh", "nomesh colorbg", "nomesh colorbg uniplane", "noplane", "noplane colorbg")
for kind in KINDS:
    hs = host.HostScene(variant(kind))
    ctx = device.Context(0); ctx.upload(hs.flatten())
    cam = hs.camera(W, H)
    for mr in (0, 3):
        prm = device.make_params(W, H, mr)
        best = 1e9
        for i in range(4):
            ctx.render(cam, prm); best = min(best, ctx.stats().ms_total)
        s = ctx.stats()
        print(f"{kind:28s} maxrec {mr}: {best:.3f} ms  rays p/s/r {s.rays_primary}/{s.rays_shadow}/{s.rays_reflect}", flush=True)
    ctx.close()
