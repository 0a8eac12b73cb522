#!/usr/bin/env python3
"""Small target for rocprofv3: K frames of one config through the C-ABI (no torch import)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import device, host, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sd, kw = scenes.config(cfg)
hs = host.HostScene(sd)
ctx = device.Context(0); ctx.upload(hs.flatten())
W, H, mr = kw["width"], kw["height"], kw["max_recursion"]
cam = hs.camera(W, H)
prm = device.make_params(W, H, mr)
for i in range(frames):
    r = ctx.render(cam, prm)
s = ctx.stats()
print("cfg", cfg, "ms_total", s.ms_total, "rays", s.rays_primary + s.rays_shadow + s.rays_reflect)
