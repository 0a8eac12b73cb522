#!/usr/bin/env python3
"""Small target for rocprofv3: K frames of one config through the C-ABI, or (third argument B > 1) K calls of
B-frame batches left in HBM: the shape of work `bench.py` times."""
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
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if batch > 1:
    import torch
    out = torch.empty((batch, H, W), dtype=torch.int32, device="cuda:0")
    cams = [cam] * batch
    for i in range(frames):
        ctx.render_batch_device(cams, prm, out.data_ptr(), sync=True)
else:
    for i in range(frames):
        r = ctx.render(cam, prm)
s = ctx.stats()
print("cfg", cfg, "ms_total", s.ms_total, "rays", s.rays_primary + s.rays_shadow + s.rays_reflect)
