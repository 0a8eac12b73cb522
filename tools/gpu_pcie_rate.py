#!/usr/bin/env python3
"""PCIe-inclusive frame rate of config 3: rthip_render (the boundary that hands a HOST pixel buffer back, as
AbstractRaycaster::RenderSync does) against rthip_render_device (output left in HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import device, host, scenes
sd, kw = scenes.config(3)
hs = host.HostScene(sd); desc = hs.flatten()
W, H, mr = kw["width"], kw["height"], kw["max_recursion"]
cam = hs.camera(W, H); prm = device.make_params(W, H, mr)
ctx = device.Context(0); ctx.upload(desc)
out = np.zeros((H, W), np.uint32)
for _ in range(5): ctx.render(cam, prm, out=out)
rays = ctx.stats(); rays = rays.rays_primary + rays.rays_shadow + rays.rays_reflect
t0 = time.perf_counter(); n = 100
for _ in range(n): ctx.render(cam, prm, out=out)
host_ms = (time.perf_counter() - t0) / n * 1e3
pinned = ctx.host_pixels(W, H)
for _ in range(5): ctx.render(cam, prm, out=pinned)
t0 = time.perf_counter()
for _ in range(n): ctx.render(cam, prm, out=pinned)
pin_ms = (time.perf_counter() - t0) / n * 1e3
assert np.array_equal(pinned, out)
ctx.host_free(pinned)
fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
for _ in range(5): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
t0 = time.perf_counter()
for _ in range(n): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
dev_ms = (time.perf_counter() - t0) / n * 1e3
print(f"config 3, one frame at a time: host buffer out {host_ms:.3f} ms/frame = {rays/host_ms/1e3:.0f} Mrays/s (PCIe-inclusive), page-locked {pin_ms:.3f} ms = {rays/pin_ms/1e3:.0f} Mrays/s; "
      f"HBM-resident {dev_ms:.3f} ms/frame = {rays/dev_ms/1e3:.0f} Mrays/s")
