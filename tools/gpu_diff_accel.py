#!/usr/bin/env python3
"""GPU box: config 4 (2 M triangles, 4K) rendered through both mesh paths - accel = 0, the 4-wide BVH, and accel = 1,
the reference's kd-tree - and the two frames compared pixel by pixel and id by id (they must be identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import device, host, scenes
sd, kw = scenes.config(4)
hs = host.HostScene(sd)
ctx = device.Context(0); ctx.upload(hs.flatten())
W, H, mr = kw["width"], kw["height"], kw["max_recursion"]
cam = hs.camera(W, H)
a = ctx.render(cam, device.make_params(W, H, mr, accel=0), ids=True)
b = ctx.render(cam, device.make_params(W, H, mr, accel=1), ids=True)
dp = a["pixels"] != b["pixels"]; di = (a["shape"] != b["shape"]) | (a["tri"] != b["tri"])
print("pixel diffs", dp.sum(), "id diffs", di.sum(), "of", W * H)
ys, xs = np.nonzero(dp | di)
sel = np.arange(len(xs))[:: max(1, len(xs) // 400)]
np.savez("gpurun_out/cfg4_diff.npz", x=xs[sel], y=ys[sel], bvh_px=a["pixels"][ys[sel], xs[sel]], kd_px=b["pixels"][ys[sel], xs[sel]],
         bvh_shape=a["shape"][ys[sel], xs[sel]], kd_shape=b["shape"][ys[sel], xs[sel]], bvh_tri=a["tri"][ys[sel], xs[sel]], kd_tri=b["tri"][ys[sel], xs[sel]])
print(list(zip(xs[:10], ys[:10])))
