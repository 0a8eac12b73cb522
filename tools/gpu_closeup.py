#!/usr/bin/env python3
"""Lone frames of BASELINE config 3 from the close-up camera (0,0,2.2) - the bunny fills the view, ~1.5 M rays per level need the walk - and a
lone call of 8 frames from the scene's own camera: median device time (HIP events around the call's launches) and the lists' lengths.
usage: [RTHIP_... env] python tools/gpu_closeup.py [label]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import device, host, scenes
label = sys.argv[1] if len(sys.argv) > 1 else ""
out = [label]
for name, cam_z, B in (("close-up x1", 2.2, 1), ("camera (0,0,5) x8", None, 8), ("close-up x4", 2.2, 4)):
    sd, kw = scenes.config(3, cam_z=cam_z) if cam_z else scenes.config(3)
    W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
    hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
    cam, prm = hs.camera(W, H), device.make_params(W, H, MR)
    fb = torch.zeros(B * W * H, dtype=torch.int32, device="cuda")
    ms = []
    for k in range(11):
        if B == 1: ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        else: ctx.render_batch_device([cam] * B, prm, fb.data_ptr(), sync=True)
        if k >= 3: ms.append(ctx.stats().ms_total / B)
    st = ctx.stats()
    out.append(f"{name}: {statistics.median(ms):.3f} ms/frame (hard rays {st.rays_hard // B} per frame, sum {int(fb.to(torch.int64).sum())})")
    ctx.close()
print(" | ".join(out), flush=True)
