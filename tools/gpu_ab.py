#!/usr/bin/env python3
"""A/B timing of a librthip build (RTHIP_SO=path): median device time of lone frames of configs 4 and 3 and of an
8-frame batch of config 3, with a checksum of the pixels (must not change between builds).
usage: RTHIP_SO=build_ab/x.so python tools/gpu_ab.py [label]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RE_BVH_DEVICE_MIN_TRIS", "0")
import torch
from raytracing_amd import device, host, scenes
label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("RTHIP_SO", "default")
cfgs = [int(c) for c in os.environ.get("AB_CONFIGS", "4,3").split(",")]
out = [label]
for cfg in cfgs:
    sd, kw = scenes.config(cfg)
    W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
    hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
    cam, prm = hs.camera(W, H), device.make_params(W, H, MR)
    for B in ((1, 8) if cfg == 3 else (1,)):
        fb = torch.zeros(B * W * H, dtype=torch.int32, device="cuda")
        ms = []
        for k in range(9):
            if B == 1: ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
            else: ctx.render_batch_device([cam] * B, prm, fb.data_ptr(), sync=True)
            if k >= 2: ms.append(ctx.stats().ms_total / B)
        out.append(f"cfg{cfg} x{B}: {statistics.median(ms):.3f} ms/frame (sum {int(fb.to(torch.int64).sum())})")
    ctx.close()
print(" | ".join(out), flush=True)
