#!/usr/bin/env python3
"""target for rocprofv3 --kernel-trace: K lone frames of a config left in HBM (graph replay after the second)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import device, host, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sd, kw = scenes.config(cfg)
hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
cam, prm = hs.camera(W, H), device.make_params(W, H, MR)
fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
for _ in range(K): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
print("done", ctx.stats().ms_total)
