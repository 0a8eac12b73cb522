#!/usr/bin/env python3
"""Latency of ONE config-3 frame when it is split into P interleaved-tile parts that P contexts (own streams) render
concurrently into the same frame buffer (layout 0: no assemble).  usage: gpu_split_frame.py [cfg] [P ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import device, host, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sd, kw = scenes.config(cfg)
hs = host.HostScene(sd); desc = hs.flatten()
W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
cam = hs.camera(W, H)
ref = None
for P in [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 6, 8]:
    ctxs = [device.Context(0) for _ in range(P)]
    for c in ctxs: c.upload(desc)
    fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    prms = [device.make_params(W, H, MR, part_index=i, part_count=P) for i in range(P)]
    def frame():
        for c, p in zip(ctxs, prms): c.render_device(cam, p, fb.data_ptr(), sync=False)
        for c in ctxs: c.synchronize()
    for _ in range(6): frame()
    ts = []
    for _ in range(30):
        t = time.perf_counter(); frame(); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    cs = int(fb.to(torch.int64).sum().item())
    ref = cs if ref is None else ref
    print(f"cfg {cfg} parts={P}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  checksum {'same' if cs == ref else 'DIFFERENT'}", flush=True)
    for c in ctxs: c.close()
