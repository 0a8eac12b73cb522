#!/usr/bin/env python3
"""Host-side cost of enqueueing one frame: tiny frames (GPU work negligible), several contexts in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import device, host, scenes
sd, _ = scenes.config(3)
hs = host.HostScene(sd)
desc = hs.flatten()
for (W, H) in ((64, 32), (480, 270), (960, 540)):
    cam = hs.camera(W, H)
    prm = device.make_params(W, H, 3)
    for F in (1, 3):
        ctxs = [device.Context(0) for _ in range(F)]
        fbs = [torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(F)]
        for c in ctxs:
            c.upload(desc)
        for k in range(30):
            ctxs[k % F].render_device(cam, prm, fbs[k % F].data_ptr(), sync=False)
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        n = 600
        for k in range(n):
            ctxs[k % F].render_device(cam, prm, fbs[k % F].data_ptr(), sync=False)
        t1 = time.perf_counter()
        for c in ctxs:
            c.synchronize()
        t2 = time.perf_counter()
        print(f"{W}x{H} inflight {F}: enqueue {1e3*(t1-t0)/n:.3f} ms/frame, total {1e3*(t2-t0)/n:.3f} ms/frame", flush=True)
        for c in ctxs:
            c.close()
