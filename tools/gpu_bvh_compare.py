#!/usr/bin/env python3
"""Host-built (binned SAH, threads) against device-built (RTHIP_BVH_BUILDER: binned SAH by default, ploc) 4-wide BVH: scene start-up time and frame time.
usage: gpu_bvh_compare.py [cfg ...]   (default 4 3)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import device, host, scenes
for cfg in [int(a) for a in sys.argv[1:]] or [4, 3]:
    sd, kw = scenes.config(cfg)
    W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
    for mode, env in (("host SAH", "0"), ("device " + os.environ.get("RTHIP_BVH_BUILDER", "sah"), "1")):
        os.environ["RE_BVH_DEVICE_MIN_TRIS"] = env
        t0 = time.perf_counter()
        hs = host.HostScene(sd); desc = hs.flatten()
        t1 = time.perf_counter()
        ctx = device.Context(0); ctx.upload(desc); ctx.synchronize()
        t2 = time.perf_counter()
        m, nodes, _ = ctx.bvh(0)
        cam, prm = hs.camera(W, H), device.make_params(W, H, MR)
        fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        for _ in range(3): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); ctx.render_device(cam, prm, fb.data_ptr(), sync=True); ts.append((time.perf_counter() - t) * 1e3)
        ts.sort()
        print(f"config {cfg} {mode:12s}: compile+flatten {t1 - t0:.3f} s, upload (+ device build) {t2 - t1:.3f} s, {len(nodes)} nodes depth {m.bvh_depth}, "
              f"frame {ts[len(ts) // 2]:.3f} ms, checksum {int(fb.to(torch.int64).sum().item())}", flush=True)
        ctx.close()
