#!/usr/bin/env python3
"""Per-part ray balance of the interleaved-tile partition (">= 7x at 8 GPUs requires ... per-GPU ray counts within
a few %"): every part of N = 2, 4, 8 rendered on this one GPU (rthip_params.part_index / part_count), its rays (primary + shadow +
reflected) and device time against the mean of the N parts.  usage: gpu_part_balance.py [config ...]   (default 3 4 5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import device, host, scenes
for cfg in [int(a) for a in sys.argv[1:]] or [3, 4, 5]:
    sd, kw = scenes.config(cfg)
    W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
    hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
    cam = hs.camera(W, H)
    for N in (2, 4, 8):
        rays, ms = [], []
        for p in range(N):
            prm = device.make_params(W, H, MR, part_index=p, part_count=N, layout=1)
            fb = torch.zeros(device.part_pixels(prm), dtype=torch.int32, device="cuda")
            for _ in range(2):
                ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
            st = ctx.stats()
            rays.append(int(st.rays_primary + st.rays_shadow + st.rays_reflect)); ms.append(st.ms_total)
        mean = sum(rays) / N
        dev = max(abs(r - mean) for r in rays) / mean
        print(f"config {cfg} {W}x{H} N={N}: rays per part {rays} | worst deviation from the mean {100 * dev:.2f} % | slowest / mean device time "
              f"{max(ms) / (sum(ms) / N):.3f} ({', '.join(f'{m:.3f}' for m in ms)} ms)", flush=True)
    ctx.close(); del hs
