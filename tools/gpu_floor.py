#!/usr/bin/env python3
"""Launch floors: config 3 (or the config given) rendered as lone frames at shrinking sizes; per level the rays, the hard rays and the
duration of every kernel (RTHIP_PROFILE_LAUNCHES=1: HIP events around plain launches), so that the part of a kernel's time that does not
go away with its rays shows; then the same frames through the replayed graph (device-event time of the whole frame)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
import torch
from raytracing_amd import device, host, scenes
sd, kw = scenes.config(cfg)
MR = kw["max_recursion"]
hs = host.HostScene(sd)
flat = hs.flatten()
sizes = [(1920, 1080), (960, 540), (480, 270), (240, 135), (64, 64)]
for profile in ("1", "0"):
    os.environ["RTHIP_PROFILE_LAUNCHES"] = profile
    ctx = device.Context(0); ctx.upload(flat)
    for W, H in sizes:
        fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        cam = hs.camera(W, H)
        prm = device.make_params(W, H, MR)
        ms = []
        for _ in range(6):
            ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
            ms.append(ctx.stats().ms_total)
        s = ctx.stats()
        if profile == "1":
            print(f"{W}x{H} plain launches with events: frame {sorted(ms)[len(ms) // 2]:.3f} ms, walk {s.ms_trace:.3f} easy {s.ms_easy:.3f}")
            for l in range(MR + 1):
                print(f"   level {l}: rays {s.level_rays[l]:8d}  walk {1e3 * s.level_ms_trace[l]:6.1f} us  easy {1e3 * s.level_ms_easy[l]:6.1f} us  shade {1e3 * s.level_ms_shade[l]:6.1f} us")
        else:
            print(f"{W}x{H} replayed graph: frame {sorted(ms)[len(ms) // 2]:.3f} ms (min {min(ms):.3f}) hard {s.rays_hard} replays {s.graph_replays}", flush=True)
    ctx.close()
