#!/usr/bin/env python3
"""The longest-lived k_trace_quad wave of a frame (RTHIP_COUNTERS=2: shader-clock cycles from the wave's start to its end, its node
steps and leaf iterations, the cycles it waited for node fetches), at several frame sizes and recursion limits: is a small launch's
duration the dependent chain of its slowest wave?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTHIP_COUNTERS"] = "2"
os.environ["RTHIP_PROFILE_LAUNCHES"] = "1"
import torch
from raytracing_amd import device, host, scenes
sd, kw = scenes.config(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
for W, H in ((1920, 1080), (240, 135), (64, 64)):
    fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    cam = hs.camera(W, H)
    for mr in (0, 1, 3):
        prm = device.make_params(W, H, mr)
        for _ in range(3): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        s = ctx.stats()
        print(f"{W}x{H} MaxRecursion {mr}: hard {s.rays_hard:7d}  walk kernels {1e3 * s.ms_trace:6.1f} us ({', '.join(f'{1e3 * s.level_ms_trace[l]:.1f}' for l in range(mr + 1))})  "
              f"longest wave {s.wave_cycles_max} cycles = {s.wave_cycles_max / 2400:.1f} us at 2.4 GHz | {s.wave_cycles_max / 100:.1f} us at 100 MHz; its node steps {s.wave_steps_max}, leaf iterations {s.wave_tris_max}, "
              f"longest ray {s.kd_max_nodes_ray} nodes, cycles waiting for node fetches {s.fetch_cycles}", flush=True)
