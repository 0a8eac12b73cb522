#!/usr/bin/env python3
"""Per-level kernel times (RTHIP_PROFILE_LAUNCHES=1), re-traced ray count and, with RTHIP_COUNTERS=1, traversal counters of the
interrupt test's frame (config 4 height field of `cells` x `cells`, 4096 x 4096, MaxRecursion 16).  usage: gpu_irq_probe2.py [cells]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTHIP_PROFILE_LAUNCHES"] = "1"
import torch
from raytracing_amd import device, host, scenes
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 384
hs = host.HostScene(scenes.config(4, cells=cells)[0])
ctx = device.Context(0); ctx.upload(hs.flatten())
w = h = 4096
cam, prm = hs.camera(w, h), device.make_params(w, h, 16)
fb = torch.zeros(w * h, dtype=torch.int32, device="cuda")
for _ in range(2): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
s = ctx.stats()
print("ms_total", s.ms_total, "hard", s.rays_hard, "redo", getattr(s, "rays_redo", -1), "nodes", s.kd_nodes_visited, "tris", s.kd_tris_tested, "pops", s.kd_pops, "max nodes of a ray", s.kd_max_nodes_ray)
for l in range(6):
    print(l, "rays", s.level_rays[l], "quad", round(s.level_ms_trace[l], 3), "easy", round(s.level_ms_easy[l], 3), "redo+shade", round(s.level_ms_shade[l], 3))
