#!/usr/bin/env python3
"""Per-frame traversal counters + per-level timings of config 3/2 on the GPU (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTHIP_COUNTERS"] = "1"
os.environ["RTHIP_PROFILE_LAUNCHES"] = "1"
import numpy as np
from raytracing_amd import device, host, scenes
import torch
for cfg, kw in ((3, {}), (3, {"cam_z": 2.2}), (2, {})):
    sd, _ = scenes.config(cfg, **kw)
    hs = host.HostScene(sd)
    ctx = device.Context(0); ctx.upload(hs.flatten())
    W, H = 1920, 1080
    cam = hs.camera(W, H)
    fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    for mr in (0, 3):
        prm = device.make_params(W, H, mr)
        for i in range(3):
            ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        s = ctx.stats()
        rays = s.rays_primary + s.rays_shadow + s.rays_reflect
        print(f"cfg{cfg}{kw} maxrec={mr}: ms_total={s.ms_total:.3f} ms_trace={s.ms_trace:.3f} launches={s.launches_trace} rays={rays} "
              f"(p={s.rays_primary} s={s.rays_shadow} r={s.rays_reflect}) Mrays/s={rays/s.ms_total/1e3:.1f}")
        print(f"   kd nodes={s.kd_nodes_visited} pops={s.kd_pops} leaves={s.kd_leaves_visited} tris={s.kd_tris_tested} shapes={s.shape_tests} mesh_hits={s.mesh_hits}"
              f"  per ray: nodes={s.kd_nodes_visited/rays:.2f} tris={s.kd_tris_tested/rays:.2f} shapes={s.shape_tests/rays:.2f}")
    ctx.close()
