#!/usr/bin/env python3
"""Per-level diagnostics on the GPU: rays per level, trace-kernel time per level, worst single-ray node count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTHIP_PROFILE_LAUNCHES"] = "1"
from raytracing_amd import device, host, scenes
import torch
cfgs = [int(a) for a in sys.argv[1:]] or [3, 4]
for cfg in cfgs:
    sd, kw = scenes.config(cfg)
    hs = host.HostScene(sd)
    desc = hs.flatten()
    W, H, mr = kw["width"], kw["height"], kw["max_recursion"]
    cam = hs.camera(W, H)
    fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    for counters, mrr in [("0", mr), ("0", 0), ("1", mr), ("2", mr)]:
        prm = device.make_params(W, H, mrr)
        os.environ["RTHIP_COUNTERS"] = counters
        ctx = device.Context(0); ctx.upload(desc)
        for i in range(3):
            ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        s = ctx.stats()
        print(f"cfg{cfg} maxrec={mrr} counters={counters} ms_total={s.ms_total:.3f} ms_trace={s.ms_trace:.3f} max_nodes_ray={s.kd_max_nodes_ray} "
              f"nodes={s.kd_nodes_visited} tris={s.kd_tris_tested} leaves={s.kd_leaves_visited} pops={s.kd_pops}\n"
              f"   wave steps max={s.wave_steps_max} total={s.wave_steps}  wave tri-iters max={s.wave_tris_max} total={s.wave_tris}  longest wave {s.wave_cycles_max} cycles; node-fetch wait {s.fetch_cycles} cycles = {s.fetch_cycles / max(1, s.wave_steps):.0f}/step")
        print("   level rays:", list(s.level_rays[:mr + 1]))
        print("   easy ms   :", [round(x, 3) for x in s.level_ms_easy[:mr + 1]], " hard rays:", s.rays_hard)
        print("   quad ms   :", [round(x, 3) for x in s.level_ms_trace[:mr + 1]])
        print("   shade ms  :", [round(x, 3) for x in s.level_ms_shade[:mr + 1]], flush=True)
        ctx.close()
