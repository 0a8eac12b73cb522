#!/usr/bin/env python3
"""The longest-lived wave of the mesh-walk kernel over a config's lone frame (RTHIP_COUNTERS=2): its lifetime in shader
clocks, the cycles it waited for BVH node fetches, its node steps and leaf steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTHIP_COUNTERS"] = "2"
os.environ.setdefault("RE_BVH_DEVICE_MIN_TRIS", "0")
import torch
from raytracing_amd import device, host, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sd, kw = scenes.config(cfg)
W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
cam = hs.camera(W, H)
for mr in range(MR + 1):
    prm = device.make_params(W, H, mr)
    for _ in range(2): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
    s = ctx.stats()
    print(f"maxrec {mr}: longest wave {s.wave_cycles_max} cycles, waited for node fetches {s.fetch_cycles} ({100.0 * s.fetch_cycles / max(1, s.wave_cycles_max):.0f} %), "
          f"node steps {s.wave_steps}, leaf steps {s.wave_tris_max}, most nodes of one ray {s.kd_max_nodes_ray}, frame {s.ms_total:.3f} ms", flush=True)
