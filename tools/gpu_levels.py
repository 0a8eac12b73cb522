#!/usr/bin/env python3
"""Per recursion level of a config's lone frame: rays in the queue, duration of the mesh-walk kernel, of k_trace_easy and
of the shade kernels (RTHIP_PROFILE_LAUNCHES=1), and with RTHIP_COUNTERS=1 the walk's step / wave-step counters per level
(one frame rendered per MaxRecursion 0..MR, counters differenced)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
counters = len(sys.argv) > 2 and sys.argv[2] == "counters"
os.environ["RTHIP_PROFILE_LAUNCHES"] = "1"
if counters: os.environ["RTHIP_COUNTERS"] = "1"
os.environ.setdefault("RE_BVH_DEVICE_MIN_TRIS", "0")
import torch
from raytracing_amd import device, host, scenes
sd, kw = scenes.config(cfg)
W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
cam = hs.camera(W, H)
prev = None
for mr in (range(MR + 1) if counters else (MR,)):
    prm = device.make_params(W, H, mr)
    for _ in range(2): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
    s = ctx.stats()
    if not counters:
        print(f"cfg {cfg}: ms_total {s.ms_total:.3f} trace {s.ms_trace:.3f} easy {s.ms_easy:.3f} hard {s.rays_hard} checksum {int(fb.to(torch.int64).sum())}")
        for l in range(mr + 1):
            print(f"  level {l}: rays {s.level_rays[l]:9d} walk {s.level_ms_trace[l]:.3f} ms easy {s.level_ms_easy[l]:.3f} ms shade+redo {s.level_ms_shade[l]:.3f} ms")
    else:
        cur = dict(hard=s.rays_hard, steps=s.kd_nodes_visited, wsteps=s.wave_steps, leaves=s.kd_leaves_visited, wtris=s.wave_tris, tris=s.kd_tris_tested)
        d = {k: cur[k] - (prev[k] if prev else 0) for k in cur}
        print(f"  level {mr}: hard {d['hard']:9d} steps {d['steps']:10d} ({d['steps'] / max(1, d['hard']):.1f}/ray) wave steps {d['wsteps']:9d} "
              f"(quads active per wave step {d['steps'] / max(1, d['wsteps']):.2f}/16) leaves {d['leaves']} wave leaf iterations {d['wtris']} "
              f"({d['leaves'] / max(1, d['wtris']):.2f}/16) tris {d['tris']}", flush=True)
        prev = cur
