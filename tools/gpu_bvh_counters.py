#!/usr/bin/env python3
"""Traversal counters of a config's lone frame with the host-built and the device-built BVH (RTHIP_COUNTERS=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTHIP_COUNTERS"] = "1"
import torch
from raytracing_amd import device, host, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd, kw = scenes.config(cfg)
W, H, MR = kw["width"], kw["height"], kw["max_recursion"]
for mode, env in (("host", "0"), ("device", "1")):
    os.environ["RE_BVH_DEVICE_MIN_TRIS"] = env
    hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
    m, nodes, _ = ctx.bvh(0)
    ch = nodes["c"]["child"].ravel()
    leaf = ch[(ch & 0x80000000) != 0]; leaf = leaf[leaf != 0xffffffff]
    cnt = ((leaf >> 28) & 7) + 1
    kids = (nodes["c"]["child"] != 0xffffffff).sum(axis=1)
    fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    cam, prm = hs.camera(W, H), device.make_params(W, H, MR)
    for _ in range(2): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
    s = ctx.stats()
    print(f"cfg {cfg} {mode:6s}: nodes {len(nodes)} depth {m.bvh_depth} leaves {len(leaf)} tris/leaf {cnt.mean():.2f} children/node {kids.mean():.2f} | "
          f"hard rays {s.rays_hard} steps {s.kd_nodes_visited} ({s.kd_nodes_visited / max(1, s.rays_hard):.1f}/ray) pops {s.kd_pops} leaves visited {s.kd_leaves_visited} "
          f"tris {s.kd_tris_tested} ({s.kd_tris_tested / max(1, s.rays_hard):.1f}/ray) wave steps {s.wave_steps} wave tris {s.wave_tris} ms {s.ms_total:.2f}", flush=True)
    ctx.close()
