#!/usr/bin/env python3
"""Start-up of a config with its BVHs built on the device: flatten, upload + device build (three times), node count and depth.
Under `rocprofv3 --kernel-trace --stats` the kernel table shows where the build's device time goes.
usage: gpu_bvh_build_time.py [cfg]   (RTHIP_BVH_BUILDER=ploc for the clustering builder)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RE_BVH_DEVICE_MIN_TRIS"] = "1"
from raytracing_amd import device, host, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd, kw = scenes.config(cfg)
t0 = time.perf_counter()
hs = host.HostScene(sd); desc = hs.flatten()
t1 = time.perf_counter()
print(f"config {cfg}: compile + flatten {t1 - t0:.3f} s", flush=True)
for k in range(3):
    ctx = device.Context(0)
    t = time.perf_counter(); ctx.upload(desc); ctx.synchronize(); dt = time.perf_counter() - t
    m, nodes, _ = ctx.bvh(0)
    print(f"  upload + device build {dt * 1e3:.1f} ms, mesh 0: {len(nodes)} nodes, depth {m.bvh_depth}", flush=True)
    ctx.close()
