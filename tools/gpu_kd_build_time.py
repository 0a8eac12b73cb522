#!/usr/bin/env python3
"""Start-up of config 4 (2 097 152 triangles) with the reference kd-tree built on the device vs on host threads: Scene::Compile +
flatten, upload (+ device builds), the device time of the kd build, and a byte comparison of the two trees.
usage: gpu_kd_build_time.py [cells]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import device, host, scenes
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sd, _ = scenes.config(4, cells=cells)
trees = {}
for name, env in (("device", "1"), ("host", "0"), ("device again", "1")):
    os.environ["RE_KD_DEVICE_MIN_TRIS"] = env
    os.environ["RE_BVH_DEVICE_MIN_TRIS"] = "1"
    t0 = time.perf_counter()
    hs = host.HostScene(sd)
    desc = hs.flatten()
    t1 = time.perf_counter()
    ctx = device.Context(0)
    ctx.upload(desc)
    ctx.synchronize()
    t2 = time.perf_counter()
    u = ctx.upload_info()
    m, nodes, bounds, refs, planes = ctx.kd(0)
    trees[name] = (nodes.tobytes(), bounds.tobytes(), refs.tobytes(), planes.tobytes())
    print(f"kd on {name}: compile+flatten {t1 - t0:.3f} s, upload+device builds {t2 - t1:.3f} s (library: {u.upload_ms:.1f} ms; kd device time {u.kd_device_ms:.2f} ms, "
          f"{u.kd_levels} levels), {m.node_count} nodes, {m.ref_count} refs, depth {m.max_depth}, planes {list(m.plane_count)}", flush=True)
    ctx.close()
    del hs
print("device tree == host tree:", trees["device"] == trees["host"], " device again == device:", trees["device again"] == trees["device"])
