#!/usr/bin/env python3
"""target for rocprofv3 --kernel-trace --stats: one upload of config 4 with both trees built on the device"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RE_KD_DEVICE_MIN_TRIS"] = "1"
os.environ["RE_BVH_DEVICE_MIN_TRIS"] = "1"
os.environ["RE_BUILD_TIMING"] = "1"
from raytracing_amd import device, host, scenes
sd, _ = scenes.config(4, cells=int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
t0 = time.perf_counter(); hs = host.HostScene(sd); t1 = time.perf_counter(); desc = hs.flatten(); t2 = time.perf_counter()
ctx = device.Context(0); ctx.upload(desc); ctx.synchronize(); t3 = time.perf_counter()
u = ctx.upload_info()
print(f"scene construction {t1 - t0:.3f} s, compile+flatten {t2 - t1:.3f} s, upload {t3 - t2:.3f} s, kd device {u.kd_device_ms:.1f} ms")
