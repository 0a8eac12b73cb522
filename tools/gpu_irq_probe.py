#!/usr/bin/env python3
"""Timeline of RenderStatus::Percent around an Interrupt() on the 4096 x 4096 frame of the interrupt test: which stage the
device is in when the request arrives and how long it runs on (found a 30 ms tail in the redo kernel in round 2)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import device, host, scenes
hs = host.HostScene(scenes.config(4, cells=384)[0])
ctx = device.Context(0); ctx.upload(hs.flatten())
w = h = 4096
cam, prm = hs.camera(w, h), device.make_params(w, h, 16)
fb = torch.zeros(w * h, dtype=torch.int32, device="cuda")
ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
t0 = time.perf_counter(); ctx.render_device(cam, prm, fb.data_ptr(), sync=True); full = time.perf_counter() - t0
print("full", full * 1e3, "stats hard", ctx.stats().rays_hard)
seen = []; stop = threading.Event()
def poll():
    while not stop.is_set():
        seen.append((time.perf_counter(), ctx.L.rthip_progress(ctx.h))); time.sleep(0.0003)
res = {}
def render():
    t1 = time.perf_counter()
    try: ctx.render_device(cam, prm, fb.data_ptr(), sync=True); res["code"] = 0
    except device.RthipError as e: res["code"] = e.code
    res["t"] = time.perf_counter() - t1
ctx.L.rthip_progress.restype = __import__("ctypes").c_float
p = threading.Thread(target=poll); p.start()
t = threading.Thread(target=render); ts = time.perf_counter(); t.start()
time.sleep(full * 0.25); ti = time.perf_counter(); ctx.L.rthip_interrupt(ctx.h); tj = time.perf_counter(); t.join(); te = time.perf_counter(); stop.set(); p.join()
print("interrupt call took", (tj - ti) * 1e3, "ms; issued at", (ti - ts) * 1e3, "; render returned at", (te - ts) * 1e3, res)
last = None
for tt, pr in seen:
    if pr != last: print(f"  t={1e3 * (tt - ts):7.2f} ms progress {pr:.4f}"); last = pr
