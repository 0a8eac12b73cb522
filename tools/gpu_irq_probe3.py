#!/usr/bin/env python3
"""Timeline of an interrupted frame (the interrupt test's frame: config-4 height field of 384 x 384 cells, 4096 x 4096, MaxRecursion 16):
a thread samples rthip_progress (the stage the device has begun: 3 per level + resolve) every ~30 us while the frame is rendered and
Interrupt() is called at a given share of the frame's time; prints when each stage began and when the call came back.
usage: [RTHIP_QUAD_COOP=0|1] python tools/gpu_irq_probe3.py [share ...]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import device, host, scenes
shares = [float(x) for x in sys.argv[1:]] or [0.1, 0.25, 0.4]
hs = host.HostScene(scenes.config(4, cells=384)[0])
ctx = device.Context(0)
ctx.upload(hs.flatten())
w = h = 4096
MR = int(os.environ.get("PROBE_MAXREC", "16"))
cam, prm = hs.camera(w, h), device.make_params(w, h, MR)
fb = torch.zeros(w * h, dtype=torch.int32, device="cuda")
for _ in range(3): ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
t0 = time.perf_counter(); ctx.render_device(cam, prm, fb.data_ptr(), sync=True); full = time.perf_counter() - t0
n_stages = 3 * (MR + 1) + 1
names = ["easy", "walk", "shade"]


def run(share):
    seen, stop, result = [], threading.Event(), {}

    def poll():
        while not stop.is_set():
            seen.append((time.perf_counter(), ctx.L.rthip_progress(ctx.h)))

    def render():
        try:
            ctx.render_device(cam, prm, fb.data_ptr(), sync=True); result["code"] = 0
        except device.RthipError as e:
            result["code"] = e.code
        result["t"] = time.perf_counter()
    ctx.L.rthip_progress.restype = __import__("ctypes").c_float
    tp = threading.Thread(target=poll); tr = threading.Thread(target=render)
    t1 = time.perf_counter(); tp.start(); tr.start()
    if share is not None:
        time.sleep(full * share)
        ti = time.perf_counter(); ctx.L.rthip_interrupt(ctx.h)
    else:
        ti = None
    tr.join(); stop.set(); tp.join()
    line, last = [], -1
    for t, p in seen:
        s = int(round(p * n_stages))
        if s != last and 0 < s <= n_stages:
            line.append(f"{'L%d %s' % ((s - 1) // 3, names[(s - 1) % 3]) if s < n_stages else 'resolve'}@{(t - t1) * 1e3:.1f}")
            last = s
    print(f"share {share}: code {result['code']}, Interrupt() at {((ti or t1) - t1) * 1e3:.1f} ms, back at {(result['t'] - t1) * 1e3:.1f} ms; stages begun: " + " ".join(line[:24]), flush=True)
    ctx.render_device(cam, prm, fb.data_ptr(), sync=True)


print(f"full frame {full * 1e3:.1f} ms", flush=True)
run(None)
for s in shares:
    run(s)
ctx.close()
