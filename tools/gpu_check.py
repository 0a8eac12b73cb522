#!/usr/bin/env python3
"""First-contact GPU check: device arithmetic probes + parity of small frames against the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import device, host, scenes
import oracle

ctx = device.Context(0)
print(device.lib().rthip_version().decode(), flush=True)
rng = np.random.default_rng(1)
n = 1 << 20
a = rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n)); b = rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))
with np.errstate(all="ignore"):
    checks = {0: a + b, 1: a * b, 2: a / b, 3: np.sqrt(np.abs(a)), 4: np.floor(a % 1e6), 6: a.astype(np.float32).astype(np.float64)}
for op, want in checks.items():
    x = np.abs(a) if op == 3 else (a % 1e6 if op == 4 else a)
    got = ctx.probe_arith(op, x, b)
    print("arith op", op, "mismatch", int((got.view(np.uint64) != want.view(np.uint64)).sum()), flush=True)
x = rng.uniform(-100, 100, n); y = rng.choice([1.0, 2.0, 16.0, 1.5, 3.0], n)
want = np.fmod(x.astype(np.float32), y.astype(np.float32)).astype(np.float64)
print("fmodf mismatch", int((ctx.probe_arith(5, x, y).view(np.uint64) != want.view(np.uint64)).sum()))
x = rng.uniform(0, 26, n)
got = ctx.probe_arith(7, x)
print("sin mismatch vs numpy", int((got != np.sin(x)).sum()), "max ulp-ish", float(np.max(np.abs(got - np.sin(x)))))

O = oracle.load_oracle()
def cmp(name, sd, w, h, **kw):
    hs = host.HostScene(sd)
    t = time.time(); got = hs.render(w, h, ids=True, **kw); t1 = time.time() - t
    ref = sd.build(O).render(w, h, ids=True, threads=16, **kw)
    d = np.abs(got["pixels"].view(np.uint8).astype(int).reshape(h, w, 4) - ref["pixels"].view(np.uint8).astype(int).reshape(h, w, 4)).max(-1)
    print(f"{name}: px!= {(got['pixels'] != ref['pixels']).sum()} >1LSB {(d > 1).sum()} shape!= {(got['shape'] != ref['shape']).sum()} "
          f"tri!= {(got['tri'] != ref['tri']).sum()} rays {got['rays']} vs {ref['rays']} wall {t1:.3f}s stats {hs.last_stats()}", flush=True)
sd, kw = scenes.config(1); cmp("cfg1", sd, 640, 480, max_recursion=1)
sd, kw = scenes.config(2); cmp("cfg2", sd, 480, 270, max_recursion=3)
sd, kw = scenes.config(3); cmp("cfg3", sd, 480, 270, max_recursion=3)
sd, kw = scenes.config(3, cam_z=2.2); cmp("cfg3close", sd, 320, 180, max_recursion=3)
sd, kw = scenes.demo3_point_lights(); cmp("demo3", sd, 320, 180, max_recursion=3)
sd, kw = scenes.nested_transforms(); cmp("nested", sd, 320, 200, max_recursion=4)
cmp("nested ssaa", sd, 160, 100, max_recursion=2, aa=1)
cmp("nested dbgN", sd, 160, 100, renderer=1)
cmp("nested dbgC", sd, 160, 100, renderer=2)
# full-size timing
sd, kw = scenes.config(3)
hs = host.HostScene(sd)
for i in range(3):
    got = hs.render(1920, 1080, max_recursion=3)
    st = hs.last_stats(); rays = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
    print("cfg3 1080p", st, "Mrays/s (device)", rays / st["device_ms"] / 1e3, flush=True)
sd, kw = scenes.config(2)
hs = host.HostScene(sd)
for i in range(2):
    got = hs.render(1920, 1080, max_recursion=3)
    st = hs.last_stats(); rays = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
    print("cfg2 1080p", st, "Mrays/s (device)", rays / st["device_ms"] / 1e3, flush=True)
