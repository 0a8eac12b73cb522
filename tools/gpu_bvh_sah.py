#!/usr/bin/env python3
"""Surface-area cost of the host-built and the device-built 4-wide BVH of a config's first mesh: expected interior-node
visits and leaf visits of a random line through the root box (sum over child slots of area(child) / area(root))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import device, host, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = {"cells": int(sys.argv[2])} if len(sys.argv) > 2 else {}
sd, _ = scenes.config(cfg, **kw)
for mode, env in (("host", "0"), ("device", "1")):
    os.environ["RE_BVH_DEVICE_MIN_TRIS"] = env
    hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
    m, nodes, _ = ctx.bvh(0)
    c = nodes["c"]
    lo, hi, ch = c["lo"].astype(np.float64), c["hi"].astype(np.float64), c["child"]
    present = ch != 0xffffffff
    ext = np.where(present[..., None], hi - lo, 0.0)
    area = 2 * (ext[..., 0] * ext[..., 1] + ext[..., 0] * ext[..., 2] + ext[..., 1] * ext[..., 2])
    rlo = np.where(present[0][:, None], lo[0], np.inf).min(0); rhi = np.where(present[0][:, None], hi[0], -np.inf).max(0)
    e = rhi - rlo; root = 2 * (e[0] * e[1] + e[0] * e[2] + e[1] * e[2])
    leaf = present & ((ch & 0x80000000) != 0)
    inner = present & ~leaf
    cnt = ((ch >> 28) & 7) + 1
    print(f"cfg {cfg} {mode:6s}: nodes {len(nodes)} depth {m.bvh_depth} leaves {int(leaf.sum())} tris/leaf {cnt[leaf].mean():.2f} | expected node visits "
          f"{1 + area[inner].sum() / root:.2f}, leaf visits {area[leaf].sum() / root:.2f}, triangle tests {(area[leaf] * cnt[leaf]).sum() / root:.2f}", flush=True)
    # per depth: interior-child area / root area (children follow their parent in the array: one forward pass gives depths)
    depth = np.zeros(len(nodes), np.int32)
    for k in range(4):
        idx = np.nonzero(inner[:, k])[0]
        # children have larger indices than parents, so a single ordered sweep would do; vectorised: iterate to a fixed point
    changed = True
    while changed:
        changed = False
        for k in range(4):
            idx = np.nonzero(inner[:, k])[0]
            d = depth[idx] + 1
            tgt = ch[idx, k]
            if (depth[tgt] != d).any():
                depth[tgt] = d; changed = True
    per = []
    for dd in range(int(depth.max()) + 1):
        sel = depth == dd
        per.append((int(sel.sum()), area[sel][inner[sel]].sum() / root, area[sel][leaf[sel]].sum() / root))
    print("   depth: nodes, interior-child area, leaf-child area:", " | ".join(f"{i}: {n} {a:.2f} {l:.2f}" for i, (n, a, l) in enumerate(per)), flush=True)
    ctx.close()
