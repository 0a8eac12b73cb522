#!/usr/bin/env python3
"""Target / summariser for the per-component VALU instruction count of k_trace_easy (rocprofv3 --pmc SQ_INSTS_VALU ...): one
1080p frame of a VARIANT of config 3's scene that adds one piece of the kernel's work at a time.
usage: prof_easy_components.py run <variant>         (under rocprofv3; prints the ray counts)
       prof_easy_components.py summary <dir>         (reads <dir>/<variant>/**/counter_collection.csv and <dir>/<variant>.log)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = ["v0_raygen_black", "v1_sky", "v2_plane_uniform_nolight", "v3_plane_checker_nolight", "v4_plane_checker_sun", "v5_plus_bunny_level0", "v6_config3"]
W, H = 1920, 1080


def scene(variant):
    from raytracing_amd import scenes
    sd = scenes.SceneDesc(variant)
    sd.camera((0, 0, 5))
    k = VARIANTS.index(variant)
    if k == 6:
        return scenes.config(3)[0], 3
    sd.position(0, -1, 0)
    black = sd.uniform_material(0, 0.7)
    white = sd.uniform_material(0xffffff, 0.7)
    if k >= 2:
        sd.plane(sd.interpolated_material(sd.checker(1), black, white) if k >= 3 else white, 0, 1, 0)
    if k >= 5:
        v, nrm = scenes.load_bunny()
        sd.mesh(sd.uniform_material(0xff2222, 0.5), v, nrm, scenes.NORMAL_VERTEX)
    if k >= 1:
        sun = sd.directional_light(0xffcccc, 1, 1, 1, enabled=1 if k >= 4 else 0)
        sd.skybox(0xff0000, 0x0000ff, sun, seed=301)
    return sd, 0


if sys.argv[1] == "run":
    from raytracing_amd import device, host
    import torch
    sd, mr = scene(sys.argv[2])
    hs = host.HostScene(sd); ctx = device.Context(0); ctx.upload(hs.flatten())
    fb = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    ctx.render_device(hs.camera(W, H), device.make_params(W, H, mr), fb.data_ptr(), sync=True)
    s = ctx.stats()
    print(f"RAYS {sys.argv[2]} primary {s.rays_primary} shadow {s.rays_shadow} reflect {s.rays_reflect} hard {s.rays_hard} level_rays {[s.level_rays[i] for i in range(mr + 1)]}")
else:
    import csv, glob, re
    d = sys.argv[2]
    print("| variant | rays through k_trace_easy | of them listed for the walk | shadow queries | VALU wave-instructions | per ray (x 64 lanes / rays) | added by this step | active lanes / VALU instr |")
    print("|---|---|---|---|---|---|---|---|")
    prev = None
    for v in VARIANTS:
        log = open(f"{d}/{v}.log").read()
        m = re.search(r"RAYS \S+ primary (\d+) shadow (\d+) reflect (\d+) hard (\d+) level_rays \[([^\]]*)\]", log)
        primary, shadow, reflect, hard = (int(m.group(i)) for i in range(1, 5))
        levels = [int(x) for x in m.group(5).split(",")]
        rays = sum(levels)   # every ray of every level goes through k_trace_easy once
        acc = {}
        for f in glob.glob(f"{d}/{v}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_trace_easy" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        insts = acc.get("SQ_INSTS_VALU", 0.0)
        per_ray = insts * 64 / rays
        lanes = acc.get("SQ_THREAD_CYCLES_VALU", 0.0) / insts / 4 if insts else 0   # thread-cycles: 4 cycles per instruction and active lane
        print(f"| `{v}` | {rays:,} | {hard:,} | {shadow:,} | {insts / 1e6:.1f} M | {per_ray:.0f} | {'' if prev is None else f'{per_ray - prev:+.0f}'} | {lanes:.0f} |")
        prev = per_ray
