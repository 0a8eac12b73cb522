"""The two kernels that trace the rays which need a mesh walk (rthip.hip): k_trace_quad - everything of a ray in the four lanes
of its quad, 16 rays per wave - and k_trace_coop - per-ray work one ray per lane, up to 64 rays per wave, the walks 16 at a time.
Which one a context uses follows its scene (small trees: k_trace_coop; RTHIP_QUAD_COOP forces it), and inside k_trace_coop the
length of the level's list picks 64 / 32 rays per wave or the quad code (RTHIP_COOP_T64 / RTHIP_COOP_T32).  The choice must
never show in a frame: every mode here gives the oracle's pixels, ids and ray counts (Scene.cpp:163-237, Raytracer.cpp:187-315)."""
import numpy as np
import pytest

from raytracing_amd import device, host, scenes

import edge_scenes
import fuzz_scenes
from test_ties import tie_scene

pytestmark = pytest.mark.gpu

BIG = str(2**31 - 1)
MODES = {
    "quad": {"RTHIP_QUAD_COOP": "0"},
    "coop64": {"RTHIP_QUAD_COOP": "1", "RTHIP_COOP_T32": "1", "RTHIP_COOP_T64": "1"},   # every list 64 rays per wave
    "coop32": {"RTHIP_QUAD_COOP": "1", "RTHIP_COOP_T32": "1", "RTHIP_COOP_T64": BIG},   # every list 32 rays per wave
    "coop_short": {"RTHIP_QUAD_COOP": "1", "RTHIP_COOP_T32": BIG, "RTHIP_COOP_T64": BIG},  # k_trace_coop, every list too short: its fallback to the quad code
    "default": {},
}


def _render(monkeypatch, mode, sd, w, h, **kw):
    for k in ("RTHIP_QUAD_COOP", "RTHIP_COOP_T32", "RTHIP_COOP_T64"):
        monkeypatch.delenv(k, raising=False)
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)   # (read when the context is created: rthip_create)
    return host.HostScene(sd).render(w, h, ids=True, **kw)


def _same(got, ref, what):
    assert np.array_equal(got["shape"], ref["shape"]), f"{what}: {(got['shape'] != ref['shape']).sum()} shape ids differ"
    assert np.array_equal(got["tri"], ref["tri"]), f"{what}: {(got['tri'] != ref['tri']).sum()} triangle ids differ"
    assert np.array_equal(got["pixels"], ref["pixels"]), f"{what}: {(got['pixels'] != ref['pixels']).sum()} pixels differ"
    assert got["rays"] == ref["rays"], f"{what}: {got['rays']} rays, the reference traces {ref['rays']}"


@pytest.mark.parametrize("mode", list(MODES))
def test_bunny_close_up_and_multi_mesh_scene(monkeypatch, oracle_lib, mode):
    """one mesh under a sun (closest + one shadow walk per ray), and five rotated / scaled meshes + 16 spheres + plane at depth 6
    and with SSAA (several meshes per query, analytic shadow casters, rays that need no walk at all)"""
    sd = scenes.config(3, cam_z=2.4)[0]
    ref = sd.build(oracle_lib).render(224, 126, ids=True, threads=8, max_recursion=3)
    _same(_render(monkeypatch, mode, sd, 224, 126, max_recursion=3), ref, f"{mode} bunny")
    sd = scenes.config(5, cells=24)[0]
    for kw in (dict(max_recursion=6), dict(max_recursion=2, aa=1)):
        ref = sd.build(oracle_lib).render(160, 90, ids=True, threads=8, **kw)
        _same(_render(monkeypatch, mode, sd, 160, 90, **kw), ref, f"{mode} multi-mesh {kw}")


@pytest.mark.parametrize("mode", ["coop64", "coop32", "coop_short"])
def test_random_scenes_edge_scenes_ties_and_degenerate_rays(monkeypatch, oracle_lib, mode):
    """what only the redo list settles (exact ties, rays in a kd plane) and the generator's scenes: point / spot lights (per-ray
    shadow directions), nested transforms, empty meshes, a single triangle, recursion 16"""
    for seed in fuzz_scenes.FUZZ_SEEDS[:12]:
        sd = fuzz_scenes.random_scene(seed)
        kw = dict(max_recursion=4, aa=1 if seed % 5 == 0 else 0)
        ref = sd.build(oracle_lib).render(96, 64, ids=True, threads=8, **kw)
        _same(_render(monkeypatch, mode, sd, 96, 64, **kw), ref, f"{mode} fuzz {seed}")
    for name, (mk, runs) in edge_scenes.EDGE_CASES.items():
        w, h, kw = runs[0]
        sd = mk()
        ref = sd.build(oracle_lib).render(w, h, ids=True, threads=4, **kw)
        _same(_render(monkeypatch, mode, sd, w, h, **kw), ref, f"{mode} {name}")
    sd, _ = tie_scene(2)
    ref = sd.build(oracle_lib).render(64, 48, ids=True, threads=4, max_recursion=2)
    _same(_render(monkeypatch, mode, sd, 64, 48, max_recursion=2), ref, f"{mode} ties")
    sd = scenes.SceneDesc("grid mesh symmetric about x = 0, camera on the plane x = 0")
    m = sd.uniform_material(0x88aa66, 0.8)
    v, n = scenes.heightfield_mesh(cells=32, seed=5, extent=4.0)
    sd.mesh(m, v, n, scenes.NORMAL_VERTEX)
    sd.directional_light(0xffffff, 1, 2, 1)
    sd.color_background(0x102030)
    sd.look_at((0, 5, 9), (0, 0, 0))
    ref = sd.build(oracle_lib).render(128, 96, ids=True, threads=8, max_recursion=2)
    _same(_render(monkeypatch, mode, sd, 128, 96, max_recursion=2), ref, f"{mode} kd-plane rays")


def test_long_lists_of_a_batch_take_the_wide_path_and_equal_the_single_frames(monkeypatch):
    """the default thresholds at work: a call of 6 frames of the benchmark scene at 1080p lists > 600 000 rays per level for the
    walk (64 rays per wave), a lone frame of it < 300 000 (the quad code): every frame of the call equals its lone render"""
    import torch
    for k in ("RTHIP_QUAD_COOP", "RTHIP_COOP_T32", "RTHIP_COOP_T64"):
        monkeypatch.delenv(k, raising=False)
    sd, kw = scenes.config(3)
    w, h, mr = kw["width"], kw["height"], kw["max_recursion"]
    hs = host.HostScene(sd)
    ctx = device.Context(0)
    try:
        ctx.upload(hs.flatten())
        prm = device.make_params(w, h, mr)
        cams = []
        for k in range(6):
            cam = hs.camera(w, h)
            cam.position[0] = -0.5 + 0.2 * k
            cams.append(cam)
        one = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        singles, rays, hard = [], 0, []
        for cam in cams:
            ctx.render_device(cam, prm, one.data_ptr(), sync=True)
            singles.append(one.clone())
            st = ctx.stats()
            rays += st.rays_primary + st.rays_shadow + st.rays_reflect
            hard.append(st.rays_hard)
        assert max(hard) < 4 * 300000 and sum(hard) > 4 * 600000 * 0.5, hard   # (rays_hard sums the four levels)
        buf = torch.zeros(len(cams) * w * h, dtype=torch.int32, device="cuda")
        for rep in range(3):   # plain launches, the capture, a replay
            buf.zero_()
            ctx.render_batch_device(cams, prm, buf.data_ptr(), sync=True)
            for k in range(len(cams)):
                assert torch.equal(buf[k * w * h:(k + 1) * w * h], singles[k]), f"frame {k} of the call (rep {rep}) differs from its lone render"
        st = ctx.stats()
        assert st.rays_primary + st.rays_shadow + st.rays_reflect == rays
    finally:
        ctx.close()


def test_counting_contexts_count_the_same_walks_in_both_kernels(monkeypatch):
    """RTHIP_COUNTERS=1 (bench.py's flop accounting): BVH steps, triangle tests and walked rays do not depend on the kernel"""
    import torch
    sd = scenes.config(3, cam_z=2.4)[0]
    hs = host.HostScene(sd)
    w, h = 320, 180
    got = {}
    for mode in ("quad", "coop64"):
        for k, v in MODES[mode].items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("RTHIP_COUNTERS", "1")
        ctx = device.Context(0)
        monkeypatch.setenv("RTHIP_COUNTERS", "0")
        try:
            ctx.upload(hs.flatten())
            fb = torch.zeros(w * h, dtype=torch.int32, device="cuda")
            ctx.render_device(hs.camera(w, h), device.make_params(w, h, 3), fb.data_ptr(), sync=True)
            st = ctx.stats()
            got[mode] = (int(st.kd_nodes_visited), int(st.kd_tris_tested), int(st.rays_hard), int(fb.to(torch.int64).sum()))
        finally:
            ctx.close()
    assert got["quad"] == got["coop64"] and got["quad"][0] > 100000, got


def test_a_wrong_guess_from_the_last_call_costs_no_ray(monkeypatch):
    """A call launches only ONE walk kernel per level once the last call of its shape has reported its list lengths (render_frame:
    walk_mode).  When the view then changes abruptly the guess is wrong - k_trace_coop alone meets lists of a few rays, k_trace_quad
    alone meets long ones - and the frames must still be the ones a context that never guesses renders."""
    import torch
    sd = scenes.config(3, cam_z=2.4)[0]
    hs = host.HostScene(sd)
    desc = hs.flatten()
    w, h = 224, 126
    prm = device.make_params(w, h, 3)
    near = hs.camera(w, h)                                              # the bunny fills the view: ~10 000 listed rays per level
    away = hs.camera_at((0.0, 3.0, 2.4), (0.0, 0.3, 1.0), w, h)          # sky only: nothing listed
    plane = hs.camera_at((0.0, 0.6, 6.0), (0.0, -0.25, -1.0), w, h)      # mostly the checker plane: a few hundred shadow rays near the model
    hs.camera(w, h)
    seq = [near, near, near, away, away, near, plane, near, away, plane, plane, near]
    for k in ("RTHIP_COOP_T32", "RTHIP_COOP_T64"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("RTHIP_QUAD_COOP", "0")
    ref_ctx = device.Context(0)
    monkeypatch.setenv("RTHIP_QUAD_COOP", "1")
    monkeypatch.setenv("RTHIP_COOP_T32", "3000")
    monkeypatch.setenv("RTHIP_COOP_T64", "6000")
    ctx = device.Context(0)
    try:
        ref_ctx.upload(desc)
        ctx.upload(desc)
        a = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        b = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        listed = set()
        for k, cam in enumerate(seq):
            ref_ctx.render_device(cam, prm, a.data_ptr(), sync=True)
            ctx.render_device(cam, prm, b.data_ptr(), sync=True)
            assert torch.equal(a, b), f"call {k}: {int((a != b).sum())} pixels differ"
            ra, rb = ref_ctx.stats(), ctx.stats()
            assert (ra.rays_primary, ra.rays_shadow, ra.rays_reflect, ra.rays_hard) == (rb.rays_primary, rb.rays_shadow, rb.rays_reflect, rb.rays_hard)
            listed.add(int(rb.rays_hard) > 4 * 1.4 * 3000)
        assert listed == {False, True}, "the views no longer fall on both sides of the threshold"
    finally:
        ref_ctx.close()
        ctx.close()


def test_which_kernel_a_scenes_long_lists_get(monkeypatch):
    """rthip_upload_info.walk_coop: every scene with meshes has its long lists walked by k_trace_coop (the bunny +8 %, config 5's five
    meshes -6 % of a frame, config 4's one big mesh unchanged: profiles/r03_coop_walk.txt); RTHIP_COOP_MAX_TREE_MB restricts that to
    scenes whose trees (BVH nodes + leaf-order triangles) fit, RTHIP_QUAD_COOP forces either side"""
    for k in ("RTHIP_QUAD_COOP", "RTHIP_COOP_MAX_TREE_MB"):
        monkeypatch.delenv(k, raising=False)
    keep = [host.HostScene(scenes.config(3)[0]),                          # the bunny: 4 968 triangles, 0.8 MB
            host.HostScene(scenes.config(4, cells=200)[0]),               # 80 000 triangles: ~13 MB
            host.HostScene(scenes.config(2)[0])]                          # (the descriptors point into these objects)
    small, big, no_mesh = (hs.flatten() for hs in keep)

    def side(desc):
        ctx = device.Context(0)
        try:
            ctx.upload(desc)
            return int(ctx.upload_info().walk_coop)
        finally:
            ctx.close()
    assert side(small) == 1 and side(big) == 1 and side(no_mesh) == 0
    monkeypatch.setenv("RTHIP_COOP_MAX_TREE_MB", "8")
    assert side(small) == 1 and side(big) == 0
    monkeypatch.setenv("RTHIP_QUAD_COOP", "0")
    assert side(small) == 0


@pytest.mark.parametrize("table", ["0", "1"])
def test_directional_lights_local_directions_from_the_upload_table(monkeypatch, oracle_lib, table):
    """SceneView::shadow_ld (rthip.hip: k_light_dirs): a directional light's direction in every shape's frame is computed once per
    (light, shape) at upload instead of per (shadow ray, shape).  With the table (default) and without it (RTHIP_SHADOW_LD=0) every
    walk kernel gives the oracle's frames: rotated / scaled meshes and spheres under several suns, a disabled sun, point lights
    beside them (per-ray directions: no table row), nested transforms from the generator."""
    monkeypatch.setenv("RTHIP_SHADOW_LD", table)   # (read when the scene is uploaded)
    sd = scenes.SceneDesc("three suns, one disabled, and a point light over rotated / scaled shapes")
    a = sd.uniform_material(0xcc8844, 0.6)
    b = sd.uniform_material(0xffffff, 0.3)
    v, n = scenes.heightfield_mesh(cells=24, seed=9, extent=3.0)
    sd.mesh(a, v, n, scenes.NORMAL_VERTEX, node=sd.node(-1, (0.5, 0.0, -1.0), (0.2, 0.7, -0.1), (1.0, 1.6, 0.8)))
    sd.mesh(b, v, n, scenes.NORMAL_FACE, node=sd.node(-1, (-2.0, 1.5, -3.0), (-0.4, 0.1, 0.9), (0.5, 0.5, 0.7)))
    sd.scale(1.3, 0.7, 1.1); sd.position(1.5, 1.2, 0.5); sd.sphere(b)
    sd.scale(1, 1, 1); sd.position(-1.5, 0.8, 1.0); sd.sphere(a)
    sd.position(0, -1.2, 0); sd.plane(sd.interpolated_material(sd.checker(2), a, b), 0, 1, 0)
    sd.directional_light(0xffeecc, 1, 2, 1)
    sd.directional_light(0x4466ff, -2, 1, 0.5)
    sd.directional_light(0xff0000, 0, 1, -1, enabled=0)
    sd.point_light(0x88ff88, 9, pos=(0.5, 4, 2))
    sd.ambient_light(0x101010)
    sd.color_background(0x203040)
    sd.look_at((0.5, 3.5, 7), (0, 0.3, 0))
    ref = sd.build(oracle_lib).render(160, 100, ids=True, threads=8, max_recursion=4)
    for mode in ("quad", "coop64", "coop32", "default"):
        _same(_render(monkeypatch, mode, sd, 160, 100, max_recursion=4), ref, f"table {table} {mode} suns")
    for seed in fuzz_scenes.FUZZ_SEEDS[12:20]:
        sd = fuzz_scenes.random_scene(seed)
        ref = sd.build(oracle_lib).render(96, 64, ids=True, threads=8, max_recursion=3)
        for mode in ("quad", "coop64"):
            _same(_render(monkeypatch, mode, sd, 96, 64, max_recursion=3), ref, f"table {table} {mode} fuzz {seed}")


def test_redo_pass_behind_k_shade(monkeypatch, oracle_lib):
    """The redo list (exact ties, rays in a kd split plane) is re-traced and shaded by k_redo behind k_shade (rthip.hip): with
    every walk kernel in front of it, with SSAA (ids come from sub-sample 4 only), and on the edge scenes."""
    sd, _ = tie_scene(2)
    ref = sd.build(oracle_lib).render(64, 48, ids=True, threads=4, max_recursion=2)
    for mode in ("quad", "coop64", "default"):
        _same(_render(monkeypatch, mode, sd, 64, 48, max_recursion=2), ref, f"{mode} ties")
    sd = scenes.SceneDesc("grid mesh symmetric about x = 0, camera on the plane x = 0")
    m = sd.uniform_material(0x88aa66, 0.8)
    v, n = scenes.heightfield_mesh(cells=32, seed=5, extent=4.0)
    sd.mesh(m, v, n, scenes.NORMAL_VERTEX)
    sd.directional_light(0xffffff, 1, 2, 1)
    sd.color_background(0x102030)
    sd.look_at((0, 5, 9), (0, 0, 0))
    for kw in (dict(max_recursion=2), dict(max_recursion=1, aa=1)):
        ref = sd.build(oracle_lib).render(128, 96, ids=True, threads=8, **kw)
        for mode in ("quad", "coop32"):
            _same(_render(monkeypatch, mode, sd, 128, 96, **kw), ref, f"{mode} kd-plane rays {kw}")


@pytest.mark.parametrize("lean", ["0", "1"])
def test_lean_kernels_for_scenes_without_worley_marble_or_material_stacks(monkeypatch, oracle_lib, lean):
    """k_trace_easy / k_shade have a second instantiation without Worley's cell loop, Marble's sin() and the material stack
    (rt_core.hpp: noise_sample<LEAN>), picked at upload for scenes whose shapes' materials are uniform or one checker / Perlin
    interpolation of two uniform ones (rthip.hip: lean_scene; RTHIP_LEAN=0: never).  Such scenes give the oracle's frames through
    either; a scene that needs the full kernels (Demo 1: marble, Worley, a nested interpolation) must not be mistaken for one."""
    monkeypatch.setenv("RTHIP_LEAN", lean)   # (read when the scene is uploaded)
    for n, w, h, kw in ((1, 96, 72, dict(max_recursion=1)), (3, 160, 90, dict(max_recursion=3)), (3, 96, 54, dict(max_recursion=2, aa=1)),
                        (2, 128, 72, dict(max_recursion=3))):
        sd = scenes.config(n)[0]
        ref = sd.build(oracle_lib).render(w, h, ids=True, threads=8, **kw)
        _same(_render(monkeypatch, "default", sd, w, h, **kw), ref, f"lean {lean} config {n} {kw}")
    sd = scenes.config(5, cells=16)[0]
    ref = sd.build(oracle_lib).render(128, 72, ids=True, threads=8, max_recursion=5)
    for mode in ("quad", "coop64"):
        _same(_render(monkeypatch, mode, sd, 128, 72, max_recursion=5), ref, f"lean {lean} {mode} config 5")
    # a Perlin-interpolated material (lean) next to the debug renderers, which evaluate materials too
    sd = scenes.SceneDesc("perlin-interpolated sphere over a checker plane")
    a = sd.uniform_material(0x224488, 0.7)
    b = sd.uniform_material(0xffeedd, 0.2)
    sd.position(0, 0.5, 0); sd.sphere(sd.interpolated_material(sd.perlin(2, seed=77), a, b))
    sd.position(0, -1, 0); sd.plane(sd.interpolated_material(sd.checker(2), a, b), 0, 1, 0)
    sun = sd.directional_light(0xffffff, 1, 2, 1)
    sd.skybox(0x4444ff, 0xffffff, sun, seed=78)
    sd.look_at((0, 2, 5), (0, 0, 0))
    for kw in (dict(max_recursion=3), dict(max_recursion=0, renderer=2), dict(max_recursion=0, renderer=1)):
        ref = sd.build(oracle_lib).render(96, 64, ids=True, threads=4, **kw)
        _same(_render(monkeypatch, "default", sd, 96, 64, **kw), ref, f"lean {lean} perlin material {kw}")
