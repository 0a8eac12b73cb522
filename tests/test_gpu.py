"""GPU parity tests proper: everything goes through the C-ABI of include/rthip.h (directly via
raytracing_amd.device, or below re::HipRaytracer via the host facade) and is compared with the golden
vectors of the genuine reference and with the CPU oracle on the same seeded inputs."""
import ctypes as C
import time

import numpy as np
import pytest
import torch

from conftest import FRAME_CASES, KAT_SCENES, SPARSE_CASES, GOLDEN, assert_frame_parity
from raytracing_amd import device, host, scenes
from raytracing_amd.sbapi import NotInScene

pytestmark = pytest.mark.gpu

# scenes whose shading calls libm sin() (Marble): device sin may differ from glibc by an ulp
# noise kinds reachable from a shape's material in each KAT scene
KAT_NOISE_KINDS = {"nested": {"perlin", "worley", "checker"}, "cfg2": {"checker", "perlin", "worley", "marble"}, "cfg3": {"checker"}}
LIBM_SCENES = {"cfg2", "cfg2_dbgC", "demo3", "cfg2_1080p"}


@pytest.fixture(scope="module")
def ctx():
    c = device.Context(0)
    yield c
    c.close()


def test_device_arithmetic_is_ieee_and_unfused(ctx):
    """the whole parity argument rests on + - * / sqrt floor fmodf and casts being bit-exact on the device"""
    rng = np.random.default_rng(1)
    n = 1 << 18
    a = rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))
    b = rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))
    a[:8] = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-310, 5e-324]
    b[:8] = [0.0, 1.0, -0.0, 3.0, 2.0, 0.5, 3.0, 2.0]
    bits = lambda x: np.asarray(x, np.float64).view(np.uint64)
    with np.errstate(all="ignore"):
        assert np.array_equal(bits(ctx.probe_arith(0, a, b)), bits(a + b))
        assert np.array_equal(bits(ctx.probe_arith(1, a, b)), bits(a * b))
        q, qd = a / b, ctx.probe_arith(2, a, b)
        ok = ~np.isnan(q)
        assert np.array_equal(bits(qd[ok]), bits(q[ok])) and np.isnan(qd[~ok]).all()
        assert np.array_equal(bits(ctx.probe_arith(3, np.abs(a))), bits(np.sqrt(np.abs(a))))
        x = rng.uniform(-1e6, 1e6, n)
        assert np.array_equal(bits(ctx.probe_arith(4, x)), bits(np.floor(x)))
        assert np.array_equal(bits(ctx.probe_arith(6, a)), bits(a.astype(np.float32).astype(np.float64)))
        assert np.array_equal(bits(ctx.probe_arith(8, x)), bits(np.trunc(x)))
        y = rng.choice([1.0, 2.0, 16.0, 1.5, 3.0, 0.1], n)
        want = np.fmod(x.astype(np.float32) * np.float32(1e-4), y.astype(np.float32)).astype(np.float64)
        assert np.array_equal(bits(ctx.probe_arith(5, (x.astype(np.float32) * np.float32(1e-4)).astype(np.float64), y)), bits(want))
        fin = np.isfinite(a) & np.isfinite(b)
        assert np.array_equal(ctx.probe_arith(9, a, b)[fin], np.fmax(a, b)[fin])
        assert np.array_equal(ctx.probe_arith(10, a, b)[fin], np.fmin(a, b)[fin])
    # a*b+c must NOT be contracted into an fma: (1+e)*(1-e)-1 is -e*e fused, 0 unfused... checked via the frames below


@pytest.mark.parametrize("name", list(FRAME_CASES))
def test_frames_through_hipraytracer_match_reference_golden(golden, name):
    mk, w, h, kw = FRAME_CASES[name]
    hs = host.HostScene(mk())
    got = hs.render(w, h, ids=True, **kw)   # re::HipRaytracer::RenderSync -> rthip_render
    f = golden["frames"]
    diff, ids_bad = assert_frame_parity(got, f[name + "_pixels"], f[name + "_shape"].astype(np.int32), f[name + "_tri"], name)
    assert got["rays"] == int(f[name + "_rays"][0]), "ray count differs from the reference's"
    if name not in LIBM_SCENES:
        assert diff == 0 and ids_bad == 0, f"{name}: expected bit-exact pixels ({diff} differ)"


@pytest.mark.parametrize("name", list(SPARSE_CASES))
def test_full_resolution_sparse_pixels_match_reference_golden(golden, name):
    mk, w, h, kw = SPARSE_CASES[name]
    s = golden["sparse"]
    xs, ys = s[name + "_x"].astype(np.int32), s[name + "_y"].astype(np.int32)
    got = host.HostScene(mk()).render(w, h, ids=True, sub=(xs, ys), **kw)
    diff, ids_bad = assert_frame_parity(got, s[name + "_pixels"], s[name + "_shape"].astype(np.int32), s[name + "_tri"], name)
    if name not in LIBM_SCENES:
        assert diff == 0 and ids_bad == 0


def test_config4_two_million_triangles_sparse_pixels_match_reference_golden():
    """BASELINE configs[3]: 2 097 152-triangle height field + 16 reflective spheres, 3840x2160, 4 bounces.
    The reference needs hours for the frame (exhaustive kd walk); 2048 seeded pixels of it are the golden."""
    g = np.load(GOLDEN + "/sparse_cfg4.npz")
    sd, kw = scenes.config(4)
    xs, ys = g["x"].astype(np.int32), g["y"].astype(np.int32)
    got = host.HostScene(sd).render(kw["width"], kw["height"], max_recursion=kw["max_recursion"], ids=True, sub=(xs, ys))
    diff, ids_bad = assert_frame_parity(got, g["pixels"], g["shape"].astype(np.int32), g["tri"], "config 4")
    assert diff == 0 and ids_bad == 0


def test_config5_ten_million_triangles_8k_sparse_pixels_match_reference_golden():
    """BASELINE configs[4]: five 2 097 152-triangle height fields under rotated / scaled nodes + checker plane + 16
    reflective spheres, 7680x4320, 8 bounces — on ONE GPU here (the frame is traced in two chunks).  1024 seeded pixels
    of the reference's frame are the golden (tools/make_golden.py big5)."""
    g = np.load(GOLDEN + "/sparse_cfg5.npz")
    sd, kw = scenes.config(5)
    xs, ys = g["x"].astype(np.int32), g["y"].astype(np.int32)
    got = host.HostScene(sd).render(kw["width"], kw["height"], max_recursion=kw["max_recursion"], ids=True, sub=(xs, ys))
    diff, ids_bad = assert_frame_parity(got, g["pixels"], g["shape"].astype(np.int32), g["tri"], "config 5")
    assert diff == 0 and ids_bad == 0
    assert len(np.unique(g["shape"])) >= 5 and (g["tri"] >= 0).mean() > 0.3   # the sample sees several meshes


@pytest.mark.parametrize("name", list(KAT_SCENES))
def test_known_answers_of_single_device_functions(golden, name):
    k = golden["kat"]
    sd = KAT_SCENES[name]()
    hs = host.HostScene(sd)
    s = hs.built
    o, d, p = k[name + "_ray_o"], k[name + "_ray_d"], k[name + "_points"]
    c = s.cast_rays(o, d)                                   # rthip_probe_cast  (S1, P1, P2, T2, T4)
    for key in ("hit", "shape", "tri", "point", "local", "normal"):
        assert np.array_equal(c[key], k[name + "_cast_" + key]), key
    assert np.array_equal(s.background_eval(d), k[name + "_background"])   # rthip_probe_background (B1, B2)
    used_noises = set()
    for ni, info in sd.noise_info.items():                  # rthip_probe_noise (N0-N4)
        try:
            got = s.noise_sample(ni, p)
        except NotInScene:
            continue                                        # noise not referenced by a shape material: not on the device
        used_noises.add(info[0])
        want = k[f"{name}_noise{ni}"]
        if info[0] == "marble":                             # libm sin: 1 ulp
            assert np.abs(got - want).max() <= 4e-16 and (got != want).mean() < 0.2
        else:
            assert np.array_equal(got, want), info
    col = hs.built.raycast(o, d, 3)                         # rthip_probe_raycast (R4-R6)
    if name == "cfg2":
        assert np.abs(col - k[name + "_raycast"]).max() < 1e-12
    else:
        assert np.array_equal(col, k[name + "_raycast"])
    n_checked = 0
    for mi in range(sd._n["mat"]):                          # rthip_probe_material (M1, M2)
        try:
            mc, ma = s.material_eval(mi, p)
        except NotInScene:
            continue
        n_checked += 1
        assert np.abs(mc - k[f"{name}_mat{mi}_color"]).max() < 1e-13 and np.abs(ma - k[f"{name}_mat{mi}_abs"]).max() < 1e-13
        if name != "cfg2":
            assert np.array_equal(mc, k[f"{name}_mat{mi}_color"]) and np.array_equal(ma, k[f"{name}_mat{mi}_abs"])
    assert n_checked > 0
    # every noise kind the scene's shapes use was probed (a kind silently dropping out would hide a device regression)
    assert used_noises == KAT_NOISE_KINDS[name], (used_noises, KAT_NOISE_KINDS[name])


@pytest.mark.parametrize("name", ["cfg3", "cfg3close", "nested"])
def test_reference_kdtree_traversal_gives_the_same_frames(golden, name):
    """accel = 1: the literal flattening of KDTreeTriangle (ordered, pruned walk with the reference's box test)"""
    mk, w, h, kw = FRAME_CASES[name]
    hs = host.HostScene(mk())
    hs.set_accel(True)
    got = hs.render(w, h, ids=True, **kw)
    f = golden["frames"]
    diff, ids_bad = assert_frame_parity(got, f[name + "_pixels"], f[name + "_shape"].astype(np.int32), f[name + "_tri"], name)
    assert diff == 0 and ids_bad == 0 and got["rays"] == int(f[name + "_rays"][0])


def test_rays_in_a_kd_split_plane_follow_the_reference(oracle_lib):
    """A ray with an exactly-zero direction component whose origin lies exactly on a kd box plane makes the
    reference's slab test evaluate 0/0 (Common.cpp:349-356): it then misses geometry a geometric structure
    finds.  The device routes exactly those rays through the kd-tree (kd_degenerate_ray)."""
    sd = scenes.SceneDesc("grid mesh symmetric about x = 0, camera on the plane x = 0")
    m = sd.uniform_material(0x88aa66, 0.8)
    v, n = scenes.heightfield_mesh(cells=32, seed=5, extent=4.0)
    sd.mesh(m, v, n, scenes.NORMAL_VERTEX)
    sd.directional_light(0xffffff, 1, 2, 1)
    sd.color_background(0x102030)
    sd.look_at((0, 5, 9), (0, 0, 0))
    w, h = 128, 96
    got = host.HostScene(sd).render(w, h, max_recursion=2, ids=True)
    ref = sd.build(oracle_lib).render(w, h, max_recursion=2, ids=True, threads=8)
    col = w // 2
    assert (ref["shape"][:, col] < 0).sum() > (ref["shape"][:, col - 1] < 0).sum(), "fixture no longer degenerate"
    diff, ids_bad = assert_frame_parity(got, ref["pixels"], ref["shape"], ref["tri"], sd.name)
    assert diff == 0 and ids_bad == 0


def test_device_matches_oracle_on_fresh_random_views(oracle_lib):
    """same seeded inputs through the oracle and the device, views that are not in the fixtures"""
    rng = np.random.default_rng(2024)
    for mk in (lambda: scenes.config(3, cam_z=3.0)[0], lambda: scenes.nested_transforms()[0]):
        for _ in range(2):
            sd = mk()
            pos = np.array(sd.camera_pos) + rng.uniform(-1.5, 1.5, 3)
            sd.look_at(tuple(pos), tuple(rng.uniform(-0.5, 0.5, 3) + np.array([0, 0, -3.0])))
            got = host.HostScene(sd).render(192, 108, max_recursion=3, ids=True)
            ref = sd.build(oracle_lib).render(192, 108, max_recursion=3, ids=True, threads=8)
            diff, ids_bad = assert_frame_parity(got, ref["pixels"], ref["shape"], ref["tri"], sd.name)
            assert diff == 0 and ids_bad == 0 and got["rays"] == ref["rays"]


def test_multi_mesh_scene_matches_oracle(oracle_lib):
    """BASELINE configs[4] geometry at reduced size: five rotated / non-uniformly scaled height-field meshes,
    16 reflective spheres and a checker plane (several meshes per quad walk, exact-tie rule across shapes,
    any-hit stop under rotation), deep recursion, and 3x3 SSAA of the same scene."""
    sd = scenes.config(5, cells=24)[0]
    for kw in (dict(max_recursion=6), dict(max_recursion=2, aa=1)):
        got = host.HostScene(sd).render(160, 90, ids=True, **kw)
        ref = sd.build(oracle_lib).render(160, 90, ids=True, threads=8, **kw)
        diff, ids_bad = assert_frame_parity(got, ref["pixels"], ref["shape"], ref["tri"], sd.name)
        assert diff == 0 and ids_bad == 0 and got["rays"] == ref["rays"]
        assert (got["tri"] >= 0).mean() > 0.2          # the meshes fill a good part of the view


def test_graph_replay_gives_the_same_frames_as_plain_launches(monkeypatch):
    """the frame's launch sequence is captured into a HIP graph and replayed; a changed camera / size / recursion
    depth re-captures.  Same pixels and ray counts as with RTHIP_GRAPH=0."""
    sd = scenes.config(3, cam_z=2.5)[0]
    hs = host.HostScene(sd)
    desc = hs.flatten()
    # (a graph is captured when the same frame is asked for twice in a row and replayed from the third time on)
    plans = [(256, 144, 3), (256, 144, 3), (256, 144, 3), (256, 144, 1), (192, 108, 3), (192, 108, 3), (192, 108, 3), (256, 144, 3)]
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RTHIP_GRAPH", mode)
        c = device.Context(0)
        c.upload(desc)
        res = []
        for k, (w, h, mr) in enumerate(plans):
            r = c.render(hs.camera(w, h), device.make_params(w, h, mr))
            st = c.stats()
            res.append((r["pixels"].copy(), st.rays_primary + st.rays_shadow + st.rays_reflect))
        c.close()
        out[mode] = res
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[1] > 0
    assert all(np.array_equal(out["1"][0][0], out["1"][k][0]) for k in (1, 2, 7)) and np.array_equal(out["1"][4][0], out["1"][6][0])


def test_graph_replay_follows_a_moving_camera(monkeypatch):
    """The cameras are not part of a captured graph (they reach the kernels through the context's page-locked camera ring):
    a camera that moves every frame - the Sandbox preview, Sandbox/Sandbox.cpp:315-359 - is served by ONE graph, captured at
    the second frame and replayed from then on; every frame equals the plain-launch render of its camera.  40 asynchronous
    calls are queued back to back (more than the ring's 16 slots), single frames and batches of 3."""
    import torch
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    desc = hs.flatten()
    w, h, n = 160, 96, 40
    prm = device.make_params(w, h, 3)
    cams = []
    for k in range(n):
        cam = hs.camera(w, h)
        cam.position[0] = -1.0 + 0.05 * k
        cam.position[1] = 0.02 * (k % 5)
        cam.position[2] = 2.4 + 0.01 * k
        cams.append(cam)
    monkeypatch.setenv("RTHIP_GRAPH", "0")
    plain = device.Context(0)
    plain.upload(desc)
    want = [plain.render(cam, prm)["pixels"].copy() for cam in cams]
    plain.close()
    assert len({x.tobytes() for x in want}) == n
    monkeypatch.setenv("RTHIP_GRAPH", "1")
    c = device.Context(0)
    try:
        c.upload(desc)
        # nothing waits between the calls (every call of a graph writes the buffer baked into it: one output buffer, copied out behind each call on the context's stream)
        ext = torch.cuda.ExternalStream(c.stream())
        out = torch.zeros(n * w * h, dtype=torch.int32, device="cuda")
        one = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        for k in range(n):
            c.render_device(cams[k], prm, one.data_ptr(), sync=False)
            with torch.cuda.stream(ext):
                out[k * w * h:(k + 1) * w * h].copy_(one, non_blocking=True)
        c.synchronize()
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint32).reshape(n, h, w)
        for k in range(n):
            assert np.array_equal(got[k], want[k]), f"frame {k} of the moving camera"
        st = c.stats()
        # (one graph - or two: the first calls launch both walk kernels per level, and once a call has reported its list lengths
        # only the kernel they are for is launched, which is another launch sequence; see render_frame: walk_mode)
        assert 1 <= st.graphs_captured <= 2 and st.graph_replays >= n - 4, (st.graphs_captured, st.graph_replays)
        captured, replays = st.graphs_captured, st.graph_replays
        # batches of 3 cameras: another frame shape, one more graph, replayed for every later run of 3
        out3 = torch.zeros(3 * w * h, dtype=torch.int32, device="cuda")
        for k0 in range(0, 12, 3):
            c.render_batch_device(cams[k0:k0 + 3], prm, out3.data_ptr(), sync=True)
            got3 = out3.cpu().numpy().view(np.uint32).reshape(3, h, w)
            for j in range(3):
                assert np.array_equal(got3[j], want[k0 + j]), f"batch frame {k0 + j}"
        st = c.stats()
        assert captured < st.graphs_captured <= captured + 2 and st.graph_replays >= replays + 1, (st.graphs_captured, st.graph_replays)
    finally:
        c.close()


def test_chunked_frames_are_identical(monkeypatch):
    """frames larger than the per-level workspace are rendered chunk by chunk (8K x SSAA in production); forcing a
    tiny chunk must not change a pixel or a ray count"""
    sd = scenes.config(3, cam_z=2.5)[0]
    hs = host.HostScene(sd)
    desc, cam = hs.flatten(), hs.camera(320, 180)
    out = {}
    for chunk in ("0", "6000"):
        monkeypatch.setenv("RTHIP_CHUNK_RAYS", chunk)
        c = device.Context(0)
        c.upload(desc)
        for aa in (0, 1):
            r = c.render(cam, device.make_params(320, 180, 3, aa=aa), ids=True)
            st = c.stats()
            out[chunk, aa] = (r["pixels"].copy(), r["shape"].copy(), r["tri"].copy(), st.rays_primary, st.rays_shadow, st.rays_reflect)
        c.close()
    for aa in (0, 1):
        a, b = out["0", aa], out["6000", aa]
        assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3:] == b[3:]
    assert out["0", 0][4] > 0 and out["0", 1][3] == 9 * out["0", 0][3]


def test_tile_partition_is_bit_identical_and_assembles(ctx):
    """multi-GPU path on one GPU: every part rendered on its own (frame layout and tile-compact layout),
    then rthip_assemble; the result must equal the single-part frame bit for bit."""
    import torch
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    ctx.upload(hs.flatten())
    w, h = 400, 232   # not a multiple of the 32x8 tile: exercises ragged edge tiles
    cam = hs.camera(w, h)
    full = ctx.render(cam, device.make_params(w, h, 3))["pixels"]
    for parts in (2, 3, 8):
        acc = np.zeros((h, w), np.uint32)
        for p in range(parts):
            ctx.render(cam, device.make_params(w, h, 3, part_index=p, part_count=parts), out=acc)
        assert np.array_equal(acc, full), f"{parts} parts, frame layout"
        prm = device.make_params(w, h, 3, part_count=parts, layout=1)
        n = device.part_pixels(prm)
        gathered = torch.zeros(parts * n, dtype=torch.int32, device="cuda")
        for p in range(parts):
            pp = device.make_params(w, h, 3, part_index=p, part_count=parts, layout=1)
            ctx.render_device(cam, pp, gathered.data_ptr() + 4 * n * p, sync=True)
        frame = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        ctx.assemble(prm, gathered.data_ptr(), frame.data_ptr(), sync=True)
        assert np.array_equal(frame.cpu().numpy().view(np.uint32).reshape(h, w), full), f"{parts} parts, gathered layout"
        # the host restatement of the layout (used by the gloo test) agrees with the device
        from raytracing_amd import parallel
        assert np.array_equal(parallel.assemble_parts(gathered.cpu().numpy().view(np.uint32).reshape(parts, n), w, h), full)


def test_frame_batches_are_bit_identical_to_single_frames(ctx):
    """rthip_render_batch_device: several cameras traced as one set of launches.  Every frame of the batch must equal
    the single-frame render of its camera bit for bit (frame layout, SSAA, tile-compact parts + rthip_assemble_batch,
    a chunked batch), and the ray statistics must add up."""
    import torch
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    ctx.upload(hs.flatten())
    w, h = 200, 120  # ragged against the 32x8 tiles
    cams = []
    for k in range(5):
        cam = hs.camera(w, h)
        cam.position[0] = -0.6 + 0.3 * k
        cam.position[2] = 2.2 + 0.4 * k
        cams.append(cam)
    for aa in (0, 1):
        prm = device.make_params(w, h, 3, aa=aa)
        singles, rays = [], 0
        for cam in cams:
            singles.append(ctx.render(cam, prm)["pixels"].copy())
            st = ctx.stats()
            rays += st.rays_primary + st.rays_shadow + st.rays_reflect
        assert any(not np.array_equal(singles[0], x) for x in singles[1:])
        buf = torch.zeros(len(cams) * w * h, dtype=torch.int32, device="cuda")
        for rep in range(2):  # second call replays the captured graph
            buf.zero_()
            ctx.render_batch_device(cams, prm, buf.data_ptr(), sync=True)
            got = buf.cpu().numpy().view(np.uint32).reshape(len(cams), h, w)
            for k in range(len(cams)):
                assert np.array_equal(got[k], singles[k]), f"aa={aa} frame {k} rep {rep}"
        st = ctx.stats()
        assert st.rays_primary + st.rays_shadow + st.rays_reflect == rays and st.pixels == len(cams) * w * h
    # parts + gather layout: [part][frame][slot] -> frames
    prm = device.make_params(w, h, 3)
    singles = [ctx.render(cam, prm)["pixels"].copy() for cam in cams[:3]]
    for parts in (2, 3):
        pl = device.make_params(w, h, 3, part_count=parts, layout=1)
        n = device.part_pixels(pl)
        gathered = torch.zeros(parts * 3 * n, dtype=torch.int32, device="cuda")
        for p in range(parts):
            pp = device.make_params(w, h, 3, part_index=p, part_count=parts, layout=1)
            ctx.render_batch_device(cams[:3], pp, gathered.data_ptr() + 4 * 3 * n * p, sync=True)
        frames = torch.zeros(3 * w * h, dtype=torch.int32, device="cuda")
        ctx.assemble_batch(pl, 3, gathered.data_ptr(), frames.data_ptr(), sync=True)
        got = frames.cpu().numpy().view(np.uint32).reshape(3, h, w)
        for k in range(3):
            assert np.array_equal(got[k], singles[k]), f"{parts} parts frame {k}"
    with pytest.raises(device.RthipError):
        ctx.render_batch_device(cams * 13, prm, buf.data_ptr(), sync=True)  # 65 > RTHIP_MAX_BATCH


def test_batch_of_forty_cameras_equals_forty_single_frames(ctx):
    """more cameras than one launch's kernel arguments hold (16): they reach the device through the context's camera array"""
    import torch
    hs = host.HostScene(scenes.config(3, cam_z=2.6)[0])
    ctx.upload(hs.flatten())
    w, h, n = 96, 56, 40
    prm = device.make_params(w, h, 2)
    cams = []
    for k in range(n):
        cam = hs.camera(w, h)
        cam.position[0] += 0.04 * k - 0.8
        cam.position[1] += 0.01 * (k % 7)
        cams.append(cam)
    buf = torch.zeros(n * w * h, dtype=torch.int32, device="cuda")
    ctx.render_batch_device(cams, prm, buf.data_ptr(), sync=True)
    got = buf.cpu().numpy().view(np.uint32).reshape(n, h, w)
    one = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    for k in (0, 1, 15, 16, 17, 31, 32, 39):
        ctx.render_device(cams[k], prm, one.data_ptr(), sync=True)
        assert np.array_equal(got[k], one.cpu().numpy().view(np.uint32).reshape(h, w)), f"frame {k}"
    assert len({got[k].tobytes() for k in range(n)}) == n


def test_chunked_frame_batch(monkeypatch):
    monkeypatch.setenv("RTHIP_CHUNK_RAYS", "7000")
    import torch
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    c = device.Context(0)
    c.upload(hs.flatten())
    w, h = 160, 96
    cams = []
    for k in range(3):
        cam = hs.camera(w, h)
        cam.position[1] = 0.2 * k
        cams.append(cam)
    prm = device.make_params(w, h, 2)
    buf = torch.zeros(3 * w * h, dtype=torch.int32, device="cuda")
    c.render_batch_device(cams, prm, buf.data_ptr(), sync=True)
    got = buf.cpu().numpy().view(np.uint32).reshape(3, h, w)
    for k in range(3):  # (single frames through the same chunked context: test_chunked_frames_are_identical covers those)
        assert np.array_equal(got[k], c.render(cams[k], prm)["pixels"])
    c.close()


def test_full_size_properties_config3(ctx):
    """BASELINE size (1920x1080, MaxRecursion 3): determinism, the reference's ray count, recursion-depth
    consistency, and a checksum of the frame that two different launch geometries must agree on."""
    hs = host.HostScene(scenes.config(3)[0])
    ctx.upload(hs.flatten())
    w, h = 1920, 1080
    cam = hs.camera(w, h)
    a = ctx.render(cam, device.make_params(w, h, 3), ids=True)
    st = ctx.stats()
    assert st.rays_primary == w * h
    # 4 589 295 rays per frame measured on the reference by wrapping Scene::CastRay
    assert st.rays_primary + st.rays_shadow + st.rays_reflect == 4589295
    b = ctx.render(cam, device.make_params(w, h, 3, tile_w=64, tile_h=16))
    assert np.array_equal(a["pixels"], b["pixels"])
    # pixels whose primary ray misses everything do not depend on MaxRecursion
    c = ctx.render(cam, device.make_params(w, h, 0))
    miss = a["shape"] < 0
    assert miss.any() and np.array_equal(a["pixels"][miss], c["pixels"][miss])
    assert ((a["pixels"] >> 24) == 0xff).all()
    # 6.4 % of the primary rays hit the bunny from this camera (132 285)
    assert int((a["tri"] >= 0).sum()) == int((a["shape"] == 1).sum()) > 100000


def test_repeated_renders_keep_the_device_geometry():
    """HipRaytracer renders frame after frame (the Sandbox preview): when only nodes / the camera moved, the second
    frame re-uploads the small tables and keeps the mesh geometry on the device (rthip_update_scene); the image must
    be the one a fresh renderer makes of the moved scene."""
    for mk in (lambda: scenes.config(3, cam_z=2.5)[0], lambda: scenes.config(5, cells=24)[0], lambda: scenes.config(2)[0]):
        hs = host.HostScene(mk())
        second, kept = hs.render_moved(200, 112, 0.35, max_recursion=3)
        assert kept, "the second frame shipped the geometry again"
        fresh = hs.render(200, 112, max_recursion=3)["pixels"]
        assert np.array_equal(second, fresh)
        again, kept = hs.render_moved(200, 112, -0.2, max_recursion=3)   # a new renderer: first frame full, second kept
        assert kept and np.array_equal(again, hs.render(200, 112, max_recursion=3)["pixels"])
        assert not np.array_equal(again, second)


def test_normal_mode_flipped_between_two_renders_reaches_the_device(oracle_lib):
    """Mesh::NormalMode is a public field the reference reads live per hit (Scene.cpp:456); flipping it does not invalidate
    the mesh.  The second Render through the same HipRaytracer must shade with the new mode (and still keep the geometry
    on the device): equal to a fresh renderer's frame of the flipped scene, different from the first frame."""
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    first = hs.render(200, 112, max_recursion=3)["pixels"]
    second, kept = hs.render_flip_normal_mode(200, 112, max_recursion=3)
    assert kept, "a NormalMode change alone must not ship the geometry again"
    fresh = hs.render(200, 112, max_recursion=3)["pixels"]
    assert np.array_equal(second, fresh)
    assert (second != first).sum() > 500, "face normals instead of vertex normals must change the bunny's shading"


def test_scene_with_a_mesh_that_has_no_bvh_is_refused_by_the_bvh_path(ctx):
    """a C-ABI caller that supplies only the kd-tree: accel = 0 would silently leave the mesh out of the image, so the
    render is refused; accel = 1 (kd-tree everywhere) renders it"""
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    full = C.cast(hs.flatten(), C.POINTER(device.SceneDesc)).contents
    assert full.n_meshes == 1 and full.meshes[0].tri_count == 4968 and full.meshes[0].bvh_node_count > 0
    kd_only = device.SceneDesc.from_buffer_copy(full)
    meshes = (device.Mesh * 1)(device.Mesh.from_buffer_copy(full.meshes[0]))
    meshes[0].bvh_node_count = 0
    kd_only.meshes = meshes
    ctx.upload(C.addressof(kd_only), build_bvh=False)   # (with RTHIP_UPLOAD_BUILD_BVH the device would build the missing BVH)
    cam = hs.camera(96, 54)
    with pytest.raises(device.RthipError) as e:
        ctx.render(cam, device.make_params(96, 54, 2))
    assert e.value.code == -1 and "no BVH" in str(e.value)
    got = ctx.render(cam, device.make_params(96, 54, 2, accel=1), ids=True)
    ctx.upload(hs.flatten())
    want = ctx.render(cam, device.make_params(96, 54, 2), ids=True)
    assert (want["tri"] >= 0).sum() > 300
    for k in ("pixels", "shape", "tri"):
        assert np.array_equal(got[k], want[k]), k


def test_skybox_sun_index_below_minus_one_is_rejected(ctx):
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    bad = device.SceneDesc.from_buffer_copy(C.cast(hs.flatten(), C.POINTER(device.SceneDesc)).contents)
    assert bad.background.type == 2
    bad.background.sun_light = -2
    with pytest.raises(device.RthipError) as e:
        ctx.upload(C.addressof(bad))
    assert e.value.code == -1 and "sun light" in str(e.value)


def test_update_scene_rejects_other_geometry(ctx):
    big = host.HostScene(scenes.config(3)[0])
    small = host.HostScene(scenes.config(4, cells=8)[0])
    ctx.upload(big.flatten())
    with pytest.raises(device.RthipError):
        ctx.update(small.flatten())          # another mesh table
    ctx.update(big.flatten())                # the same scene again: fine (pointers are ignored)
    w, h = 96, 54
    a = ctx.render(big.camera(w, h), device.make_params(w, h, 2))["pixels"].copy()
    ctx.upload(big.flatten())
    assert np.array_equal(a, ctx.render(big.camera(w, h), device.make_params(w, h, 2))["pixels"])


def test_probes_reject_indices_outside_the_scene(ctx):
    hs = host.HostScene(scenes.config(2)[0])
    d = hs.flatten()
    ctx.upload(d)
    pts = np.zeros((4, 3))
    for bad in (-1, 10 ** 6):
        with pytest.raises(device.RthipError):
            ctx.probe_noise(bad, pts)
        with pytest.raises(device.RthipError):
            ctx.probe_material(bad, pts)


def test_render_contract_async_status_interrupt():
    """AbstractRaycaster::Render contract (Raytracer.cpp:34-99): returns at once, status pollable, promise
    satisfied exactly once with Finished = true — also after Interrupt()."""
    hs = host.HostScene(scenes.config(2)[0])
    job = host.RenderJob(hs, 1920, 1080, max_recursion=3)
    st0 = job.status()
    assert st0["has_pixels"]
    px, st = job.wait()
    assert st["finished"] and not st["interrupted"] and st["percent"] == 1.0
    assert job.status()["finished"]
    ref = hs.render(1920, 1080, max_recursion=3)["pixels"]
    assert np.array_equal(px, ref)
    job.close()
    # Interrupt() right after Render(): the promise is kept, with Interruped set (Raytracer.cpp:76-79, 91-94)
    job = host.RenderJob(hs, 1920, 1080, max_recursion=8, aa=1)
    job.interrupt()
    px, st = job.wait()
    assert st["finished"] and st["interrupted"] and st["percent"] == 0.0
    job.close()
    # and the renderer is usable afterwards (a new Render starts uninterrupted, Raytracer.cpp:36)
    job = host.RenderJob(hs, 320, 180, max_recursion=3)
    px, st = job.wait()
    assert st["finished"] and not st["interrupted"] and np.array_equal(px, hs.render(320, 180, max_recursion=3)["pixels"])
    job.close()


def test_interrupt_and_progress_reach_a_frame_that_is_one_workspace_chunk():
    """Interrupt() / RenderStatus::Percent at kernel granularity (Raytracer.cpp:91-99, 150-154): a 4096x4096 frame is ONE
    workspace chunk (2^24 level-0 rays) whose ~70 launches are all queued before the first finishes; the device polls the
    interrupt word per workgroup iteration, so the frame comes back early, and the stage word moves Percent while it runs."""
    import threading
    hs = host.HostScene(scenes.config(4, cells=384)[0])
    ctx = device.Context(0)
    try:
        ctx.upload(hs.flatten())
        w = h = 4096
        cam, prm = hs.camera(w, h), device.make_params(w, h, 16)
        fb = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        ctx.render_device(cam, prm, fb.data_ptr(), sync=True)     # warm-up: workspace allocated
        seen = []
        stop = threading.Event()

        def poll():
            while not stop.is_set():
                seen.append(ctx.L.rthip_progress(ctx.h))
                time.sleep(0.0005)
        t = threading.Thread(target=poll); t.start()
        t0 = time.perf_counter()
        ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        full = time.perf_counter() - t0
        stop.set(); t.join()
        between = [p for p in seen if 0.0 < p < 1.0]
        assert len(set(between)) >= 3, f"Percent did not move during a {full * 1e3:.1f} ms frame: {sorted(set(seen))[:8]}"
        assert between == sorted(between), "Percent went backwards"
        want = fb.clone()
        # interrupt a quarter of the way in: the call returns RTHIP_E_INTERRUPTED well before a whole frame's time.  (The timing
        # half of this is measured against a host thread's sleep on a shared box, and the one stretch of this frame that is not
        # polled from inside - a redo-pass iteration of k_shade, 256 rays per workgroup through the reference's kd walk, ~5 ms
        # here - sits right behind the 25 % point: best of three attempts; code and pixel buffer are checked on every one.)
        timings = []
        for attempt in range(3):
            fb.fill_(0x11111111)
            result = {}

            def render():
                t1 = time.perf_counter()
                try:
                    ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
                    result["code"] = 0
                except device.RthipError as e:
                    result["code"] = e.code
                result["t"] = time.perf_counter() - t1
            t = threading.Thread(target=render); t.start()
            time.sleep(full * 0.25)
            ti = time.perf_counter()
            ctx.L.rthip_interrupt(ctx.h)
            t.join()
            after = time.perf_counter() - ti
            assert result["code"] == -4, result
            assert int((fb == 0x11111111).sum().item()) == w * h, "an abandoned frame must not be resolved into the pixel buffer"
            timings.append((result["t"], after))
            if result["t"] < 0.75 * full and after < 0.4 * full:
                break
        else:
            raise AssertionError(f"full frame {full * 1e3:.1f} ms; (interrupted frame, time after Interrupt()) in ms: "
                                 f"{[(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in timings]}")
        # the next frame starts uninterrupted and is the same frame as before
        ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        assert torch.equal(fb, want)
    finally:
        ctx.close()


def test_c_abi_error_reporting(ctx):
    hs = host.HostScene(scenes.config(1)[0])
    fresh = device.Context(0)
    cam = hs.camera(64, 48)
    with pytest.raises(device.RthipError) as e:
        fresh.render(cam, device.make_params(64, 48, 1))
    assert e.value.code == -5   # RTHIP_E_STATE: no scene uploaded
    fresh.upload(hs.flatten())
    for prm, code in ((device.make_params(64, 48, 99), -3), (device.make_params(64, 48, 1, part_index=2, part_count=2), -1),
                      (device.make_params(0, 48, 1), -1), (device.make_params(64, 48, 1, tile_w=12), -1),
                      (device.make_params(64, 48, 1, aa=7), -1)):
        with pytest.raises(device.RthipError) as e:
            fresh.render(cam, prm)
        assert e.value.code == code, (prm.width, prm.max_recursion, e.value)
    assert fresh.render(cam, device.make_params(64, 48, 1))["pixels"].shape == (48, 64)   # still usable afterwards
    with pytest.raises(device.RthipError):
        device.Context(1 << 20)
    fresh.close()


def test_hip_raytracer_raycast_answers_single_rays(golden):
    """re::HipRaytracer::Raycast(scene, ray) (the virtual every AbstractRaycaster has, Raytracer.h:78): one ray at a time through
    the renderer class - scene compiled, flattened and uploaded by the first call, kept for the following ones - gives the
    genuine Raytracer::Raycast's colours of the golden vectors (bunny + checker plane + SkyBox: mesh hits, plane, sky, bounces)."""
    k = golden["kat"]
    hs = host.HostScene(KAT_SCENES["cfg3"]())
    o, d = k["cfg3_ray_o"], k["cfg3_ray_d"]
    pick = np.linspace(0, len(o) - 1, 24).astype(int)
    got = hs.raycast(np.concatenate([o[pick], d[pick]], axis=1), 3)
    assert np.array_equal(got, k["cfg3_raycast"][pick])
    assert len(np.unique(got.round(6), axis=0)) > 5, "the picked rays all see the same thing"
