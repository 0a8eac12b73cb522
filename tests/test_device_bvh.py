"""The 4-wide BVH built ON THE DEVICE (rthip_upload_scene_ex + RTHIP_UPLOAD_BUILD_BVH, raytracing_amd/csrc/rthip_bvh.hip;
SURVEY.md §8f rank 3).  The BVH takes no part in a result's arithmetic, so frames must stay bit-identical to the goldens of the
genuine reference whichever builder made it; the structure itself must be complete (every triangle in exactly one leaf) and
conservative (every triangle inside all the outward-rounded fp32 boxes on its path)."""
import numpy as np
import pytest

from conftest import FRAME_CASES, assert_frame_parity
from raytracing_amd import device, host, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["sah", "ploc"])
def builder(request, monkeypatch):
    """both device builders of the binary tree: top-down binned SAH (the default) and PLOC clustering"""
    monkeypatch.setenv("RTHIP_BVH_BUILDER", request.param)
    return request.param


def _check_bvh(mesh, nodes, index, verts):
    n_tris = len(index)
    assert mesh.bvh_node_count == len(nodes) and mesh.tri_count == n_tris
    assert np.array_equal(np.sort(index), np.arange(n_tris, dtype=np.uint32))   # a permutation: no triangle lost or duplicated
    seen = np.zeros(n_tris, bool)
    stack = [(0, np.full(3, -np.inf), np.full(3, np.inf), 0)]
    visited, deepest = 0, 0
    while stack:
        ni, lo, hi, depth = stack.pop()
        visited += 1
        deepest = max(deepest, depth)
        n_children = 0
        for ch in nodes[ni]["c"]:
            c, blo, bhi = int(ch["child"]), ch["lo"], ch["hi"]
            if c == 0xffffffff:
                continue
            n_children += 1
            clo, chi = np.maximum(lo, blo.astype(np.float64)), np.minimum(hi, bhi.astype(np.float64))
            if c & 0x80000000:
                first, cnt = c & 0x0fffffff, ((c >> 28) & 7) + 1
                assert cnt <= 4 and first + cnt <= n_tris
                t = index[first:first + cnt]
                assert not seen[t].any()
                seen[t] = True
                assert (verts[t] >= clo).all() and (verts[t] <= chi).all(), "triangle sticks out of its boxes"
            else:
                assert ni < c < len(nodes)                                     # children follow their parent
                stack.append((c, clo, chi, depth + 1))
        assert n_children >= 1
    assert visited == len(nodes) and seen.all()
    assert deepest == mesh.bvh_depth <= 48
    assert mesh.abs_max >= np.abs(verts).max()


def test_device_built_bvh_of_the_bunny_is_complete_and_conservative(monkeypatch, builder):
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "1")
    hs = host.HostScene(scenes.config(3)[0])
    ctx = device.Context(0)
    try:
        ctx.upload(hs.flatten())
        m, nodes, index = ctx.bvh(0)
        assert 400 < len(nodes) < 4968
        _check_bvh(m, nodes, index, scenes.load_bunny()[0].reshape(-1, 3, 3))
    finally:
        ctx.close()


def test_device_built_bvh_of_a_height_field_and_of_tiny_meshes(builder):
    v, _ = scenes.heightfield_mesh(384, seed=1234)      # 294 912 triangles: above the host library's 2^18 limit by default
    hs = host.HostScene(scenes.config(4, cells=384)[0])
    ctx = device.Context(0)
    try:
        ctx.upload(hs.flatten())
        m, nodes, index = ctx.bvh(0)
        _check_bvh(m, nodes, index, v.reshape(-1, 3, 3))
    finally:
        ctx.close()


def _leaf_sets(nodes, index):
    ch = nodes["c"]["child"].ravel()
    leaf = ch[((ch & 0x80000000) != 0) & (ch != 0xffffffff)]
    first, cnt = leaf & 0x0fffffff, ((leaf >> 28) & 7) + 1
    return sorted(tuple(sorted(int(t) for t in index[f:f + c])) for f, c in zip(first, cnt))


def test_device_sah_tree_is_the_same_from_run_to_run(monkeypatch):
    """the SAH builder's partition lands in the order of its atomics; the tree must not depend on it (a height field: equal
    centroid coordinates and equal costs everywhere) - same leaves, same number of nodes, same depth in two builds"""
    monkeypatch.setenv("RTHIP_BVH_BUILDER", "sah")
    hs = host.HostScene(scenes.config(4, cells=384)[0])
    desc = hs.flatten()
    got = []
    for _ in range(2):
        ctx = device.Context(0)
        try:
            ctx.upload(desc)
            m, nodes, index = ctx.bvh(0)
            got.append((len(nodes), m.bvh_depth, _leaf_sets(nodes, index)))
        finally:
            ctx.close()
    assert got[0][0] == got[1][0] and got[0][1] == got[1][1]
    assert got[0][2] == got[1][2]


@pytest.mark.parametrize("n_tris", [1, 2, 4, 5, 9, 64, 65, 131])
def test_device_built_bvh_of_meshes_of_a_few_triangles(monkeypatch, oracle_lib, builder, n_tris):
    """(also the regression test of the quad walk's stack size: a BVH that is a lone root pushes up to three children at once;
    the large overlapping triangles here make that happen, and 3 * depth + 1 entries lost hits in both builders' trees)"""
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "1")
    rng = np.random.default_rng(n_tris)
    v = rng.uniform(-2, 2, (n_tris, 9))
    if n_tris > 64:   # a block of identical triangles: coincident centroids, no plane separates them (split by position)
        v[10:50] = v[10]
    sd = scenes.SceneDesc("few triangles")
    sd.camera((0, 0, 6))
    sd.mesh(sd.uniform_material(0xffffff, 0.9), v, np.tile([0.0, 0, 1], (n_tris, 3)), scenes.NORMAL_FACE)
    sd.ambient_light(0xffffff)
    sd.color_background(0x102030)
    hs = host.HostScene(sd)
    ctx = device.Context(0)
    try:
        ctx.upload(hs.flatten())
        m, nodes, index = ctx.bvh(0)
        _check_bvh(m, nodes, index, v.reshape(-1, 3, 3))
        got = ctx.render(hs.camera(96, 64), device.make_params(96, 64, 1), ids=True)
    finally:
        ctx.close()
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "0")
    host_built = host.HostScene(sd).render(96, 64, max_recursion=1, ids=True)
    want = sd.build(oracle_lib).render(96, 64, max_recursion=1, ids=True, threads=4)
    assert n_tris < 4 or (want["tri"] >= 0).sum() > 200   # (a lone random triangle may face away: back-face culled)
    for k in ("pixels", "shape", "tri"):
        assert np.array_equal(got[k], want[k]), f"device-built BVH: {k}"
        assert np.array_equal(host_built[k], want[k]), f"host-built BVH: {k}"


@pytest.mark.parametrize("kind", ["identical", "collinear", "flat", "huge", "two_clumps"])
def test_device_built_bvh_of_degenerate_meshes(monkeypatch, oracle_lib, builder, kind):
    """inputs that leave a builder without a plane to choose: coincident centroids (split by position all the way down), centroids
    on one line, a mesh without extent on an axis, coordinates whose box areas overflow fp32, two far clumps of identical triangles"""
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "1")
    rng = np.random.default_rng(7)
    n = 700
    tri = np.array([[-0.5, -0.4, 0.0], [0.6, -0.3, 0.1], [0.0, 0.7, -0.1]])
    if kind == "identical":
        v = np.tile(tri.reshape(1, 9), (n, 1))
    elif kind == "collinear":
        v = (tri[None] * 0.2 + np.linspace(-3, 3, n)[:, None, None] * np.array([1.0, 0.5, 0.25])[None, None]).reshape(n, 9)
    elif kind == "flat":
        c = rng.uniform(-3, 3, (n, 1, 3)); c[:, :, 2] = 0.0
        t = tri.copy(); t[:, 2] = 0.0
        v = (t[None] * 0.3 + c).reshape(n, 9)
    elif kind == "huge":
        v = ((tri[None] * 0.4 + rng.uniform(-3, 3, (n, 1, 3))) * 1e25).reshape(n, 9)
    else:
        v = np.tile(tri.reshape(1, 9), (n, 1))
        v[n // 3:] += np.tile([40.0, 0.0, 0.0], 3)
    sd = scenes.SceneDesc("degenerate mesh: " + kind)
    sd.camera((0, 0, 9e25 if kind == "huge" else 9))
    sd.mesh(sd.uniform_material(0xffffff, 0.9), v, np.tile([0.0, 0, 1], (n, 3)), scenes.NORMAL_FACE)
    sd.ambient_light(0xffffff)
    sd.color_background(0x102030)
    hs = host.HostScene(sd)
    ctx = device.Context(0)
    try:
        ctx.upload(hs.flatten())
        m, nodes, index = ctx.bvh(0)
        _check_bvh(m, nodes, index, v.reshape(-1, 3, 3))
        got = ctx.render(hs.camera(96, 64), device.make_params(96, 64, 1), ids=True)
    finally:
        ctx.close()
    want = sd.build(oracle_lib).render(96, 64, max_recursion=1, ids=True, threads=4)
    assert (want["tri"] >= 0).sum() > 20
    for k in ("pixels", "shape", "tri"):
        assert np.array_equal(got[k], want[k]), f"{kind}: {k}"


@pytest.mark.parametrize("name", ["cfg3", "cfg3close", "nested", "nested_ssaa"])
def test_frames_with_device_built_bvhs_match_the_reference_golden(golden, monkeypatch, name):
    """re::HipRaytracer end to end with EVERY mesh's BVH built on the device: the golden frames of the genuine reference"""
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "1")
    mk, w, h, kw = FRAME_CASES[name]
    got = host.HostScene(mk()).render(w, h, ids=True, **kw)
    f = golden["frames"]
    diff, ids_bad = assert_frame_parity(got, f[name + "_pixels"], f[name + "_shape"].astype(np.int32), f[name + "_tri"], name)
    assert diff == 0 and ids_bad == 0 and got["rays"] == int(f[name + "_rays"][0])


def test_host_and_device_built_bvhs_give_the_same_frame_and_geometry_is_kept(monkeypatch):
    sd = scenes.config(5, cells=96)[0]     # five rotated / scaled meshes of 18 432 triangles + plane + spheres
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "0")
    want = host.HostScene(sd).render(320, 180, max_recursion=4, ids=True)
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "1")
    hs = host.HostScene(sd)
    got = hs.render(320, 180, max_recursion=4, ids=True)
    for k in ("pixels", "shape", "tri"):
        assert np.array_equal(got[k], want[k]), k
    assert got["rays"] == want["rays"]
    second, kept = hs.render_moved(320, 180, 0.3, max_recursion=4)   # the next frame keeps the device geometry, device-built BVHs included
    assert kept and np.array_equal(second, hs.render(320, 180, max_recursion=4)["pixels"])


def _two_mesh_scene():
    """a height field (18 432 triangles) FIRST, then a small mesh (200 triangles) above it, plane, sun"""
    big, big_n = scenes.heightfield_mesh(96, seed=5)
    rng = np.random.default_rng(11)
    c = rng.uniform(-2.5, 2.5, (200, 1, 3)) + np.array([0.0, 2.5, 0.0])
    tri = np.array([[-0.3, 0.0, 0.25], [0.3, 0.0, 0.25], [0.0, 0.0, -0.35]])   # counter-clockwise seen from +y
    small = (tri[None] + c).reshape(200, 9)
    sd = scenes.SceneDesc("two meshes: device-built BVH first, host-built BVH second")
    sd.camera((0, 6, 12), (0, -0.45, -1))
    sd.mesh(sd.uniform_material(0x88aa66, 0.8), big, big_n, scenes.NORMAL_VERTEX)
    sd.mesh(sd.uniform_material(0xff4422, 0.6), small, np.tile([0.0, 1, 0], (200, 3)), scenes.NORMAL_FACE)
    sd.directional_light(0xffffff, 1, 2, 1)
    sd.ambient_light(0x202020)
    sd.color_background(0x102030)
    return sd


def test_mixed_scene_big_mesh_device_bvh_small_mesh_host_bvh_keeps_geometry(monkeypatch, oracle_lib):
    """RE_BVH_DEVICE_MIN_TRIS between the two mesh sizes, the big mesh first: the upload rewrites the host-BVH mesh's
    bvh_tri_base; the next frame through the same renderer (rthip_update_scene) must still recognise the mesh table (round-2
    advisor finding: every second frame failed with 'mesh table differs'), and plain rthip_upload_scene_ex without the build
    flag must not read past the caller's short bvh_tris array."""
    sd = _two_mesh_scene()
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "1000")
    hs = host.HostScene(sd)
    want = sd.build(oracle_lib).render(160, 100, max_recursion=2, ids=True, threads=8)
    got = hs.render(160, 100, max_recursion=2, ids=True)
    assert (want["shape"] == 0).sum() > 500 and (want["shape"] == 1).sum() > 50
    for k in ("pixels", "shape", "tri"):
        assert np.array_equal(got[k], want[k]), k
    second, kept = hs.render_moved(160, 100, 0.25, max_recursion=2)
    assert kept, "the second frame re-uploaded the geometry"
    assert np.array_equal(second, hs.render(160, 100, max_recursion=2)["pixels"])
    third, kept = hs.render_moved(160, 100, 0.25, max_recursion=2)   # and again (the failure alternated)
    assert kept and np.array_equal(third, hs.render(160, 100, max_recursion=2)["pixels"])
    # the same flattened (mixed) scene WITHOUT the build flag: refused at render time for accel = 0 (a mesh without BVH), fine through the kd-tree
    ctx = device.Context(0)
    try:
        ctx.upload(hs.flatten(), build_bvh=False)
        with pytest.raises(device.RthipError):
            ctx.render(hs.camera(160, 100), device.make_params(160, 100, 2))
        kd = ctx.render(hs.camera(160, 100), device.make_params(160, 100, 2, accel=1), ids=True)
        o = np.array([[0.0, 6.0, 12.0]]); d = np.array([[0.0, -0.45, -1.0]]) / np.linalg.norm([0.0, -0.45, -1.0])
        probe = ctx.probe_cast(o, d)   # (probes take the kd-tree when a mesh has no BVH instead of silently missing it)
    finally:
        ctx.close()
    assert np.array_equal(kd["pixels"], hs.render(160, 100, max_recursion=2)["pixels"])
    assert probe["hit"][0] == 1 and probe["shape"][0] in (0, 1)
