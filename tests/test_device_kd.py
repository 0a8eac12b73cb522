"""The REFERENCE kd-tree built ON THE DEVICE (rthip_upload_scene_ex + RTHIP_UPLOAD_BUILD_KD, raytracing_amd/csrc/rthip_kd.hip;
SURVEY.md §8f rank 3: "GPU-side KD build, same splits").  Unlike the BVH this tree takes part in results (accel = 1 walks it,
kd-degenerate rays and exact ties are settled in its order), so the device arrays must be BYTE-IDENTICAL to what the host
builder (rtbuild::KdBuilder, the restatement of KDTreeTriangle's constructor, Scene.cpp:18-95, pinned against the genuine
reference's tree in tests/test_host.py and tests/test_ref_binding.py) and the flattener produce: nodes, bounds, leaf references,
plane coordinates, depths."""
import numpy as np
import pytest

from conftest import FRAME_CASES, assert_frame_parity
from raytracing_amd import device, host, scenes

pytestmark = pytest.mark.gpu


def _mesh_scene(verts, name="kd mesh"):
    sd = scenes.SceneDesc(name)
    sd.camera((0, 0, 6))
    sd.mesh(sd.uniform_material(0xffffff, 0.9), verts, np.tile([0.0, 0, 1], (len(verts), 3)), scenes.NORMAL_FACE)
    sd.ambient_light(0xffffff)
    sd.color_background(0x102030)
    return sd


def _kd_of(monkeypatch, sd, on_device, mesh=0):
    monkeypatch.setenv("RE_KD_DEVICE_MIN_TRIS", "1" if on_device else "0")
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "0")   # (the BVH is not the subject here)
    hs = host.HostScene(sd)
    ctx = device.Context(0)
    try:
        ctx.upload(hs.flatten())
        info = ctx.upload_info()
        assert info.kd_meshes == (1 if on_device else 0)
        return ctx.kd(mesh), info
    finally:
        ctx.close()


def _assert_same_tree(dev, ref, what):
    (md, nd, bd, rd, pd), (mh, nh, bh, rh, ph) = dev, ref
    assert (md.node_count, md.ref_count, md.max_depth, list(md.plane_count)) == (mh.node_count, mh.ref_count, mh.max_depth, list(mh.plane_count)), what
    assert nd.tobytes() == nh.tobytes(), f"{what}: kd_nodes differ ({(nd != nh).sum()} of {len(nh)} records)"
    assert bd.tobytes() == bh.tobytes(), f"{what}: kd_bounds differ"
    assert rd.tobytes() == rh.tobytes(), f"{what}: kd_refs differ"
    assert pd.tobytes() == ph.tobytes(), f"{what}: kd_planes differ"


def test_device_kd_tree_of_the_bunny_is_the_reference_tree(monkeypatch, golden):
    sd = scenes.config(3)[0]
    dev, info = _kd_of(monkeypatch, sd, True)
    ref, _ = _kd_of(monkeypatch, sd, False)
    _assert_same_tree(dev, ref, "bunny")
    # 6 881 nodes, 26 456 references, depth 16 (the genuine reference's tree, tests/golden/bunny_kd.npz)
    assert (dev[0].node_count, dev[0].ref_count, dev[0].max_depth) == (6881, 26456, 16)
    assert info.kd_nodes == 6881 and info.kd_refs == 26456 and info.kd_levels == 17


@pytest.mark.parametrize("exact", ["1", "0"])
@pytest.mark.parametrize("cells", [24, 96, 384])
def test_device_kd_tree_of_height_fields(monkeypatch, cells, exact):
    """the config-4 mesh generator at 1 152, 18 432 and 294 912 triangles: x / z lie on a grid, so their sequential sums are exact
    and the builder takes them as integer reductions (RTHIP_KD_EXACT_SUMS=0: the dependent chain for every node, as for y)"""
    monkeypatch.setenv("RTHIP_KD_EXACT_SUMS", exact)
    sd = _mesh_scene(scenes.heightfield_mesh(cells, seed=1234)[0], f"height field {cells}")
    dev, info = _kd_of(monkeypatch, sd, True)
    ref, _ = _kd_of(monkeypatch, sd, False)
    _assert_same_tree(dev, ref, f"height field {cells}")
    assert dev[0].node_count >= 3


@pytest.mark.parametrize("seed", range(8))
def test_device_kd_tree_of_random_meshes(monkeypatch, seed):
    """fuzz: 2 - 3 000 random triangles of mixed sizes (big ones straddle many splits: the 1.5 n rule fires at every depth), some
    duplicated, some degenerate"""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 3000))
    c = rng.uniform(-4, 4, (n, 1, 3))
    size = rng.choice([0.02, 0.2, 2.0], (n, 1, 1), p=[0.6, 0.3, 0.1])
    v = (c + rng.normal(0, 1, (n, 3, 3)) * size).reshape(n, 9)
    if n > 40:
        v[5:15] = v[5]            # identical triangles
        v[20, 3:] = np.tile(v[20, :3], 2)   # a point
    sd = _mesh_scene(v, f"random mesh {seed}")
    dev, _ = _kd_of(monkeypatch, sd, True)
    ref, _ = _kd_of(monkeypatch, sd, False)
    _assert_same_tree(dev, ref, f"random mesh {seed} ({n} triangles)")


@pytest.mark.parametrize("bits", [12, 30, 37, 38, 39, 40, 41, 46, 52])
def test_device_kd_tree_on_grid_coordinates_around_the_exactness_limit(monkeypatch, bits):
    """coordinates that are integers of `bits` bits times 2^-7: the sequential sum of 3n of them is exact as long as it stays
    below 2^53 units - the builder's integer route - and rounds beyond (the chain).  Around the limit (3n = 9 000 is 14 bits:
    39 + 14 = 53) both routes must give the host builder's tree; zeros, negative zeros and a subnormal are mixed in."""
    rng = np.random.default_rng(bits)
    n = 3000
    k = rng.integers(-(1 << bits), 1 << bits, (n, 9), dtype=np.int64)
    k[:, 3:] = k[:, :3].repeat(2).reshape(n, 6)[:, [0, 2, 4, 1, 3, 5]] + rng.integers(-(1 << max(bits - 6, 1)), 1 << max(bits - 6, 1), (n, 6))  # small triangles
    v = k.astype(np.float64) * 2.0 ** -7
    v[7, 0] = 0.0; v[8, 1] = -0.0; v[9, 2] = 5e-324
    sd = _mesh_scene(v, f"grid coordinates, {bits} bits")
    dev, _ = _kd_of(monkeypatch, sd, True)
    ref, _ = _kd_of(monkeypatch, sd, False)
    _assert_same_tree(dev, ref, f"{bits}-bit grid")
    assert dev[0].node_count > 50


@pytest.mark.parametrize("kind", ["one", "two", "points_chain", "identical", "empty_side"])
def test_device_kd_tree_of_degenerate_meshes(monkeypatch, kind):
    """the stop conditions and the one-sided nodes: a single triangle (leaf at the root), two triangles, point triangles whose mean
    rounds BELOW their common coordinate (nothing goes left, everything right, on every axis: a chain of one-child nodes down
    to the depth cap of 100, Scene.cpp:25), identical triangles (2 n > 1.5 n: a leaf), a split that leaves one side empty"""
    tri = np.array([-0.5, -0.4, 0.0, 0.6, -0.3, 0.1, 0.0, 0.7, -0.1])
    if kind == "one":
        v = tri[None]
    elif kind == "two":
        v = np.stack([tri, tri + 1.5])
    elif kind == "points_chain":
        v = np.full((3, 9), 0.1)          # nine times 0.1 sums to 0.8999999999999999: the mean is below 0.1
    elif kind == "identical":
        v = np.tile(tri[None], (50, 1))
    else:
        v = np.concatenate([np.tile(tri[None], (3, 1)), (tri + 100.0)[None]])
    sd = _mesh_scene(v, "degenerate kd: " + kind)
    dev, info = _kd_of(monkeypatch, sd, True)
    ref, _ = _kd_of(monkeypatch, sd, False)
    _assert_same_tree(dev, ref, kind)
    if kind == "points_chain":
        assert dev[0].max_depth == 100 and dev[0].node_count == 101 and info.kd_levels == 101
    if kind == "one":
        assert dev[0].node_count == 1


@pytest.mark.parametrize("name", ["cfg3", "cfg3close", "nested"])
def test_frames_through_the_device_built_kd_tree_match_the_reference_golden(golden, monkeypatch, name):
    """re::HipRaytracer end to end with every mesh's kd-tree AND BVH built on the device, walked as the reference walks its tree
    (UseReferenceKdTree: accel = 1) and through the default BVH path (whose kd-degenerate rays and ties use the same tree)"""
    monkeypatch.setenv("RE_KD_DEVICE_MIN_TRIS", "1")
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "1")
    mk, w, h, kw = FRAME_CASES[name]
    f = golden["frames"]
    for use_kd in (True, False):
        hs = host.HostScene(mk())
        hs.set_accel(use_kd)
        got = hs.render(w, h, ids=True, **kw)
        diff, ids_bad = assert_frame_parity(got, f[name + "_pixels"], f[name + "_shape"].astype(np.int32), f[name + "_tri"], name)
        assert diff == 0 and ids_bad == 0 and got["rays"] == int(f[name + "_rays"][0])


def test_device_kd_scene_keeps_its_geometry_and_mixed_scenes_work(monkeypatch, oracle_lib):
    """a big mesh whose trees the device builds in front of a small mesh that brings its own (thresholds between the two sizes):
    the frame equals the oracle's through both acceleration structures, and the next frame through the same renderer keeps the
    device geometry (rthip_update_scene against a mesh table whose kd fields the upload filled in)"""
    from test_device_bvh import _two_mesh_scene
    sd = _two_mesh_scene()
    monkeypatch.setenv("RE_KD_DEVICE_MIN_TRIS", "1000")
    monkeypatch.setenv("RE_BVH_DEVICE_MIN_TRIS", "1000")
    want = sd.build(oracle_lib).render(160, 100, max_recursion=2, ids=True, threads=8)
    for use_kd in (False, True):
        hs = host.HostScene(sd)
        hs.set_accel(use_kd)
        got = hs.render(160, 100, max_recursion=2, ids=True)
        for k in ("pixels", "shape", "tri"):
            assert np.array_equal(got[k], want[k]), (use_kd, k)
    second, kept = hs.render_moved(160, 100, 0.25, max_recursion=2)
    assert kept and np.array_equal(second, hs.render(160, 100, max_recursion=2)["pixels"])


def test_a_mesh_without_kd_nodes_is_refused_without_the_build_flag(monkeypatch):
    monkeypatch.setenv("RE_KD_DEVICE_MIN_TRIS", "1")
    hs = host.HostScene(scenes.config(3)[0])
    ctx = device.Context(0)
    try:
        with pytest.raises(device.RthipError) as e:
            ctx.upload(hs.flatten(), build_kd=False)
        assert e.value.code == -1 and "RTHIP_UPLOAD_BUILD_KD" in str(e.value)
    finally:
        ctx.close()
