"""Exact distance ties between triangles: rays that run through the shared diagonal of axis-aligned
coplanar quads are accepted by BOTH triangles of the quad at bit-identical distance; the reference keeps the one its kd
walk tests first (Scene.cpp:469-493: leaves in pre-order, strict '<'), which is neither the lower triangle index nor a
property a pruned or reordered walk sees.  The two triangles of a quad carry different vertex normals here, so the winner
decides the pixel as well as the id.  The fast walks detect the tie and hand the query to the reference-order walk
(rt_core.hpp: mesh_closest_reference_order; from the quad kernel through the redo list)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
from raytracing_amd import device, host, scenes

W, H = 64, 256   # W a power of two: the column x = W / 2 has xFactor == 0 exactly
MAXREC = 1
NB = 16          # blocks per side, 2 x 2 units each, centred on the axis
LAYERS = 8


def tie_scene(seed=1):
    """A ground plane y = 0 tiled with 2 x 2 blocks: one BIG triangle on one side of the block's diagonal x = z, two SMALL
    ones on the other.  The camera sits above the diagonal plane x = z and looks along it, so every ray of the image's
    centre column stays in that plane bit for bit (direction.x == direction.z, origin.x == origin.z): it runs along the
    shared edges, is accepted by the big triangle and by a small one at the same distance, and crosses several kd cells on
    its way down.  The rivals have different bounding boxes, so the reference's walk may meet the big one in an earlier
    leaf than the small one.  LAYERS copies of the tiling one unit apart in depth keep the round-robin kd build splitting
    (a coplanar mesh stops at depth 2 with both rivals in one leaf, where the index order decides).  The mesh's triangle
    order is shuffled."""
    sd = scenes.SceneDesc(f"tie scene {seed}")
    mat = sd.uniform_material(0xc0c0c0, 1.0)
    sd.directional_light(0xffffff, -0.3, -1.0, -0.5)
    sd.color_background(0x203040)
    tris, nrm = [], []
    for layer in range(LAYERS):
        y = -float(layer)
        for bz in range(NB):
            for bx in range(NB):
                x0, z0 = 2.0 * (bx - NB // 2), 2.0 * (bz - NB // 2)
                tris.append((x0, y, z0, x0 + 2, y, z0 + 2, x0 + 2, y, z0))            # big
                nrm.append((0.3, 1.0, 0.0) * 3)
                tris.append((x0, y, z0, x0, y, z0 + 1, x0 + 1, y, z0 + 1))            # small, first half of the diagonal
                nrm.append((-0.3, 1.0, 0.0) * 3)
                tris.append((x0 + 1, y, z0 + 1, x0 + 1, y, z0 + 2, x0 + 2, y, z0 + 2))  # small, second half
                nrm.append((0.0, 1.0, 0.3) * 3)
    order = np.arange(len(tris))
    g = scenes.PCG32(seed)
    for k in range(len(order) - 1, 0, -1):   # Fisher-Yates: the index order says nothing about the kd order
        j = g.next_u32() % (k + 1)
        order[k], order[j] = order[j], order[k]
    v = np.array(tris, dtype=np.float64)[order]
    n = np.array(nrm, dtype=np.float64)[order]
    sd.mesh(mat, v, n, scenes.NORMAL_VERTEX, node=sd.node(-1, (0.0, 0.0, 0.0), (0, 0, 0), (1, 1, 1)))
    sd.look_at((9.0, 2.5, 9.0), (0.0, 0.0, 0.0))
    return sd, order


def test_the_scene_has_exact_ties_and_the_reference_does_not_keep_the_lower_index(oracle_lib):
    """the premise: down the image's centre column the oracle's winner is the big triangle for some rays and a small one
    for others, and it is the HIGHER mesh index of the two rivals for a good share of them"""
    higher = ties = bigs = 0
    x = W // 2
    for seed in (1, 2, 3):
        sd, order = tie_scene(seed)
        ref = sd.build(oracle_lib).render(W, H, ids=True, threads=4, max_recursion=MAXREC)
        pos = np.empty_like(order); pos[order] = np.arange(len(order))   # original triangle -> index in the mesh
        for y in range(H):
            t = int(ref["tri"][y, x])
            if t < 0:
                continue   # above the horizon
            orig = int(order[t])
            if orig >= 3 * NB * NB:
                continue   # (beyond the tiling's edge a ray can reach a lower layer)
            blk, kind = divmod(orig, 3)
            assert blk % NB == blk // NB, "the ray should be on the diagonal of a diagonal block"
            if kind == 0:   # its rival: the small triangle the neighbouring column sees
                near = [int(order[int(ref["tri"][y, xx])]) for xx in (x - 1, x + 1) if ref["tri"][y, xx] >= 0]
                near = [o for o in near if o // 3 == blk and o % 3 != 0]
                if not near:
                    continue
                rival = pos[near[0]]
            else:
                rival = pos[3 * blk]
            ties += 1
            bigs += int(kind == 0)
            higher += int(rival < t)
    assert ties > 150 and bigs > 20 and ties - bigs > 20 and higher > 10, (ties, bigs, higher)   # `higher`: the lower index lost


def test_oracle_matches_the_genuine_reference_on_ties(oracle_lib, ref_lib):
    for seed in (1, 2, 3):
        sd, _ = tie_scene(seed)
        a = sd.build(oracle_lib).render(W, H, ids=True, threads=4, max_recursion=MAXREC)
        b = sd.build(ref_lib).render(W, H, ids=True, threads=4, max_recursion=MAXREC)
        assert np.array_equal(a["tri"], b["tri"]) and np.array_equal(a["shape"], b["shape"])
        assert np.array_equal(a["pixels"], b["pixels"]) and a["rays"] == b["rays"]


@pytest.mark.parametrize("accel", [0, 1])
def test_host_core_resolves_ties_like_the_oracle(host_lib, oracle_lib, accel):
    hc = C.CDLL(os.path.join(ROOT, "tests", "hostcore", "libhostcore.so"))
    for seed in (1, 2, 3):
        sd, _ = tie_scene(seed)
        ref = sd.build(oracle_lib).render(W, H, ids=True, threads=4, max_recursion=MAXREC)
        hc.hc_set_accel(accel)
        hs = host.HostScene(sd)
        cam = hs.camera(W, H)
        prm = device.make_params(W, H, MAXREC)
        px = np.zeros((H, W), np.uint32); sh = np.zeros((H, W), np.int32); tr = np.zeros((H, W), np.int32)
        rays = C.c_uint64(0)
        vp = C.c_void_p
        assert hc.hc_render(vp(hs.flatten()), C.byref(cam), C.byref(prm), px.ctypes.data_as(vp), sh.ctypes.data_as(vp),
                            tr.ctypes.data_as(vp), C.byref(rays), None, 4) == 0
        hc.hc_set_accel(0)
        assert np.array_equal(tr, ref["tri"]), f"seed {seed}: {(tr != ref['tri']).sum()} triangle ids differ"
        assert np.array_equal(px, ref["pixels"]), f"seed {seed}: {(px != ref['pixels']).sum()} pixels differ"
        assert rays.value == ref["rays"]


@pytest.mark.gpu
@pytest.mark.parametrize("use_kd", [False, True])
def test_device_resolves_ties_like_the_oracle(oracle_lib, use_kd):
    for seed in (1, 2, 3):
        for min_tris in ("0", "1"):   # host-built and device-built BVH
            os.environ["RE_BVH_DEVICE_MIN_TRIS"] = min_tris
            try:
                sd, _ = tie_scene(seed)
                ref = sd.build(oracle_lib).render(W, H, ids=True, threads=4, max_recursion=MAXREC)
                hs = host.HostScene(sd)
                hs.set_accel(use_kd)
                got = hs.render(W, H, ids=True, max_recursion=MAXREC)
            finally:
                os.environ.pop("RE_BVH_DEVICE_MIN_TRIS", None)
            assert np.array_equal(got["shape"], ref["shape"])
            assert np.array_equal(got["tri"], ref["tri"]), f"seed {seed} kd={use_kd} device-bvh={min_tris}: {(got['tri'] != ref['tri']).sum()} triangle ids differ"
            assert np.array_equal(got["pixels"], ref["pixels"]), f"seed {seed} kd={use_kd}: {(got['pixels'] != ref['pixels']).sum()} pixels differ"
            assert got["rays"] == ref["rays"]
