"""CPU: the oracle (plain-C restatement, oracle/rt_oracle.c) against the golden vectors generated from
the genuine reference (tools/make_golden.py), and against oracle/_ref live when it is built."""
import ctypes as C

import numpy as np
import pytest

from conftest import FRAME_CASES, KAT_SCENES, SPARSE_CASES
from raytracing_amd import scenes


@pytest.mark.parametrize("name", list(FRAME_CASES))
def test_oracle_frames_match_reference_golden(oracle_lib, golden, name):
    mk, w, h, kw = FRAME_CASES[name]
    r = mk().build(oracle_lib).render(w, h, ids=True, threads=4, **kw)
    f = golden["frames"]
    assert np.array_equal(r["pixels"], f[name + "_pixels"])       # bit-exact: same libm, same operation order
    assert np.array_equal(r["shape"], f[name + "_shape"])
    assert np.array_equal(r["tri"], f[name + "_tri"])
    assert r["rays"] == int(f[name + "_rays"][0])


@pytest.mark.parametrize("name", list(SPARSE_CASES))
def test_oracle_sparse_fullres_match_reference_golden(oracle_lib, golden, name):
    mk, w, h, kw = SPARSE_CASES[name]
    s = golden["sparse"]
    xs, ys = s[name + "_x"].astype(np.int32), s[name + "_y"].astype(np.int32)
    r = mk().build(oracle_lib).render(w, h, ids=True, sub=(xs, ys), **kw)
    assert np.array_equal(r["pixels"], s[name + "_pixels"])
    assert np.array_equal(r["shape"], s[name + "_shape"])
    assert np.array_equal(r["tri"], s[name + "_tri"])


@pytest.mark.parametrize("name", list(KAT_SCENES))
def test_oracle_known_answers(oracle_lib, golden, name):
    k = golden["kat"]
    sd = KAT_SCENES[name]()
    s = sd.build(oracle_lib)
    o, d, p = k[name + "_ray_o"], k[name + "_ray_d"], k[name + "_points"]
    c = s.cast_rays(o, d)
    for key in ("hit", "point", "local", "normal", "shape", "tri"):
        assert np.array_equal(c[key], k[name + "_cast_" + key]), key
    assert np.array_equal(s.raycast(o, d, 3), k[name + "_raycast"])
    assert np.array_equal(s.background_eval(d), k[name + "_background"])
    for ni in sd.noise_info:
        assert np.array_equal(s.noise_sample(ni, p), k[f"{name}_noise{ni}"]), sd.noise_info[ni]
    for mi in range(sd._n["mat"]):
        col, ab = s.material_eval(mi, p)
        assert np.array_equal(col, k[f"{name}_mat{mi}_color"]) and np.array_equal(ab, k[f"{name}_mat{mi}_abs"])
    for node in range(1, sd._n["node"]):
        assert np.array_equal(s.node_matrices(node), k[f"{name}_node{node}_matrices"])


def test_oracle_bunny_kdtree_matches_reference(oracle_lib, golden):
    kd = scenes.config(3)[0].build(oracle_lib).kd_dump(2)
    g = golden["bunny_kd"]
    assert len(kd["depth"]) == 6881 and len(kd["refs"]) == 26456 and kd["depth"].max() == 16
    for key in ("depth", "ntris", "split", "bounds", "refs"):
        assert np.array_equal(kd[key], g[key]), key


def test_oracle_vs_reference_live(oracle_lib, ref_lib):
    """container-only: a scene NOT in the fixtures, both libraries rendering it now."""
    sd, _ = scenes.nested_transforms()
    sd.camera((1.5, 2.5, 4), (-0.2, -0.3, -1))
    a = sd.build(oracle_lib).render(96, 64, max_recursion=5, ids=True, aa=1)
    b = sd.build(ref_lib).render(96, 64, max_recursion=5, ids=True, aa=1)
    assert np.array_equal(a["pixels"], b["pixels"]) and np.array_equal(a["shape"], b["shape"]) and a["rays"] == b["rays"]


def test_rand_tables_follow_reference_ctor_order(oracle_lib, ref_lib):
    """tables drawn by the ctors themselves (no injection): same libc rand() consumption order."""
    libc = C.CDLL(None)

    def tables(lib):
        libc.srand(1234)
        L = lib.lib
        h = L.sb_scene_new()
        L.sb_noise_perlin(h, 2.0, None)
        L.sb_noise_worley(h, 2.0, 3, None)
        L.sb_noise_marble(h, 2.0, 2.0, 4.0, None)
        from raytracing_amd.sbapi import BuiltScene
        s = BuiltScene(lib, h)
        return s.perlin_table(0), s.worley_points(1, 3), s.perlin_table(2)
    for x, y in zip(tables(oracle_lib), tables(ref_lib)):
        assert np.array_equal(x, y)


def test_powf_square_vs_float_product():
    """SkyBox's powf(x, 2.0f) (Scene.cpp:134) becomes the float product x*x on the device (a *libm* row of
    SURVEY.md §8a B2).  Over ALL 167 772 floats the border interval [0.98, 0.99) can produce, glibc 2.35's
    powf differs from the correctly rounded product on 117 inputs (0.07 %), each by one float ulp; a sample
    of the interval is re-checked here.  Effect on an 8-bit pixel: far inside the 1-LSB bar."""
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    thr = np.float32(0.98)
    border = (np.float32(1.0) - thr) / np.float32(2.0)
    f = thr + np.arange(0, 167772, 13, dtype=np.float32) * np.float32(2.0 ** -24)
    xs = ((f[f < thr + border] - thr) / border).astype(np.float32)
    bad = 0
    for x in xs:
        a, b = np.float32(libm.powf(float(x), 2.0)), x * x
        if a != b:
            bad += 1
            assert abs(int(a.view(np.int32)) - int(b.view(np.int32))) == 1
    assert bad <= len(xs) * 2e-3
