"""Scene inputs: the product's Wavefront OBJ reader, scene-script evaluator and scenes.xml
reader (raytracing_amd/host/scene_io.cpp) against the behaviour of Sandbox/WavefrontLoader.cpp and
Sandbox/Sandbox.cpp:578-763.  CPU tests use the host-compiled device core; the GPU test renders through
re::HipRaytracer and checks against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, assert_frame_parity
from raytracing_amd import device, host, scenes

REF_RES = "/root/reference/Sandbox/res"

# the scene of scenes._demo1 (the reference's "Demo 1 - Spheres" plus BASELINE's Perlin sphere), written for this test
DEMO1_SCRIPT = """
-- materials
red = reUniformMaterial(0xf44336, 0.6)  green = reUniformMaterial(0x4CAF50, 0.6)
blue = reUniformMaterial(0x2196F3, 0.6);
local dark_red = reUniformMaterial(0x8d1007, 0.9)
dark_green = reUniformMaterial(0x265728, 0.9)
dark_blue = reUniformMaterial(0x074b83, 0.9)
glass_black = reUniformMaterial(0x000000, 0.8)
glass_white = reUniformMaterial(0xffffff, 0.4)
mirror = reUniformMaterial(0xffffff, 0.0)   solid_black = reUniformMaterial(0, 1)
reCameraPos(0, 3, 4)
sun = reDirectionalLight(0xffddee, 1, 1, -1)
reSkyBox(0x4444ff, 0xffffff, sun);
--[[ noises: construction order matters,
     every constructor draws from rand() ]]
marble = reMarble(2, 2, 4)
worley = reWorley(2, 10)
rePosition(-4, 0, -4) reSphere(red)
rePosition(0, 0, -4) reSphere(green)
rePosition(4, 0, -4) reSphere(blue)
rePosition(-4, 0, -8) reSphere(reInterpolatedMaterial(marble, dark_red, red))
rePosition(0, 0, -8) reSphere(reInterpolatedMaterial(marble, dark_green, green))
rePosition(4, 0, -8) reSphere(reInterpolatedMaterial(marble, dark_blue, blue))
rePosition(-4, 0, -12) reSphere(reInterpolatedMaterial(worley, dark_red, red))
rePosition(0, 0, -12) reSphere(reInterpolatedMaterial(worley, dark_green, green))
rePosition(4, 0, -12) reSphere(reInterpolatedMaterial(worley, dark_blue, blue))
perlin = rePerlin(2)
rePosition(0, 2.5, -8) reSphere(reInterpolatedMaterial(perlin, dark_blue, glass_white))
rePosition(0, -1, 0) rePlane(reInterpolatedMaterial(reCheckerBoard(16), glass_black, glass_white), 0, 1, 0)
"""


def _demo1_desc_with_tables_of(hs):
    sd = scenes.config(2)[0]
    sd.adopt_tables(hs.built)
    return sd


def _hostcore_frame(hs, w, h, max_recursion):
    L = C.CDLL(os.path.join(ROOT, "tests", "hostcore", "libhostcore.so"))
    vp = C.c_void_p
    L.hc_render.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.hc_set_accel(0)
    cam = hs.camera(w, h)
    prm = device.make_params(w, h, max_recursion)
    px = np.zeros((h, w), np.uint32); sh = np.zeros((h, w), np.int32); tr = np.zeros((h, w), np.int32)
    rays = C.c_uint64(0)
    assert L.hc_render(vp(hs.flatten()), C.byref(cam), C.byref(prm), px.ctypes.data_as(vp), sh.ctypes.data_as(vp),
                       tr.ctypes.data_as(vp), C.byref(rays), None, 8) == 0
    return px, sh, tr, rays.value


def test_obj_reader_face_formats_groups_and_errors(host_lib, tmp_path):
    obj = tmp_path / "m.obj"
    obj.write_text("# comment\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1.5e0\nvt 0.5 0.25\nvn 0 0 1\nvn 0 1 0\n"
                   "f 1 2 3\n"                    # v            -> group 'default'
                   "g second\nf 1/1 2/1 4/1\n"     # v/vt
                   "f 1//1 3//2 4//2\n"            # v//vn
                   "f 2/1/2 3/1/1 4/1/1 1/1/1\n")  # v/vt/vn, a quad: only the first three vertices become the triangle
    v, n = host.load_obj(str(obj), "default")
    assert v.tolist() == [[0, 0, 0, 1, 0, 0, 0, 1, 0]] and not n.any()       # missing normals are zero vectors
    v, n = host.load_obj(str(obj), "second")
    assert v.tolist() == [[0, 0, 0, 1, 0, 0, 0, 0, 1.5], [0, 0, 0, 0, 1, 0, 0, 0, 1.5], [1, 0, 0, 0, 1, 0, 0, 0, 1.5]]
    assert n.tolist() == [[0] * 9, [0, 0, 1, 0, 1, 0, 0, 1, 0], [0, 1, 0, 0, 0, 1, 0, 0, 1]]
    with pytest.raises(RuntimeError, match="Invalid obj group"):
        host.load_obj(str(obj), "nope")
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")          # the reference indexes past its vertex list here (UB): an error instead
    with pytest.raises(RuntimeError, match="line 2.*out of range"):
        host.load_obj(str(bad), "default")
    with pytest.raises(RuntimeError, match="cannot open"):
        host.load_obj(str(tmp_path / "missing.obj"), "default")


@pytest.mark.skipif(not os.path.exists(REF_RES + "/bunny.obj_"), reason="reference checkout not present")
def test_obj_reader_reads_the_reference_bunny_like_the_reference(host_lib):
    """tests/golden/bunny_mesh.npz was written through the genuine reference's LoadWavefront (tools/make_bunny_fixture.py)"""
    v, n = host.load_obj(REF_RES + "/bunny.obj_", "bunny")
    gv, gn = scenes.load_bunny()
    assert v.shape == (4968, 9) and np.array_equal(v, gv.reshape(-1, 9)) and np.array_equal(n, gn.reshape(-1, 9))


def test_scene_script_builds_the_same_scene_as_the_api_calls(host_lib):
    """script -> re::Scene -> flatten -> device core (host build) == the same scene built call by call with the same
    noise tables; Perlin / Worley / SkyBox tables come from rand() in construction order like the reference's"""
    libc = C.CDLL(None)
    libc.srand(1)
    hs = host.HostScene.from_script(DEMO1_SCRIPT)
    libc.srand(1)
    want_sky = np.array([libc.rand() % 256 for _ in range(256)], np.uint8)   # SkyBox's Perlin is the first rand() consumer
    assert np.array_equal(hs.built.skybox_table(), want_sky)
    ref = host.HostScene(_demo1_desc_with_tables_of(hs))
    a = _hostcore_frame(hs, 96, 54, 3)
    b = _hostcore_frame(ref, 96, 54, 3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3]
    assert len(np.unique(a[0])) > 500                                       # a real image, not a constant


def test_scene_script_errors_name_the_line(host_lib):
    for code, msg in (("m = reUniformMaterial(1, 0.5)\nreSphere(3)", "line 2: Invalid material: 3"),
                      ("reSkyBox(1, 2, 0)", "Invalid light: 0"),
                      ("x = reNope(1)", r"attempt to call a nil value \(global 'reNope'\)"),
                      ("reSphere(undefined_name)", "variable 'undefined_name' is nil"),
                      ("rePosition(1, 2)", "rePosition expects 3 argument"),
                      ("m = reUniformMaterial(1, 0.5)\nreObjMesh(m, \"/nonexistent.obj\", \"g\")", "cannot open"),
                      ("reSphere(", "unexpected symbol")):
        with pytest.raises(RuntimeError, match=msg):
            host.HostScene.from_script(code)


def test_scenes_xml_reader(host_lib, tmp_path):
    x = tmp_path / "scenes.xml"
    x.write_text('<scenes>\n  <scene name="A &amp; B">m = reUniformMaterial(0xff, 1)\nreSphere(m) -- a &lt; b\n</scene>\n'
                 '  <scene name="two">reCameraPos(0, 0, 1)</scene>\n</scenes>\n')
    got = host.scenes_xml(str(x))
    assert [n for n, _ in got] == ["A & B", "two"]
    assert got[0][1] == "m = reUniformMaterial(0xff, 1)\nreSphere(m) -- a < b\n" and got[1][1] == "reCameraPos(0, 0, 1)"
    host.HostScene.from_script(got[0][1])


@pytest.mark.skipif(not os.path.exists(REF_RES + "/scenes.xml"), reason="reference checkout not present")
def test_reference_scene_store_loads_and_demo4_is_config3(host_lib):
    """every scene of the reference's own scenes.xml evaluates; 'Demo 4 - Bunny' (through the OBJ reader) flattens to
    the scene BASELINE config 3 is defined as"""
    store = host.scenes_xml(REF_RES + "/scenes.xml")
    assert [n for n, _ in store][0] == "Demo 1 - Spheres" and len(store) == 4
    for name, code in store:
        hs = host.HostScene.from_script(code, res_dir="/root/reference/Sandbox")
        assert hs.L.rehost_shape_count(hs.h) > 0, name
    demo4 = host.HostScene.from_script(dict(store)["Demo 4 - Bunny"], res_dir="/root/reference/Sandbox")
    sd = scenes.config(3)[0]
    sd.adopt_tables(demo4.built)
    a = _hostcore_frame(demo4, 120, 68, 3)
    b = _hostcore_frame(host.HostScene(sd), 120, 68, 3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


def test_ppm_writer(host_lib, tmp_path):
    px = np.array([[0xff0000ff, 0xff00ff00], [0xffff0000, 0xff102030]], np.uint32)   # 0xAABBGGRR
    host.write_ppm(str(tmp_path / "f.ppm"), px)
    raw = (tmp_path / "f.ppm").read_bytes()
    assert raw == b"P6\n2 2\n255\n" + bytes([255, 0, 0, 0, 255, 0, 0, 0, 255, 0x30, 0x20, 0x10])


@pytest.mark.gpu
def test_script_scene_renders_on_the_device_like_the_oracle(oracle_lib, tmp_path):
    hs = host.HostScene.from_script(DEMO1_SCRIPT)
    got = hs.render(240, 135, max_recursion=3, ids=True)
    sd = _demo1_desc_with_tables_of(hs)
    ref = sd.build(oracle_lib).render(240, 135, max_recursion=3, ids=True, threads=8)
    diff, ids_bad = assert_frame_parity(got, ref["pixels"], ref["shape"], ref["tri"], "script demo 1")
    assert ids_bad == 0 and got["rays"] == ref["rays"]      # Marble's sin is a libm row: pixels within the parity bar
    # and the headless command line end to end
    import subprocess
    script = tmp_path / "demo1.lua"
    script.write_text(DEMO1_SCRIPT)
    out = tmp_path / "demo1.ppm"
    r = subprocess.run([os.path.join(ROOT, "raytracing_amd", "lib", "rt_render"), "--script", str(script), "--size", "240x135",
                        "--maxrec", "3", "--out", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "rays" in r.stdout and out.stat().st_size == len(b"P6\n240 135\n255\n") + 240 * 135 * 3
