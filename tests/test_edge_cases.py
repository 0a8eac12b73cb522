"""Edge cases of the Raycast path against the oracle (tests/edge_scenes.py): on the CPU through the flattener + the
g++ build of rt_core.hpp (tests/hostcore), on the GPU through re::HipRaytracer and the C ABI.  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
from edge_scenes import EDGE_CASES
from raytracing_amd import device, host


def _cases():
    for name, (mk, runs) in EDGE_CASES.items():
        for k, (w, h, kw) in enumerate(runs):
            yield pytest.param(mk, w, h, kw, id=f"{name}-{w}x{h}-{k}")


def _oracle_frame(oracle_lib, sd, w, h, kw):
    return sd.build(oracle_lib).render(w, h, ids=True, threads=4, **kw)


@pytest.mark.parametrize("mk,w,h,kw", list(_cases()))
def test_edge_case_host_core_matches_oracle(host_lib, oracle_lib, mk, w, h, kw):
    sd = mk()
    ref = _oracle_frame(oracle_lib, sd, w, h, kw)
    hc = C.CDLL(os.path.join(ROOT, "tests", "hostcore", "libhostcore.so"))
    for accel in (0, 1):
        hc.hc_set_accel(accel)
        hs = host.HostScene(sd)
        cam = hs.camera(w, h)
        prm = device.make_params(w, h, kw.get("max_recursion", 3), kw.get("aa", 0))
        px = np.zeros((h, w), np.uint32); sh = np.zeros((h, w), np.int32); tr = np.zeros((h, w), np.int32)
        rays = C.c_uint64(0)
        vp = C.c_void_p
        rc = hc.hc_render(vp(hs.flatten()), C.byref(cam), C.byref(prm), px.ctypes.data_as(vp), sh.ctypes.data_as(vp),
                          tr.ctypes.data_as(vp), C.byref(rays), None, 4)
        assert rc == 0
        assert np.array_equal(px, ref["pixels"]), f"accel {accel}: {(px != ref['pixels']).sum()} pixels differ"
        assert np.array_equal(sh, ref["shape"]) and np.array_equal(tr, ref["tri"])
        assert rays.value == ref["rays"]
    hc.hc_set_accel(0)


@pytest.mark.gpu
@pytest.mark.parametrize("mk,w,h,kw", list(_cases()))
def test_edge_case_device_matches_oracle(oracle_lib, mk, w, h, kw):
    sd = mk()
    ref = _oracle_frame(oracle_lib, sd, w, h, kw)
    for use_kd in (False, True):
        hs = host.HostScene(sd)
        hs.set_accel(use_kd)
        got = hs.render(w, h, ids=True, **kw)
        assert np.array_equal(got["pixels"], ref["pixels"]), f"kd={use_kd}: {(got['pixels'] != ref['pixels']).sum()} pixels differ"
        assert np.array_equal(got["shape"], ref["shape"]) and np.array_equal(got["tri"], ref["tri"])
        assert got["rays"] == ref["rays"]


@pytest.mark.gpu
def test_edge_case_batches_and_parts_on_the_device():
    """a batch of one-pixel frames, a batch wider than the image has tiles, and more parts than tiles"""
    import torch
    from edge_scenes import lights_off
    hs = host.HostScene(lights_off())
    ctx = device.Context(0)
    ctx.upload(hs.flatten())
    for (w, h) in ((1, 1), (31, 7), (40, 9)):
        cam = hs.camera(w, h)
        prm = device.make_params(w, h, 2)
        single = ctx.render(cam, prm)["pixels"]
        buf = torch.zeros(device.MAX_BATCH * w * h, dtype=torch.int32, device="cuda")
        ctx.render_batch_device([cam] * device.MAX_BATCH, prm, buf.data_ptr(), sync=True)
        got = buf.cpu().numpy().view(np.uint32).reshape(device.MAX_BATCH, h, w)
        assert all(np.array_equal(got[k], single) for k in range(device.MAX_BATCH))
        parts = 5   # 31x7 and 1x1 have a single 32x8 tile: four of the five parts are empty
        acc = np.zeros((h, w), np.uint32)
        for p in range(parts):
            ctx.render(cam, device.make_params(w, h, 2, part_index=p, part_count=parts), out=acc)
        assert np.array_equal(acc, single)
    ctx.close()


@pytest.mark.parametrize("mk,w,h,kw", list(_cases()))
def test_edge_case_oracle_matches_the_genuine_reference(oracle_lib, ref_lib, mk, w, h, kw):
    """pins the oracle on the edge cases (skipped where oracle/_ref is absent)"""
    sd = mk()
    a = _oracle_frame(oracle_lib, sd, w, h, kw)
    b = sd.build(ref_lib).render(w, h, ids=True, threads=4, **kw)
    assert np.array_equal(a["pixels"], b["pixels"]) and np.array_equal(a["shape"], b["shape"]) and np.array_equal(a["tri"], b["tri"])
    assert a["rays"] == b["rays"]
