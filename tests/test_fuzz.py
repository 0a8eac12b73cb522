"""Seeded random scenes (tests/fuzz_scenes.py) against the oracle, bit for bit: on the CPU through the flattener and
the g++ build of rt_core.hpp, on the GPU through re::HipRaytracer (BVH path and reference kd-tree path)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
from fuzz_scenes import FUZZ_SEEDS, random_scene
from raytracing_amd import device, host

W, H, MAXREC = 112, 80, 4


def _oracle(oracle_lib, sd, aa):
    return sd.build(oracle_lib).render(W, H, ids=True, threads=4, max_recursion=MAXREC, aa=aa)


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_fuzz_host_core_matches_oracle(host_lib, oracle_lib, seed):
    sd = random_scene(seed)
    ref = _oracle(oracle_lib, sd, 0)
    hc = C.CDLL(os.path.join(ROOT, "tests", "hostcore", "libhostcore.so"))
    hc.hc_set_accel(0)
    hs = host.HostScene(sd)
    cam = hs.camera(W, H)
    prm = device.make_params(W, H, MAXREC)
    px = np.zeros((H, W), np.uint32); sh = np.zeros((H, W), np.int32); tr = np.zeros((H, W), np.int32)
    rays = C.c_uint64(0)
    vp = C.c_void_p
    assert hc.hc_render(vp(hs.flatten()), C.byref(cam), C.byref(prm), px.ctypes.data_as(vp), sh.ctypes.data_as(vp),
                        tr.ctypes.data_as(vp), C.byref(rays), None, 4) == 0
    assert np.array_equal(sh, ref["shape"]) and np.array_equal(tr, ref["tri"]), f"seed {seed}: ids differ"
    assert np.array_equal(px, ref["pixels"]), f"seed {seed}: {(px != ref['pixels']).sum()} pixels differ"
    assert rays.value == ref["rays"]
    assert (ref["shape"] >= 0).mean() > 0.05, "the random camera sees nothing"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_fuzz_device_matches_oracle(oracle_lib, seed):
    sd = random_scene(seed)
    for aa in ((0, 1) if seed % 4 == 0 else (0,)):
        ref = _oracle(oracle_lib, sd, aa)
        for use_kd in (False, True):
            hs = host.HostScene(sd)
            hs.set_accel(use_kd)
            got = hs.render(W, H, ids=True, max_recursion=MAXREC, aa=aa)
            assert np.array_equal(got["shape"], ref["shape"]) and np.array_equal(got["tri"], ref["tri"]), f"seed {seed} kd={use_kd}: ids differ"
            assert np.array_equal(got["pixels"], ref["pixels"]), f"seed {seed} kd={use_kd} aa={aa}: {(got['pixels'] != ref['pixels']).sum()} pixels differ"
            assert got["rays"] == ref["rays"]


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_fuzz_oracle_matches_the_genuine_reference(oracle_lib, ref_lib, seed):
    """pins the oracle itself: the C restatement and the reference compiled from its own sources (oracle/_ref,
    skipped where that build is absent) render every random scene to the same bits"""
    sd = random_scene(seed)
    a = _oracle(oracle_lib, sd, 0)
    b = sd.build(ref_lib).render(W, H, ids=True, threads=4, max_recursion=MAXREC, aa=0)
    assert np.array_equal(a["shape"], b["shape"]) and np.array_equal(a["tri"], b["tri"])
    assert np.array_equal(a["pixels"], b["pixels"]) and a["rays"] == b["rays"]
