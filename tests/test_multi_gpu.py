"""Several GPUs from ONE process (rthip_multi_*): interleaved tiles, parts gathered to the first device,
device assemble, one D2H.  On a one-GPU box the same GPU is listed several times (the gather degenerates to device-to-device
copies) and a one-device renderer is forced through RCCL (self ncclSend / ncclRecv), so both transports run; the frame must be
bit-identical to the single-context frame and the parts' ray counts must add up."""
import os

import numpy as np
import pytest

from raytracing_amd import device, host, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [2, 3, 8])
def test_frame_split_over_n_contexts_of_the_gpu_equals_the_single_frame(n):
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    desc = hs.flatten()
    w, h = 333, 190   # ragged: tiles hang over the right and bottom edges
    cam, prm = hs.camera(w, h), device.make_params(w, h, 3)
    one = device.Context(0)
    one.upload(desc)
    want = one.render(cam, prm)["pixels"]
    st = one.stats()
    want_rays = st.rays_primary + st.rays_shadow + st.rays_reflect
    one.close()
    m = device.MultiContext([0] * n)
    try:
        assert m.transport() == "peer copies"   # RCCL refuses the same device twice
        m.upload(desc)
        for _ in range(2):   # twice: the second call replays buffers / graphs
            got = m.render(cam, prm)
            assert np.array_equal(got["pixels"], want)
            assert got["rays"] == want_rays
        small = device.make_params(64, 40, 2)
        a = m.render(hs.camera(64, 40), small)["pixels"]
        one = device.Context(0); one.upload(desc)
        assert np.array_equal(a, one.render(hs.camera(64, 40), small)["pixels"])
        one.close()
    finally:
        m.close()


def test_rccl_transport_on_one_device_self_send_recv():
    os.environ["RTHIP_MULTI_TRANSPORT"] = "rccl"
    try:
        hs = host.HostScene(scenes.config(2)[0])
        desc = hs.flatten()
        w, h = 320, 180
        cam, prm = hs.camera(w, h), device.make_params(w, h, 3)
        m = device.MultiContext([0])
        try:
            assert m.transport().startswith("RCCL")
            m.upload(desc)
            got = m.render(cam, prm)
        finally:
            m.close()
    finally:
        del os.environ["RTHIP_MULTI_TRANSPORT"]
    one = device.Context(0); one.upload(desc)
    assert np.array_equal(got["pixels"], one.render(cam, prm)["pixels"])
    one.close()


def test_hipraytracer_devices_knob_renders_the_golden_frame(golden):
    """re::HipRaytracer with Devices = {0, 0, 0, 0}: the drop-in class drives the multi-GPU path; golden config-3 frame"""
    from conftest import FRAME_CASES
    mk, w, h, kw = FRAME_CASES["cfg3"]
    hs = host.HostScene(mk())
    hs.set_devices([0, 0, 0, 0])
    got = hs.render(w, h, **kw)
    f = golden["frames"]
    assert np.array_equal(got["pixels"], f["cfg3_pixels"])
    assert got["rays"] == int(f["cfg3_rays"][0])
