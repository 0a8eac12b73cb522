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


def _poses(n):
    import math
    out = []
    for k in range(n):
        th = 0.3 * math.sin(2 * math.pi * k / max(n, 1)) + 0.02 * k
        out.append([2.5 * math.sin(th), 0.1 * (k % 3), 2.5 * math.cos(th), -math.sin(th), -0.02 * (k % 3), -math.cos(th)])
    return np.array(out)


def _single_frames(hs, desc, w, h, poses, max_recursion):
    one = device.Context(0)
    try:
        one.upload(desc)
        prm = device.make_params(w, h, max_recursion)
        return [one.render(hs.camera_at(p[:3], p[3:], w, h), prm)["pixels"].copy() for p in poses], prm
    finally:
        one.close()


@pytest.mark.parametrize("devices", [[0, 0, 0, 0], [0, 0]])
def test_batch_over_several_contexts_equals_single_renders(devices):
    """rthip_multi_render_batch: a run of cameras split over the devices in ONE round (one set of launches per device, the parts
    cross once, one assemble, one D2H); every frame must equal the single-context render of its camera, the ray counts add up"""
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    desc = hs.flatten()
    w, h, n = 200, 120, 7   # ragged against the 32x8 tiles
    poses = _poses(n)
    want, prm = _single_frames(hs, desc, w, h, poses, 3)
    assert len({x.tobytes() for x in want}) == n
    cams = [hs.camera_at(p[:3], p[3:], w, h) for p in poses]
    m = device.MultiContext(devices)
    try:
        m.upload(desc)
        for rep in range(3):   # the third call replays the devices' launch graphs (with the cameras of a different run)
            run = cams if rep != 1 else cams[::-1]
            got = m.render_batch(run, prm)
            for k in range(n):
                assert np.array_equal(got["pixels"][k], want[k if rep != 1 else n - 1 - k]), f"rep {rep} frame {k}"
        one = device.Context(0); one.upload(desc)
        rays = 0
        for c in cams:
            one.render(c, prm)
            st = one.stats()
            rays += st.rays_primary + st.rays_shadow + st.rays_reflect
        one.close()
        assert got["rays"] == rays
        assert np.array_equal(m.render(cams[2], prm)["pixels"], want[2])   # a single frame afterwards
        with pytest.raises(device.RthipError):
            m.render_batch(cams * 10, prm)   # 70 > RTHIP_MAX_BATCH
    finally:
        m.close()


def test_rthip_render_batch_hands_back_host_frames():
    hs = host.HostScene(scenes.config(2)[0])
    desc = hs.flatten()
    w, h, n = 160, 90, 5
    poses = _poses(n)
    poses[:, 1] += 3.0
    poses[:, 2] += 2.0
    want, prm = _single_frames(hs, desc, w, h, poses, 2)
    c = device.Context(0)
    try:
        c.upload(desc)
        got = c.render_batch([hs.camera_at(p[:3], p[3:], w, h) for p in poses], prm)
    finally:
        c.close()
    for k in range(n):
        assert np.array_equal(got[k], want[k]), f"frame {k}"


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_hipraytracer_camera_path(devices):
    """re::HipRaytracer::RenderPathSync: a queued camera path of 70 poses (more than one run of 64) through the drop-in class, on one
    device and split over Devices; every frame equals the RenderSync frame of its camera"""
    sd = scenes.config(3, cam_z=2.5)[0]
    hs = host.HostScene(sd)
    if devices:
        hs.set_devices(devices)
    w, h, n = 96, 64, 70
    poses = _poses(n)
    frames = hs.render_path(w, h, poses, max_recursion=2)
    assert frames.shape == (n, h, w) and len({f.tobytes() for f in frames}) > n // 2
    ref = host.HostScene(sd)
    desc = ref.flatten()
    want, _ = _single_frames(ref, desc, w, h, poses[[0, 1, 33, 63, 64, 69]], 2)
    for k, f in zip([0, 1, 33, 63, 64, 69], want):
        assert np.array_equal(frames[k], f), f"pose {k}"


def test_rccl_gather_on_distinct_devices():
    """The RCCL branch of the in-process multi-GPU path with N > 1 REAL devices: skipped on a one-GPU box, runs wherever two GPUs
    are visible (the 8-GPU scaling node).  Frames over {0, 1} and over every visible device with RTHIP_MULTI_TRANSPORT=rccl -
    single frames and a run of frames - must be bit-identical to the one-GPU frames."""
    import torch
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("needs two GPUs (ncclCommInitAll refuses one device twice)")
    hs = host.HostScene(scenes.config(3, cam_z=2.5)[0])
    desc = hs.flatten()
    w, h, n = 333, 190, 6
    poses = _poses(n)
    want, prm = _single_frames(hs, desc, w, h, poses, 3)
    cams = [hs.camera_at(p[:3], p[3:], w, h) for p in poses]
    os.environ["RTHIP_MULTI_TRANSPORT"] = "rccl"
    try:
        for devs in ([0, 1], list(range(ngpu))):
            m = device.MultiContext(devs)
            try:
                assert m.transport().startswith("RCCL")
                m.upload(desc)
                for rep in range(2):
                    assert np.array_equal(m.render(cams[0], prm)["pixels"], want[0]), (devs, rep)
                    got = m.render_batch(cams, prm)
                    for k in range(n):
                        assert np.array_equal(got["pixels"][k], want[k]), (devs, rep, k)
            finally:
                m.close()
    finally:
        del os.environ["RTHIP_MULTI_TRANSPORT"]
