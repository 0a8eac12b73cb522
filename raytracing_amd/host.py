"""Python handle on the product host library (raytracing_amd/lib/librt_host.so): the from-scratch
re:: API mirror + flattener + re::HipRaytracer, reached through its sb_* C facade (sbapi.SbLib)
plus a few product-only entry points (rehost_*)."""
import ctypes as C
import os

import numpy as np

from . import device
from .sbapi import BuiltScene, SbLib

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_SO = os.path.join(_HERE, "lib", "librt_host.so")
_host = None


def load_host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_SO):
            raise RuntimeError(f"{HOST_SO} is missing: run __graft_entry__.build() first")
        device.lib()  # librt_host.so links librthip.so; fail loudly if that is missing
        sb = SbLib(HOST_SO)
        L = sb.lib
        L.rehost_flatten.argtypes = [C.c_void_p]; L.rehost_flatten.restype = C.c_void_p
        L.rehost_camera.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(device.Camera)]
        L.rehost_set_device.argtypes = [C.c_void_p, C.c_int]; L.rehost_set_device.restype = None
        L.rehost_set_accel.argtypes = [C.c_void_p, C.c_int]; L.rehost_set_accel.restype = None
        L.rehost_shape_count.argtypes = [C.c_void_p]
        L.rehost_shape_ordinal.argtypes = [C.c_void_p, C.c_int]
        L.rehost_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.rehost_last_error.restype = C.c_char_p
        from .sbapi import RenderParams
        L.rehost_render_async.argtypes = [C.c_void_p, C.POINTER(RenderParams)]; L.rehost_render_async.restype = C.c_void_p
        L.rehost_job_status.argtypes = [C.c_void_p, C.POINTER(C.c_float)]; L.rehost_job_status.restype = None
        L.rehost_job_interrupt.argtypes = [C.c_void_p]; L.rehost_job_interrupt.restype = None
        L.rehost_job_wait.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.rehost_job_free.argtypes = [C.c_void_p]; L.rehost_job_free.restype = None
        _host = sb
    return _host


class HostScene:
    """A scene built in the product host library from a scenes.SceneDesc."""

    def __init__(self, scene_desc, device_index=0):
        self.sb = load_host()
        self.built = scene_desc.build(self.sb)
        self.L = self.sb.lib
        self.h = self.built.h
        self.L.rehost_set_device(self.h, device_index)

    def set_accel(self, use_reference_kd):
        self.L.rehost_set_accel(self.h, int(bool(use_reference_kd)))

    def flatten(self):
        """`const rthip_scene_desc*` of the compiled scene (valid until the scene is freed)."""
        p = self.L.rehost_flatten(self.h)
        if not p:
            raise RuntimeError("flatten failed: " + self.L.rehost_last_error().decode())
        return p

    def camera(self, width, height, fovy=0.0):
        cam = device.Camera()
        self.L.rehost_camera(self.h, width, height, fovy, C.byref(cam))
        return cam

    def render(self, *a, **kw):
        """HipRaytracer::RenderSync through the sb_render facade."""
        try:
            return self.built.render(*a, **kw)
        except RuntimeError as e:
            raise RuntimeError(str(e) + ": " + self.L.rehost_last_error().decode())

    def last_stats(self):
        out = (C.c_double * 8)()
        self.L.rehost_last_stats(self.h, out)
        keys = ["rays_primary", "rays_shadow", "rays_reflect", "pixels", "device_ms", "flatten_ms", "upload_ms"]
        return dict(zip(keys, list(out)))


class RenderJob:
    """Asynchronous re::HipRaytracer::Render (returns at once; status can be polled; can be interrupted)."""

    def __init__(self, hs, width, height, max_recursion=3, aa=0):
        from .sbapi import RenderParams
        self.hs, self.w, self.h = hs, width, height
        p = RenderParams(width, height, 0.0, max_recursion, aa, 1, 0)
        self.j = hs.L.rehost_render_async(hs.h, C.byref(p))

    def status(self):
        out = (C.c_float * 4)()
        self.hs.L.rehost_job_status(self.j, out)
        return {"finished": bool(out[0]), "interrupted": bool(out[1]), "percent": float(out[2]), "has_pixels": bool(out[3])}

    def interrupt(self):
        self.hs.L.rehost_job_interrupt(self.j)

    def wait(self):
        px = np.zeros((self.h, self.w), np.uint32)
        out = (C.c_float * 3)()
        rc = self.hs.L.rehost_job_wait(self.j, px.ctypes.data_as(C.c_void_p), out)
        if rc != 0:
            raise RuntimeError(f"render failed ({rc}): " + self.hs.L.rehost_last_error().decode())
        return px, {"finished": bool(out[0]), "interrupted": bool(out[1]), "percent": float(out[2])}

    def close(self):
        if self.j:
            self.hs.L.rehost_job_free(self.j)
            self.j = None
