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
        L.rehost_set_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]; L.rehost_set_devices.restype = None
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
        L.rehost_scene_from_script.argtypes = [C.c_char_p, C.c_char_p]; L.rehost_scene_from_script.restype = C.c_void_p
        L.rehost_scenes_xml.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.rehost_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
        L.rehost_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.rehost_render_moved.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.POINTER(C.c_int)]
        L.rehost_render_flip_normal_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.rehost_render_path.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rehost_raycast.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
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

    @classmethod
    def from_script(cls, code, res_dir="", device_index=0):
        """A scene evaluated by the product's scene-script reader (host/scene_io.cpp: the reXxx call subset of the
        reference's Sandbox scripts, Sandbox.cpp:578-763)."""
        self = cls.__new__(cls)
        self.sb = load_host()
        self.L = self.sb.lib
        h = self.L.rehost_scene_from_script(code.encode(), res_dir.encode())
        if not h:
            raise RuntimeError("scene script: " + self.L.rehost_last_error().decode())
        self.built = BuiltScene(self.sb, h)
        self.h = self.built.h
        self.L.rehost_set_device(self.h, device_index)
        return self

    def set_devices(self, devices):
        """HipRaytracer::Devices: render() splits the frame over these GPUs from this one process (rthip_multi_*)"""
        arr = (C.c_int * len(devices))(*devices)
        self.L.rehost_set_devices(self.h, arr, len(devices))

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

    def camera_at(self, pos, direction, width, height, fovy=0.0):
        """rthip_camera for Scene::Camera = {pos, direction} (the scene keeps that camera, like Sandbox::Update moving it,
        Sandbox/Sandbox.cpp:315-350); the flattened scene is not touched."""
        from .sbapi import _vec3
        self.L.sb_camera(self.h, _vec3(tuple(map(float, pos))), _vec3(tuple(map(float, direction))))
        return self.camera(width, height, fovy)

    def render(self, *a, **kw):
        """HipRaytracer::RenderSync through the sb_render facade."""
        try:
            return self.built.render(*a, **kw)
        except RuntimeError as e:
            raise RuntimeError(str(e) + ": " + self.L.rehost_last_error().decode())

    def render_moved(self, width, height, dx, max_recursion=3):
        """Two frames through ONE HipRaytracer; between them every node under the root moves by dx along x and the
        camera by dx / 2.  Returns (pixels of the second frame, whether it kept the device geometry).  The scene stays
        moved, so a later render() shows what a fresh renderer makes of it."""
        from .sbapi import RenderParams
        p = RenderParams(width, height, 0.0, max_recursion, 0, 8, 0)
        px = np.zeros((height, width), np.uint32)
        kept = C.c_int(0)
        rc = self.L.rehost_render_moved(self.h, C.byref(p), C.c_double(dx), px.ctypes.data_as(C.c_void_p), C.byref(kept))
        if rc != 0:
            raise RuntimeError(f"rehost_render_moved failed ({rc}): " + self.L.rehost_last_error().decode())
        return px, bool(kept.value)

    def render_path(self, width, height, poses, max_recursion=3, aa=0):
        """HipRaytracer::RenderPathSync: one frame per (position, direction) pose of `poses` ([n, 6]) over the scene as it is, traced
        in runs (over the GPUs of set_devices, if any) -> [n, height, width] uint32"""
        from .sbapi import RenderParams
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
        p = RenderParams(width, height, 0.0, max_recursion, aa, 8, 0)
        px = np.zeros((len(poses), height, width), np.uint32)
        rc = self.L.rehost_render_path(self.h, C.byref(p), len(poses), poses.ctypes.data_as(C.c_void_p), px.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(f"rehost_render_path failed ({rc}): " + self.L.rehost_last_error().decode())
        return px

    def raycast(self, rays, max_recursion=3):
        """HipRaytracer::Raycast, one call per ray of `rays` ([n, 6]: origin, direction) -> [n, 3] colours"""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        rgb = np.zeros((len(rays), 3), np.float64)
        rc = self.L.rehost_raycast(self.h, max_recursion, len(rays), rays.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(f"rehost_raycast failed ({rc}): " + self.L.rehost_last_error().decode())
        return rgb

    def render_flip_normal_mode(self, width, height, max_recursion=3):
        """Two frames through ONE HipRaytracer; between them Mesh::NormalMode of every mesh flips (Face <-> Vertex).  Returns
        (pixels of the second frame, whether it kept the device geometry).  The scene keeps the flipped modes."""
        from .sbapi import RenderParams
        p = RenderParams(width, height, 0.0, max_recursion, 0, 8, 0)
        px = np.zeros((height, width), np.uint32)
        kept = C.c_int(0)
        rc = self.L.rehost_render_flip_normal_mode(self.h, C.byref(p), px.ctypes.data_as(C.c_void_p), C.byref(kept))
        if rc != 0:
            raise RuntimeError(f"rehost_render_flip_normal_mode failed ({rc}): " + self.L.rehost_last_error().decode())
        return px, bool(kept.value)

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


def scenes_xml(path):
    """[(name, code)] of a scenes.xml store (Sandbox/res/scenes.xml format) read by the product's reader."""
    L = load_host().lib
    n = L.rehost_scenes_xml(path.encode(), -1, None, 0, None, 0)
    if n < 0:
        raise RuntimeError(L.rehost_last_error().decode())
    out = []
    for i in range(n):
        name = C.create_string_buffer(1 << 10)
        code = C.create_string_buffer(1 << 20)
        L.rehost_scenes_xml(path.encode(), i, name, len(name), code, len(code))
        out.append((name.value.decode(), code.value.decode()))
    return out


def load_obj(path, group):
    """(verts, normals) float64 [n, 9] of an OBJ group as reObjMesh takes them (product's Wavefront reader)."""
    L = load_host().lib
    n = L.rehost_load_obj(path.encode(), group.encode(), None, None, 0)
    if n < 0:
        raise RuntimeError(L.rehost_last_error().decode())
    v = np.zeros((n, 9)); nr = np.zeros((n, 9))
    L.rehost_load_obj(path.encode(), group.encode(), v.ctypes.data_as(C.c_void_p), nr.ctypes.data_as(C.c_void_p), n)
    return v, nr


def write_ppm(path, pixels):
    px = np.ascontiguousarray(pixels, dtype=np.uint32)
    L = load_host().lib
    if L.rehost_write_ppm(path.encode(), px.ctypes.data_as(C.c_void_p), px.shape[1], px.shape[0]) != 0:
        raise RuntimeError(L.rehost_last_error().decode())
