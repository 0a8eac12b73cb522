"""ctypes binding of the shared "scene builder" C entry points (sb_*).

The same signature set is exported by three libraries:
  * raytracing_amd/lib/librt_host.so  — the product: from-scratch re:: host API + HipRaytracer
  * oracle/liboracle.so               — CPU oracle, plain-C restatement (tests / cpu_baseline only)
  * oracle/_ref/libref.so             — the genuine reference compiled from its sources (tests only)
so one Python scene description (raytracing_amd.scenes.SceneDesc) can drive all of them.
This module only knows how to call a library handed to it; it loads nothing by itself.
"""
import ctypes as C

import numpy as np

c_dp = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fovy", C.c_double),
                ("max_recursion", C.c_int), ("aa_mode", C.c_int), ("num_threads", C.c_int),
                ("renderer", C.c_int)]


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def _vec3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _f64(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


class SbLib:
    """Thin wrapper over one library exporting the sb_* entry points."""

    def __init__(self, path):
        self.path = str(path)
        L = self.lib = C.CDLL(self.path)
        vp = C.c_void_p
        sig = {
            "sb_scene_new": (vp, []),
            "sb_scene_free": (None, [vp]),
            "sb_camera": (None, [vp, c_dp, c_dp]),
            "sb_mat_uniform": (C.c_int, [vp, C.c_uint, C.c_double]),
            "sb_mat_interp": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
            "sb_noise_checker": (C.c_int, [vp, C.c_double]),
            "sb_noise_perlin": (C.c_int, [vp, C.c_double, c_u8p]),
            "sb_noise_worley": (C.c_int, [vp, C.c_double, C.c_int, c_dp]),
            "sb_noise_marble": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, c_u8p]),
            "sb_node": (C.c_int, [vp, C.c_int, c_dp, c_dp, c_dp]),
            "sb_sphere": (C.c_int, [vp, C.c_int, C.c_int]),
            "sb_plane": (C.c_int, [vp, C.c_int, C.c_int, c_dp]),
            "sb_mesh": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp]),
            "sb_light": (C.c_int, [vp, C.c_int, C.c_uint, c_dp, c_dp, C.c_double, C.c_int]),
            "sb_bg_color": (None, [vp, C.c_uint]),
            "sb_bg_skybox": (None, [vp, C.c_uint, C.c_uint, C.c_int, c_u8p]),
            "sb_get_perlin_table": (C.c_int, [vp, C.c_int, c_u8p]),
            "sb_get_skybox_table": (C.c_int, [vp, c_u8p]),
            "sb_get_worley_points": (C.c_int, [vp, C.c_int, c_dp]),
            "sb_render": (C.c_int, [vp, C.POINTER(RenderParams), c_u32p, c_i32p, c_i32p,
                                    C.POINTER(C.c_uint64), c_i32p, c_i32p, C.c_int]),
            "sb_cast_rays": (C.c_int, [vp, C.c_int, c_dp, c_dp, c_i32p, c_dp, c_dp, c_dp, c_i32p, c_i32p]),
            "sb_noise_sample": (C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp]),
            "sb_material_eval": (C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp, c_dp]),
            "sb_background_eval": (C.c_int, [vp, C.c_int, c_dp, c_dp]),
            "sb_raycast": (C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp, c_dp]),
            "sb_screen_rays": (C.c_int, [vp, C.c_int, C.c_int, C.c_double, C.c_int, c_dp, c_dp, c_dp]),
            "sb_kd_dump": (C.c_int, [vp, C.c_int, c_i32p, c_i32p, c_dp, c_dp, c_i32p, C.POINTER(C.c_int64)]),
            "sb_node_matrices": (C.c_int, [vp, C.c_int, c_dp]),
            "sb_impl_name": (C.c_char_p, []),
        }
        self.missing = []
        for name, (res, args) in sig.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args

    @property
    def name(self):
        return self.lib.sb_impl_name().decode()


class NotInScene(LookupError):
    """a probe of an object the flattened scene does not contain (nothing references it)"""


class BuiltScene:
    """A scene instantiated inside one sb_* library."""

    def __init__(self, sblib, handle):
        self.sb = sblib
        self.L = sblib.lib
        self.h = handle

    def close(self):
        if self.h:
            self.L.sb_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- rendering -------------------------------------------------------------
    def render(self, width, height, max_recursion=3, aa=0, threads=8, renderer=0, fovy=0.0,
               ids=False, sub=None):
        p = RenderParams(width, height, fovy, max_recursion, aa, threads, renderer)
        if sub is not None:
            sx = np.ascontiguousarray(sub[0], dtype=np.int32)
            sy = np.ascontiguousarray(sub[1], dtype=np.int32)
            n = len(sx)
            shape = (n,)
        else:
            sx = sy = None
            n = 0
            shape = (height, width)
        px = np.zeros(shape, dtype=np.uint32)
        so = np.full(shape, -1, dtype=np.int32) if ids else None
        ti = np.full(shape, -1, dtype=np.int32) if ids else None
        rays = C.c_uint64(0)
        rc = self.L.sb_render(
            self.h, C.byref(p), px.ctypes.data_as(c_u32p),
            so.ctypes.data_as(c_i32p) if ids else None, ti.ctypes.data_as(c_i32p) if ids else None,
            C.byref(rays), sx.ctypes.data_as(c_i32p) if n else None,
            sy.ctypes.data_as(c_i32p) if n else None, n)
        if rc != 0:
            raise RuntimeError(f"sb_render failed ({rc}) in {self.sb.name}")
        return {"pixels": px, "shape": so, "tri": ti, "rays": int(rays.value)}

    # ---- probes ---------------------------------------------------------------
    def cast_rays(self, o, d):
        o = _f64(o, 3); d = _f64(d, 3); n = len(o)
        hit = np.zeros(n, np.int32); pt = np.zeros((n, 3)); lp = np.zeros((n, 3)); nr = np.zeros((n, 3))
        sh = np.zeros(n, np.int32); tr = np.zeros(n, np.int32)
        rc = self.L.sb_cast_rays(self.h, n, _dp(o), _dp(d), hit.ctypes.data_as(c_i32p), _dp(pt), _dp(lp), _dp(nr),
                                 sh.ctypes.data_as(c_i32p), tr.ctypes.data_as(c_i32p))
        assert rc == 0
        return {"hit": hit, "point": pt, "local": lp, "normal": nr, "shape": sh, "tri": tr}

    def noise_sample(self, noise, p):
        p = _f64(p, 3); out = np.zeros(len(p))
        rc = self.L.sb_noise_sample(self.h, noise, len(p), _dp(p), _dp(out))
        if rc == -100:  # the noise exists but no shape's material references it, so the device scene lacks it
            raise NotInScene(f"noise {noise}")
        assert rc == 0, rc
        return out

    def material_eval(self, mat, p):
        p = _f64(p, 3); col = np.zeros((len(p), 3)); ab = np.zeros(len(p))
        rc = self.L.sb_material_eval(self.h, mat, len(p), _dp(p), _dp(col), _dp(ab))
        if rc == -100:
            raise NotInScene(f"material {mat}")
        assert rc == 0, rc
        return col, ab

    def background_eval(self, d):
        d = _f64(d, 3); col = np.zeros((len(d), 3))
        assert self.L.sb_background_eval(self.h, len(d), _dp(d), _dp(col)) == 0
        return col

    def raycast(self, o, d, max_recursion=3):
        o = _f64(o, 3); d = _f64(d, 3); col = np.zeros((len(o), 3))
        assert self.L.sb_raycast(self.h, max_recursion, len(o), _dp(o), _dp(d), _dp(col)) == 0
        return col

    def screen_rays(self, width, height, xy, fovy=0.0):
        xy = _f64(xy, 2); o = np.zeros((len(xy), 3)); d = np.zeros((len(xy), 3))
        assert self.L.sb_screen_rays(self.h, width, height, fovy, len(xy), _dp(xy), _dp(o), _dp(d)) == 0
        return o, d

    def kd_dump(self, node):
        nrefs = C.c_int64(0)
        nn = self.L.sb_kd_dump(self.h, node, None, None, None, None, None, C.byref(nrefs))
        if nn < 0:
            raise RuntimeError("sb_kd_dump failed")
        depth = np.zeros(nn, np.int32); ntris = np.zeros(nn, np.int32); split = np.zeros(nn)
        bounds = np.zeros((nn, 6)); refs = np.zeros(nrefs.value, np.int32)
        self.L.sb_kd_dump(self.h, node, depth.ctypes.data_as(c_i32p), ntris.ctypes.data_as(c_i32p), _dp(split),
                          _dp(bounds), refs.ctypes.data_as(c_i32p), C.byref(nrefs))
        return {"depth": depth, "ntris": ntris, "split": split, "bounds": bounds, "refs": refs}

    def node_matrices(self, node):
        out = np.zeros(48)
        assert self.L.sb_node_matrices(self.h, node, _dp(out)) == 0
        return out.reshape(3, 4, 4)

    def perlin_table(self, noise):
        t = np.zeros(256, np.uint8)
        assert self.L.sb_get_perlin_table(self.h, noise, t.ctypes.data_as(c_u8p)) == 0
        return t

    def skybox_table(self):
        t = np.zeros(256, np.uint8)
        assert self.L.sb_get_skybox_table(self.h, t.ctypes.data_as(c_u8p)) == 0
        return t

    def worley_points(self, noise, div):
        t = np.zeros((div ** 3, 3))
        assert self.L.sb_get_worley_points(self.h, noise, _dp(t)) == 0
        return t
