"""ctypes binding of the C-ABI in include/rthip.h (raytracing_amd/lib/librthip.so).

Fails loudly when the HIP library is missing or a call fails: there is no CPU path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
RTHIP_SO = os.environ.get("RTHIP_SO") or os.path.join(_HERE, "lib", "librthip.so")  # env override: A/B builds in perf scripts

c_dp = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)

# every symbol include/rthip.h declares
SYMBOLS = ["rthip_create", "rthip_destroy", "rthip_last_error", "rthip_upload_scene", "rthip_render",
           "rthip_render_device", "rthip_stream", "rthip_synchronize", "rthip_progress", "rthip_interrupt",
           "rthip_get_stats", "rthip_probe_cast", "rthip_probe_noise", "rthip_probe_material",
           "rthip_probe_background", "rthip_probe_raycast", "rthip_probe_arith", "rthip_version",
           "rthip_part_pixels", "rthip_assemble", "rthip_render_batch_device", "rthip_assemble_batch", "rthip_host_alloc",
           "rthip_host_free", "rthip_update_scene", "rthip_render_device_noalloc", "rthip_collect_stats", "rthip_multi_create",
           "rthip_multi_destroy", "rthip_multi_device_count", "rthip_multi_context", "rthip_multi_transport", "rthip_multi_upload_scene",
           "rthip_multi_render", "rthip_multi_progress", "rthip_multi_interrupt", "rthip_upload_scene_ex", "rthip_probe_bvh",
           "rthip_debug_device_allocs", "rthip_probe_kd", "rthip_get_upload_info", "rthip_render_batch", "rthip_multi_render_batch"]
MAX_BATCH = 64  # RTHIP_MAX_BATCH
UPLOAD_BUILD_BVH, UPLOAD_KEEP_GEOMETRY, UPLOAD_BUILD_KD = 1, 2, 4  # RTHIP_UPLOAD_*


class Camera(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("forward", C.c_double * 3), ("right", C.c_double * 3),
                ("up", C.c_double * 3), ("h2", C.c_double), ("w2", C.c_double)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_recursion", C.c_uint32),
                ("aa_mode", C.c_int32), ("renderer", C.c_int32), ("part_index", C.c_uint32),
                ("part_count", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("layout", C.c_int32), ("accel", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("pixels", C.c_uint64), ("ms_total", C.c_float), ("ms_trace", C.c_float),
                ("launches_trace", C.c_uint32), ("pad", C.c_uint32), ("kd_nodes_visited", C.c_uint64),
                ("kd_tris_tested", C.c_uint64), ("kd_leaves_visited", C.c_uint64), ("kd_pops", C.c_uint64),
                ("shape_tests", C.c_uint64), ("mesh_hits", C.c_uint64), ("kd_max_nodes_ray", C.c_uint64),
                ("wave_steps_max", C.c_uint64), ("wave_tris_max", C.c_uint64), ("wave_steps", C.c_uint64), ("wave_tris", C.c_uint64),
                ("wave_cycles_max", C.c_uint64), ("fetch_cycles", C.c_uint64),
                ("level_rays", C.c_uint32 * 17), ("level_ms_trace", C.c_float * 17), ("level_ms_shade", C.c_float * 17), ("level_ms_easy", C.c_float * 17),
                ("ms_easy", C.c_float), ("rays_redo", C.c_uint32), ("rays_hard", C.c_uint64), ("shape_tests_easy", C.c_uint64),
                ("graphs_captured", C.c_uint32), ("graph_replays", C.c_uint32),
                ("shape_hits", (C.c_uint64 * 64) * 2), ("background_evals", C.c_uint64 * 2),
                ("rays_shadow_by_kernel", C.c_uint64 * 2), ("rays_reflect_by_kernel", C.c_uint64 * 2),
                ("shape_tests_full", C.c_uint64), ("shape_tests_easy_full", C.c_uint64)]


class Mesh(C.Structure):  # rthip_mesh
    _fields_ = [("node_base", C.c_uint32), ("node_count", C.c_uint32), ("ref_base", C.c_uint32), ("ref_count", C.c_uint32),
                ("tri_base", C.c_uint32), ("tri_count", C.c_uint32), ("normal_mode", C.c_int32), ("max_depth", C.c_uint32),
                ("bvh_node_base", C.c_uint32), ("bvh_node_count", C.c_uint32), ("bvh_tri_base", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("abs_max", C.c_float), ("pad", C.c_uint32), ("plane_base", C.c_uint32 * 3), ("plane_count", C.c_uint32 * 3)]


class Background(C.Structure):  # rthip_background
    _fields_ = [("type", C.c_int32), ("table", C.c_int32), ("sun_light", C.c_int32), ("pad", C.c_int32),
                ("color0", C.c_double * 3), ("color1", C.c_double * 3)]


class SceneDesc(C.Structure):  # rthip_scene_desc (tests build variants of a flattened scene through this view)
    _fields_ = [("shapes", C.c_void_p), ("n_shapes", C.c_uint32), ("meshes", C.POINTER(Mesh)), ("n_meshes", C.c_uint32),
                ("kd_nodes", C.c_void_p), ("n_kd_nodes", C.c_uint64), ("kd_bounds", C.c_void_p),
                ("kd_refs", C.c_void_p), ("n_kd_refs", C.c_uint64), ("kd_planes", C.c_void_p), ("n_kd_planes", C.c_uint64),
                ("tris", C.c_void_p), ("n_tris", C.c_uint64), ("tri_normals", C.c_void_p),
                ("bvh_nodes", C.c_void_p), ("n_bvh_nodes", C.c_uint64), ("bvh_tris", C.c_void_p), ("bvh_tri_index", C.c_void_p),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32), ("noises", C.c_void_p), ("n_noises", C.c_uint32),
                ("perlin_tables", C.c_void_p), ("n_perlin_tables", C.c_uint32), ("worley_points", C.c_void_p), ("n_worley_points", C.c_uint64),
                ("lights", C.c_void_p), ("n_lights", C.c_uint32), ("background", Background),
                ("tri_verts", C.c_void_p), ("mesh_bounds", C.c_void_p)]


class UploadInfo(C.Structure):  # rthip_upload_info
    _fields_ = [("kd_meshes", C.c_uint32), ("kd_nodes", C.c_uint32), ("kd_levels", C.c_uint32), ("walk_coop", C.c_uint32), ("kd_refs", C.c_uint64),
                ("kd_device_ms", C.c_float), ("bvh_meshes", C.c_uint32), ("bvh_nodes", C.c_uint32), ("upload_ms", C.c_float)]


class Shape(C.Structure):  # rthip_shape
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("mesh", C.c_int32), ("shape_id", C.c_uint32), ("inv", C.c_double * 12),
                ("fwd", C.c_double * 12), ("nrm", C.c_double * 12), ("plane_n", C.c_double * 4), ("rt_lin", C.c_double), ("rt_t", C.c_double),
                ("bound_c", C.c_double * 3), ("bound_r2", C.c_double)]


class MatOp(C.Structure):  # rthip_matop
    _fields_ = [("op", C.c_int32), ("noise", C.c_int32), ("first_op", C.c_int32), ("pad", C.c_int32), ("color", C.c_double * 3), ("absorptance", C.c_double)]


class Noise(C.Structure):  # rthip_noise
    _fields_ = [("type", C.c_int32), ("table", C.c_int32), ("points_base", C.c_int32), ("divisions", C.c_int32), ("octaves", C.c_uint32), ("pad", C.c_int32),
                ("domain", C.c_double), ("persistance", C.c_double), ("step", C.c_double), ("frequency", C.c_double), ("turbolence", C.c_double)]


def scene_tables(desc_ptr):
    """(shapes, material ops, noises, background type) of a flattened scene as ctypes arrays (views into the caller's memory)"""
    d = SceneDesc.from_address(desc_ptr)
    shapes = (Shape * d.n_shapes).from_address(d.shapes) if d.n_shapes else []
    mats = (MatOp * d.n_materials).from_address(d.materials) if d.n_materials else []
    noises = (Noise * d.n_noises).from_address(d.noises) if d.n_noises else []
    return shapes, mats, noises, d.background.type


class RthipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rthip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Loads librthip.so (built by raytracing_amd/csrc/Makefile or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(RTHIP_SO):
        raise RuntimeError(f"{RTHIP_SO} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'); "
                           "the Raycast path has no CPU fallback")
    try:
        # One HIP runtime per process: torch ships its own libamdhip64 / libhsa-runtime64 and finds no GPU when another
        # copy (the /opt/rocm one librthip.so would pull in) has opened the device first.  Loaded after torch, librthip.so
        # binds to torch's copy.  Without torch in the process there is nothing to collide with.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(RTHIP_SO)
    vp = C.c_void_p
    L.rthip_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rthip_destroy.argtypes = [vp]
    L.rthip_last_error.restype = C.c_char_p
    L.rthip_version.restype = C.c_char_p
    L.rthip_debug_device_allocs.restype = C.c_uint64
    L.rthip_upload_scene.argtypes = [vp, vp]
    L.rthip_update_scene.argtypes = [vp, vp]
    L.rthip_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), c_u32p, c_i32p, c_i32p]
    L.rthip_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.c_int]
    L.rthip_render_device_noalloc.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.c_int]
    L.rthip_upload_scene_ex.argtypes = [vp, vp, C.c_uint32]
    L.rthip_probe_bvh.argtypes = [vp, C.c_uint32, C.POINTER(Mesh), vp, C.c_uint64, vp, C.c_uint64]
    L.rthip_probe_kd.argtypes = [vp, C.c_uint32, C.POINTER(Mesh), vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64]
    L.rthip_get_upload_info.argtypes = [vp, C.POINTER(UploadInfo)]
    L.rthip_collect_stats.argtypes = [vp, C.POINTER(Params), C.POINTER(Stats)]
    L.rthip_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.rthip_multi_destroy.argtypes = [vp]
    L.rthip_multi_device_count.argtypes = [vp]
    L.rthip_multi_context.argtypes = [vp, C.c_int]
    L.rthip_multi_context.restype = vp
    L.rthip_multi_transport.argtypes = [vp]
    L.rthip_multi_transport.restype = C.c_char_p
    L.rthip_multi_upload_scene.argtypes = [vp, vp, C.c_uint32]
    L.rthip_multi_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), c_u32p, C.POINTER(Stats)]
    L.rthip_multi_render_batch.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.POINTER(Params), c_u32p, C.POINTER(Stats)]
    L.rthip_render_batch.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.POINTER(Params), c_u32p]
    L.rthip_multi_progress.argtypes = [vp]
    L.rthip_multi_progress.restype = C.c_float
    L.rthip_multi_interrupt.argtypes = [vp]
    L.rthip_part_pixels.argtypes = [C.POINTER(Params)]
    L.rthip_part_pixels.restype = C.c_uint64
    L.rthip_assemble.argtypes = [vp, C.POINTER(Params), vp, vp, C.c_int]
    L.rthip_render_batch_device.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.POINTER(Params), vp, C.c_int]
    L.rthip_assemble_batch.argtypes = [vp, C.POINTER(Params), C.c_uint32, vp, vp, C.c_int]
    L.rthip_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.rthip_host_free.argtypes = [vp, vp]
    L.rthip_stream.argtypes = [vp]
    L.rthip_stream.restype = vp
    L.rthip_synchronize.argtypes = [vp]
    L.rthip_progress.argtypes = [vp]
    L.rthip_progress.restype = C.c_float
    L.rthip_interrupt.argtypes = [vp]
    L.rthip_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rthip_probe_cast.argtypes = [vp, C.c_uint32, c_dp, c_dp, c_i32p, c_dp, c_dp, c_dp, c_i32p, c_i32p]
    L.rthip_probe_noise.argtypes = [vp, C.c_int32, C.c_uint32, c_dp, c_dp]
    L.rthip_probe_material.argtypes = [vp, C.c_int32, C.c_uint32, c_dp, c_dp, c_dp]
    L.rthip_probe_background.argtypes = [vp, C.c_uint32, c_dp, c_dp]
    L.rthip_probe_raycast.argtypes = [vp, C.c_uint32, C.c_uint32, c_dp, c_dp, c_dp]
    L.rthip_probe_arith.argtypes = [vp, C.c_int32, C.c_uint32, c_dp, c_dp, c_dp]
    _lib = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def make_params(width, height, max_recursion=3, aa=0, renderer=0, part_index=0, part_count=1, tile_w=0, tile_h=0, layout=0, accel=0):
    return Params(width, height, max_recursion, aa, renderer, part_index, part_count, tile_w, tile_h, layout, accel)


def part_pixels(params):
    return int(lib().rthip_part_pixels(C.byref(params)))


class Context:
    """One rthip_ctx (one per process per GPU)."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        self._check(self.L.rthip_create(device, C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise RthipError(rc, self.L.rthip_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.rthip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, desc_ptr, build_bvh=True, build_kd=True):
        """desc_ptr: `const rthip_scene_desc*` (e.g. from host.HostScene.flatten()).  build_bvh / build_kd: meshes that come without
        a BVH / without their kd-tree (the host library leaves the big ones to the device) get them built on the device
        (RTHIP_UPLOAD_BUILD_BVH / RTHIP_UPLOAD_BUILD_KD)."""
        self._check(self.L.rthip_upload_scene_ex(self.h, desc_ptr, (UPLOAD_BUILD_BVH if build_bvh else 0) | (UPLOAD_BUILD_KD if build_kd else 0)))

    def upload_info(self):
        u = UploadInfo()
        self._check(self.L.rthip_get_upload_info(self.h, C.byref(u)))
        return u

    def kd(self, mesh=0):
        """the kd-tree of a mesh as the device holds it: (mesh-table entry, nodes as a structured array [split, link, info], bounds [n, 6],
        refs, planes = the three axes' runs one after the other)"""
        m = Mesh()
        self._check(self.L.rthip_probe_kd(self.h, mesh, C.byref(m), None, None, 0, None, 0, None, 0))
        nodes = np.zeros(m.node_count, dtype=np.dtype([("split", "<f8"), ("link", "<u4"), ("info", "<u4")]))
        bounds = np.zeros((m.node_count, 6))
        refs = np.zeros(m.ref_count, np.uint32)
        planes = np.zeros(sum(m.plane_count), np.float64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self.L.rthip_probe_kd(self.h, mesh, None, vp(nodes), vp(bounds), len(nodes), vp(refs), len(refs), vp(planes), len(planes)))
        return m, nodes, bounds, refs, planes

    def bvh(self, mesh=0):
        """(mesh-table entry, nodes as a structured array [n]['c'][4], leaf-order triangle indices) of a mesh's BVH on the device"""
        m = Mesh()
        self._check(self.L.rthip_probe_bvh(self.h, mesh, C.byref(m), None, 0, None, 0))
        child = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("child", "<u4"), ("pad", "<u4")])
        nodes = np.zeros(m.bvh_node_count, dtype=np.dtype([("c", child, 4)]))
        index = np.zeros(m.tri_count, np.uint32)
        self._check(self.L.rthip_probe_bvh(self.h, mesh, None, nodes.ctypes.data_as(C.c_void_p), len(nodes), index.ctypes.data_as(C.c_void_p), len(index)))
        return m, nodes, index

    def update(self, desc_ptr):
        """like upload() for a scene whose mesh geometry is the one already uploaded (rthip_update_scene)"""
        self._check(self.L.rthip_update_scene(self.h, desc_ptr))

    def render(self, cam, params, ids=False, out=None):
        n = params.width * params.height
        px = out if out is not None else np.zeros((params.height, params.width), np.uint32)
        sh = np.full((params.height, params.width), -1, np.int32) if ids else None
        tr = np.full((params.height, params.width), -1, np.int32) if ids else None
        self._check(self.L.rthip_render(self.h, C.byref(cam), C.byref(params), px.ctypes.data_as(c_u32p),
                                        sh.ctypes.data_as(c_i32p) if ids else None,
                                        tr.ctypes.data_as(c_i32p) if ids else None))
        assert px.size == n
        return {"pixels": px, "shape": sh, "tri": tr}

    def render_batch(self, cams, params):
        """rthip_render_batch: a run of frames into host memory -> [n, height, width] uint32"""
        arr = cams if isinstance(cams, C.Array) else (Camera * len(cams))(*cams)
        px = np.zeros((len(arr), params.height, params.width), np.uint32)
        self._check(self.L.rthip_render_batch(self.h, arr, len(arr), C.byref(params), px.ctypes.data_as(c_u32p)))
        return px

    def render_device(self, cam, params, d_pixels_ptr, sync=False):
        self._check(self.L.rthip_render_device(self.h, C.byref(cam), C.byref(params), C.c_void_p(d_pixels_ptr), int(sync)))

    def render_device_noalloc(self, cam, params, d_pixels_ptr, sync=False):
        """rthip_render_device with the workspace in the library's static device array: no device allocation in the call"""
        self._check(self.L.rthip_render_device_noalloc(self.h, C.byref(cam), C.byref(params), C.c_void_p(d_pixels_ptr), int(sync)))

    def assemble(self, params, d_gathered_ptr, d_frame_ptr, sync=False):
        self._check(self.L.rthip_assemble(self.h, C.byref(params), C.c_void_p(d_gathered_ptr), C.c_void_p(d_frame_ptr), int(sync)))

    def render_batch_device(self, cams, params, d_pixels_ptr, sync=False):
        """cams: sequence of Camera (1..MAX_BATCH); d_pixels_ptr: len(cams) frame (or part) buffers one after the other."""
        arr = cams if isinstance(cams, C.Array) else (Camera * len(cams))(*cams)
        self._check(self.L.rthip_render_batch_device(self.h, arr, len(arr), C.byref(params), C.c_void_p(d_pixels_ptr), int(sync)))

    def assemble_batch(self, params, n_frames, d_gathered_ptr, d_frames_ptr, sync=False):
        self._check(self.L.rthip_assemble_batch(self.h, C.byref(params), n_frames, C.c_void_p(d_gathered_ptr), C.c_void_p(d_frames_ptr), int(sync)))

    def host_pixels(self, width, height):
        """(height, width) uint32 array in page-locked host memory (free with host_free before close())"""
        p = C.c_void_p()
        self._check(self.L.rthip_host_alloc(self.h, 4 * width * height, C.byref(p)))
        arr = np.ctypeslib.as_array(C.cast(p, c_u32p), shape=(height, width))
        return arr

    def host_free(self, array):
        self._check(self.L.rthip_host_free(self.h, C.c_void_p(array.ctypes.data)))

    def synchronize(self):
        self._check(self.L.rthip_synchronize(self.h))

    def stream(self):
        return self.L.rthip_stream(self.h)

    def stats(self):
        s = Stats()
        self._check(self.L.rthip_get_stats(self.h, C.byref(s)))
        return s

    def probe_arith(self, op, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        b = None if b is None else np.ascontiguousarray(b, np.float64)
        out = np.zeros_like(a)
        self._check(self.L.rthip_probe_arith(self.h, op, a.size, _dp(a), _dp(b), _dp(out)))
        return out

    def probe_cast(self, o, d):
        o = np.ascontiguousarray(o, np.float64).reshape(-1, 3); d = np.ascontiguousarray(d, np.float64).reshape(-1, 3)
        n = len(o)
        hit = np.zeros(n, np.int32); sh = np.zeros(n, np.int32); tr = np.zeros(n, np.int32)
        pt = np.zeros((n, 3)); lp = np.zeros((n, 3)); nr = np.zeros((n, 3))
        self._check(self.L.rthip_probe_cast(self.h, n, _dp(o), _dp(d), hit.ctypes.data_as(c_i32p), _dp(pt), _dp(lp), _dp(nr),
                                            sh.ctypes.data_as(c_i32p), tr.ctypes.data_as(c_i32p)))
        return {"hit": hit, "point": pt, "local": lp, "normal": nr, "shape": sh, "tri": tr}

    def probe_noise(self, noise, p):
        p = np.ascontiguousarray(p, np.float64).reshape(-1, 3); out = np.zeros(len(p))
        self._check(self.L.rthip_probe_noise(self.h, noise, len(p), _dp(p), _dp(out)))
        return out

    def probe_material(self, mat, p):
        p = np.ascontiguousarray(p, np.float64).reshape(-1, 3); col = np.zeros((len(p), 3)); ab = np.zeros(len(p))
        self._check(self.L.rthip_probe_material(self.h, mat, len(p), _dp(p), _dp(col), _dp(ab)))
        return col, ab

    def probe_background(self, d):
        d = np.ascontiguousarray(d, np.float64).reshape(-1, 3); col = np.zeros((len(d), 3))
        self._check(self.L.rthip_probe_background(self.h, len(d), _dp(d), _dp(col)))
        return col

    def probe_raycast(self, o, d, max_recursion=3):
        o = np.ascontiguousarray(o, np.float64).reshape(-1, 3); d = np.ascontiguousarray(d, np.float64).reshape(-1, 3)
        col = np.zeros((len(o), 3))
        self._check(self.L.rthip_probe_raycast(self.h, max_recursion, len(o), _dp(o), _dp(d), _dp(col)))
        return col


class MultiContext:
    """rthip_multi: several GPUs of the node driven from this one process (interleaved tiles, RCCL / peer-copy gather to the
    first device, device assemble)."""

    def __init__(self, devices):
        self.L = lib()
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = self.L.rthip_multi_create(arr, len(devices), C.byref(h))
        if rc != 0:
            raise RthipError(rc, self.L.rthip_last_error().decode())
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise RthipError(rc, self.L.rthip_last_error().decode())

    def transport(self):
        return self.L.rthip_multi_transport(self.h).decode()

    def upload(self, desc_ptr, keep_geometry=False):
        self._check(self.L.rthip_multi_upload_scene(self.h, desc_ptr, UPLOAD_KEEP_GEOMETRY if keep_geometry else UPLOAD_BUILD_BVH))

    def render(self, cam, params):
        px = np.zeros((params.height, params.width), np.uint32)
        st = Stats()
        self._check(self.L.rthip_multi_render(self.h, C.byref(cam), C.byref(params), px.ctypes.data_as(c_u32p), C.byref(st)))
        return {"pixels": px, "rays": st.rays_primary + st.rays_shadow + st.rays_reflect, "stats": st}

    def render_batch(self, cams, params):
        """a run of frames in one round over the devices -> [n, height, width] uint32"""
        arr = cams if isinstance(cams, C.Array) else (Camera * len(cams))(*cams)
        px = np.zeros((len(arr), params.height, params.width), np.uint32)
        st = Stats()
        self._check(self.L.rthip_multi_render_batch(self.h, arr, len(arr), C.byref(params), px.ctypes.data_as(c_u32p), C.byref(st)))
        return {"pixels": px, "rays": st.rays_primary + st.rays_shadow + st.rays_reflect, "stats": st}

    def close(self):
        if getattr(self, "h", None):
            self.L.rthip_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
