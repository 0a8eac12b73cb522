"""Scene descriptions for the Raycast hot path, independent of any backend.

`SceneDesc` records the calls the reference's Lua DSL would make (Sandbox/Sandbox.cpp:595-750:
reUniformMaterial, reInterpolatedMaterial, reCheckerBoard/rePerlin/reWorley/reMarble,
rePosition/reScale, reSphere/rePlane/reObjMesh, re*Light, reCameraPos, reSkyBox) and replays
them onto any library exporting the sb_* entry points (see sbapi.py).

Noise tables are carried explicitly (drawn here from a PCG32 stream) so that scenes do not
depend on libc rand().  `config(n)` builds the five BASELINE.json configs as
frozen in SURVEY.md §8(d).
"""
import os

import numpy as np

from .sbapi import BuiltScene, _vec3, c_dp, c_u8p

LIGHT_DIRECTIONAL, LIGHT_AMBIENT, LIGHT_POINT = 0, 1, 2  # Raytracer/Scene.h:17
NORMAL_FACE, NORMAL_VERTEX = 0, 1                        # Raytracer/Scene.h:18
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class PCG32:
    """Minimal PCG-XSH-RR 64/32 (O'Neill), scalar stream; deterministic on every platform."""
    MULT = 6364136223846793005
    MASK = (1 << 64) - 1

    def __init__(self, seed, seq=54):
        self.state = 0
        self.inc = ((seq << 1) | 1) & self.MASK
        self.next_u32()
        self.state = (self.state + seed) & self.MASK
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.MASK
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def u32_array(self, n):
        """n draws, vectorised by jumping: each element = output at state_k (LCG closed form)."""
        # sequential but in pure ints is too slow for 1M draws; use numpy uint64 wraparound
        out = np.empty(n, dtype=np.uint32)
        state = np.uint64(self.state)
        mult = np.uint64(self.MULT)
        inc = np.uint64(self.inc)
        # block-jump: compute states for a block with cumulative affine maps
        B = 1 << 16
        # affine map powers a_k, c_k s.t. s_k = a_k*s_0 + c_k
        a = np.empty(B, dtype=np.uint64)
        c = np.empty(B, dtype=np.uint64)
        with np.errstate(over="ignore"):
            ak, ck = np.uint64(1), np.uint64(0)
            for k in range(B):
                a[k], c[k] = ak, ck
                ak, ck = ak * mult, ck * mult + inc
            a_blk, c_blk = ak, ck
            done = 0
            while done < n:
                m = min(B, n - done)
                olds = a[:m] * state + c[:m]
                xs = (((olds >> np.uint64(18)) ^ olds) >> np.uint64(27)).astype(np.uint32)
                rot = (olds >> np.uint64(59)).astype(np.uint32)
                out[done:done + m] = (xs >> rot) | (xs << ((np.uint32(32) - rot) & np.uint32(31)))
                state = (a[m - 1] * state + c[m - 1]) * mult + inc if m < B else a_blk * state + c_blk
                done += m
        self.state = int(state)
        return out


def perlin_table(seed):
    """256 entries in [0,256) like Perlin::NewSeed (noise/Perlin.cpp:17-23), PCG32 instead of rand()."""
    g = PCG32(seed)
    return np.array([g.next_u32() % 256 for _ in range(256)], dtype=np.uint8)


def worley_points(seed, div):
    """div^3 feature points, one per cell, z/y/x loop order, x fastest (noise/Worley.cpp:9-26)."""
    g = PCG32(seed)
    step = 1.0 / div
    pts = np.zeros((div ** 3, 3))
    i = 0
    for z in range(div):
        for y in range(div):
            for x in range(div):
                for k, c in enumerate((x, y, z)):
                    u = g.next_u32() / 4294967296.0
                    lo, hi = c * step, (c + 1) * step
                    pts[i, k] = lo + (hi - lo) * u
                i += 1
    return pts


class SceneDesc:
    def __init__(self, name=""):
        self.name = name
        self.ops = []
        self._n = {"mat": 0, "noise": 0, "node": 1, "light": 0}
        self.pos = (0.0, 0.0, 0.0)
        self.scl = (1.0, 1.0, 1.0)
        self.meshes = {}
        self.camera_pos = (0.0, 0.0, 0.0)
        self.camera_dir = (0.0, 0.0, -1.0)
        self.noise_info = {}

    def _add(self, kind, op):
        self.ops.append(op)
        i = self._n[kind]
        self._n[kind] += 1
        return i

    # materials / noises -----------------------------------------------------------
    def uniform_material(self, rgb, absorptance):
        return self._add("mat", ("mat_uniform", int(rgb), float(absorptance)))

    def interpolated_material(self, noise, m0, m1):
        return self._add("mat", ("mat_interp", noise, m0, m1))

    def checker(self, domain):
        i = self._add("noise", ("noise_checker", float(domain)))
        self.noise_info[i] = ("checker", float(domain))
        return i

    def perlin(self, domain, seed):
        i = self._add("noise", ("noise_perlin", float(domain), perlin_table(seed)))
        self.noise_info[i] = ("perlin", float(domain))
        return i

    def worley(self, domain, div, seed):
        i = self._add("noise", ("noise_worley", float(domain), int(div), worley_points(seed, div)))
        self.noise_info[i] = ("worley", float(domain), int(div))
        return i

    def marble(self, domain, freq, turb, seed):
        i = self._add("noise", ("noise_marble", float(domain), float(freq), float(turb), perlin_table(seed)))
        self.noise_info[i] = ("marble", float(domain))
        return i

    # transforms / shapes ------------------------------------------------------------
    def position(self, x, y, z):
        self.pos = (float(x), float(y), float(z))

    def scale(self, x, y, z):
        self.scl = (float(x), float(y), float(z))

    def node(self, parent=-1, pos=None, rot=(0, 0, 0), scale=None):
        pos = self.pos if pos is None else pos
        scale = self.scl if scale is None else scale
        return self._add("node", ("node", parent, tuple(map(float, pos)), tuple(map(float, rot)),
                                  tuple(map(float, scale))))

    def sphere(self, mat, node=None):
        node = self.node() if node is None else node
        self.ops.append(("sphere", node, mat))
        return node

    def plane(self, mat, nx, ny, nz, node=None):
        node = self.node() if node is None else node
        self.ops.append(("plane", node, mat, (float(nx), float(ny), float(nz))))
        return node

    def mesh(self, mat, verts, normals, normal_mode=NORMAL_VERTEX, node=None):
        node = self.node() if node is None else node
        verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
        normals = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 9)
        self.meshes[node] = len(verts)
        self.ops.append(("mesh", node, mat, normal_mode, verts, normals))
        return node

    # lights / camera / background -----------------------------------------------------
    def ambient_light(self, rgb):
        return self._add("light", ("light", LIGHT_AMBIENT, int(rgb), (0.0, 1.0, 0.0), (0.0, 0.0, 0.0), 5.0, 1))

    def directional_light(self, rgb, x, y, z, enabled=1):
        d = np.array([x, y, z], dtype=np.float64)
        ln = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])  # Vector3::Normalized (Sandbox.cpp:714)
        d = d / ln if ln > 0 else d * 0
        return self._add("light", ("light", LIGHT_DIRECTIONAL, int(rgb), tuple(d), (0.0, 0.0, 0.0), 5.0, enabled))

    def point_light(self, rgb, attenuation, pos=None):
        pos = self.pos if pos is None else pos
        return self._add("light", ("light", LIGHT_POINT, int(rgb), (0.0, 1.0, 0.0), tuple(map(float, pos)),
                                   float(attenuation), 1))

    def camera(self, pos, direction=(0, 0, -1)):
        self.camera_pos = tuple(map(float, pos))
        self.camera_dir = tuple(map(float, direction))

    def look_at(self, pos, target):
        d = np.array(target, dtype=np.float64) - np.array(pos, dtype=np.float64)
        ln = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])  # Scene.h:84 LookAt
        self.camera(pos, tuple(d / ln))

    def color_background(self, rgb):
        self.ops.append(("bg_color", int(rgb)))

    def skybox(self, rgb0, rgb1, sun, seed):
        self.ops.append(("bg_skybox", int(rgb0), int(rgb1), sun, perlin_table(seed)))

    def adopt_tables(self, built):
        """Replaces this description's noise tables by those of an already built scene (sbapi.BuiltScene) with the
        same noises in the same order, e.g. one whose constructors drew them from rand() like the reference's."""
        ni = 0
        for k, op in enumerate(self.ops):
            if op[0] in ("noise_checker", "noise_perlin", "noise_worley", "noise_marble"):
                if op[0] in ("noise_perlin", "noise_marble"):
                    self.ops[k] = op[:-1] + (built.perlin_table(ni),)
                elif op[0] == "noise_worley":
                    self.ops[k] = op[:-1] + (built.worley_points(ni, op[2]),)
                ni += 1
            elif op[0] == "bg_skybox":
                self.ops[k] = op[:-1] + (built.skybox_table(),)

    # replay -------------------------------------------------------------------------
    def build(self, sblib):
        L = sblib.lib
        h = L.sb_scene_new()
        if not h:
            raise RuntimeError("sb_scene_new failed")
        keep = []

        def dp(a):
            keep.append(a)
            return a.ctypes.data_as(c_dp)

        def u8(a):
            keep.append(a)
            return a.ctypes.data_as(c_u8p)

        def chk(rc, what):
            if rc is not None and rc < 0:
                raise RuntimeError(f"{what} failed ({rc}) in {sblib.name}")

        for op in self.ops:
            k = op[0]
            if k == "mat_uniform":
                chk(L.sb_mat_uniform(h, op[1], op[2]), k)
            elif k == "mat_interp":
                chk(L.sb_mat_interp(h, op[1], op[2], op[3]), k)
            elif k == "noise_checker":
                chk(L.sb_noise_checker(h, op[1]), k)
            elif k == "noise_perlin":
                chk(L.sb_noise_perlin(h, op[1], u8(op[2])), k)
            elif k == "noise_worley":
                chk(L.sb_noise_worley(h, op[1], op[2], dp(op[3])), k)
            elif k == "noise_marble":
                chk(L.sb_noise_marble(h, op[1], op[2], op[3], u8(op[4])), k)
            elif k == "node":
                chk(L.sb_node(h, op[1], _vec3(op[2]), _vec3(op[3]), _vec3(op[4])), k)
            elif k == "sphere":
                chk(L.sb_sphere(h, op[1], op[2]), k)
            elif k == "plane":
                chk(L.sb_plane(h, op[1], op[2], _vec3(op[3])), k)
            elif k == "mesh":
                chk(L.sb_mesh(h, op[1], op[2], op[3], len(op[4]), dp(op[4]), dp(op[5])), k)
            elif k == "light":
                chk(L.sb_light(h, op[1], op[2], _vec3(op[3]), _vec3(op[4]), op[5], op[6]), k)
            elif k == "bg_color":
                L.sb_bg_color(h, op[1])
            elif k == "bg_skybox":
                L.sb_bg_skybox(h, op[1], op[2], op[3], u8(op[4]))
            else:
                raise ValueError(k)
        L.sb_camera(h, _vec3(self.camera_pos), _vec3(self.camera_dir))
        return BuiltScene(sblib, h)


# ------------------------------------------------------------------------------------
# deterministic trig for the synthetic meshes: only + - * / so every platform produces
# bit-identical vertices (numpy's sin/cos differ in the last ulp between SIMD back ends)
_TWO_PI = 6.283185307179586
_S = [-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800,
      -1.0 / 1307674368000, 1.0 / 355687428096000]


def det_sin(x):
    x = np.asarray(x, dtype=np.float64)
    k = np.floor(x / _TWO_PI + 0.5)
    r = x - k * _TWO_PI           # [-pi, pi]
    r2 = r * r
    p = np.full_like(r, _S[-1])
    for c in _S[-2::-1]:
        p = p * r2 + c
    return r + r * r2 * p


def det_cos(x):
    return det_sin(np.asarray(x, dtype=np.float64) + 1.5707963267948966)


def heightfield_mesh(cells=1024, seed=1234, extent=8.0):
    """SURVEY §8(d) config 4 mesh: cells x cells height field, 2 triangles per cell, x,z in
    [-extent, extent], y = 0.35 sin(1.7x) cos(1.3z) + 0.1 h(i,j), h = PCG32(seed) in [-1,1];
    counter-clockwise seen from +y; vertex normals from the analytic gradient of the smooth part."""
    n = cells + 1
    g = PCG32(seed)
    h = g.u32_array(n * n).astype(np.float64) / 4294967296.0 * 2.0 - 1.0
    h = h.reshape(n, n)                          # h[j(z), i(x)]
    xs = -extent + (2.0 * extent) * (np.arange(n, dtype=np.float64) / cells)
    X, Z = np.meshgrid(xs, xs)                   # X[j,i] = xs[i], Z[j,i] = xs[j]
    Y = 0.35 * det_sin(1.7 * X) * det_cos(1.3 * Z) + 0.1 * h
    dYdx = 0.35 * 1.7 * det_cos(1.7 * X) * det_cos(1.3 * Z)
    dYdz = -0.35 * 1.3 * det_sin(1.7 * X) * det_sin(1.3 * Z)
    N = np.stack([-dYdx, np.ones_like(dYdx), -dYdz], axis=-1)
    N = N / np.sqrt((N * N).sum(-1, keepdims=True))
    P = np.stack([X, Y, Z], axis=-1)
    p00, p10 = P[:-1, :-1], P[:-1, 1:]           # (z j, x i), (z j, x i+1)
    p01, p11 = P[1:, :-1], P[1:, 1:]
    n00, n10, n01, n11 = N[:-1, :-1], N[:-1, 1:], N[1:, :-1], N[1:, 1:]
    # CCW from +y: (v1-v0) x (v2-v1) must point to +y -> order p00, p01, p11 and p00, p11, p10
    tv = np.stack([np.stack([p00, p01, p11], axis=2), np.stack([p00, p11, p10], axis=2)], axis=2)
    tn = np.stack([np.stack([n00, n01, n11], axis=2), np.stack([n00, n11, n10], axis=2)], axis=2)
    return tv.reshape(-1, 9).copy(), tn.reshape(-1, 9).copy()


def load_bunny():
    """Triangles of Sandbox/res/bunny.obj_ group 'bunny' as the reference's reObjMesh builds them
    (Sandbox.cpp:669-693): tests/golden/bunny_mesh.npz, written by tools/make_bunny_fixture.py."""
    z = np.load(os.path.join(GOLDEN_DIR, "bunny_mesh.npz"))
    return z["verts"], z["normals"]


# ------------------------------------------------------------------------------------
def _demo1(sd, with_perlin_sphere):
    """'Demo 1 - Spheres' (Sandbox/res/scenes.xml:2-39), construction order kept."""
    red = sd.uniform_material(0xf44336, 0.6)
    green = sd.uniform_material(0x4CAF50, 0.6)
    blue = sd.uniform_material(0x2196F3, 0.6)
    dark_red = sd.uniform_material(0x8d1007, 0.9)
    dark_green = sd.uniform_material(0x265728, 0.9)
    dark_blue = sd.uniform_material(0x074b83, 0.9)
    glass_black = sd.uniform_material(0x000000, 0.8)
    glass_white = sd.uniform_material(0xffffff, 0.4)
    sd.uniform_material(0xffffff, 0.0)   # mirror (unused in the demo)
    sd.uniform_material(0x000000, 1)     # solid_black (unused)
    sd.camera((0, 3, 4))
    sun = sd.directional_light(0xffddee, 1, 1, -1)
    sd.skybox(0x4444ff, 0xffffff, sun, seed=101)
    marble = sd.marble(2, 2, 4, seed=102)
    worley = sd.worley(2, 10, seed=103)
    for x, m in ((-4, red), (0, green), (4, blue)):
        sd.position(x, 0, -4); sd.sphere(m)
    for x, a, b in ((-4, dark_red, red), (0, dark_green, green), (4, dark_blue, blue)):
        sd.position(x, 0, -8); sd.sphere(sd.interpolated_material(marble, a, b))
    for x, a, b in ((-4, dark_red, red), (0, dark_green, green), (4, dark_blue, blue)):
        sd.position(x, 0, -12); sd.sphere(sd.interpolated_material(worley, a, b))
    if with_perlin_sphere:
        # one extra sphere with InterpolatedMaterial(Perlin(2), ...)
        perlin = sd.perlin(2, seed=104)
        sd.position(0, 2.5, -8); sd.sphere(sd.interpolated_material(perlin, dark_blue, glass_white))
    sd.position(0, -1, 0)
    sd.plane(sd.interpolated_material(sd.checker(16), glass_black, glass_white), 0, 1, 0)


def config(n, **kw):
    """Returns (SceneDesc, render kwargs) for BASELINE.json configs[n-1]."""
    if n == 1:
        sd = SceneDesc("config1 sphere+plane, ColorBackground, 640x480, 1 bounce")
        sd.camera((0, 1, 5))
        a = sd.uniform_material(0xf44336, 0.6)
        b = sd.uniform_material(0xffffff, 0.7)
        sd.position(0, 0, 0); sd.sphere(a)
        sd.position(0, -1, 0); sd.plane(b, 0, 1, 0)
        sd.directional_light(0xffddee, 1, 1, -1)
        sd.ambient_light(0x202020)
        sd.color_background(0x4444ff)
        return sd, dict(width=640, height=480, max_recursion=1)
    if n == 2:
        sd = SceneDesc("config2 Demo 1 spheres + Perlin sphere, SkyBox, 1080p, 3 bounces")
        _demo1(sd, with_perlin_sphere=True)
        return sd, dict(width=1920, height=1080, max_recursion=3)
    if n == 3:
        sd = SceneDesc("config3 Demo 4 bunny (Sandbox/res/scenes.xml:124-140), 1080p, 3 bounces")
        cam_z = kw.get("cam_z", 5)
        sd.camera((0, 0, cam_z))
        sd.position(0, -1, 0)
        black = sd.uniform_material(0, 0.7)
        white = sd.uniform_material(0xffffff, 0.7)
        sd.plane(sd.interpolated_material(sd.checker(1), black, white), 0, 1, 0)
        bunny_mat = sd.uniform_material(0xff2222, 0.5)
        v, nrm = load_bunny()
        sd.mesh(bunny_mat, v, nrm, NORMAL_VERTEX)
        sun = sd.directional_light(0xffcccc, 1, 1, 1)
        sd.skybox(0xff0000, 0x0000ff, sun, seed=301)
        return sd, dict(width=1920, height=1080, max_recursion=3)
    if n in (4, 5):
        cells = kw.get("cells", 1024)
        sd = SceneDesc(f"config{n} synthetic height-field mesh(es) {cells}x{cells} + 16 reflective spheres")
        ground = sd.uniform_material(0x88aa66, 0.8)
        chrome = sd.uniform_material(0xffffff, 0.2)
        sun = sd.directional_light(0xffffff, 1, 2, 1)
        sd.skybox(0x4444ff, 0xffffff, sun, seed=401)
        if n == 4:
            v, nrm = heightfield_mesh(cells, seed=1234)
            sd.position(0, 0, 0); sd.mesh(ground, v, nrm, NORMAL_VERTEX)
        else:
            place = [((0, 0, 0), (0, 0, 0), (1, 1, 1)),
                     ((-17, 0, -6), (0, 0.35, 0), (1, 1.5, 1)),
                     ((17, 0, -6), (0.1, -0.4, 0), (1, 1, 1.2)),
                     ((0, -0.5, -22), (0, 0.8, 0.05), (1.5, 2, 1)),
                     ((0, 9, -30), (-0.6, 0, 0), (1.2, 1, 1))]
            for s, (pos, rot, scl) in enumerate(place, start=1):
                v, nrm = heightfield_mesh(cells, seed=s)
                sd.mesh(ground, v, nrm, NORMAL_VERTEX, node=sd.node(-1, pos, rot, scl))
            sd.position(0, -3, 0)
            black = sd.uniform_material(0, 0.7)
            white = sd.uniform_material(0xffffff, 0.7)
            sd.plane(sd.interpolated_material(sd.checker(4), black, white), 0, 1, 0)
        sd.scale(1, 1, 1)
        for j in range(4):
            for i in range(4):
                sd.position(-6 + 4 * i, 2, -6 + 4 * j); sd.sphere(chrome)
        sd.look_at((0, 6, 14), (0, 0, 0))
        if n == 4:
            return sd, dict(width=3840, height=2160, max_recursion=4)
        return sd, dict(width=7680, height=4320, max_recursion=8)
    raise ValueError(n)


def demo3_point_lights():
    """'Demo 3 - Spheres (night/lights)' (scenes.xml:79-123): exercises Point lights."""
    sd = SceneDesc("demo3 night + 4 point lights")
    red = sd.uniform_material(0xf44336, 0.6)
    green = sd.uniform_material(0x4CAF50, 0.6)
    blue = sd.uniform_material(0x2196F3, 0.6)
    dark_red = sd.uniform_material(0x8d1007, 0.9)
    dark_green = sd.uniform_material(0x265728, 0.9)
    dark_blue = sd.uniform_material(0x074b83, 0.9)
    glass_black = sd.uniform_material(0x000000, 0.8)
    glass_white = sd.uniform_material(0xffffff, 0.4)
    sd.camera((0, 3, 4))
    moon = sd.directional_light(0x353532, 1, 1, -1)
    sd.skybox(0x000033, 0x444444, moon, seed=201)
    marble = sd.marble(2, 2, 4, seed=202)
    worley = sd.worley(2, 10, seed=203)
    for x, m in ((-4, red), (0, green), (4, blue)):
        sd.position(x, 0, -4); sd.sphere(m)
    for x, a, b in ((-4, dark_red, red), (0, dark_green, green), (4, dark_blue, blue)):
        sd.position(x, 0, -8); sd.sphere(sd.interpolated_material(marble, a, b))
    for x, a, b in ((-4, dark_red, red), (0, dark_green, green), (4, dark_blue, blue)):
        sd.position(x, 0, -12); sd.sphere(sd.interpolated_material(worley, a, b))
    sd.position(0, -1, 0)
    sd.plane(sd.interpolated_material(sd.checker(16), glass_black, glass_white), 0, 1, 0)
    for p, c in (((-2, 3, -6), 0xff0000), ((2, 3, -6), 0x00ff00), ((-2, 3, -10), 0x0000ff), ((2, 3, -10), 0x00ffff)):
        sd.point_light(c, 10, pos=p)
    return sd, dict(width=640, height=360, max_recursion=3)


def nested_transforms():
    """Not in the demos: nested nodes with rotation + non-uniform scale, a shape on an interior node,
    a nested InterpolatedMaterial tree, a disabled light and a tiny mesh in Face mode."""
    sd = SceneDesc("nested transforms / rotations / material tree")
    a = sd.uniform_material(0xcc8844, 0.7)
    b = sd.uniform_material(0x2244cc, 0.3)
    c = sd.uniform_material(0xffffff, 0.9)
    inner = sd.interpolated_material(sd.perlin(1.5, seed=501), a, b)
    outer = sd.interpolated_material(sd.worley(3, 4, seed=502), inner, c)
    sd.directional_light(0xffffff, 0.3, 1, 0.5)
    sd.directional_light(0xff0000, -1, 1, 0, enabled=0)
    sd.point_light(0x88ff88, 12, pos=(3, 4, 2))
    sd.ambient_light(0x101010)
    sd.skybox(0x224488, 0xeeeeff, 0, seed=503)
    parent = sd.node(-1, (0, 0.5, -6), (0.3, 0.6, -0.2), (1.5, 1, 1))
    sd.sphere(outer, node=parent)
    child = sd.node(parent, (2.2, 0.4, 0), (0.0, 1.1, 0.4), (0.5, 0.8, 0.5))
    sd.sphere(a, node=child)
    grand = sd.node(child, (0, 2.5, 0), (0.7, 0, 0), (1, 1, 2))
    verts = np.array([[-1, 0, -1, -1, 0, 1, 1, 0, 1], [-1, 0, -1, 1, 0, 1, 1, 0, -1],
                      [-1, 0, -1, 0, 1.5, 0, -1, 0, 1], [1, 0, 1, 0, 1.5, 0, 1, 0, -1],
                      [-1, 0, 1, 0, 1.5, 0, 1, 0, 1], [1, 0, -1, 0, 1.5, 0, -1, 0, -1]], dtype=np.float64)
    sd.mesh(b, verts, np.zeros_like(verts), NORMAL_FACE, node=grand)
    sd.plane(sd.interpolated_material(sd.checker(2), c, a), 0.1, 1, 0.05, node=sd.node(-1, (0, -1.5, 0), (0, 0, 0.05), (1, 1, 1)))
    sd.camera((0.5, 2, 3), (-0.05, -0.2, -1))
    return sd, dict(width=320, height=200, max_recursion=4)
