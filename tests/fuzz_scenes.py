"""Seeded random scenes for the fuzz parity tests: spheres / planes / small meshes (bumpy grids, octahedra, soups of large
overlapping triangles) under nested nodes with random
rotation and non-uniform (sometimes extreme) scale, every light type, uniform and noise-interpolated materials,
both backgrounds.  Everything derives from the seed through PCG32, so a failing seed reproduces anywhere."""
import numpy as np

from raytracing_amd import scenes


class _Rng:
    def __init__(self, seed):
        self.g = scenes.PCG32(seed)

    def u(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * (self.g.next_u32() / 2.0**32)

    def i(self, n):
        return self.g.next_u32() % n

    def color(self):
        return self.g.next_u32() & 0xffffff


def _small_mesh(r, kind):
    if kind == 0:   # bumpy grid, vertex normals
        v, n = scenes.heightfield_mesh(cells=6 + r.i(6), seed=1 + r.i(1000), extent=1.0 + r.u(0, 2))
        return v, n, scenes.NORMAL_VERTEX
    if kind == 2:   # a soup of a few LARGE random triangles that overlap and cross each other (round 2: found the quad walk's
                    # stack one level short on BVHs that are a lone root), face or vertex normals
        n_tris = 1 + r.i(24)
        v = np.array([[r.u(-1.5, 1.5) for _ in range(9)] for _ in range(n_tris)])
        n = np.array([[r.u(-1, 1) for _ in range(9)] for _ in range(n_tris)])
        return v, n, scenes.NORMAL_VERTEX if r.i(2) else scenes.NORMAL_FACE
    # closed octahedron, face normals, outward winding
    p = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    faces = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    v = np.array([np.concatenate([p[a], p[b], p[c]]) for a, b, c in faces])
    return v, np.zeros_like(v), scenes.NORMAL_FACE


def random_scene(seed):
    r = _Rng(seed)
    sd = scenes.SceneDesc(f"fuzz scene {seed}")
    mats = [sd.uniform_material(r.color(), r.u(0.05, 1.0)) for _ in range(3)]
    mats.append(sd.uniform_material(0xffffff, 0.0))                       # a perfect mirror
    noises = [sd.checker(r.u(0.5, 4)), sd.perlin(r.u(0.5, 3), seed=seed * 7 + 1), sd.worley(r.u(1, 4), 2 + r.i(4), seed=seed * 7 + 2),
              sd.marble(r.u(1, 3), r.u(1, 4), r.u(1, 5), seed=seed * 7 + 3)]
    for k in range(3):
        mats.append(sd.interpolated_material(noises[r.i(len(noises))], mats[r.i(len(mats))], mats[r.i(len(mats))]))
    sun = sd.directional_light(r.color() | 0x808080, r.u(-1, 1), r.u(0.2, 1), r.u(-1, 1))
    if r.i(2):
        sd.point_light(r.color() | 0x404040, r.u(3, 25), pos=(r.u(-4, 4), r.u(1, 6), r.u(-6, 2)))
    if r.i(2):
        sd.ambient_light(r.color() & 0x303030)
    if r.i(3) == 0:
        sd.directional_light(0xffffff, 0, 1, 0, enabled=0)
    if r.i(2):
        sd.skybox(r.color(), r.color(), sun, seed=seed * 7 + 4)
    else:
        sd.color_background(r.color())
    nodes = [-1]
    for k in range(5 + r.i(6)):
        parent = nodes[r.i(len(nodes))]
        extreme = r.i(6) == 0
        scl = tuple(r.u(0.05, 0.3) if extreme and r.i(2) else r.u(0.4, 2.5) for _ in range(3))
        node = sd.node(parent, (r.u(-3, 3), r.u(-1, 2.5), r.u(-9, -2)) if parent == -1 else (r.u(-1.5, 1.5), r.u(-1, 1.5), r.u(-1.5, 1.5)),
                       (r.u(-3.2, 3.2), r.u(-3.2, 3.2), r.u(-3.2, 3.2)) if r.i(3) else (0, 0, 0), scl)
        nodes.append(node)
        kind = r.i(4)
        mat = mats[r.i(len(mats))]
        if kind <= 1:
            sd.sphere(mat, node=node)
        elif kind == 2:
            v, n, mode = _small_mesh(r, r.i(3))
            sd.mesh(mat, v, n, mode, node=node)
        else:
            sd.plane(mat, r.u(-0.3, 0.3), 1, r.u(-0.3, 0.3), node=sd.node(-1, (0, r.u(-3, -1), 0), (0, 0, r.u(-0.1, 0.1)), (1, 1, 1)))
    sd.look_at((r.u(-2, 2), r.u(0, 3), r.u(2, 5)), (r.u(-1, 1), r.u(-0.5, 1), -5))
    return sd


FUZZ_SEEDS = list(range(1, 33))
