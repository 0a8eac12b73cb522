// rt_core.hpp — per-ray device functions of the Raycast hot path (gfx950), written once as
// __host__ __device__ so tests can also compile them with g++ in a GPU-less container
// (tests/hostcore).  The product only ever runs them inside the HIP kernels of rthip.hip.
//
// Numerics contract: IEEE double, the reference's operation order, no FMA
// contraction (-ffp-contract=off), float round-trips reproduced where the reference has them.
// Each function cites the reference file:line whose arithmetic it reproduces; the structure
// (flat pre-order shape table, ordered kd traversal with an explicit stack, postfix material
// programs, iterative bounce) is this library's own.
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "rthip.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#define RT_HD_NOINLINE __host__ __device__ __noinline__
#else
#define RT_HD inline
#define RT_HD_NOINLINE inline
#endif

namespace rtc {

struct V3 {
    double x, y, z;
};
struct Col {
    double r, g, b;
};

// Device-resident flattened scene (pointers into HBM; see DESIGN.md "data layout").
struct SceneView {
    const rthip_shape *shapes;
    uint32_t n_shapes;
    const rthip_mesh *meshes;
    const rthip_kdnode *kd_nodes;
    const double *kd_bounds;
    const uint32_t *kd_refs;
    const double *kd_planes;
    const rthip_tri *tris;
    const double *tri_normals;
    const rthip_bvhnode *bvh_nodes;
    const rthip_tri *bvh_tris;
    const uint32_t *bvh_tri_index;
    int32_t accel;  // 0 = BVH, 1 = reference kd-tree
    const rthip_matop *materials;
    const rthip_noise *noises;
    const uint8_t *perlin_tables;
    const double *worley_points;
    const rthip_light *lights;
    uint32_t n_lights;
    rthip_background bg;
    // [n_lights][n_shapes][3]: the direction of a DIRECTIONAL light in the frame of every shape, normalized(inv * direction) - what
    // Scene::CastRay computes per (shadow ray, shape) although it is the same for every ray; filled at upload by a kernel that runs
    // the same functions (rthip.hip: k_light_dirs), so bit for bit what a lane would compute.  nullptr: not available.
    const double *shadow_ld = nullptr;
    // [n_shapes][3]: the world-space normal of a PLANE, normalized(nrm * plane_n) - the same for every hit of it (hit_details);
    // filled by the same kernel.  nullptr: not available.
    const double *plane_wn = nullptr;
};

// Shape and light records are indexed by wave-uniform loop counters.  On the device they are read
// through the constant address space (same memory, but the loads are invariant and therefore scalar:
// one s_load per wave, the matrices land in SGPRs instead of 24 VGPRs per lane).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) rthip_shape rt_uniform_shape;
typedef const __attribute__((address_space(4))) rthip_light rt_uniform_light;
typedef const __attribute__((address_space(4))) rthip_mesh rt_uniform_mesh;
__device__ __forceinline__ rthip_mesh rt_load_mesh(const rthip_mesh *p) {
    rthip_mesh m;
    __builtin_memcpy(&m, (rt_uniform_mesh *)p, sizeof m);
    return m;
}
__device__ __forceinline__ rt_uniform_shape *rt_uniform_cast(const rthip_shape *p) { return (rt_uniform_shape *)p; }
__device__ __forceinline__ rt_uniform_light *rt_uniform_cast(const rthip_light *p) { return (rt_uniform_light *)p; }
#else
typedef const rthip_shape rt_uniform_shape;
typedef const rthip_light rt_uniform_light;
RT_HD rt_uniform_shape *rt_uniform_cast(const rthip_shape *p) { return p; }
RT_HD rt_uniform_light *rt_uniform_cast(const rthip_light *p) { return p; }
RT_HD rthip_mesh rt_load_mesh(const rthip_mesh *p) { return *p; }
#endif

// Traversal stack: `stride` lets the device interleave the lanes of a workgroup in LDS
// (entry k of lane t lives at base[k*stride], base already offset by t).  The BVH walk also keeps
// the entry parameter of every pending subtree (tbase, same indexing; may be null = no culling on
// pop).  A push beyond `cap` sets `overflow`: the caller re-runs the ray where the stack is full-depth.
struct KdStack {
    uint32_t *base;
    uint32_t stride;
    uint32_t cap;
    uint32_t n;
    float *tbase;
    bool overflow;
    RT_HD void clear() { n = 0; }
    RT_HD bool empty() const { return n == 0; }
    RT_HD void push(uint32_t v) {
        if (n < cap) {
            base[n * stride] = v;
            n++;
        } else {
            overflow = true;
        }
    }
    RT_HD uint32_t pop() {
        n--;
        return base[n * stride];
    }
    RT_HD void push_t(uint32_t v, float t) {
        if (n < cap) {
            base[n * stride] = v;
            if (tbase) tbase[n * stride] = t;
            n++;
        } else {
            overflow = true;
        }
    }
    RT_HD uint32_t pop_t(float &t) {
        n--;
        t = tbase ? tbase[n * stride] : 0.0f;
        return base[n * stride];
    }
};
RT_HD KdStack make_stack(uint32_t *base, uint32_t stride, uint32_t cap, float *tbase) {
    KdStack st;
    st.base = base;
    st.stride = stride;
    st.cap = cap;
    st.n = 0;
    st.tbase = tbase;
    st.overflow = false;
    return st;
}

struct KdCounters {
    uint32_t nodes, leaves, tris, pops, shapes, mesh_hits;
    uint32_t wave_steps, wave_tris;  // loop iterations the WAVE executed (one lane counts each): divergence diagnostic
    uint32_t fetch_cycles;           // cycles the wave waited for BVH node fetches (one lane counts)
    uint32_t shapes_full;            // shape-table entries that passed the bounding-ball test (world -> local transform + intersection / root boxes)
};
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t rt_first_active() {
    return (uint32_t)(__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1));
}
#else
RT_HD uint32_t rt_first_active() { return 1u; }
#endif

// ----------------------------------------------------------------------------- math (C1)
RT_HD V3 mk(double x, double y, double z) { return V3{x, y, z}; }
RT_HD V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }    // Common.cpp:119
RT_HD V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }    // Common.cpp:123
RT_HD V3 scale(V3 a, double t) { return mk(a.x * t, a.y * t, a.z * t); }    // Common.cpp:245
RT_HD V3 divs(V3 a, double t) { return mk(a.x / t, a.y / t, a.z / t); }     // Common.cpp:250
RT_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // Common.cpp:138
RT_HD double sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }      // Common.cpp:37
RT_HD double length(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // Common.cpp:42
RT_HD V3 normalized(V3 a) {                                                   // Common.cpp:47-58
    double l = length(a);
    if (l > 0) return mk(a.x / l, a.y / l, a.z / l);
    return mk(0, 0, 0);
}
// Common.cpp:60-63: *this - normal * (normal ^ *this) * 2.0f
RT_HD V3 reflect(V3 d, V3 n) { return sub(d, scale(scale(n, dot(n, d)), 2.0)); }
RT_HD double comp(V3 a, uint32_t ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }

// Common.h:22-25 Clamp(v, 0, 1) = std::max(0, std::min(1, v)) with the std:: tie/NaN behaviour
RT_HD double clamp01(double v) {
    double m = (v < 1.0) ? v : 1.0;
    return (0.0 < m) ? m : 0.0;
}
RT_HD Col mkc(double r, double g, double b) { return Col{clamp01(r), clamp01(g), clamp01(b)}; }  // Common.cpp:260
RT_HD Col c_add(Col a, Col b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }                 // Common.cpp:212
RT_HD Col c_mul(Col a, Col b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }                 // Common.cpp:221
RT_HD Col c_scale(Col a, double k) { return mkc(a.r * k, a.g * k, a.b * k); }                  // Common.cpp:230
RT_HD Col c_mix(Col a, Col b, double t) {                                                       // Common.cpp:239-243
    double _t = 1 - t;
    return mkc(_t * a.r + t * b.r, _t * a.g + t * b.g, _t * a.b + t * b.b);
}
RT_HD Col c_from(const double *p) { return Col{p[0], p[1], p[2]}; }
// Common.cpp:275-282 GetHexValue: truncating, R in the low byte
RT_HD uint32_t c_pack(Col c) {
    return 0xff000000u | ((uint32_t)(c.r * 255.0)) | (((uint32_t)(c.g * 255.0)) << 8) |
           (((uint32_t)(c.b * 255.0)) << 16);
}

// Common.cpp:169-180 Matrix4 * Vector4 restricted to rows 0..2; the accumulator starts at 0 and
// all four products are added in order (w = 1 for points, 0 for directions).
RT_HD V3 xform(const double *m, V3 p, double w) {
    V3 r;
    r.x = (((0.0 + m[0] * p.x) + m[1] * p.y) + m[2] * p.z) + m[3] * w;
    r.y = (((0.0 + m[4] * p.x) + m[5] * p.y) + m[6] * p.z) + m[7] * w;
    r.z = (((0.0 + m[8] * p.x) + m[9] * p.y) + m[10] * p.z) + m[11] * w;
    return r;
}

// ----------------------------------------------------------------------------- noise (N0-N4)
RT_HD double lerp(double a, double b, double t) { return a * (1 - t) + b * t; }              // Common.h:27-30
RT_HD double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }                 // Perlin.cpp:59-62

// Perlin.cpp:57-83 SampleAtFrequency + Perlin.cpp:49-55 GetSeed.  frequency F is a power of two, so the reference's divisions by
// it are exact scalings (x / 2^k == x * 2^-k bit for bit), and its cell corner floor(pos / F) * F is floor(pos) with the low bits
// cleared: (ipx, ipy, ipz) = floor(pos) as integers (once per sample), min = ip & ~(F - 1), max = min + F - the same integers the
// reference converts from doubles (pos >= 0 here: every caller wraps the point into the domain first, Noise.cpp:3-19).
// The 24 chained table lookups of the reference collapse to 14 distinct ones.  (The 256-byte table spans the 64 LDS banks exactly
// once, so byte lookups never conflict; the same table as 256 doubles - no conversion, no scaling: 16 instructions fewer per
// octave - spans them eight times and was 1.5 - 4 % SLOWER per frame: conflicts cost more than the instructions saved.)
struct PerlinCell {
    uint32_t ipx, ipy, ipz;
};
// The corner values are the table bytes / 256 (Perlin.cpp:49-55).  The interpolation is linear in them and a scaling by 2^-8 is
// exact and commutes with every rounding (no result here comes near the subnormals), so the eight scalings are done once, on
// the interpolated value: the corners go in as the integers 0..255 and the result is bit for bit the reference's.
RT_HD double perlin_corner(const uint8_t *tab, uint32_t idx) { return (double)tab[idx & 255u]; }
RT_HD double perlin_trilinear(V3 pos, double fx, double fy, double fz, double inv_fq, double a, double b, double c, double d, double e,
                              double f, double g, double h) {
    double u = fade((pos.x - fx) * inv_fq), v = fade((pos.y - fy) * inv_fq), w = fade((pos.z - fz) * inv_fq);
    return lerp(lerp(lerp(a, b, u), lerp(c, d, u), v), lerp(lerp(e, f, u), lerp(g, h, u), v), w) * (1.0 / 256.0);
}
RT_HD double perlin_at_freq(const uint8_t *tab, V3 pos, PerlinCell ip, uint32_t F, double inv_fq) {
    const uint32_t mask = ~(F - 1u);
    const uint32_t x0 = ip.ipx & mask, y0 = ip.ipy & mask, z0 = ip.ipz & mask;
    const uint32_t x1 = x0 + F, y1 = y0 + F, z1 = z0 + F;
    uint32_t ix0 = tab[x0 & 255u], ix1 = tab[x1 & 255u];
    uint32_t i00 = tab[(ix0 + y0) & 255u], i10 = tab[(ix1 + y0) & 255u];
    uint32_t i01 = tab[(ix0 + y1) & 255u], i11 = tab[(ix1 + y1) & 255u];
    double a = perlin_corner(tab, i00 + z0), b = perlin_corner(tab, i10 + z0);
    double c = perlin_corner(tab, i01 + z0), d = perlin_corner(tab, i11 + z0);
    double e = perlin_corner(tab, i00 + z1), f = perlin_corner(tab, i10 + z1);
    double g = perlin_corner(tab, i01 + z1), h = perlin_corner(tab, i11 + z1);
    return perlin_trilinear(pos, (double)x0, (double)y0, (double)z0, inv_fq, a, b, c, d, e, f, g, h);
}
// Perlin.cpp:25-47 SampleNormalized
RT_HD double perlin_normalized(const uint8_t *tab, uint32_t octaves, double persistance, V3 p) {
    double amplitude = 1.0, acc = 0.0, result = 0.0;
    V3 rpos = scale(p, 256.0);
    PerlinCell ip;
    ip.ipx = (uint32_t)floor(rpos.x); ip.ipy = (uint32_t)floor(rpos.y); ip.ipz = (uint32_t)floor(rpos.z);
    {   // octave 0, frequency 256 = the table's period: the eight corners are the same table entry (every corner coordinate is
        // a multiple of 256), so one chain of three lookups serves all of them; the interpolation still runs (a * (1 - t) + a * t
        // rounds, it need not give a back)
        const uint32_t x0 = ip.ipx & ~255u, y0 = ip.ipy & ~255u, z0 = ip.ipz & ~255u;
        const double a = perlin_corner(tab, tab[tab[0]]);
        result += amplitude * perlin_trilinear(rpos, (double)x0, (double)y0, (double)z0, 1.0 / 256.0, a, a, a, a, a, a, a, a);
        acc += amplitude;
        amplitude *= persistance;
    }
    uint32_t F = 128u;
    double inv_fq = 1.0 / 128.0;
    for (uint32_t i = 1; i < octaves; i++) {
        result += amplitude * perlin_at_freq(tab, rpos, ip, F, inv_fq);
        F >>= 1;
        inv_fq *= 2.0;
        acc += amplitude;
        amplitude *= persistance;
    }
    return result / acc;
}
// Worley.cpp:29-57 SampleNormalized, :59-84 helpers
RT_HD double worley_normalized(const double *pts, int div, double step, V3 p) {
    double maxDistance = step * (double)sqrtf(3.0f);
    int x = (int)(p.x * div), y = (int)(p.y * div), z = (int)(p.z * div);
    double distance = (double)FLT_MAX;
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++) {
                int cx = x + dx, cy = y + dy, cz = z + dz;
                int ox = cx < 0 ? -1 : (cx >= div ? 1 : 0);
                int oy = cy < 0 ? -1 : (cy >= div ? 1 : 0);
                int oz = cz < 0 ? -1 : (cz >= div ? 1 : 0);
                int wx = ((cx % div) + div) % div, wy = ((cy % div) + div) % div, wz = ((cz % div) + div) % div;
                const double *q = pts + 3 * (size_t)(wz * div * div + wy * div + wx);
                V3 fp = mk(q[0] + (double)ox, q[1] + (double)oy, q[2] + (double)oz);
                double d = sqlen(sub(fp, p));
                if (d < distance) distance = d;
            }
    return distance / maxDistance * 10.0;
}
// fmodf(x, d), which is exact by definition, without the library's general algorithm (~50 instructions on the device) when d is a
// power of two - the domains scenes use (1, 2, 4, 16): x * (1 / d) is then exact, k = trunc(x / d) an integer below 2^23, d * k exact,
// and x - d * k is a multiple of ulp(x) below d, hence representable; the sign of the result is x's, as fmodf's (a zero included).
RT_HD float fmodf_exact(float x, float d) {
    uint32_t db;
    __builtin_memcpy(&db, &d, 4);
    const uint32_t e = (db >> 23) & 255u;
    if ((db & 0x807fffffu) == 0u && e >= 64u && e <= 190u) {  // d = 2^(e - 127)
        const uint32_t ib = (254u - e) << 23;
        float inv;
        __builtin_memcpy(&inv, &ib, 4);
        const float t = x * inv;
        if (fabsf(t) < 8388608.0f) return copysignf(fabsf(x - d * truncf(t)), x);  // (false for inf / NaN too)
    }
    return fmodf(x, d);
}
// Noise.cpp:3-19: fmodf rounds point and domain to float; its float result is divided by the double domain
RT_HD double noise_wrap(double x, double domain) {
    x = (double)fmodf_exact((float)x, (float)domain) / domain;
    return x < 0 ? 1.0 + x : x;
}
// LEAN (here and in material_eval / shade_stage / debug_shade): the instantiation for scenes whose shapes' materials are uniform
// or ONE interpolation of two uniform ones by a checker board or Perlin noise (rthip.hip: lean_scene decides at upload; the demo
// scenes with a ground plane and plain objects).  Worley's 27-cell loop, Marble's sin() (~900 instructions of argument reduction)
// and the material stack are then not part of the kernel, whose register allocation they dominate although such a scene never runs
// them: k_trace_easy spills 7 registers instead of 61, k_shade none instead of 26 (profiles/r03_lean_kernels.txt).
template <bool LEAN = false>
RT_HD double noise_sample(const SceneView &sv, const uint8_t *tabs, int32_t ni, V3 p) {
    const rthip_noise &n = sv.noises[ni];
    V3 q = mk(noise_wrap(p.x, n.domain), noise_wrap(p.y, n.domain), noise_wrap(p.z, n.domain));
    if (n.type == RTHIP_NOISE_CHECKER) {  // CheckerBoard.cpp:3-16
        int t0 = q.x <= 0.5, t2 = q.z <= 0.5;
        return (double)(t2 ^ t0);
    }
    if (LEAN || n.type == RTHIP_NOISE_PERLIN) return perlin_normalized(tabs + 256 * (size_t)n.table, n.octaves, n.persistance, q);
    if (n.type == RTHIP_NOISE_WORLEY) return worley_normalized(sv.worley_points + 3 * (size_t)n.points_base, n.divisions, n.step, q);
    // Marble.cpp:11-15
    double f = perlin_normalized(tabs + 256 * (size_t)n.table, n.octaves, n.persistance, q) * n.turbolence;
    return fabs(sin(f * n.frequency * (double)3.141592f));
}

// ----------------------------------------------------------------------------- materials (M1, M2)
// Evaluates the postfix program ending at op `root`: Lerp<Color> = a*(1-t) (+) b*t with clamping
// Color ops, Lerp<real> unclamped (Material.cpp:21-29).  Noise is sampled once per INTERP node
// (the reference re-samples it per query; Sample is a pure function of the point).
template <bool LEAN = false>
RT_HD void material_eval(const SceneView &sv, const uint8_t *tabs, int32_t root, V3 p, Col &color, double &absorptance) {
    const rthip_matop &r = sv.materials[root];
    if (r.op == RTHIP_MAT_UNIFORM) {
        color = c_from(r.color);
        absorptance = r.absorptance;
        return;
    }
    if (LEAN || root - r.first_op == 2) {  // INTERP of two UNIFORMs: no stack needed
        const rthip_matop &a = sv.materials[root - 2], &b = sv.materials[root - 1];
        double t = noise_sample<LEAN>(sv, tabs, r.noise, p);
        color = c_add(c_scale(c_from(a.color), 1 - t), c_scale(c_from(b.color), t));
        absorptance = lerp(a.absorptance, b.absorptance, t);
        return;
    }
    Col cs[RTHIP_MAX_MAT_STACK];
    double as[RTHIP_MAX_MAT_STACK];
    int sp = 0;
    for (int32_t i = r.first_op; i <= root; i++) {
        const rthip_matop &m = sv.materials[i];
        if (m.op == RTHIP_MAT_UNIFORM) {
            if (sp < RTHIP_MAX_MAT_STACK) {
                cs[sp] = c_from(m.color);
                as[sp] = m.absorptance;
                sp++;
            }
        } else if (sp >= 2) {
            double t = noise_sample(sv, tabs, m.noise, p);
            Col a = cs[sp - 2], b = cs[sp - 1];
            double aa = as[sp - 2], ab = as[sp - 1];
            sp--;
            cs[sp - 1] = c_add(c_scale(a, 1 - t), c_scale(b, t));
            as[sp - 1] = lerp(aa, ab, t);
        }
    }
    color = cs[0];
    absorptance = as[0];
}

// ----------------------------------------------------------------------------- background (B1, B2)
// Scene.cpp:111-151 SkyBox::GetColor; Scene.h:43-49 ColorBackground
RT_HD Col background_color(const SceneView &sv, const uint8_t *tabs, V3 direction) {
    const rthip_background &bg = sv.bg;
    if (bg.type == RTHIP_BG_NONE) return Col{0, 0, 0};  // Raytracer.cpp:296-299
    if (bg.type == RTHIP_BG_COLOR) return c_from(bg.color0);
    const float thresold = 0.98f;
    const float border = (1.0f - thresold) / 2.0f;
    const float thresoldPlusBorder = thresold + border;
    V3 q = divs(add(direction, mk(1, 1, 1)), 2.0);
    V3 w = mk(noise_wrap(q.x, 1.0), noise_wrap(q.y, 1.0), noise_wrap(q.z, 1.0));
    double sample = perlin_normalized(tabs + 256 * (size_t)bg.table, 8, 0.5, w);
    Col color = c_mix(c_from(bg.color0), c_from(bg.color1), sample);
    if (bg.sun_light >= 0 && sv.lights[bg.sun_light].type == RTHIP_LIGHT_DIRECTIONAL) {
        const rthip_light &sun = sv.lights[bg.sun_light];
        float f = fmaxf(0.0f, (float)dot(direction, mk(sun.direction[0], sun.direction[1], sun.direction[2])));
        if (f < thresold) return color;
        if (f < thresoldPlusBorder) {
            // powf(x, 2.0f) of the reference: glibc's powf is correctly rounded here and equals the
            // float product x*x on all but 117 of the 167 772 possible inputs, where it differs by one float ulp
            // (exhaustive check: tests/test_oracle.py::test_powf_square_vs_float_product)
            float x = (f - thresold) / border;
            f = x * x;
            return c_mix(color, c_from(sun.color), (double)f);
        }
        return c_from(sun.color);
    }
    return color;
}

// ----------------------------------------------------------------------------- primitives (P1, P2, T2)
struct LocalHit {
    bool hit;
    V3 point;
    V3 normal;
    int32_t tri;
};

// Scene.cpp:294-330 Sphere::Intersect (unit sphere at the local origin)
RT_HD bool sphere_intersect(V3 o, V3 d, V3 &point) {
    double projection = dot(sub(mk(0, 0, 0), o), d);
    double squaredDistance = dot(o, o) - projection * projection;
    if (squaredDistance > 1.0) return false;
    double offset = sqrt(1.0 - squaredDistance);
    double t1 = projection - offset, t2 = projection + offset;
    if (t1 < 0 && t2 < 0) return false;
    point = add(o, scale(d, (t1 >= 0.0 ? t1 : t2)));
    return true;
}
// Scene.cpp:332-362 Plane::Intersect (one-sided, through the local origin, un-normalised Normal)
RT_HD bool plane_intersect(V3 n, V3 o, V3 d, V3 &point) {
    double distance = dot(o, n);
    if (distance < 0) return false;
    double cosine = dot(n, d);
    if (cosine >= 0) return false;
    point = add(o, scale(d, distance / -cosine));
    return true;
}

// Only what decides the argmin stays live during a traversal; the hit point and the barycentrics of the
// winner are recomputed afterwards by tri_point (same arithmetic, so bit-identical), which keeps the
// traversal loops small in registers.
struct TriBest {
    double d;      // |O - P|^2 exactly as Scene.cpp:477
    uint32_t tri;  // mesh-relative triangle index
    uint32_t ref;  // position in the pre-order leaf-reference array (kd walk) / the triangle index (BVH walk)
    bool hit;
    bool tie;      // another triangle was accepted at exactly the distance `d` (see mesh_closest_reference_order)
    double ext;    // largest bounding-box edge of the triangles accepted at exactly `d` (bounds where their kd leaves can be)
};
// largest edge of the triangle's bounding box, from its cached edges (vertices v0, v0 + e0, v0 + e1)
RT_HD double tri_extent(const rthip_tri &t) {
    double m = 0.0;
    for (int a = 0; a < 3; a++) {
        const double lo = fmin(0.0, fmin(t.e0[a], t.e1[a])), hi = fmax(0.0, fmax(t.e0[a], t.e1[a]));
        m = fmax(m, hi - lo);
    }
    return m;
}

// Scene.cpp:421-467 IntersectTriangle + :495-507 Baricentric: plane projection, back-face /
// behind-plane cull, inclusive barycentric test.  Updates `best` with the strict '<' of Scene.cpp:479.
// EXACT TIES (two triangles accepted at bit-identical d, e.g. a ray through the shared diagonal of an
// axis-aligned quad): the reference keeps the one its walk tests first, and that order - leaves in
// pre-order, Left before Right (Scene.cpp:486-490), a straddling triangle at its first VISITED leaf - is
// a property of the ray and the tree, not of the triangle index or of the reference position alone (a
// pruned or reordered walk may never see the earlier instance).  So the fast walks only DETECT a tie at
// the winning distance (`best.tie`; any provisional winner) and closest-hit queries that end with one are
// repeated by mesh_closest_reference_order, which is the reference's walk itself.
RT_HD void tri_test(const rthip_tri &t, V3 o, V3 d, uint32_t tri, uint32_t ref, TriBest &best) {
    V3 v0 = mk(t.v0[0], t.v0[1], t.v0[2]);
    V3 nf = mk(t.face_normal[0], t.face_normal[1], t.face_normal[2]);
    V3 l = sub(o, v0);
    double distance = dot(l, nf);
    if (distance < 0) return;
    double cosine = dot(d, nf);
    if (cosine >= 0) return;
    V3 projection = add(o, scale(d, distance / -cosine));
    V3 v2 = sub(projection, v0);
    double d20 = dot(v2, mk(t.e0[0], t.e0[1], t.e0[2]));
    double d21 = dot(v2, mk(t.e1[0], t.e1[1], t.e1[2]));
    double v = (t.d11 * d20 - t.d01 * d21) * t.inv_den;
    double w = (t.d00 * d21 - t.d01 * d20) * t.inv_den;
    double u = 1.0 - v - w;
    if (u >= 0 && v >= 0 && w >= 0) {
        double dd = sqlen(sub(o, projection));
        if (best.hit && dd == best.d && tri != best.tri) {
            best.tie = true;
            best.ext = fmax(best.ext, tri_extent(t));
        }
        if (dd < best.d || (best.hit && dd == best.d && ref < best.ref)) {
            if (dd < best.d) {
                best.tie = false;
                best.ext = tri_extent(t);
            }
            best.d = dd;
            best.tri = tri;
            best.ref = ref;
            best.hit = true;
        }
    }
}

// Same test for the BVH walk; the mesh-relative triangle index is only fetched for accepted hits.  On an exact tie
// the lower index is kept provisionally and `best.tie` is raised (see tri_test).
RT_HD void tri_test_bvh(const rthip_tri &t, V3 o, V3 d, const uint32_t *index, uint32_t pos, TriBest &best) {
    V3 v0 = mk(t.v0[0], t.v0[1], t.v0[2]);
    V3 nf = mk(t.face_normal[0], t.face_normal[1], t.face_normal[2]);
    V3 l = sub(o, v0);
    double distance = dot(l, nf);
    if (distance < 0) return;
    double cosine = dot(d, nf);
    if (cosine >= 0) return;
    V3 projection = add(o, scale(d, distance / -cosine));
    V3 v2 = sub(projection, v0);
    double d20 = dot(v2, mk(t.e0[0], t.e0[1], t.e0[2]));
    double d21 = dot(v2, mk(t.e1[0], t.e1[1], t.e1[2]));
    double v = (t.d11 * d20 - t.d01 * d21) * t.inv_den;
    double w = (t.d00 * d21 - t.d01 * d20) * t.inv_den;
    double u = 1.0 - v - w;
    if (u >= 0 && v >= 0 && w >= 0) {
        double dd = sqlen(sub(o, projection));
        if (dd <= best.d) {
            uint32_t ti = index[pos];
            if (best.hit && dd == best.d && ti != best.tri) {
                best.tie = true;
                best.ext = fmax(best.ext, tri_extent(t));
            }
            if (dd < best.d || (best.hit && ti < best.tri)) {
                if (dd < best.d) {
                    best.tie = false;
                    best.ext = tri_extent(t);
                }
                best.d = dd;
                best.tri = ti;
                best.ref = ti;
                best.hit = true;
            }
        }
    }
}

// Hit point and barycentrics of a triangle already known to be hit (Scene.cpp:446-450, 495-507)
RT_HD void tri_point(const rthip_tri &t, V3 o, V3 d, V3 &projection, double &u, double &v, double &w) {
    V3 v0 = mk(t.v0[0], t.v0[1], t.v0[2]);
    V3 nf = mk(t.face_normal[0], t.face_normal[1], t.face_normal[2]);
    double distance = dot(sub(o, v0), nf);
    double cosine = dot(d, nf);
    projection = add(o, scale(d, distance / -cosine));
    V3 v2 = sub(projection, v0);
    double d20 = dot(v2, mk(t.e0[0], t.e0[1], t.e0[2]));
    double d21 = dot(v2, mk(t.e1[0], t.e1[1], t.e1[2]));
    v = (t.d11 * d20 - t.d01 * d21) * t.inv_den;
    w = (t.d00 * d21 - t.d01 * d20) * t.inv_den;
    u = 1.0 - v - w;
}

// Common.cpp:345-363 BoundingBox::Intersect on precomputed plane parameters
// t[0..5] = (Min.x-O.x)/D.x, (Max.x-O.x)/D.x, (Min.y..)/D.y, (Max.y..)/D.y, (Min.z..)/D.z, (Max.z..)/D.z
RT_HD bool slab_hit(const double *t, double &t7) {
    t7 = fmax(fmax(fmin(t[0], t[1]), fmin(t[2], t[3])), fmin(t[4], t[5]));
    double t8 = fmin(fmin(fmax(t[0], t[1]), fmax(t[2], t[3])), fmax(t[4], t[5]));
    double t9 = (t8 < 0 || t7 > t8) ? -1.0 : t7;
    return t9 != -1.0;
}
RT_HD void slab_params(const double *b, V3 o, V3 d, double *t) {
    t[0] = (b[0] - o.x) / d.x;
    t[1] = (b[3] - o.x) / d.x;
    t[2] = (b[1] - o.y) / d.y;
    t[3] = (b[4] - o.y) / d.y;
    t[4] = (b[2] - o.z) / d.z;
    t[5] = (b[5] - o.z) / d.z;
}
// A subtree whose box the ray enters at parameter t7 (|D| = 1) cannot hold a hit point closer than
// t7; it is skipped once the best squared distance is safely below t7^2 (1e-9 relative slack covers
// the rounding of both sides).  Skipping never changes the argmin the reference computes.
RT_HD bool kd_prunable(double t7, const TriBest &best) {
    return best.hit && t7 > 0.0 && t7 * t7 * (1.0 - 1e-9) > best.d;
}

// T4: Mesh::Intersect / IntersectInternal (Scene.cpp:372-378, 469-493).  The reference visits every
// node whose box (and all ancestor boxes) pass BoundingBox::Intersect and keeps the strictly
// closest triangle hit.  This traversal applies the same box test with the same arithmetic to every
// node it visits, but descends near-child first and skips subtrees that cannot beat the current
// best, so it reaches the same argmin with a fraction of the node/triangle tests.  A child's box
// differs from its parent's in one plane (BoundingBox::Split, Common.cpp:313-338), so descending
// costs one division; the six plane parameters are recomputed from the stored bounds on a pop.
RT_HD void mesh_closest(const SceneView &sv, const rthip_mesh &m, V3 o, V3 d, KdStack &st, TriBest &best,
                                 KdCounters *cnt) {
    best.hit = false;
    best.tie = false;
    best.ext = 0.0;
    best.d = DBL_MAX;
    best.tri = 0xffffffffu;
    best.ref = 0xffffffffu;
    if (m.node_count == 0) return;
    const rthip_kdnode *nodes = sv.kd_nodes + m.node_base;
    const double *bounds = sv.kd_bounds + 6 * (size_t)m.node_base;
    const uint32_t *refs = sv.kd_refs + m.ref_base;
    const rthip_tri *tris = sv.tris + m.tri_base;
    double t[6], t7;
    slab_params(bounds, o, d, t);
    if (cnt) cnt->pops++;
    if (!slab_hit(t, t7)) return;
    st.clear();
    uint32_t cur = 0;
    for (;;) {
        rthip_kdnode nd = nodes[cur];
        if (cnt) cnt->nodes++;
        bool descend = false;
        if (nd.info & RTHIP_KD_LEAF) {
            uint32_t count = nd.info >> RTHIP_KD_COUNT_SHIFT;
            if (cnt) {
                cnt->leaves++;
                cnt->tris += count;
            }
            for (uint32_t i = 0; i < count; i++) {
                uint32_t ti = refs[nd.link + i];
                tri_test(tris[ti], o, d, ti, nd.link + i, best);
            }
        } else {
            uint32_t ax = nd.info & RTHIP_KD_AXIS_MASK;
            double ts = (nd.split - comp(o, ax)) / comp(d, ax);
            // left child: Max[ax] = split; right child: Min[ax] = split
            // every index below is a compile-time constant after unrolling: runtime-indexed
            // register arrays would be demoted to scratch memory on gfx950
            double tl[6], tr[6], t7l = 0, t7r = 0;
            const int ihi = (int)(2 * ax + 1), ilo = (int)(2 * ax);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                tl[k] = (k == ihi) ? ts : t[k];
                tr[k] = (k == ilo) ? ts : t[k];
            }
            bool hl = (nd.info & RTHIP_KD_HAS_LEFT) && slab_hit(tl, t7l) && !kd_prunable(t7l, best);
            bool hr = (nd.info & RTHIP_KD_HAS_RIGHT) && slab_hit(tr, t7r) && !kd_prunable(t7r, best);
            if (hl && hr) {
                bool left_first = t7l <= t7r;
                st.push(left_first ? nd.link : cur + 1);
                cur = left_first ? cur + 1 : nd.link;
#pragma unroll
                for (int k = 0; k < 6; k++) t[k] = left_first ? tl[k] : tr[k];
                descend = true;
            } else if (hl) {
                cur = cur + 1;
#pragma unroll
                for (int k = 0; k < 6; k++) t[k] = tl[k];
                descend = true;
            } else if (hr) {
                cur = nd.link;
#pragma unroll
                for (int k = 0; k < 6; k++) t[k] = tr[k];
                descend = true;
            }
        }
        if (descend) continue;
        // pop the next pending subtree that can still matter
        bool found = false;
        while (!st.empty()) {
            cur = st.pop();
            if (cnt) cnt->pops++;
            slab_params(bounds + 6 * (size_t)cur, o, d, t);
            if (slab_hit(t, t7) && !kd_prunable(t7, best)) {
                found = true;
                break;
            }
        }
        if (!found) break;
    }
}

// Mesh::IntersectInternal as written (Scene.cpp:469-493): every node whose box passes BoundingBox::Intersect, Left
// subtree before Right, the triangles of a leaf in reference order, strict '<' - no reordering, no distance pruning.
// Only queries whose fast walk ended in an exact tie come here, so that the triangle returned is the one the
// reference returns (its id, and with vertex normals its normal).
// What IS left out: subtrees that cannot reference a tied triangle.  The fast walk has tested every triangle
// accepted at the winning distance (its culling never skips a box that could hold a closer-or-EQUAL hit), all of
// them contain the hit point `at` (to rounding), and `ext` is their largest bounding-box edge; the kd build only
// hands a triangle to a child whose box its bounding box overlaps (Scene.cpp:59-70), so every leaf that
// references one of them lies within `ext` of `at` on every axis.  Skipping the others leaves the order of the
// leaves that matter untouched - and keeps a ray that runs along a grid line of a 300 K-triangle height field
// (a tie at every crossing) from testing every triangle on its way: 40 ms for one lane.
// `tied` / `n_tied`: the triangles accepted at exactly the winning distance `dist`, when the caller knows them
// (tied_triangles below): the leaves are then only SCANNED for the first reference to one of them (4 bytes per
// reference instead of a 128-byte record and an fp64 test - the reference's tree keeps leaves of thousands of
// triangles where its 50 % rule stops a split, and one lane testing 10 000 of them in turn takes 15 ms).
// n_tied = 0: every triangle of every visited leaf is tested, as the reference does.
RT_HD void mesh_closest_reference_order(const SceneView &sv, const rthip_mesh &m, V3 o, V3 d, KdStack &st, TriBest &best, V3 at,
                                        double ext, double dist, const uint32_t *tied, uint32_t n_tied, KdCounters *cnt) {
    best.hit = false;
    best.tie = false;
    best.ext = 0.0;
    best.d = DBL_MAX;
    best.tri = 0xffffffffu;
    best.ref = 0xffffffffu;
    if (m.node_count == 0) return;
    const rthip_kdnode *nodes = sv.kd_nodes + m.node_base;
    const double *bounds = sv.kd_bounds + 6 * (size_t)m.node_base;
    const uint32_t *refs = sv.kd_refs + m.ref_base;
    const rthip_tri *tris = sv.tris + m.tri_base;
    const double reach = ext * (1.0 + 1e-9) + 1e-9 * (fabs(at.x) + fabs(at.y) + fabs(at.z) + ext) + 1e-300;
    const double qlo[3] = {at.x - reach, at.y - reach, at.z - reach}, qhi[3] = {at.x + reach, at.y + reach, at.z + reach};
    auto near_hit_point = [&](const double *b) {  // b = Min xyz, Max xyz
        return b[0] <= qhi[0] && b[3] >= qlo[0] && b[1] <= qhi[1] && b[4] >= qlo[1] && b[2] <= qhi[2] && b[5] >= qlo[2];
    };
    double t[6], t7;
    slab_params(bounds, o, d, t);
    if (!slab_hit(t, t7)) return;
    st.clear();
    uint32_t cur = 0;
    for (;;) {
        const rthip_kdnode nd = nodes[cur];
        if (cnt) cnt->nodes++;
        bool descend = false;
        if (nd.info & RTHIP_KD_LEAF) {
            const uint32_t count = nd.info >> RTHIP_KD_COUNT_SHIFT;
            if (cnt) {
                cnt->leaves++;
                if (n_tied == 0) cnt->tris += count;
            }
            if (n_tied) {  // the first reference to a tied triangle is the first triangle to reach the winning distance
                for (uint32_t i0 = 0; i0 < count; i0 += 8) {  // eight loads in flight: the scan is one lane's latency chain
                    uint32_t r8[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) r8[j] = refs[nd.link + (i0 + j < count ? i0 + j : count - 1u)];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) {
                        bool member = false;
                        for (uint32_t k = 0; k < n_tied; k++) member = member || tied[k] == r8[j];
                        if (member && i0 + j < count) {
                            best.hit = true;
                            best.d = dist;
                            best.tri = r8[j];
                            best.ref = nd.link + i0 + j;
                            return;
                        }
                    }
                }
            } else {
                for (uint32_t i = 0; i < count; i++) {
                    const uint32_t ti = refs[nd.link + i];
                    TriBest one;
                    one.hit = false;
                    one.tie = false;
                    one.ext = 0.0;
                    one.d = DBL_MAX;
                    one.tri = one.ref = 0xffffffffu;
                    tri_test(tris[ti], o, d, ti, nd.link + i, one);
                    if (one.hit && one.d < best.d) best = one;  // Scene.cpp:479
                }
            }
        } else {
            bool hl = false, hr = false;
            if (nd.info & RTHIP_KD_HAS_LEFT) {
                const double *b = bounds + 6 * (size_t)(cur + 1);
                slab_params(b, o, d, t);
                hl = slab_hit(t, t7) && near_hit_point(b);
            }
            if (nd.info & RTHIP_KD_HAS_RIGHT) {
                const double *b = bounds + 6 * (size_t)nd.link;
                slab_params(b, o, d, t);
                hr = slab_hit(t, t7) && near_hit_point(b);
            }
            if (hl) {
                if (hr) st.push(nd.link);
                cur = cur + 1;
                descend = true;
            } else if (hr) {
                cur = nd.link;
                descend = true;
            }
        }
        if (descend) continue;
        if (st.empty()) break;
        cur = st.pop();
    }
}

// T4 through the device-oriented BVH: same result as mesh_closest (argmin over the mesh's triangles
// of the reference's d = |O-P|^2 with the reference's triangle test), different culling structure.
// Box tests are fp32 and CONSERVATIVE — child boxes are stored rounded outward, and every slab interval
// is widened by `pad`*|1/d|, so a box that could contain a closer-or-equal hit is never skipped; they
// take no part in the result's arithmetic.
// Error budget of one plane parameter a = fma(plane, inv, -(o*inv)) against the exact (plane - O)/D:
// |a - exact| <= 2^-23 * (1.5|plane| + 2.5|O|) * |1/D|  (roundings of O, D, 1/D, o*inv and the fma),
// i.e. < 0.32 * 2^-20 * (abs_max + |O|) * |inv|; pad = 2^-20 * (abs_max + max|O|) leaves 3x margin.
struct BvhRay {
    float ix, iy, iz;     // clamped reciprocal direction
    float cx, cy, cz;     // o * inv
    float sx, sy, sz;     // slab widening per axis
};
RT_HD float bvh_inv(double d) {
    float f = (float)d;
    float r = 1.0f / f;
    const float big = 1e18f;  // keeps plane*inv - o*inv and pad*|inv| finite (no inf - inf) up to |coordinate| ~ 1e20;
                              // the slab widening pad*big still covers every parameter below 1e12 * scene magnitude
    if (!(fabsf(r) < big)) r = (f < 0.0f || (f == 0.0f && d < 0.0)) ? -big : big;
    return r;
}
RT_HD float rt_fmaf(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return (float)((double)a * (double)b + (double)c);  // test build: double-rounded, within the pad's margin
#endif
}
RT_HD void bvh_slab(const float *lo, const float *hi, const BvhRay &r, float tmax, float &tn, bool &hit) {
    float ax = rt_fmaf(lo[0], r.ix, -r.cx), bx = rt_fmaf(hi[0], r.ix, -r.cx);
    float ay = rt_fmaf(lo[1], r.iy, -r.cy), by = rt_fmaf(hi[1], r.iy, -r.cy);
    float az = rt_fmaf(lo[2], r.iz, -r.cz), bz = rt_fmaf(hi[2], r.iz, -r.cz);
    float nx = fminf(ax, bx) - r.sx, fx = fmaxf(ax, bx) + r.sx;
    float ny = fminf(ay, by) - r.sy, fy = fmaxf(ay, by) + r.sy;
    float nz = fminf(az, bz) - r.sz, fz = fmaxf(az, bz) + r.sz;
    tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    float tf = fminf(fminf(fx, fy), fminf(fz, tmax));
    hit = tn <= tf;
}
RT_HD BvhRay bvh_ray(V3 o, V3 d, float abs_max) {
    BvhRay r;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    r.ix = bvh_inv(d.x); r.iy = bvh_inv(d.y); r.iz = bvh_inv(d.z);
    r.cx = ox * r.ix; r.cy = oy * r.iy; r.cz = oz * r.iz;
    const float pad = 9.6e-7f * (fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + abs_max) + 1e-30f;  // 2^-20 * magnitude
    r.sx = pad * fabsf(r.ix); r.sy = pad * fabsf(r.iy); r.sz = pad * fabsf(r.iz);
    return r;
}
// upper bound (fp32) of the ray parameter of a hit with squared distance d: sqrt rounded up, two
// float ulps of margin
RT_HD float bvh_tmax_of(double d) {
    float t = (float)sqrt(d);
    return t + t * 2.4e-7f + 1e-37f;
}
// One ray per lane over the 4-wide BVH ("while-while": descend through interior nodes until a leaf is
// held, then test its triangles).  This is the portable form: the CPU test build runs it, and on the
// device it serves the probes and the redo kernel; the frame kernel walks the same nodes with four
// lanes per ray (rthip.hip: mesh_closest_quad).  Nearest child first; the other hit children go to
// the stack far-first with their entry parameters and are culled against the current best when popped.
// t_stop: in an any-hit query (Raytracer.cpp:305-310 only consumes .Hit) the walk stops looking
// beyond t_stop once a hit is known; hits up to t_stop are still resolved exactly because the
// self-intersection test of Scene.cpp:200 applies to the CLOSEST hit (see cast_ray_id).
RT_HD void bvh_cswap(float &ta, uint32_t &ca, float &tb, uint32_t &cb) {
    if (tb < ta) {
        float t = ta; ta = tb; tb = t;
        uint32_t c = ca; ca = cb; cb = c;
    }
}
template <bool COUNT>
RT_HD void mesh_closest_bvh(const SceneView &sv, const rthip_mesh &m, V3 o, V3 d, KdStack &st, TriBest &best,
                            float t_stop, KdCounters *cnt) {
    best.hit = false;
    best.tie = false;
    best.ext = 0.0;
    best.d = DBL_MAX;
    best.tri = 0xffffffffu;
    best.ref = 0xffffffffu;
    if (m.bvh_node_count == 0) return;
    const rthip_bvhnode *nodes = sv.bvh_nodes + m.bvh_node_base;
    const rthip_tri *tris = sv.bvh_tris + m.bvh_tri_base;
    const uint32_t *index = sv.bvh_tri_index + m.bvh_tri_base;
    const BvhRay r = bvh_ray(o, d, m.abs_max);
    float tmax = 3.0e38f;
    st.clear();
    const float kMiss = 3.4e38f;
    uint32_t cur = 0;
    for (;;) {
        while (!(cur & RTHIP_BVH_LEAF)) {  // RTHIP_BVH_ABSENT (= done) has the leaf bit set
            const rthip_bvhnode &nd = nodes[cur];
            if (COUNT) cnt->nodes++;
            float t0, t1, t2, t3;
            bool h;
            uint32_t c0 = nd.c[0].child, c1 = nd.c[1].child, c2 = nd.c[2].child, c3 = nd.c[3].child;
            bvh_slab(nd.c[0].lo, nd.c[0].hi, r, tmax, t0, h); if (!h || c0 == RTHIP_BVH_ABSENT) t0 = kMiss;
            bvh_slab(nd.c[1].lo, nd.c[1].hi, r, tmax, t1, h); if (!h || c1 == RTHIP_BVH_ABSENT) t1 = kMiss;
            bvh_slab(nd.c[2].lo, nd.c[2].hi, r, tmax, t2, h); if (!h || c2 == RTHIP_BVH_ABSENT) t2 = kMiss;
            bvh_slab(nd.c[3].lo, nd.c[3].hi, r, tmax, t3, h); if (!h || c3 == RTHIP_BVH_ABSENT) t3 = kMiss;
            bvh_cswap(t0, c0, t1, c1);  // sorting network, ascending entry parameter
            bvh_cswap(t2, c2, t3, c3);
            bvh_cswap(t0, c0, t2, c2);
            bvh_cswap(t1, c1, t3, c3);
            bvh_cswap(t1, c1, t2, c2);
            if (t3 < kMiss) st.push_t(c3, t3);
            if (t2 < kMiss) st.push_t(c2, t2);
            if (t1 < kMiss) st.push_t(c1, t1);
            if (t0 < kMiss) {
                cur = c0;
            } else {
                cur = RTHIP_BVH_ABSENT;
                while (!st.empty()) {
                    float t;
                    uint32_t c = st.pop_t(t);
                    if (COUNT) cnt->pops++;
                    if (t <= tmax) { cur = c; break; }
                }
            }
        }
        if (cur == RTHIP_BVH_ABSENT) break;
        {   // leaf
            const uint32_t first = cur & RTHIP_BVH_FIRST_MASK, count = ((cur >> RTHIP_BVH_COUNT_SHIFT) & 7u) + 1u;
            if (COUNT) {
                cnt->leaves++;
                cnt->tris += count;
            }
            const bool had = best.hit;
            const double before = best.d;
            for (uint32_t i = 0; i < count; i++) tri_test_bvh(tris[first + i], o, d, index, first + i, best);
            if (best.hit && (!had || best.d < before)) tmax = fminf(bvh_tmax_of(best.d), t_stop);
        }
        cur = RTHIP_BVH_ABSENT;
        while (!st.empty()) {
            float t;
            uint32_t c = st.pop_t(t);
            if (COUNT) cnt->pops++;
            if (t <= tmax) { cur = c; break; }
        }
    }
}

// The triangles of the mesh accepted at exactly squared distance `dist` (the winning one: nothing is closer), through
// the BVH with the search bounded at that distance from the start: a handful of boxes around the hit point.
// Returns their number, kMaxTied + 1 when there are more than `tied` holds.
static const uint32_t kMaxTied = 8;
RT_HD uint32_t tied_triangles(const SceneView &sv, const rthip_mesh &m, V3 o, V3 d, KdStack &st, double dist, uint32_t *tied) {
    const rthip_bvhnode *nodes = sv.bvh_nodes + m.bvh_node_base;
    const rthip_tri *tris = sv.bvh_tris + m.bvh_tri_base;
    const uint32_t *index = sv.bvh_tri_index + m.bvh_tri_base;
    const BvhRay r = bvh_ray(o, d, m.abs_max);
    const float tmax = bvh_tmax_of(dist);
    uint32_t n = 0;
    st.clear();
    uint32_t cur = 0;
    for (;;) {
        if (!(cur & RTHIP_BVH_LEAF)) {
            const rthip_bvhnode &nd = nodes[cur];
            for (int c = 0; c < 4; c++) {
                float tn;
                bool h;
                bvh_slab(nd.c[c].lo, nd.c[c].hi, r, tmax, tn, h);
                if (h && nd.c[c].child != RTHIP_BVH_ABSENT) st.push_t(nd.c[c].child, tn);
            }
        } else {
            const uint32_t first = cur & RTHIP_BVH_FIRST_MASK, count = ((cur >> RTHIP_BVH_COUNT_SHIFT) & 7u) + 1u;
            for (uint32_t i = 0; i < count; i++) {
                TriBest one;
                one.hit = false;
                one.tie = false;
                one.ext = 0.0;
                one.d = DBL_MAX;
                one.tri = one.ref = 0xffffffffu;
                tri_test(tris[first + i], o, d, 0u, 0u, one);
                if (one.hit && one.d == dist) {
                    if (n == kMaxTied) return kMaxTied + 1u;
                    tied[n++] = index[first + i];
                }
            }
        }
        if (st.empty()) break;
        float t;
        cur = st.pop_t(t);
    }
    return n;
}

// true iff `v` is exactly one of the mesh's kd box-plane coordinates on `axis` (binary search)
RT_HD bool kd_on_plane(const SceneView &sv, const rthip_mesh &m, uint32_t axis, double v) {
    const double *p = sv.kd_planes + m.plane_base[axis];
    uint32_t lo = 0, hi = m.plane_count[axis];
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        double x = p[mid];
        if (x == v) return true;
        if (x < v) lo = mid + 1; else hi = mid;
    }
    return false;
}
// The reference's slab test divides by the direction (Common.cpp:349-354): a component that is exactly
// 0 with the origin exactly on a box plane gives 0/0 = NaN, which fmin/fmax drop, and the walk then
// accepts/rejects boxes in a way only the reference's own tree reproduces.
RT_HD bool kd_degenerate_ray(const SceneView &sv, const rthip_mesh &m, V3 o, V3 d) {
    bool r = false;
    if (d.x == 0.0) r = r || kd_on_plane(sv, m, 0, o.x);
    if (d.y == 0.0) r = r || kd_on_plane(sv, m, 1, o.y);
    if (d.z == 0.0) r = r || kd_on_plane(sv, m, 2, o.z);
    return r;
}

// ----------------------------------------------------------------------------- scene query (S1)
struct Hit {
    int32_t shape;  // pre-order ordinal into sv.shapes, -1 = miss
    int32_t tri;    // mesh-relative triangle index or -1
    V3 point, local, normal;
};

// Shape::Intersect of one shape in its local frame: hit flag, local hit point, winning triangle.
// One ray per lane (probes, the redo kernel, accel = 1, the CPU test build): meshes go through the 4-wide BVH
// unless the ray is one only the reference's kd-tree reproduces (kd_degenerate_ray) or accel = 1 asks for the
// kd-tree everywhere.  t_stop: see mesh_closest_bvh (3e38f for a closest-hit query).
template <bool COUNT>
RT_HD bool shape_intersect(const SceneView &sv, int32_t type, int32_t mesh, V3 plane_n, V3 lo, V3 ld, KdStack &st, float t_stop,
                           V3 &p, int32_t &tri, KdCounters *cnt) {
    tri = -1;
    if (type == RTHIP_SHAPE_SPHERE) return sphere_intersect(lo, ld, p);
    if (type == RTHIP_SHAPE_PLANE) return plane_intersect(plane_n, lo, ld, p);
    const rthip_mesh m = rt_load_mesh(sv.meshes + mesh);  // `mesh` is wave-uniform: scalar loads
    TriBest tb;
    if (sv.accel == 0 && !kd_degenerate_ray(sv, m, lo, ld))
        mesh_closest_bvh<COUNT>(sv, m, lo, ld, st, tb, t_stop, cnt);
    else
        mesh_closest(sv, m, lo, ld, st, tb, cnt);
    // an exact tie at the winning distance: which triangle wins is the reference walk's business (a shadow query
    // only consumes the distance: t_stop < 3e38 marks it)
    if (tb.hit && tb.tie && !(t_stop < 3.0e38f)) {
        V3 at;
        double u, v, w;
        tri_point(sv.tris[m.tri_base + tb.tri], lo, ld, at, u, v, w);
        const double dist = tb.d, ext = tb.ext;
        uint32_t tied[kMaxTied], n_tied = 0;
        if (m.bvh_node_count) {
            n_tied = tied_triangles(sv, m, lo, ld, st, dist, tied);
            if (n_tied > kMaxTied) n_tied = 0;  // (a fan of more than eight: test everything, like the reference)
        }
        mesh_closest_reference_order(sv, m, lo, ld, st, tb, at, ext, dist, tied, n_tied, cnt);
    }
    if (!tb.hit) return false;
    tri = (int32_t)tb.tri;
    double u, v, w;
    tri_point(sv.tris[m.tri_base + tb.tri], lo, ld, p, u, v, w);
    return true;
}

// host side of the any-hit bound below (called by rthip_upload_scene)
// E = fwd*inv - I of a shape's two 3x4 matrices (see rthip_shape.rt_lin / rt_t and anyhit_t_stop)
// mesh_roots: the BVH root node of every mesh (meshes[] order; unused for meshes without a BVH)
inline void roundtrip_bounds(rthip_shape &sh, const rthip_mesh *meshes, const rthip_bvhnode *mesh_roots) {
    double lin = 0.0, tr = 0.0;
    for (int i = 0; i < 3; i++) {
        double row = 0.0;
        for (int j = 0; j < 4; j++) {
            double e = 0.0, a = 0.0;
            for (int k = 0; k < 3; k++) {
                e += sh.fwd[4 * i + k] * sh.inv[4 * k + j];
                a += fabs(sh.fwd[4 * i + k] * sh.inv[4 * k + j]);
            }
            if (j == 3) { e += sh.fwd[4 * i + 3]; a += fabs(sh.fwd[4 * i + 3]); }
            if (j == i) e -= 1.0;
            const double bound = fabs(e) + 16.0 * DBL_EPSILON * (a + 1.0);  // |E_ij| plus the rounding of this very sum
            if (j < 3) row += bound; else tr = fmax(tr, bound);
        }
        lin = fmax(lin, row);
    }
    sh.rt_lin = 4.0 * lin;
    sh.rt_t = 4.0 * tr;
    if (!(sh.rt_lin < 1e300)) sh.rt_lin = 1e300;  // NaN/inf matrices: the device then never stops early
    if (!(sh.rt_t < 1e300)) sh.rt_t = 1e300;
    // Bounding ball in world space.  |M p| <= sigma_max(M) |p| and sigma_max(M)^2 = lambda_max(M^T M) <= the largest
    // absolute row sum of M^T M (exact for a rotation times a uniform scale).  Sphere: |p| <= 1 around the local
    // origin.  Mesh: the box around the BVH root's child boxes (fp32, rounded outward by the builder), |p - centre| <=
    // its half diagonal.  Plane: unbounded.
    double s2 = 0.0;
    for (int i = 0; i < 3; i++) {
        double row = 0.0;
        for (int j = 0; j < 3; j++) {
            double a = 0.0;
            for (int k = 0; k < 3; k++) a += sh.fwd[4 * k + i] * sh.fwd[4 * k + j];
            row += fabs(a);
        }
        s2 = fmax(s2, row);
    }
    double c[3] = {0.0, 0.0, 0.0}, rad2 = 1.0;
    bool unbounded = sh.type == RTHIP_SHAPE_PLANE;
    if (sh.type == RTHIP_SHAPE_MESH) {
        rad2 = 0.0;
        const rthip_mesh &m = meshes[sh.mesh];
        if (!m.bvh_node_count && m.tri_count) unbounded = true;  // kd-tree only (accel = 1 callers): no box to bound it with
        if (m.bvh_node_count) {
            const rthip_bvhnode &root = mesh_roots[sh.mesh];
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int k = 0; k < 4; k++) {
                if (root.c[k].child == RTHIP_BVH_ABSENT) continue;
                for (int a = 0; a < 3; a++) {
                    lo[a] = fmin(lo[a], (double)root.c[k].lo[a]);
                    hi[a] = fmax(hi[a], (double)root.c[k].hi[a]);
                }
            }
            if (lo[0] <= hi[0]) {
                for (int a = 0; a < 3; a++) {
                    c[a] = 0.5 * (lo[a] + hi[a]);
                    const double h = 0.5 * (hi[a] - lo[a]);
                    rad2 += h * h;
                }
                rad2 *= 1.0 + 1e-6;
            }
        }
    }
    for (int i = 0; i < 3; i++) sh.bound_c[i] = sh.fwd[4 * i] * c[0] + sh.fwd[4 * i + 1] * c[1] + sh.fwd[4 * i + 2] * c[2] + sh.fwd[4 * i + 3];
    sh.bound_r2 = s2 * rad2 * (1.0 + 1e-9);
    const bool finite = sh.bound_r2 < 1e300 && fabs(sh.bound_c[0]) < 1e300 && fabs(sh.bound_c[1]) < 1e300 && fabs(sh.bound_c[2]) < 1e300;
    if (unbounded || !finite) {  // unbounded, or NaN/inf matrices: never rejects
        sh.bound_c[0] = sh.bound_c[1] = sh.bound_c[2] = 0.0;
        sh.bound_r2 = 1e300;
    }
}

// Early rejection of a Sphere or Mesh shape in WORLD space: the shape lies inside the ball of squared radius r2
// around c (above), so a ray whose line passes outside that ball, or that starts outside it and points away,
// cannot hit the shape.  Conservative: the margin m (1e-9 relative to the squares compared) is
// a million times the rounding of this test and of the reference's local-space test (Scene.cpp:294-330), so every
// ray that one accepts passes here; NaNs never reject.  Saves the world->local transform, a normalisation (square
// root + three divisions) and the intersection / the BVH root tests for the shapes a ray is nowhere near.
RT_HD bool ball_missed(double cx, double cy, double cz, double r2, V3 o, V3 d) {
    V3 oc = sub(mk(cx, cy, cz), o);
    double cc = dot(oc, oc), b = dot(oc, d), dd = dot(d, d);
    double a = cc - r2, m = 1e-9 * (cc + r2);
    if (!(a > m)) return false;  // the origin is inside or close to the ball
    if (b < 0.0) return true;    // outside, heading away: both roots are negative
    return a * dd - b * b > m * dd;
}

// Any-hit bound.  A shadow query only needs RaycastResult.Hit, but the reference decides it from the
// CLOSEST local hit of each shape, accepted iff |O - M*P|^2 >= DBL_EPSILON (Scene.cpp:196-200): a
// self-hit closer than that masks every farther hit of the same shape.  So once any hit is known the
// mesh walk may stop looking beyond a local parameter t_stop, provided every hit that could fail the
// epsilon test lies below t_stop.  The computed world point of a local hit at parameter t is
// O + E*(O,1) + t*(M*ld) + rounding with E = M*Minv - I, so a computed distance below
// sqrt(DBL_EPSILON) = 1.5e-8 needs  t*|M*ld| <= 1.5e-8 + |E*(O,1)| + rounding.
// rt_lin >= ||E_3x3||_inf and rt_t >= |E translation|_inf come from the scene upload (fp64, x4 margin).
RT_HD float anyhit_t_stop(const double *fwd, double rt_lin, double rt_t, V3 o, V3 ld) {
    double omax = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z));
    double tmag = fmax(fmax(fabs(fwd[3]), fabs(fwd[7])), fabs(fwd[11]));
    double world = 1.5e-8 + 3.0 * rt_lin * omax + rt_t + 1e-9 * (1.0 + omax + tmag);
    V3 wd = xform(fwd, ld, 0.0);
    double stretch = length(wd);  // world distance per unit of local parameter
    double t = world / stretch * 1.001;
    if (!(t < 1e37)) return 3.0e38f;  // singular transform: no early stop
    return (float)t * 1.0000002f + 1e-37f;
}

// Scene::CastRay / CastRayRecursive (Scene.cpp:163-237) over the flat pre-order shape table:
// per shape world->local ray, Intersect, local->world point, accept iff d^2 >= DBL_EPSILON
// (Scene.cpp:200), keep the strictly smallest d^2 (earlier node wins ties, Scene.cpp:225).
// Returns the winning shape (or -1) and its triangle; any_hit stops at the first accepted shape
// (CastShadowRay only consumes .Hit, Raytracer.cpp:305-310).  `active` = false makes the call a
// no-op for this lane: the shape loop runs a wave-uniform number of times so that the shape records
// are fetched with scalar loads.
template <bool COUNT>
RT_HD int32_t cast_ray_id(const SceneView &sv, V3 o, V3 d, KdStack &st, bool any_hit, bool active, int32_t &best_tri,
                          KdCounters *cnt) {
    int32_t best = -1;
    best_tri = -1;
    double hitDistance = DBL_MAX;
    for (uint32_t s = 0; s < sv.n_shapes; s++) {
        const rt_uniform_shape *sh = rt_uniform_cast(sv.shapes) + s;
        if (!active) continue;
        if (COUNT) cnt->shapes++;
        double inv[12], fwd[12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            inv[k] = sh->inv[k];
            fwd[k] = sh->fwd[k];
        }
        const int32_t type = sh->type;
        if (type != RTHIP_SHAPE_PLANE && ball_missed(sh->bound_c[0], sh->bound_c[1], sh->bound_c[2], sh->bound_r2, o, d)) continue;
        if (COUNT) cnt->shapes_full++;
        V3 lo = xform(inv, o, 1.0);
        V3 ld = normalized(xform(inv, d, 0.0));
        V3 p;
        int32_t tri;
        float t_stop = 3.0e38f;
        if (any_hit && type == RTHIP_SHAPE_MESH) t_stop = anyhit_t_stop(fwd, sh->rt_lin, sh->rt_t, o, ld);
        if (!shape_intersect<COUNT>(sv, type, sh->mesh, mk(sh->plane_n[0], sh->plane_n[1], sh->plane_n[2]), lo, ld, st, t_stop, p,
                                    tri, cnt))
            continue;
        V3 worldPoint = xform(fwd, p, 1.0);
        double distance = sqlen(sub(o, worldPoint));
        if (!(distance >= DBL_EPSILON)) continue;
        if (!(distance < hitDistance)) continue;
        hitDistance = distance;
        best = (int32_t)s;
        best_tri = tri;
        if (any_hit) active = false;
    }
    return best;
}

// Point, local point and world normal of the winner (Scene.cpp:196-209, 327, 359, 456-463), recomputed
// with the arithmetic the search used, so the values are the ones the reference stores in RaycastResult.
RT_HD void hit_details(const SceneView &sv, V3 o, V3 d, int32_t shape, int32_t tri, Hit &h, KdCounters *cnt) {
    const rthip_shape &sh = sv.shapes[shape];
    V3 lo = xform(sh.inv, o, 1.0);
    V3 ld = normalized(xform(sh.inv, d, 0.0));
    V3 p = mk(0, 0, 0), ln;
    if (sh.type == RTHIP_SHAPE_SPHERE) {
        sphere_intersect(lo, ld, p);
        ln = normalized(p);
    } else if (sh.type == RTHIP_SHAPE_PLANE) {
        ln = mk(sh.plane_n[0], sh.plane_n[1], sh.plane_n[2]);
        plane_intersect(ln, lo, ld, p);
    } else {
        const rthip_mesh &m = sv.meshes[sh.mesh];
        const rthip_tri &t = sv.tris[m.tri_base + (uint32_t)tri];
        double u, v, w;
        tri_point(t, lo, ld, p, u, v, w);
        if (m.normal_mode == 0) {
            ln = mk(t.face_normal[0], t.face_normal[1], t.face_normal[2]);
        } else {
            if (cnt) cnt->mesh_hits++;
            const double *n = sv.tri_normals + 9 * (size_t)(m.tri_base + (uint32_t)tri);
            ln = normalized(add(add(scale(mk(n[0], n[1], n[2]), u), scale(mk(n[3], n[4], n[5]), v)), scale(mk(n[6], n[7], n[8]), w)));
        }
    }
    h.shape = shape;
    h.tri = tri;
    h.local = p;
    h.point = xform(sh.fwd, p, 1.0);
    if (sh.type == RTHIP_SHAPE_PLANE && sv.plane_wn) {  // (see SceneView::plane_wn)
        const double *n = sv.plane_wn + 3 * (size_t)shape;
        h.normal = mk(n[0], n[1], n[2]);
    } else {
        h.normal = normalized(xform(sh.nrm, ln, 0.0));
    }
}

RT_HD Hit cast_ray(const SceneView &sv, V3 o, V3 d, KdStack &st, bool any_hit, KdCounters *cnt) {
    Hit h;
    h.point = h.local = h.normal = mk(0, 0, 0);
    if (cnt)
        h.shape = cast_ray_id<true>(sv, o, d, st, any_hit, true, h.tri, cnt);
    else
        h.shape = cast_ray_id<false>(sv, o, d, st, any_hit, true, h.tri, cnt);
    if (h.shape >= 0 && !any_hit) hit_details(sv, o, d, h.shape, h.tri, h, cnt);
    return h;
}

// ----------------------------------------------------------------------------- shading (R4-R6)
// One level of Raytracer::RecursiveRaytrace (Raytracer.cpp:187-303) is split into the two wavefront
// stages the frame kernels run: trace_stage (all scene queries of the level: the closest hit and one
// shadow query per light) and shade_stage (material/noise, Lambert terms, the reflected ray).
struct HitRec {
    int32_t shape, tri;     // -1 = miss
    uint32_t shadow_mask;   // bit l set: CastShadowRay towards light l hit something (Raytracer.cpp:227, 232)
    V3 local, normal;       // RaycastResult.LocalPoint / .Normal (Scene.cpp:205-209)
};

// Closest hit, then the shadow queries.  The reference only casts shadow rays when absorptance > 0
// (Raytracer.cpp:208); casting them unconditionally cannot change the image (their results are unused
// in that case) and keeps material evaluation out of the traversal kernel.  One cast_ray call site
// serves both query kinds so the traversal code is instantiated once; the query loop runs a
// wave-uniform number of times (lanes without a hit idle through the shadow queries).
// n_shadow_lights = 0 gives the single closest-hit query of the DebugRaycaster (Raytracer.cpp:317-351).
template <bool COUNT>
RT_HD void trace_stage(const SceneView &sv, V3 o, V3 d, uint32_t n_shadow_lights, KdStack &st, HitRec &h, KdCounters *cnt) {
    h.shape = -1;
    h.tri = -1;
    h.shadow_mask = 0;
    h.local = h.normal = mk(0, 0, 0);
    V3 point = mk(0, 0, 0);
    for (int32_t q = -1; q < (int32_t)n_shadow_lights; q++) {
        V3 ro = o, rd = d;
        bool active = true;
        if (q >= 0) {
            const rt_uniform_light *light = rt_uniform_cast(sv.lights) + q;
            const int32_t ltype = light->type;
            if (!light->enabled || ltype == RTHIP_LIGHT_AMBIENT) continue;  // Raytracer.cpp:213, 234-237
            active = active && h.shape >= 0;
            ro = point;
            if (ltype == RTHIP_LIGHT_DIRECTIONAL)
                rd = mk(light->direction[0], light->direction[1], light->direction[2]);
            else
                rd = normalized(sub(mk(light->position[0], light->position[1], light->position[2]), point));
        }
        int32_t tri;
        int32_t shape = cast_ray_id<COUNT>(sv, ro, rd, st, q >= 0, active, tri, cnt);
        if (q < 0) {
            if (shape >= 0) {
                Hit c;
                hit_details(sv, o, d, shape, tri, c, cnt);
                h.shape = shape;
                h.tri = tri;
                h.local = c.local;
                h.normal = c.normal;
                point = c.point;
            }
        } else if (shape >= 0) {
            h.shadow_mask |= 1u << q;
        }
    }
}

struct ShadeOut {
    Col color;      // miss: background; hit: directLighting (Raytracer.cpp:192, 269-270)
    double refl;    // reflectance applied to the child's colour (Raytracer.cpp:283)
    bool spawn;     // a reflected ray follows (Raytracer.cpp:276)
    V3 child_o, child_d;
    uint32_t shadow_rays;  // CastShadowRay calls the reference makes for this hit
};

template <bool LEAN = false>
RT_HD void shade_stage(const SceneView &sv, const uint8_t *tabs, const HitRec &h, V3 rd, bool may_reflect, ShadeOut &out) {
    out.spawn = false;
    out.refl = 0.0;
    out.shadow_rays = 0;
    out.child_o = out.child_d = mk(0, 0, 0);
    if (h.shape < 0) {
        out.color = background_color(sv, tabs, rd);
        return;
    }
    const rthip_shape &sh = sv.shapes[h.shape];
    V3 point = xform(sh.fwd, h.local, 1.0);  // worldPoint, recomputed exactly as Scene.cpp:196
    Col albedo;
    double absorptance;
    material_eval<LEAN>(sv, tabs, sh.material, h.local, albedo, absorptance);
    double reflectance = 1.0 - absorptance;  // Material.cpp:9-12
    Col direct = Col{0, 0, 0};
    if (absorptance > 0) {
        for (uint32_t li = 0; li < sv.n_lights; li++) {
            const rthip_light &light = sv.lights[li];
            if (!light.enabled) continue;
            V3 ldir = mk(light.direction[0], light.direction[1], light.direction[2]);
            V3 lpos = mk(light.position[0], light.position[1], light.position[2]);
            if (light.type != RTHIP_LIGHT_AMBIENT) {
                out.shadow_rays++;
                if (h.shadow_mask & (1u << li)) continue;
            }
            double diffuseFactor = 0.0;
            if (light.type == RTHIP_LIGHT_DIRECTIONAL) {
                diffuseFactor = (double)fmaxf(0.0f, (float)dot(h.normal, ldir));  // Raytracer.cpp:252
            } else if (light.type == RTHIP_LIGHT_POINT) {                        // Raytracer.cpp:255-259
                V3 lightVector = sub(lpos, point);
                V3 direction = normalized(lightVector);
                double distance = length(lightVector);
                double attenuation = 1.0 - ((distance * distance) / (light.attenuation * light.attenuation));
                diffuseFactor = (double)fmaxf(0.0f, (float)dot(direction, h.normal)) * attenuation;
            } else {
                diffuseFactor = 1.0;  // Raytracer.cpp:263
            }
            Col term = c_scale(c_scale(c_mul(c_from(light.color), albedo), diffuseFactor), absorptance);
            direct = c_add(direct, term);
        }
    }
    out.color = direct;
    if (reflectance > 0 && may_reflect) {
        out.spawn = true;
        out.refl = reflectance;
        out.child_o = point;
        out.child_d = reflect(rd, h.normal);
    }
}

struct Bounce {
    Col color;
    double refl;
    bool spawn;
    V3 child_o, child_d;
    int32_t shape, tri;
    uint32_t shadow_rays;
};
// both stages back to back for one ray (probes and the host-compiled test build)
RT_HD void bounce(const SceneView &sv, V3 ro, V3 rd, bool may_reflect, KdStack &st, Bounce &out, KdCounters *cnt) {
    HitRec h;
    if (cnt)
        trace_stage<true>(sv, ro, rd, sv.n_lights, st, h, cnt);
    else
        trace_stage<false>(sv, ro, rd, sv.n_lights, st, h, cnt);
    ShadeOut so;
    shade_stage(sv, sv.perlin_tables, h, rd, may_reflect, so);
    out.color = so.color;
    out.refl = so.refl;
    out.spawn = so.spawn;
    out.child_o = so.child_o;
    out.child_d = so.child_d;
    out.shape = h.shape;
    out.tri = h.tri;
    out.shadow_rays = so.shadow_rays;
}

// Fold of one level: directLighting + (child * reflectance) with clamping Color ops (Raytracer.cpp:283, 286)
RT_HD Col fold(Col direct, Col child, double refl) { return c_add(direct, c_scale(child, refl)); }

// Whole Raytracer::Raycast for one ray, iteratively (used by probes; the frame kernels run the
// same bounce() level by level over compacted ray queues).
RT_HD Col raycast(const SceneView &sv, V3 o, V3 d, uint32_t max_recursion, KdStack &st, uint32_t *n_rays) {
    Col direct[RTHIP_MAX_RECURSION + 1];
    double refl[RTHIP_MAX_RECURSION + 1];
    uint32_t level = 0, rays = 0;
    Col c;
    for (;;) {
        Bounce b;
        bounce(sv, o, d, level < max_recursion, st, b, nullptr);
        rays += 1 + b.shadow_rays;
        if (!b.spawn) {
            // hit without reflection returns directLighting + Black (identity on clamped colours)
            c = b.shape < 0 ? b.color : c_add(b.color, Col{0, 0, 0});
            break;
        }
        direct[level] = b.color;
        refl[level] = b.refl;
        o = b.child_o;
        d = b.child_d;
        level++;
    }
    while (level > 0) {
        level--;
        c = fold(direct[level], c, refl[level]);
    }
    if (n_rays) *n_rays = rays;
    return c;
}

// DebugRaycaster::Raycast (Raytracer.cpp:317-351); mode 0 Normal, 1 Color: shading half (the query half is a
// closest-hit cast_ray)
template <bool LEAN = false>
RT_HD Col debug_shade(const SceneView &sv, const uint8_t *tabs, const HitRec &h, V3 d, int mode) {
    if (mode == 0) {
        if (h.shape >= 0) return mkc((h.normal.x + 1) / 2, (h.normal.y + 1) / 2, 1 - (h.normal.z + 1) / 2);
        return Col{0, 0, 0};
    }
    if (h.shape >= 0) {
        Col albedo;
        double a;
        material_eval<LEAN>(sv, tabs, sv.shapes[h.shape].material, h.local, albedo, a);
        return albedo;
    }
    return background_color(sv, tabs, d);
}
RT_HD Col debug_raycast(const SceneView &sv, V3 o, V3 d, int mode, KdStack &st, int32_t *shape, int32_t *tri) {
    Hit c = cast_ray(sv, o, d, st, false, nullptr);
    if (shape) *shape = c.shape;
    if (tri) *tri = c.tri;
    HitRec h;
    h.shape = c.shape;
    h.tri = c.tri;
    h.shadow_mask = 0;
    h.local = c.local;
    h.normal = c.normal;
    return debug_shade(sv, sv.perlin_tables, h, d, mode);
}

// ----------------------------------------------------------------------------- ray generation (R1)
// AbstractRaycaster::CreateScreenRay (Raytracer.cpp:102-121) with the camera basis, h2 and w2
// hoisted to the host (rthip_camera); x, y are the real pixel coordinates (integers, or +-0.5 for SSAA).
// Division of a 32-bit index by a divisor that is the same for a whole launch (tile sizes, slots per chunk): the GPU has no integer
// divide, `n / d` is ~25 instructions and the slot -> pixel mapping of every level-0 ray holds five of them.  Granlund & Montgomery
// 1994, figure 4.1 with N = 32: for l = ceil(log2 d), m' = floor(2^32 (2^l - d) / d) + 1, every 0 <= n < 2^32 has
// floor(n / d) = (t + ((n - t) >> min(l, 1))) >> max(l - 1, 0) with t = the high word of m' * n.  The host computes (m', shifts) once.
struct FastDiv {
    uint32_t m;
    uint32_t s;  // min(l, 1) | max(l - 1, 0) << 8
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    if (d == 0) d = 1;
    uint32_t l = 0;
    while (l < 32 && ((uint64_t)1 << l) < d) l++;
    f.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    f.s = (l ? 1u : 0u) | ((l ? l - 1u : 0u) << 8);
    return f;
}
RT_HD uint32_t fast_div(uint32_t n, FastDiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(f.m, n);
#else
    const uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);
#endif
    return (t + ((n - t) >> (f.s & 255u))) >> (f.s >> 8);
}

RT_HD void screen_ray(const rthip_camera &cam, uint32_t W, uint32_t H, double x, double y, V3 &o, V3 &d) {
    double xFactor = (x / (double)W) * 2.0 - 1.0;
    double yFactor = 1.0 - (y / (double)H) * 2.0;
    V3 fw = mk(cam.forward[0], cam.forward[1], cam.forward[2]);
    V3 rt = mk(cam.right[0], cam.right[1], cam.right[2]);
    V3 up = mk(cam.up[0], cam.up[1], cam.up[2]);
    o = mk(cam.position[0], cam.position[1], cam.position[2]);
    d = normalized(add(add(fw, scale(scale(rt, xFactor), cam.w2)), scale(scale(up, yFactor), cam.h2)));
}

}  // namespace rtc
