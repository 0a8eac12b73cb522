/*
 * rt_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See rt_oracle.h.
 *
 * Plain-C restatement of the Raycast hot path of erbuka/raytracing.  Each function cites
 * the reference file:line it follows.  Arithmetic is IEEE double in the reference's operation
 * order; build with -O2 -ffp-contract=off and NO -march/-ffast-math (the goldens are defined
 * by the non-fused x86-64 baseline build of the reference).
 *
 * Parity: PINNED against oracle/_ref (the genuine reference sources compiled here) — see
 * tests/test_oracle.py, tests/test_fuzz.py, tests/test_edge_cases.py and the committed vectors in tests/golden/.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef double real;
typedef struct { real x, y, z; } v3;
typedef struct { real e[4]; } v4;
typedef struct { real m[16]; } m4;
typedef struct { real r, g, b; } col;

/* ------------------------------------------------------------------ C1: math --- */
/* Common.h:9  constexpr real PI = 3.14159265359f (float literal widened) */
static const real RE_PI = (real)3.14159265359f;

/* Common.h:22-25  Clamp = std::max(min, std::min(max, value))
 * std::min(a,b) = (b < a) ? b : a ; std::max(a,b) = (a < b) ? b : a */
static real clamp01(real v) {
    real lo = (real)0.0f, hi = (real)1.0f;
    real m = (v < hi) ? v : hi;   /* std::min(max, value) */
    return (lo < m) ? m : lo;     /* std::max(min, m)     */
}
static v3 V(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }      /* Common.cpp:119 */
static v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }      /* Common.cpp:123 */
static v3 v_scale(v3 a, real t) { return V(a.x * t, a.y * t, a.z * t); }        /* Common.cpp:245 */
static v3 v_divs(v3 a, real t) { return V(a.x / t, a.y / t, a.z / t); }         /* Common.cpp:250 */
static v3 v_div(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }      /* Common.cpp:133 */
static real v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     /* Common.cpp:138 */
static v3 v_cross(v3 b, v3 c) {                                                 /* Common.cpp:143 */
    return V(b.y * c.z - b.z * c.y, b.z * c.x - b.x * c.z, b.x * c.y - b.y * c.x);
}
static real v_sqlen(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }         /* Common.cpp:37 */
static real v_len(v3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }     /* Common.cpp:42 */
static v3 v_norm(v3 a) {                                                        /* Common.cpp:47-58 */
    real l = v_len(a);
    if (l > 0) return V(a.x / l, a.y / l, a.z / l);
    return V(0, 0, 0);
}
/* Common.cpp:60-63  *this - normal * (normal ^ *this) * 2.0f */
static v3 v_reflect(v3 d, v3 n) { return v_sub(d, v_scale(v_scale(n, v_dot(n, d)), (real)2.0f)); }
static real v_get(v3 a, unsigned ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }

static col C(real r, real g, real b) { col c = {clamp01(r), clamp01(g), clamp01(b)}; return c; } /* Common.cpp:260 */
static col c_raw(real r, real g, real b) { col c = {r, g, b}; return c; }
static col c_add(col a, col b) { return C(clamp01(a.r + b.r), clamp01(a.g + b.g), clamp01(a.b + b.b)); } /* :212 */
static col c_mul(col a, col b) { return C(clamp01(a.r * b.r), clamp01(a.g * b.g), clamp01(a.b * b.b)); } /* :221 */
static col c_scale(col a, real k) { return C(clamp01(a.r * k), clamp01(a.g * k), clamp01(a.b * k)); }    /* :230 */
/* Common.cpp:239-243 Mix -> brace-init Color(r,g,b) (clamping ctor) */
static col c_mix(col a, col b, real t) {
    real _t = 1 - t;
    return C(_t * a.r + t * b.r, _t * a.g + t * b.g, _t * a.b + t * b.b);
}
/* Common.cpp:267-273  Color(unsigned): factor is a FLOAT quotient widened to real */
static col c_hex(unsigned color) {
    const real factor = 1.0f / 255.0f;
    col c;
    c.b = (color & 0x000000ff) * factor;
    c.g = ((color & 0x0000ff00) >> 8) * factor;
    c.r = ((color & 0x00ff0000) >> 16) * factor;
    return c;
}
/* Common.cpp:275-282 GetHexValue: truncation, R in the low byte */
static unsigned c_pack(col c) {
    return 0xff000000u | ((unsigned)(c.r * (real)255.0f)) | (((unsigned)(c.g * (real)255.0f)) << 8) |
           (((unsigned)(c.b * (real)255.0f)) << 16);
}

static const m4 M_IDENT = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
/* Common.cpp:152-167 */
static m4 m_mul(const m4 *l, const m4 *r) {
    m4 o;
    memset(&o, 0, sizeof o);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) o.m[i * 4 + j] += l->m[i * 4 + k] * r->m[k * 4 + j];
    return o;
}
/* Common.cpp:169-180 */
static v4 m_mulv(const m4 *m, v3 p, real w) {
    v4 in = {{p.x, p.y, p.z, w}}, o = {{0, 0, 0, 0}};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) o.e[i] += m->m[i * 4 + j] * in.e[j];
    return o;
}
static v3 v4_xyz(v4 a) { return V(a.e[0], a.e[1], a.e[2]); } /* Common.cpp:30-35 */
static m4 m_translation(v3 t) { m4 o = M_IDENT; o.m[3] = t.x; o.m[7] = t.y; o.m[11] = t.z; return o; } /* :365 */
static m4 m_scale(v3 s) { m4 o = M_IDENT; o.m[0] = s.x; o.m[5] = s.y; o.m[10] = s.z; return o; }       /* :377 */
static m4 m_transpose(const m4 *a) {                                                                  /* :388 */
    m4 o;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) o.m[i * 4 + j] = a->m[j * 4 + i];
    return o;
}
/* Common.cpp:401-423 — cosf/sinf of the float-rounded angles */
static m4 m_rotation(v3 rot) {
    m4 o = M_IDENT;
    real cyaw = cosf((float)rot.z), syaw = sinf((float)rot.z);
    real cpitch = cosf((float)rot.y), spitch = sinf((float)rot.y);
    real croll = cosf((float)rot.x), sroll = sinf((float)rot.x);
    o.m[0] = cyaw * cpitch;
    o.m[1] = cyaw * spitch * sroll - syaw * croll;
    o.m[2] = cyaw * spitch * croll + syaw * sroll;
    o.m[4] = syaw * cpitch;
    o.m[5] = syaw * spitch * sroll + cyaw * croll;
    o.m[6] = syaw * spitch * croll - cyaw * sroll;
    o.m[8] = -spitch;
    o.m[9] = cpitch * sroll;
    o.m[10] = cpitch * croll;
    return o;
}

/* ------------------------------------------------------------- bounding box --- */
typedef struct { v3 mn, mx; } bbox;
/* Common.cpp:313-338 */
static void bb_split(const bbox *b, unsigned axis, real value, bbox *l, bbox *r) {
    *l = *b;
    *r = *b;
    switch (axis) {
    case 0: l->mx.x = value; r->mn.x = value; break;
    case 1: l->mx.y = value; r->mn.y = value; break;
    case 2: l->mx.z = value; r->mn.z = value; break;
    }
}
/* Common.cpp:345-363 (only .Hit is consumed by the hot path) */
static int bb_intersect(const bbox *b, v3 o, v3 d) {
    real t1 = (b->mn.x - o.x) / d.x, t2 = (b->mx.x - o.x) / d.x;
    real t3 = (b->mn.y - o.y) / d.y, t4 = (b->mx.y - o.y) / d.y;
    real t5 = (b->mn.z - o.z) / d.z, t6 = (b->mx.z - o.z) / d.z;
    real t7 = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6));
    real t8 = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
    real t9 = (t8 < 0 || t7 > t8) ? -1 : t7;
    return t9 != -1;
}

/* ------------------------------------------------------------------- noise --- */
enum { N_CHECKER, N_PERLIN, N_WORLEY, N_MARBLE };
typedef struct {
    real seed[256];  /* Perlin.h:26 (stored as reals) */
    real persistance;
    unsigned octaves;
} perlin_t;
typedef struct noise_s {
    int type;
    real domain;
    perlin_t perlin;       /* N_PERLIN, N_MARBLE (Marble.h:15) */
    real freq, turb;       /* marble */
    int div;               /* worley */
    real step;
    v3 *pts;
} noise_t;

/* Perlin.cpp:17-23 NewSeed */
static void perlin_init(perlin_t *p, const uint8_t *table) {
    p->persistance = 0.5f;
    p->octaves = 8;
    for (int i = 0; i < 256; i++) p->seed[i] = table ? (real)table[i] : (real)(rand() % (size_t)256);
}
/* Perlin.cpp:49-55 */
static real perlin_getseed(const perlin_t *p, size_t x, size_t y, size_t z) {
    size_t ix = (size_t)p->seed[x % 256];
    size_t iy = (size_t)p->seed[(ix + y) % 256];
    size_t iz = (size_t)p->seed[(iy + z) % 256];
    return (real)iz / 256;
}
static real lerp_r(real a, real b, real t) { return a * (1 - t) + b * t; } /* Common.h:27-30 */
static real fade(real t) { return t * t * t * (t * (t * 6 - 15) + 10); }  /* Perlin.cpp:59-62 */
/* Perlin.cpp:57-83 */
static real perlin_at_freq(const perlin_t *p, v3 pos, unsigned frequency) {
    real fq = frequency;
    v3 q = v_divs(pos, fq);
    v3 mn = v_scale(V(floor(q.x), floor(q.y), floor(q.z)), fq);
    v3 mx = v_add(mn, V(fq, fq, fq));
    real a = perlin_getseed(p, (size_t)mn.x, (size_t)mn.y, (size_t)mn.z);
    real b = perlin_getseed(p, (size_t)mx.x, (size_t)mn.y, (size_t)mn.z);
    real c = perlin_getseed(p, (size_t)mn.x, (size_t)mx.y, (size_t)mn.z);
    real d = perlin_getseed(p, (size_t)mx.x, (size_t)mx.y, (size_t)mn.z);
    real e = perlin_getseed(p, (size_t)mn.x, (size_t)mn.y, (size_t)mx.z);
    real f = perlin_getseed(p, (size_t)mx.x, (size_t)mn.y, (size_t)mx.z);
    real g = perlin_getseed(p, (size_t)mn.x, (size_t)mx.y, (size_t)mx.z);
    real h = perlin_getseed(p, (size_t)mx.x, (size_t)mx.y, (size_t)mx.z);
    real u = fade((pos.x - mn.x) / fq);
    real v = fade((pos.y - mn.y) / fq);
    real w = fade((pos.z - mn.z) / fq);
    return lerp_r(lerp_r(lerp_r(a, b, u), lerp_r(c, d, u), v), lerp_r(lerp_r(e, f, u), lerp_r(g, h, u), v), w);
}
/* Perlin.cpp:25-47 */
static real perlin_sample_normalized(const perlin_t *p, v3 position) {
    real amplitude = 1.0f, amplitudeAcc = 0.0f, result = 0.0f;
    unsigned frequency = 256;
    v3 rpos = v_scale(position, (real)256);
    for (unsigned i = 0; i < p->octaves; i++) {
        result += amplitude * perlin_at_freq(p, rpos, frequency);
        frequency = frequency >> 1;
        amplitudeAcc += amplitude;
        amplitude *= p->persistance;
    }
    result /= amplitudeAcc;
    return result;
}
/* Worley.cpp:64-68, 70-84 */
static int worley_wrap(const noise_t *n, int index) {
    while (index < 0) index += n->div;
    return index % n->div;
}
static int worley_off(int v, int divisions) { return v < 0 ? -1 : (v >= divisions ? 1 : 0); }
static v3 worley_point_at(const noise_t *n, int x, int y, int z) {
    int ox = worley_off(x, n->div), oy = worley_off(y, n->div), oz = worley_off(z, n->div);
    v3 p = n->pts[worley_wrap(n, z) * n->div * n->div + worley_wrap(n, y) * n->div + worley_wrap(n, x)];
    return v_add(p, V(ox, oy, oz));
}
/* Worley.cpp:29-57 */
static real worley_sample_normalized(const noise_t *n, v3 point) {
    real maxDistance = n->step * sqrtf(3.0f);
    int x = (int)(point.x * n->div), y = (int)(point.y * n->div), z = (int)(point.z * n->div);
    real distance = FLT_MAX;
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++) {
                v3 p = worley_point_at(n, x + dx, y + dy, z + dz);
                real d = v_sqlen(v_sub(p, point));
                if (d < distance) distance = d;
            }
    return distance / maxDistance * 10.0f;
}
static real noise_sample_normalized(const noise_t *n, v3 p) {
    switch (n->type) {
    case N_CHECKER: { /* CheckerBoard.cpp:3-16 */
        int t0 = p.x <= 0.5, t1 = 0, t2 = p.z <= 0.5;
        return (real)(t2 ^ (t0 ^ t1));
    }
    case N_PERLIN: return perlin_sample_normalized(&n->perlin, p);
    case N_WORLEY: return worley_sample_normalized(n, p);
    case N_MARBLE: { /* Marble.cpp:11-15 */
        real f = perlin_sample_normalized(&n->perlin, p) * n->turb;
        return fabs(sin(f * n->freq * 3.141592f));
    }
    }
    return 0;
}
/* Noise.cpp:3-19 — fmodf rounds point and domain to float; quotient by the DOUBLE domain */
static real noise_wrap(real x, real domain) {
    x = fmodf((float)x, (float)domain) / domain;
    return x < 0 ? 1.0 + x : x;
}
static real noise_sample(const noise_t *n, v3 p) {
    v3 q = V(noise_wrap(p.x, n->domain), noise_wrap(p.y, n->domain), noise_wrap(p.z, n->domain));
    return noise_sample_normalized(n, q);
}

/* ---------------------------------------------------------------- materials --- */
typedef struct {
    int interp;  /* 0 uniform (Material.h:18-32), 1 interpolated (Material.h:35-47) */
    col color;
    real absorptance;
    int noise, m0, m1;
} material_t;

/* ------------------------------------------------------------------- shapes --- */
typedef struct {
    v3 v[3], n[3];
    v3 face_normal, edges[3];
    real d00, d01, d11, inv_den;
} tri_t;
typedef struct kd_s {
    unsigned depth;
    bbox bounds;
    struct kd_s *left, *right;
    int *tris;   /* indices into the mesh triangle list (the reference copies by value) */
    int ntris;
} kd_t;
enum { S_SPHERE, S_PLANE, S_MESH };
typedef struct {
    int type, material;
    unsigned id;
    v3 plane_normal;
    int normal_mode;
    tri_t *tris;
    int ntris;
    bbox bounds;
    kd_t *kd;
} shape_t;
typedef struct node_s {
    int parent;
    int *children;
    int nchildren;
    v3 position, rotation, scale;
    m4 transform, inverse, normal;
    shape_t *shape;
    int shape_ordinal; /* pre-order ordinal among shape nodes, filled at compile */
} node_t;
typedef struct { int type; v3 direction, position; col color; real attenuation; int enabled; } light_t;
enum { BG_NONE, BG_COLOR, BG_SKYBOX };
struct sb_scene {
    v3 cam_pos, cam_dir;
    node_t *nodes; int nnodes;
    material_t *mats; int nmats;
    noise_t *noises; int nnoises;
    light_t *lights; int nlights;
    int bg_type; col bg_color0, bg_color1; int bg_sun; perlin_t bg_perlin;
    int compiled;
};
static unsigned g_next_shape_id = 1; /* Scene.cpp:365-370 */

/* Scene.cpp:509-523 */
static void tri_update(tri_t *t) {
    t->face_normal = v_norm(v_cross(v_sub(t->v[1], t->v[0]), v_sub(t->v[2], t->v[1])));
    t->edges[0] = v_sub(t->v[1], t->v[0]);
    t->edges[1] = v_sub(t->v[2], t->v[0]);
    t->edges[2] = v_sub(t->v[2], t->v[1]);
    t->d00 = v_dot(t->edges[0], t->edges[0]);
    t->d01 = v_dot(t->edges[0], t->edges[1]);
    t->d11 = v_dot(t->edges[1], t->edges[1]);
    t->inv_den = 1.0 / (t->d00 * t->d11 - t->d01 * t->d01);
}
/* Scene.cpp:18-95 */
static kd_t *kd_build(const tri_t *all, const int *tris, int n, const bbox *bounds, int depth) {
    kd_t *k = (kd_t *)calloc(1, sizeof *k);
    k->depth = depth;
    k->bounds = *bounds;
    if (n <= 1 || depth == 100) {
        k->tris = (int *)malloc(sizeof(int) * (n ? n : 1));
        memcpy(k->tris, tris, sizeof(int) * n);
        k->ntris = n;
        return k;
    }
    unsigned ax = depth % 3;
    real median = 0;
    for (int i = 0; i < n; i++) {
        const tri_t *t = &all[tris[i]];
        median += v_get(t->v[0], ax);
        median += v_get(t->v[1], ax);
        median += v_get(t->v[2], ax);
    }
    median /= 3 * (size_t)n;
    bbox lb, rb;
    bb_split(bounds, ax, median, &lb, &rb);
    int *lt = (int *)malloc(sizeof(int) * n), *rt = (int *)malloc(sizeof(int) * n);
    int nl = 0, nr = 0;
    for (int i = 0; i < n; i++) {
        const tri_t *t = &all[tris[i]];
        real a = v_get(t->v[0], ax), b = v_get(t->v[1], ax), c = v_get(t->v[2], ax);
        if (a <= median || b <= median || c <= median) lt[nl++] = tris[i];
        if (a >= median || b >= median || c >= median) rt[nr++] = tris[i];
    }
    if ((size_t)nl + (size_t)nr > 1.5 * (size_t)n) {
        k->tris = (int *)malloc(sizeof(int) * n);
        memcpy(k->tris, tris, sizeof(int) * n);
        k->ntris = n;
    } else {
        if (nl > 0) k->left = kd_build(all, lt, nl, &lb, depth + 1);
        if (nr > 0) k->right = kd_build(all, rt, nr, &rb, depth + 1);
    }
    free(lt);
    free(rt);
    return k;
}
static void kd_free(kd_t *k) {
    if (!k) return;
    kd_free(k->left);
    kd_free(k->right);
    free(k->tris);
    free(k);
}
/* Scene.cpp:386-414 — bbox starts at (0,0,0)/(0,0,0) (Common.h:212) */
static void mesh_compile(shape_t *s) {
    if (s->kd) return; /* m_Invalidated == false */
    for (int i = 0; i < s->ntris; i++) {
        tri_t *t = &s->tris[i];
        tri_update(t);
        for (int k = 0; k < 3; k++) {
            v3 v = t->v[k];
            s->bounds.mn.x = (v.x < s->bounds.mn.x) ? v.x : s->bounds.mn.x;
            s->bounds.mn.y = (v.y < s->bounds.mn.y) ? v.y : s->bounds.mn.y;
            s->bounds.mn.z = (v.z < s->bounds.mn.z) ? v.z : s->bounds.mn.z;
            s->bounds.mx.x = (s->bounds.mx.x < v.x) ? v.x : s->bounds.mx.x;
            s->bounds.mx.y = (s->bounds.mx.y < v.y) ? v.y : s->bounds.mx.y;
            s->bounds.mx.z = (s->bounds.mx.z < v.z) ? v.z : s->bounds.mx.z;
        }
    }
    int *all = (int *)malloc(sizeof(int) * (s->ntris ? s->ntris : 1));
    for (int i = 0; i < s->ntris; i++) all[i] = i;
    s->kd = kd_build(s->tris, all, s->ntris, &s->bounds, 0);
    free(all);
}

typedef struct { int hit; v3 point, normal; int tri; } rayhit_t;

/* Scene.cpp:294-330 */
static rayhit_t sphere_intersect(v3 o, v3 d) {
    rayhit_t r = {0, {0, 0, 0}, {0, 0, 0}, -1};
    real projection = v_dot(v_sub(V(0, 0, 0), o), d);
    real squaredDistance = v_dot(o, o) - projection * projection;
    if (squaredDistance > 1.0f) return r;
    real offset = sqrt(1.0f - squaredDistance);
    real t1 = projection - offset, t2 = projection + offset;
    if (t1 < 0 && t2 < 0) return r;
    r.hit = 1;
    r.point = v_add(o, v_scale(d, (t1 >= 0.0f ? t1 : t2)));
    r.normal = v_norm(r.point);
    return r;
}
/* Scene.cpp:332-362 */
static rayhit_t plane_intersect(v3 n, v3 o, v3 d) {
    rayhit_t r = {0, {0, 0, 0}, {0, 0, 0}, -1};
    real distance = v_dot(o, n);
    if (distance < 0) return r;
    real cosine = v_dot(n, d);
    if (cosine >= 0) return r;
    r.hit = 1;
    r.point = v_add(o, v_scale(d, distance / -cosine));
    r.normal = n;
    return r;
}
/* Scene.cpp:495-507 */
static v3 tri_baricentric(const tri_t *t, v3 point) {
    v3 v2 = v_sub(point, t->v[0]);
    real d20 = v_dot(v2, t->edges[0]);
    real d21 = v_dot(v2, t->edges[1]);
    real v = (t->d11 * d20 - t->d01 * d21) * t->inv_den;
    real w = (t->d00 * d21 - t->d01 * d20) * t->inv_den;
    real u = 1.0f - v - w;
    return V(u, v, w);
}
/* Scene.cpp:421-467 */
static rayhit_t mesh_intersect_triangle(const shape_t *s, v3 o, v3 d, const tri_t *t) {
    rayhit_t r = {0, {0, 0, 0}, {0, 0, 0}, -1};
    v3 l = v_sub(o, t->v[0]);
    real distance = v_dot(l, t->face_normal);
    if (distance < 0) return r;
    real cosine = v_dot(d, t->face_normal);
    if (cosine >= 0) return r;
    v3 projection = v_add(o, v_scale(d, distance / -cosine));
    v3 bar = tri_baricentric(t, projection);
    if (bar.x >= 0 && bar.y >= 0 && bar.z >= 0) {
        r.hit = 1;
        r.point = projection;
        if (s->normal_mode == 0)
            r.normal = t->face_normal;
        else
            r.normal = v_norm(v_add(v_add(v_scale(t->n[0], bar.x), v_scale(t->n[1], bar.y)), v_scale(t->n[2], bar.z)));
    }
    return r;
}
/* Scene.cpp:469-493 — exhaustive, unordered walk; strict '<' so the first in pre-order wins ties */
static void mesh_intersect_internal(const shape_t *s, v3 o, v3 d, const kd_t *node, rayhit_t *result, real *distance) {
    if (bb_intersect(&node->bounds, o, d)) {
        for (int i = 0; i < node->ntris; i++) {
            rayhit_t r = mesh_intersect_triangle(s, o, d, &s->tris[node->tris[i]]);
            real dd = v_sqlen(v_sub(o, r.point));
            if (dd < *distance && r.hit) {
                *result = r;
                result->tri = node->tris[i];
                *distance = dd;
            }
        }
        if (node->left) mesh_intersect_internal(s, o, d, node->left, result, distance);
        if (node->right) mesh_intersect_internal(s, o, d, node->right, result, distance);
    }
}
/* Scene.cpp:372-378 */
static rayhit_t mesh_intersect(const shape_t *s, v3 o, v3 d) {
    rayhit_t r = {0, {0, 0, 0}, {0, 0, 0}, -1};
    real distance = DBL_MAX;
    mesh_intersect_internal(s, o, d, s->kd, &r, &distance);
    return r;
}
static rayhit_t shape_intersect(const shape_t *s, v3 o, v3 d) {
    switch (s->type) {
    case S_SPHERE: return sphere_intersect(o, d);
    case S_PLANE: return plane_intersect(s->plane_normal, o, d);
    default: return mesh_intersect(s, o, d);
    }
}

/* -------------------------------------------------------------- scene graph --- */
/* Scene.cpp:239-264 */
static void transform_compile(sb_scene *s, node_t *n) {
    m4 prev = M_IDENT, prevInverse = M_IDENT;
    if (n->parent >= 0) {
        prev = s->nodes[n->parent].transform;
        prevInverse = s->nodes[n->parent].inverse;
    }
    m4 t = m_translation(n->position), r = m_rotation(n->rotation), sc = m_scale(n->scale);
    m4 it = m_translation(v_sub(V(0, 0, 0), n->position));
    m4 ir = m_transpose(&r);
    m4 is = m_scale(v_div(V(1, 1, 1), n->scale));
    m4 a = m_mul(&prev, &t);
    a = m_mul(&a, &r);
    n->transform = m_mul(&a, &sc);
    m4 b = m_mul(&is, &ir);
    b = m_mul(&b, &it);
    n->inverse = m_mul(&b, &prevInverse);
    n->normal = m_transpose(&n->inverse);
}
/* Scene.cpp:279-292 — components (Transform first, then Shape), then children */
static void node_compile(sb_scene *s, int ni, int *ordinal) {
    node_t *n = &s->nodes[ni];
    transform_compile(s, n);
    n->shape_ordinal = -1;
    if (n->shape) {
        n->shape_ordinal = (*ordinal)++;
        if (n->shape->type == S_MESH) mesh_compile(n->shape);
    }
    for (int i = 0; i < n->nchildren; i++) node_compile(s, n->children[i], ordinal);
}
static void scene_compile(sb_scene *s) { /* Scene.cpp:158-161 */
    int ord = 0;
    node_compile(s, 0, &ord);
    s->compiled = 1;
}

typedef struct { int hit; v3 point, local, normal; int node; int tri; } castres_t;
static __thread uint64_t t_ray_count;

/* Scene.cpp:168-237 */
static castres_t cast_ray_recursive(const sb_scene *s, v3 o, v3 d, int ni) {
    const node_t *n = &s->nodes[ni];
    castres_t res = {0, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, -1, -1};
    real hitDistance = DBL_MAX;
    if (n->shape) {
        v3 lo = v4_xyz(m_mulv(&n->inverse, o, 1));
        v3 ld = v_norm(v4_xyz(m_mulv(&n->inverse, d, 0)));
        rayhit_t r = shape_intersect(n->shape, lo, ld);
        if (r.hit) {
            v3 worldPoint = v4_xyz(m_mulv(&n->transform, r.point, 1));
            real distance = v_sqlen(v_sub(o, worldPoint));
            if (distance >= DBL_EPSILON) {
                res.hit = 1;
                res.point = worldPoint;
                res.local = r.point;
                res.normal = v_norm(v4_xyz(m_mulv(&n->normal, r.normal, 0)));
                res.node = ni;
                res.tri = r.tri;
                hitDistance = distance;
            }
        }
    }
    for (int i = 0; i < n->nchildren; i++) {
        castres_t c = cast_ray_recursive(s, o, d, n->children[i]);
        if (c.hit) {
            real distance = v_sqlen(v_sub(o, c.point));
            if (distance < hitDistance) {
                res = c;
                hitDistance = distance;
            }
        }
    }
    return res;
}
/* Scene.cpp:163-166 */
static castres_t cast_ray(const sb_scene *s, v3 o, v3 d) {
    t_ray_count++;
    return cast_ray_recursive(s, o, d, 0);
}

/* Material.cpp:9-12, 21-29 */
static col material_color(const sb_scene *s, int mi, v3 p) {
    const material_t *m = &s->mats[mi];
    if (!m->interp) return m->color;
    col a = material_color(s, m->m0, p), b = material_color(s, m->m1, p);
    real t = noise_sample(&s->noises[m->noise], p);
    return c_add(c_scale(a, 1 - t), c_scale(b, t)); /* Lerp<Color,real> */
}
static real material_absorptance(const sb_scene *s, int mi, v3 p) {
    const material_t *m = &s->mats[mi];
    if (!m->interp) return m->absorptance;
    real a = material_absorptance(s, m->m0, p), b = material_absorptance(s, m->m1, p);
    real t = noise_sample(&s->noises[m->noise], p);
    return lerp_r(a, b, t);
}
static real material_reflectance(const sb_scene *s, int mi, v3 p) { return (real)(1.0 - material_absorptance(s, mi, p)); }

/* Scene.cpp:111-151 */
static col background_color(const sb_scene *s, v3 direction) {
    if (s->bg_type == BG_COLOR) return s->bg_color0; /* Scene.h:43-49 */
    const float thresold = 0.98f;
    const float border = (1.0f - thresold) / 2.0f;
    const float thresoldPlusBorder = thresold + border;
    /* m_Perlin = Perlin(1): Noise::Sample with domain 1 */
    v3 q = v_divs(v_add(direction, V(1, 1, 1)), 2);
    v3 w = V(noise_wrap(q.x, 1), noise_wrap(q.y, 1), noise_wrap(q.z, 1));
    real sample = perlin_sample_normalized(&s->bg_perlin, w);
    col color = c_mix(s->bg_color0, s->bg_color1, sample);
    if (s->bg_sun >= 0 && s->lights[s->bg_sun].type == 0) {
        const light_t *sun = &s->lights[s->bg_sun];
        float f = fmaxf(0, (float)v_dot(direction, sun->direction));
        if (f < thresold) return color;
        else if (f < thresoldPlusBorder) {
            f = powf((f - thresold) / border, 2.0f);
            return c_mix(color, sun->color, f);
        } else
            return sun->color;
    }
    return color;
}

/* Raytracer.cpp:305-310 */
static int cast_shadow_ray(const sb_scene *s, v3 o, v3 d) { return cast_ray(s, o, d).hit; }
/* Raytracer.cpp:187-303 */
static col recursive_raytrace(const sb_scene *s, v3 ro, v3 rd, int recursion, unsigned max_recursion) {
    castres_t rr = cast_ray(s, ro, rd);
    col directLighting = c_raw(0, 0, 0), indirectLighting = c_raw(0, 0, 0);
    if (rr.hit) {
        v3 worldPoint = rr.point, localPoint = rr.local, normal = rr.normal;
        int material = s->nodes[rr.node].shape->material;
        real absorptance = material_absorptance(s, material, localPoint);
        real reflectance = material_reflectance(s, material, localPoint);
        if (absorptance > 0) {
            for (int li = 0; li < s->nlights; li++) {
                const light_t *light = &s->lights[li];
                if (!light->enabled) continue;
                int shadowRayHit = 0;
                switch (light->type) {
                case 0: shadowRayHit = cast_shadow_ray(s, worldPoint, light->direction); break;
                case 2: shadowRayHit = cast_shadow_ray(s, worldPoint, v_norm(v_sub(light->position, worldPoint))); break;
                case 1: shadowRayHit = 0; break;
                }
                if (!shadowRayHit) {
                    real diffuseFactor = 0.0f, distance, attenuation;
                    v3 direction, lightVector;
                    switch (light->type) {
                    case 0: diffuseFactor = fmaxf(0.0f, (float)v_dot(normal, light->direction)); break;
                    case 2:
                        lightVector = v_sub(light->position, worldPoint);
                        direction = v_norm(lightVector);
                        distance = v_len(lightVector);
                        attenuation = 1.0f - ((distance * distance) / (light->attenuation * light->attenuation));
                        diffuseFactor = fmaxf(0.0f, (float)v_dot(direction, normal)) * attenuation;
                        break;
                    case 1: diffuseFactor = 1.0f; break;
                    }
                    col term = c_scale(c_scale(c_mul(light->color, material_color(s, material, localPoint)), diffuseFactor), absorptance);
                    directLighting = c_add(directLighting, term);
                }
            }
        }
        if (reflectance > 0 && (unsigned)recursion < max_recursion) {
            v3 refl = v_reflect(rd, normal);
            indirectLighting = c_scale(recursive_raytrace(s, worldPoint, refl, recursion + 1, max_recursion), reflectance);
        }
        return c_add(directLighting, indirectLighting);
    }
    if (s->bg_type != BG_NONE) return background_color(s, rd);
    return c_raw(0, 0, 0);
}
/* Raytracer.cpp:317-351 */
static col debug_raycast(const sb_scene *s, v3 o, v3 d, int mode) {
    castres_t r = cast_ray(s, o, d);
    if (mode == 0) {
        if (r.hit) return C((r.normal.x + 1) / 2, (r.normal.y + 1) / 2, 1 - (r.normal.z + 1) / 2);
    } else {
        if (r.hit) return material_color(s, s->nodes[r.node].shape->material, r.local);
        if (s->bg_type != BG_NONE) return background_color(s, d);
    }
    return c_raw(0, 0, 0);
}

/* Raytracer.cpp:102-121 */
static void create_screen_ray(const sb_scene *s, size_t W, size_t H, real fovy, real x, real y, v3 *o, v3 *d) {
    v3 forward = v_norm(s->cam_dir);
    v3 right = v_norm(v_cross(forward, V(0, 1, 0)));
    v3 up = v_norm(v_cross(right, forward));
    real h2 = atan(fovy / 2);
    real w2 = h2 * (real)W / (real)H;
    real xFactor = ((real)x / W) * 2.0f - 1.0f;
    real yFactor = 1.0f - ((real)y / H) * 2.0f;
    *o = s->cam_pos;
    *d = v_norm(v_add(v_add(forward, v_scale(v_scale(right, xFactor), w2)), v_scale(v_scale(up, yFactor), h2)));
}

typedef struct {
    const sb_scene *s;
    const sb_render_params *p;
    real fovy;
    col *colors;
    int32_t *shape_ord, *tri_idx;
    pthread_mutex_t mu;
    unsigned next_col; /* Raytracer.cpp:172-185 */
    uint64_t rays;
    const int32_t *sub_x, *sub_y;
    int n_sub;
} job_t;

static col raycast_dispatch(const job_t *j, v3 o, v3 d) {
    if (j->p->renderer == 0) return recursive_raytrace(j->s, o, d, 0, (unsigned)j->p->max_recursion);
    return debug_raycast(j->s, o, d, j->p->renderer - 1);
}
/* Raytracer.cpp:130-148 body for one pixel */
static void render_pixel(job_t *j, unsigned x, int y, size_t out_idx) {
    const sb_render_params *p = j->p;
    v3 o, d;
    if (p->aa_mode == 1) {
        col result = c_raw(0, 0, 0);
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++) {
                create_screen_ray(j->s, p->width, p->height, j->fovy, (real)(x + dx * .5f), (real)(y + dy * .5f), &o, &d);
                result = c_add(result, c_scale(raycast_dispatch(j, o, d), (1.0f / 9.0f)));
            }
        j->colors[out_idx] = result;
    } else {
        create_screen_ray(j->s, p->width, p->height, j->fovy, x, y, &o, &d);
        j->colors[out_idx] = raycast_dispatch(j, o, d);
    }
    if (j->shape_ord || j->tri_idx) {
        uint64_t keep = t_ray_count;
        create_screen_ray(j->s, p->width, p->height, j->fovy, x, y, &o, &d);
        castres_t r = cast_ray(j->s, o, d);
        t_ray_count = keep; /* id probe is not part of the frame's ray count */
        if (j->shape_ord) j->shape_ord[out_idx] = r.hit ? j->s->nodes[r.node].shape_ordinal : -1;
        if (j->tri_idx) j->tri_idx[out_idx] = r.hit ? r.tri : -1;
    }
}
/* Raytracer.cpp:123-158 */
static void *render_thread(void *arg) {
    job_t *j = (job_t *)arg;
    t_ray_count = 0;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        unsigned x = j->next_col;
        unsigned limit = j->n_sub > 0 ? (unsigned)j->n_sub : (unsigned)j->p->width;
        int more = x < limit;
        if (more) j->next_col++;
        pthread_mutex_unlock(&j->mu);
        if (!more) break;
        if (j->n_sub > 0) {
            render_pixel(j, (unsigned)j->sub_x[x], j->sub_y[x], x);
        } else {
            for (int y = 0; y < j->p->height; y++) render_pixel(j, x, y, (size_t)y * j->p->width + x);
        }
    }
    pthread_mutex_lock(&j->mu);
    j->rays += t_ray_count;
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

/* ============================================================ public API ==== */
const char *sb_impl_name(void) { return "oracle-c"; }

sb_scene *sb_scene_new(void) {
    sb_scene *s = (sb_scene *)calloc(1, sizeof *s);
    s->cam_pos = V(0, 0, 0);
    s->cam_dir = V(0, 0, -1); /* Scene.h:82-83 */
    s->bg_type = BG_NONE;
    s->bg_sun = -1;
    s->nodes = (node_t *)calloc(1, sizeof(node_t));
    s->nnodes = 1; /* root: Scene.cpp:153-156, SceneNode ctor adds a Transform */
    s->nodes[0].parent = -1;
    s->nodes[0].scale = V(1, 1, 1);
    return s;
}
void sb_scene_free(sb_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->nnodes; i++) {
        free(s->nodes[i].children);
        if (s->nodes[i].shape) {
            kd_free(s->nodes[i].shape->kd);
            free(s->nodes[i].shape->tris);
            free(s->nodes[i].shape);
        }
    }
    for (int i = 0; i < s->nnoises; i++) free(s->noises[i].pts);
    free(s->nodes); free(s->mats); free(s->noises); free(s->lights);
    free(s);
}
void sb_camera(sb_scene *s, const double pos[3], const double dir[3]) {
    s->cam_pos = V(pos[0], pos[1], pos[2]);
    s->cam_dir = V(dir[0], dir[1], dir[2]);
}
static int add_mat(sb_scene *s, material_t m) {
    s->mats = (material_t *)realloc(s->mats, sizeof(material_t) * (s->nmats + 1));
    s->mats[s->nmats] = m;
    return s->nmats++;
}
int sb_mat_uniform(sb_scene *s, unsigned rgb, double absorptance) {
    material_t m; memset(&m, 0, sizeof m);
    m.color = c_hex(rgb); m.absorptance = absorptance;
    return add_mat(s, m);
}
int sb_mat_interp(sb_scene *s, int noise, int m0, int m1) {
    if (noise < 0 || noise >= s->nnoises || m0 < 0 || m0 >= s->nmats || m1 < 0 || m1 >= s->nmats) return -1;
    material_t m; memset(&m, 0, sizeof m);
    m.interp = 1; m.noise = noise; m.m0 = m0; m.m1 = m1;
    return add_mat(s, m);
}
static int add_noise(sb_scene *s, noise_t n) {
    s->noises = (noise_t *)realloc(s->noises, sizeof(noise_t) * (s->nnoises + 1));
    s->noises[s->nnoises] = n;
    return s->nnoises++;
}
int sb_noise_checker(sb_scene *s, double domain) {
    noise_t n; memset(&n, 0, sizeof n);
    n.type = N_CHECKER; n.domain = domain;
    return add_noise(s, n);
}
int sb_noise_perlin(sb_scene *s, double domain, const uint8_t *table256) {
    noise_t n; memset(&n, 0, sizeof n);
    n.type = N_PERLIN; n.domain = domain;
    perlin_init(&n.perlin, table256);
    return add_noise(s, n);
}
/* Worley.cpp:3-27; Common.h:17-20 Random<T> */
int sb_noise_worley(sb_scene *s, double domain, int divisions, const double *points) {
    noise_t n; memset(&n, 0, sizeof n);
    n.type = N_WORLEY; n.domain = domain; n.div = divisions;
    n.step = 1.0 / divisions;
    size_t cnt = (size_t)divisions * divisions * divisions;
    n.pts = (v3 *)malloc(sizeof(v3) * cnt);
    size_t i = 0;
    for (int z = 0; z < divisions; z++)
        for (int y = 0; y < divisions; y++)
            for (int x = 0; x < divisions; x++, i++) {
                if (points) {
                    n.pts[i] = V(points[i * 3], points[i * 3 + 1], points[i * 3 + 2]);
                } else {
                    real rx = x, ry = y, rz = z, mn, mx;
                    v3 p;
                    mn = rx * n.step; mx = (rx + 1) * n.step; p.x = (real)(mn + (mx - mn) * ((double)rand() / RAND_MAX));
                    mn = ry * n.step; mx = (ry + 1) * n.step; p.y = (real)(mn + (mx - mn) * ((double)rand() / RAND_MAX));
                    mn = rz * n.step; mx = (rz + 1) * n.step; p.z = (real)(mn + (mx - mn) * ((double)rand() / RAND_MAX));
                    n.pts[i] = p;
                }
            }
    return add_noise(s, n);
}
int sb_noise_marble(sb_scene *s, double domain, double freq, double turb, const uint8_t *table256) {
    noise_t n; memset(&n, 0, sizeof n);
    n.type = N_MARBLE; n.domain = domain; n.freq = freq; n.turb = turb;
    perlin_init(&n.perlin, table256); /* Marble.cpp:3-9 */
    return add_noise(s, n);
}
int sb_node(sb_scene *s, int parent, const double pos[3], const double rot[3], const double scale[3]) {
    if (parent < 0) parent = 0;
    if (parent >= s->nnodes) return -1;
    s->nodes = (node_t *)realloc(s->nodes, sizeof(node_t) * (s->nnodes + 1));
    node_t *n = &s->nodes[s->nnodes];
    memset(n, 0, sizeof *n);
    n->parent = parent;
    n->position = V(pos[0], pos[1], pos[2]);
    n->rotation = V(rot[0], rot[1], rot[2]);
    n->scale = V(scale[0], scale[1], scale[2]);
    node_t *p = &s->nodes[parent];
    p->children = (int *)realloc(p->children, sizeof(int) * (p->nchildren + 1));
    p->children[p->nchildren++] = s->nnodes;
    s->compiled = 0;
    return s->nnodes++;
}
static shape_t *new_shape(sb_scene *s, int node, int type, int mat) {
    if (node < 0 || node >= s->nnodes || s->nodes[node].shape || mat < 0 || mat >= s->nmats) return NULL;
    shape_t *sh = (shape_t *)calloc(1, sizeof *sh);
    sh->type = type; sh->material = mat;
    sh->id = g_next_shape_id;
    g_next_shape_id = (g_next_shape_id + 1 > 1u) ? g_next_shape_id + 1 : 1u;
    sh->plane_normal = V(0, 1, 0);
    s->nodes[node].shape = sh;
    s->compiled = 0;
    return sh;
}
int sb_sphere(sb_scene *s, int node, int mat) { return new_shape(s, node, S_SPHERE, mat) ? 0 : -1; }
int sb_plane(sb_scene *s, int node, int mat, const double n[3]) {
    shape_t *sh = new_shape(s, node, S_PLANE, mat);
    if (!sh) return -1;
    sh->plane_normal = V(n[0], n[1], n[2]);
    return 0;
}
int sb_mesh(sb_scene *s, int node, int mat, int normal_mode, int ntris, const double *verts, const double *normals) {
    shape_t *sh = new_shape(s, node, S_MESH, mat);
    if (!sh) return -1;
    sh->normal_mode = normal_mode; sh->ntris = ntris;
    sh->tris = (tri_t *)calloc(ntris ? ntris : 1, sizeof(tri_t));
    for (int i = 0; i < ntris; i++)
        for (int k = 0; k < 3; k++) {
            sh->tris[i].v[k] = V(verts[i * 9 + k * 3], verts[i * 9 + k * 3 + 1], verts[i * 9 + k * 3 + 2]);
            if (normals) sh->tris[i].n[k] = V(normals[i * 9 + k * 3], normals[i * 9 + k * 3 + 1], normals[i * 9 + k * 3 + 2]);
        }
    return 0;
}
int sb_light(sb_scene *s, int type, unsigned rgb, const double dir[3], const double pos[3], double attenuation, int enabled) {
    s->lights = (light_t *)realloc(s->lights, sizeof(light_t) * (s->nlights + 1));
    light_t *l = &s->lights[s->nlights];
    l->type = type; l->color = c_hex(rgb);
    l->direction = V(dir[0], dir[1], dir[2]);
    l->position = V(pos[0], pos[1], pos[2]);
    l->attenuation = attenuation; l->enabled = enabled;
    return s->nlights++;
}
void sb_bg_color(sb_scene *s, unsigned rgb) { s->bg_type = BG_COLOR; s->bg_color0 = c_hex(rgb); }
void sb_bg_skybox(sb_scene *s, unsigned rgb0, unsigned rgb1, int sun_light, const uint8_t *table256) {
    s->bg_type = BG_SKYBOX; s->bg_color0 = c_hex(rgb0); s->bg_color1 = c_hex(rgb1);
    s->bg_sun = sun_light;
    perlin_init(&s->bg_perlin, table256); /* Scene.cpp:105-109 */
}
static void table_out(const perlin_t *p, uint8_t *out) { for (int i = 0; i < 256; i++) out[i] = (uint8_t)p->seed[i]; }
int sb_get_perlin_table(sb_scene *s, int noise, uint8_t *out256) {
    if (noise < 0 || noise >= s->nnoises) return -1;
    if (s->noises[noise].type != N_PERLIN && s->noises[noise].type != N_MARBLE) return -1;
    table_out(&s->noises[noise].perlin, out256);
    return 0;
}
int sb_get_skybox_table(sb_scene *s, uint8_t *out256) {
    if (s->bg_type != BG_SKYBOX) return -1;
    table_out(&s->bg_perlin, out256);
    return 0;
}
int sb_get_worley_points(sb_scene *s, int noise, double *out) {
    if (noise < 0 || noise >= s->nnoises || s->noises[noise].type != N_WORLEY) return -1;
    const noise_t *n = &s->noises[noise];
    size_t cnt = (size_t)n->div * n->div * n->div;
    for (size_t i = 0; i < cnt; i++) { out[i * 3] = n->pts[i].x; out[i * 3 + 1] = n->pts[i].y; out[i * 3 + 2] = n->pts[i].z; }
    return 0;
}
static void ensure_compiled(sb_scene *s) { if (!s->compiled) scene_compile(s); }

int sb_render(sb_scene *s, const sb_render_params *p, uint32_t *pixels, int32_t *shape_ord, int32_t *tri_idx,
              uint64_t *ray_count, const int32_t *sub_x, const int32_t *sub_y, int n_sub) {
    ensure_compiled(s); /* Raytracer.cpp:42 */
    job_t j; memset(&j, 0, sizeof j);
    j.s = s; j.p = p;
    j.fovy = p->fovy > 0 ? p->fovy : RE_PI / 4;
    size_t npx = n_sub > 0 ? (size_t)n_sub : (size_t)p->width * p->height;
    j.colors = (col *)calloc(npx, sizeof(col));
    j.shape_ord = shape_ord; j.tri_idx = tri_idx;
    j.sub_x = sub_x; j.sub_y = sub_y; j.n_sub = n_sub;
    pthread_mutex_init(&j.mu, NULL);
    int nt = p->num_threads > 0 ? p->num_threads : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
    for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, render_thread, &j);
    for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
    /* Raytracer.cpp:160-170 */
    for (size_t i = 0; i < npx; i++) pixels[i] = c_pack(j.colors[i]);
    if (ray_count) *ray_count = j.rays;
    free(th); free(j.colors);
    pthread_mutex_destroy(&j.mu);
    return 0;
}

int sb_cast_rays(sb_scene *s, int n, const double *o, const double *d, int32_t *out_hit, double *out_point,
                 double *out_local, double *out_normal, int32_t *out_shape, int32_t *out_tri) {
    ensure_compiled(s);
    for (int i = 0; i < n; i++) {
        castres_t r = cast_ray(s, V(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]), V(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]));
        out_hit[i] = r.hit;
        if (out_point) { out_point[i * 3] = r.point.x; out_point[i * 3 + 1] = r.point.y; out_point[i * 3 + 2] = r.point.z; }
        if (out_local) { out_local[i * 3] = r.local.x; out_local[i * 3 + 1] = r.local.y; out_local[i * 3 + 2] = r.local.z; }
        if (out_normal) { out_normal[i * 3] = r.normal.x; out_normal[i * 3 + 1] = r.normal.y; out_normal[i * 3 + 2] = r.normal.z; }
        if (out_shape) out_shape[i] = r.hit ? s->nodes[r.node].shape_ordinal : -1;
        if (out_tri) out_tri[i] = r.hit ? r.tri : -1;
    }
    return 0;
}
int sb_noise_sample(sb_scene *s, int noise, int n, const double *p, double *out) {
    if (noise < 0 || noise >= s->nnoises) return -1;
    for (int i = 0; i < n; i++) out[i] = noise_sample(&s->noises[noise], V(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]));
    return 0;
}
int sb_material_eval(sb_scene *s, int mat, int n, const double *p, double *out_color, double *out_absorptance) {
    if (mat < 0 || mat >= s->nmats) return -1;
    for (int i = 0; i < n; i++) {
        v3 q = V(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
        col c = material_color(s, mat, q);
        out_color[i * 3] = c.r; out_color[i * 3 + 1] = c.g; out_color[i * 3 + 2] = c.b;
        out_absorptance[i] = material_absorptance(s, mat, q);
    }
    return 0;
}
int sb_background_eval(sb_scene *s, int n, const double *d, double *out_color) {
    for (int i = 0; i < n; i++) {
        col c = s->bg_type != BG_NONE ? background_color(s, V(d[i * 3], d[i * 3 + 1], d[i * 3 + 2])) : c_raw(0, 0, 0);
        out_color[i * 3] = c.r; out_color[i * 3 + 1] = c.g; out_color[i * 3 + 2] = c.b;
    }
    return 0;
}
int sb_raycast(sb_scene *s, int max_recursion, int n, const double *o, const double *d, double *out_color) {
    ensure_compiled(s);
    for (int i = 0; i < n; i++) {
        col c = recursive_raytrace(s, V(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]), V(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]), 0, (unsigned)max_recursion);
        out_color[i * 3] = c.r; out_color[i * 3 + 1] = c.g; out_color[i * 3 + 2] = c.b;
    }
    return 0;
}
int sb_screen_rays(sb_scene *s, int width, int height, double fovy, int n, const double *xy, double *out_o, double *out_d) {
    real f = fovy > 0 ? fovy : RE_PI / 4;
    for (int i = 0; i < n; i++) {
        v3 o, d;
        create_screen_ray(s, (size_t)width, (size_t)height, f, xy[i * 2], xy[i * 2 + 1], &o, &d);
        out_o[i * 3] = o.x; out_o[i * 3 + 1] = o.y; out_o[i * 3 + 2] = o.z;
        out_d[i * 3] = d.x; out_d[i * 3 + 1] = d.y; out_d[i * 3 + 2] = d.z;
    }
    return 0;
}
static void kd_dump_rec(const kd_t *k, int *ni, int64_t *nr, int32_t *depth, int32_t *ntris, double *split,
                        double *bounds, int32_t *refs) {
    int i = (*ni)++;
    if (depth) depth[i] = (int32_t)k->depth;
    if (ntris) ntris[i] = k->ntris;
    if (split) {
        /* the split plane is recoverable from the left child's Max (or right child's Min) on axis depth%3 */
        unsigned ax = k->depth % 3;
        split[i] = k->left ? v_get(k->left->bounds.mx, ax) : (k->right ? v_get(k->right->bounds.mn, ax) : 0.0);
    }
    if (bounds) {
        bounds[i * 6 + 0] = k->bounds.mn.x; bounds[i * 6 + 1] = k->bounds.mn.y; bounds[i * 6 + 2] = k->bounds.mn.z;
        bounds[i * 6 + 3] = k->bounds.mx.x; bounds[i * 6 + 4] = k->bounds.mx.y; bounds[i * 6 + 5] = k->bounds.mx.z;
    }
    for (int t = 0; t < k->ntris; t++) {
        if (refs) refs[*nr] = k->tris[t];
        (*nr)++;
    }
    if (k->left) kd_dump_rec(k->left, ni, nr, depth, ntris, split, bounds, refs);
    if (k->right) kd_dump_rec(k->right, ni, nr, depth, ntris, split, bounds, refs);
}
int sb_kd_dump(sb_scene *s, int node, int32_t *node_depth, int32_t *node_ntris, double *node_split,
               double *node_bounds, int32_t *tri_refs, int64_t *n_refs) {
    ensure_compiled(s);
    if (node < 0 || node >= s->nnodes || !s->nodes[node].shape || s->nodes[node].shape->type != S_MESH) return -1;
    int ni = 0; int64_t nr = 0;
    kd_dump_rec(s->nodes[node].shape->kd, &ni, &nr, node_depth, node_ntris, node_split, node_bounds, tri_refs);
    if (n_refs) *n_refs = nr;
    return ni;
}
int sb_node_matrices(sb_scene *s, int node, double *out48) {
    ensure_compiled(s);
    if (node < 0 || node >= s->nnodes) return -1;
    memcpy(out48, s->nodes[node].transform.m, sizeof(double) * 16);
    memcpy(out48 + 16, s->nodes[node].inverse.m, sizeof(double) * 16);
    memcpy(out48 + 32, s->nodes[node].normal.m, sizeof(double) * 16);
    return 0;
}
