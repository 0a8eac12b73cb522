// tests/hostcore/hostcore.cpp — TEST ONLY.  Compiles the device core (raytracing_amd/csrc/rt_core.hpp,
// the exact source the HIP kernels inline) with g++ so its logic can be checked against the oracle
// in a container that has no GPU.  Never part of the product path: the product runs rt_core.hpp
// only inside the kernels of rthip.hip.
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "rt_core.hpp"

using namespace rtc;

static int g_accel = 0;
extern "C" void hc_set_accel(int a) { g_accel = a; }
static std::vector<rthip_shape> g_shapes;  // with the round-trip bounds rthip_upload_scene would fill in
static SceneView view_of(const rthip_scene_desc *s) {
    SceneView v;
    g_shapes.assign(s->shapes, s->shapes + s->n_shapes);
    std::vector<rthip_bvhnode> roots(s->n_meshes);
    for (uint32_t i = 0; i < s->n_meshes; i++)
        if (s->meshes[i].bvh_node_count) roots[i] = s->bvh_nodes[s->meshes[i].bvh_node_base];
    for (rthip_shape &sh : g_shapes) roundtrip_bounds(sh, s->meshes, roots.data());
    v.shapes = g_shapes.data(); v.n_shapes = s->n_shapes; v.meshes = s->meshes;
    v.kd_nodes = s->kd_nodes; v.kd_bounds = s->kd_bounds; v.kd_refs = s->kd_refs; v.kd_planes = s->kd_planes;
    v.tris = s->tris; v.tri_normals = s->tri_normals; v.bvh_nodes = s->bvh_nodes; v.bvh_tris = s->bvh_tris;
    v.bvh_tri_index = s->bvh_tri_index; v.accel = g_accel; v.materials = s->materials; v.noises = s->noises;
    v.perlin_tables = s->perlin_tables; v.worley_points = s->worley_points;
    v.lights = s->lights; v.n_lights = s->n_lights; v.bg = s->background;
    return v;
}

extern "C" {
// rtc::fmodf_exact against the C library's fmodf, bit for bit, over n float patterns starting at bits `lo` with stride `step`
uint64_t hc_fmodf_mismatches(float d, uint64_t lo, uint64_t n, uint64_t step) {
    uint64_t bad = 0;
    for (uint64_t k = 0; k < n; k++) {
        const uint32_t b = (uint32_t)(lo + k * step);
        float x;
        memcpy(&x, &b, 4);
        const float a = rtc::fmodf_exact(x, d), r = fmodf(x, d);
        uint32_t ab, rb;
        memcpy(&ab, &a, 4);
        memcpy(&rb, &r, 4);
        bad += (ab != rb) && !(a != a && r != r);
    }
    return bad;
}

// rtc::fast_div (the slot -> pixel mapping's divisions) against the hardware division: mismatches over n = lo, lo + step, ... < hi
uint64_t hc_fastdiv_mismatches(uint32_t d, uint64_t lo, uint64_t hi, uint64_t step) {
    const rtc::FastDiv f = rtc::make_fastdiv(d);
    uint64_t bad = 0;
    for (uint64_t n = lo; n < hi && n <= 0xffffffffull; n += step ? step : 1)
        bad += rtc::fast_div((uint32_t)n, f) != (uint32_t)n / d;
    return bad;
}


// frame with the same per-pixel semantics as k_bounce/k_fold/k_resolve
int hc_render(const rthip_scene_desc *scene, const rthip_camera *cam, const rthip_params *p, uint32_t *pixels,
              int32_t *shape, int32_t *tri, uint64_t *rays, uint64_t *kd_counters /*3, nullable*/, int threads) {
    SceneView sv = view_of(scene);
    std::atomic<uint32_t> next{0};
    std::atomic<uint64_t> total{0}, kn{0}, kl{0}, kt{0};
    auto work = [&]() {
        uint32_t stack_mem[128];
        float stack_t[128];
        KdStack st = make_stack(stack_mem, 1, 128, stack_t);
        uint64_t mine = 0;
        KdCounters kc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (;;) {
            uint32_t y = next.fetch_add(1);
            if (y >= p->height) break;
            for (uint32_t x = 0; x < p->width; x++) {
                V3 o, d;
                Col c;
                if (p->renderer != RTHIP_RENDER_RAYTRACER) {
                    screen_ray(*cam, p->width, p->height, (double)x, (double)y, o, d);
                    c = debug_raycast(sv, o, d, p->renderer - 1, st, nullptr, nullptr);
                    mine++;
                } else if (p->aa_mode) {
                    c = Col{0, 0, 0};
                    for (int dx = -1; dx <= 1; dx++)
                        for (int dy = -1; dy <= 1; dy++) {
                            screen_ray(*cam, p->width, p->height, (double)((float)x + (float)dx * .5f),
                                       (double)((float)(int)y + (float)dy * .5f), o, d);
                            uint32_t n = 0;
                            Col r = raycast(sv, o, d, p->max_recursion, st, &n);
                            mine += n;
                            c = c_add(c, c_scale(r, (double)(1.0f / 9.0f)));
                        }
                } else {
                    screen_ray(*cam, p->width, p->height, (double)x, (double)y, o, d);
                    uint32_t n = 0;
                    c = raycast(sv, o, d, p->max_recursion, st, &n);
                    mine += n;
                }
                size_t idx = (size_t)y * p->width + x;
                pixels[idx] = c_pack(c);
                if (shape || tri || kd_counters) {
                    screen_ray(*cam, p->width, p->height, (double)x, (double)y, o, d);
                    Hit h = cast_ray(sv, o, d, st, false, kd_counters ? &kc : nullptr);
                    if (shape) shape[idx] = h.shape;
                    if (tri) tri[idx] = h.tri;
                }
            }
        }
        total += mine;
        kn += kc.nodes; kl += kc.leaves; kt += kc.tris;
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < (threads > 0 ? threads : 1); i++) pool.emplace_back(work);
    for (auto &t : pool) t.join();
    if (rays) *rays = total;
    if (kd_counters) { kd_counters[0] = kn; kd_counters[1] = kl; kd_counters[2] = kt; }
    return 0;
}

int hc_cast(const rthip_scene_desc *scene, uint32_t n, const double *o, const double *d, int32_t *hit, double *point,
            double *local, double *normal, int32_t *shape, int32_t *tri) {
    SceneView sv = view_of(scene);
    uint32_t stack_mem[128];
    float stack_t[128];
    KdStack st = make_stack(stack_mem, 1, 128, stack_t);
    for (uint32_t i = 0; i < n; i++) {
        Hit h = cast_ray(sv, mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), st, false, nullptr);
        hit[i] = h.shape >= 0; shape[i] = h.shape; tri[i] = h.tri;
        point[3 * i] = h.point.x; point[3 * i + 1] = h.point.y; point[3 * i + 2] = h.point.z;
        local[3 * i] = h.local.x; local[3 * i + 1] = h.local.y; local[3 * i + 2] = h.local.z;
        normal[3 * i] = h.normal.x; normal[3 * i + 1] = h.normal.y; normal[3 * i + 2] = h.normal.z;
    }
    return 0;
}
int hc_noise(const rthip_scene_desc *scene, int32_t noise, uint32_t n, const double *p, double *out) {
    SceneView sv = view_of(scene);
    for (uint32_t i = 0; i < n; i++) out[i] = noise_sample(sv, sv.perlin_tables, noise, mk(p[3 * i], p[3 * i + 1], p[3 * i + 2]));
    return 0;
}
int hc_material(const rthip_scene_desc *scene, int32_t mat, uint32_t n, const double *p, double *color, double *absorptance) {
    SceneView sv = view_of(scene);
    for (uint32_t i = 0; i < n; i++) {
        Col c; double a;
        material_eval(sv, sv.perlin_tables, mat, mk(p[3 * i], p[3 * i + 1], p[3 * i + 2]), c, a);
        color[3 * i] = c.r; color[3 * i + 1] = c.g; color[3 * i + 2] = c.b; absorptance[i] = a;
    }
    return 0;
}
int hc_background(const rthip_scene_desc *scene, uint32_t n, const double *d, double *color) {
    SceneView sv = view_of(scene);
    for (uint32_t i = 0; i < n; i++) {
        Col c = background_color(sv, sv.perlin_tables, mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        color[3 * i] = c.r; color[3 * i + 1] = c.g; color[3 * i + 2] = c.b;
    }
    return 0;
}
}
