// rthip.hip — HIP kernels (gfx950 / CDNA4) + C-ABI of the Raycast hot path.  See include/rthip.h.
//
// Frame pipeline (wavefront style, one ray per lane, 64-lane waves):
//   k_trace level k  : every scene query of the level — closest hit (level 0 generates its rays)
//                      and one shadow query per light; traversal code only, hit records out.
//   k_shade level k  : material / noise / background, Lambert terms, reflected rays compacted
//                      into the level-(k+1) queue with a wave ballot + prefix popcount and ONE
//                      atomic per workgroup.  Level counts are read from device memory, so the
//                      host never synchronises between levels.
//   k_fold level k   : C_{k-1} = direct_{k-1} (+) C_k (x) refl_{k-1}  (clamping Color algebra forbids a
//                      forward throughput product), back to front.
//   k_resolve        : level-0 colours -> 0xAABBGGRR pixels (and the 3x3 SSAA sum).
// kd-tree traversal keeps its per-lane stack of pending subtrees in LDS (lane-interleaved, so
// pushes/pops of a wave are bank-conflict free).  No MFMA: the path is divergent fp64 VALU work.
//
// Built with -ffp-contract=off: the oracle is the non-fused x86-64 build of the reference.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_core.hpp"

using namespace rtc;

// ============================================================================ helpers
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(RTHIP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        if (n == 0) n = 256;
        HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

static const uint32_t kBlock = 256;           // 4 waves per workgroup
// minimum waves per SIMD the register allocator must leave room for (the kernels are
// bound by dependent fp64 chains, so resident waves matter more than a few spilled values)
#ifndef RT_TRACE_OCC
#define RT_TRACE_OCC 4
#endif
#ifndef RT_SHADE_OCC
#define RT_SHADE_OCC 4
#endif
static const uint32_t kFastStackCap = 16;      // entries per lane of the hot kernel's LDS stack (32 KB per workgroup)
static const uint32_t kMaxGridLevel = 4096;   // queue levels: grid-stride over a bounded grid
enum { ST_PRIMARY = 0, ST_SHADOW, ST_REFLECT, ST_KD_NODES, ST_KD_TRIS, ST_KD_LEAVES, ST_KD_POPS, ST_SHAPES, ST_MESH_HITS, ST_KD_MAXRAY, ST_WAVE_STEPS_MAX, ST_WAVE_TRIS_MAX, ST_WAVE_STEPS, ST_WAVE_TRIS, ST_WAVE_CYCLES_MAX, ST_COUNT };

// Per-frame constants
struct FrameParams {
    rthip_camera cam;
    uint32_t W, H, max_rec;
    int32_t aa, renderer;
    uint32_t part_index, part_count, tile_w, tile_h, tiles_x, tiles_y;
    uint32_t q0;           // first per-part tile ordinal of this chunk
    uint32_t n_pix_slots;  // tile slots in this chunk * tile_w * tile_h
    uint32_t N0;           // level-0 rays of this chunk = n_pix_slots * (aa ? 9 : 1)
    uint32_t count_kd;     // collect kd counters
    int32_t layout;        // 0 frame, 1 tile-compact
};
// Per-frame device workspace (struct-of-arrays; N = capacity of one level)
struct Work {
    double *rec;            // [levels][4][N]: colour r,g,b + reflectance of every ray of a level
    uint32_t *parent;       // [levels][N]  : index of the spawning ray in level-1
    double *rays;           // [2][6][N]    : ping-pong ray queues (o.xyz, d.xyz)
    uint32_t *counts;       // [levels+1]   : rays per level
    unsigned long long *stats;  // [ST_COUNT]
    int32_t *id_shape, *id_tri; // [n_pix_slots] primary-hit ids (optional)
    int32_t *hit_shape;         // [N] per ray of the current level: shape ordinal, -1 miss, -2 no ray
    uint32_t *hit_mask;         // [N] shadow mask
    double *hit;                // [6][N] local point, normal
    uint32_t *redo;             // [N] rays of the current level that must be re-traced by the full kernel
    uint32_t *redo_count;       // [levels] one counter per level (zeroed with `counts`)
    uint32_t N;
};

__device__ __forceinline__ double *rec_ptr(const Work &w, uint32_t level, uint32_t comp) {
    return w.rec + ((size_t)level * 4 + comp) * w.N;
}
__device__ __forceinline__ double *hit_ptr(const Work &w, uint32_t comp) { return w.hit + (size_t)comp * w.N; }
__device__ __forceinline__ double *ray_ptr(const Work &w, uint32_t level, uint32_t comp) {
    return w.rays + ((size_t)(level & 1) * 6 + comp) * w.N;
}

// slot (within chunk) -> pixel.  Tiles are tile_w x tile_h; inside a tile the pixels are ordered
// as 8x8 sub-blocks so that one 64-lane wave covers a compact 8x8 patch of the screen.
__device__ __forceinline__ bool slot_to_pixel(const FrameParams &fp, uint32_t pix_slot, uint32_t &x, uint32_t &y) {
    uint32_t per_tile = fp.tile_w * fp.tile_h;
    uint32_t q = fp.q0 + pix_slot / per_tile;
    uint32_t r = pix_slot % per_tile;
    uint32_t t = fp.part_index + q * fp.part_count;
    if (t >= fp.tiles_x * fp.tiles_y) return false;
    uint32_t tx = t % fp.tiles_x, ty = t / fp.tiles_x;
    uint32_t sb = r >> 6, l = r & 63u;
    uint32_t sbx = fp.tile_w >> 3;
    x = tx * fp.tile_w + (sb % sbx) * 8 + (l & 7u);
    y = ty * fp.tile_h + (sb / sbx) * 8 + (l >> 3);
    return x < fp.W && y < fp.H;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ============================================================================ kernels
extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];

// Stage 1 of a level: every scene query of the level.
// Level 0 generates its rays (R1).  Output: one HitRec per ray (SoA) + the ray direction.
// FAST = true is the hot instantiation: BVH only; rays that meet a degenerate (ray, mesh) pair are
// appended to the redo list and re-traced by the FAST = false instantiation (which also serves
// accel = 1, the reference kd-tree everywhere).  REDO = true: `n` and the ray indices come from the
// redo list.
template <bool FAST, bool REDO, bool COUNT>
__global__ __launch_bounds__(kBlock, RT_TRACE_OCC) void k_trace(SceneView sv, FrameParams fp, Work w, uint32_t level,
                                                                 uint32_t stack_cap) {
    // LDS: stack_cap child words per lane, then (FAST only) stack_cap entry parameters per lane
    KdStack st = make_stack(lds_dyn + threadIdx.x, kBlock, stack_cap,
                            FAST ? reinterpret_cast<float *>(lds_dyn + (size_t)stack_cap * kBlock) + threadIdx.x : nullptr);
    const uint32_t n = REDO ? w.redo_count[0] : ((level == 0) ? fp.N0 : w.counts[level]);
    const uint32_t n_shadow = fp.renderer == RTHIP_RENDER_RAYTRACER ? sv.n_lights : 0u;  // DebugRaycaster: closest hit only
    KdCounters kc = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t max_ray_nodes = 0;
    const long long t_start = COUNT ? clock64() : 0;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        const uint32_t j = base + threadIdx.x;
        const bool in_range = j < n;
        const uint32_t i = in_range ? (REDO ? w.redo[j] : j) : 0u;
        const uint32_t nodes_before = kc.nodes;
        V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
        uint32_t pix_slot = 0;
        bool live = in_range;
        if (level == 0) {
            uint32_t s = i / fp.n_pix_slots;  // SSAA sub-sample (0 when aa is off)
            pix_slot = i - s * fp.n_pix_slots;
            uint32_t x = 0, y = 0;
            if (live && !slot_to_pixel(fp, pix_slot, x, y)) {
                w.hit_shape[i] = -2;  // slot outside the image
                live = false;
            }
            double fx = (double)x, fy = (double)y;
            if (fp.aa) {  // Raytracer.cpp:134-138: dx outer, dy inner, offsets +-0.5
                int dx = (int)(s / 3) - 1, dy = (int)(s % 3) - 1;
                fx = (double)((float)x + (float)dx * .5f);
                fy = (double)((float)(int)y + (float)dy * .5f);
            }
            screen_ray(fp.cam, fp.W, fp.H, fx, fy, o, d);
            if (!REDO && live) {
                ray_ptr(w, 0, 3)[i] = d.x;  // the shade stage needs the direction (background, reflection)
                ray_ptr(w, 0, 4)[i] = d.y;
                ray_ptr(w, 0, 5)[i] = d.z;
            }
        } else if (live) {
            o = mk(ray_ptr(w, level, 0)[i], ray_ptr(w, level, 1)[i], ray_ptr(w, level, 2)[i]);
            d = mk(ray_ptr(w, level, 3)[i], ray_ptr(w, level, 4)[i], ray_ptr(w, level, 5)[i]);
        }
        // every lane of the wave walks the (wave-uniform) query and shape loops; dead lanes carry a
        // ray that is never traced (n_shapes = 0 would need a second code path)
        HitRec h;
        st.overflow = false;
        bool done = true;
        if (live) done = trace_stage<FAST, COUNT>(sv, o, d, n_shadow, st, h, COUNT ? &kc : nullptr);
        if (!live) continue;
        if (COUNT && kc.nodes - nodes_before > max_ray_nodes) max_ray_nodes = kc.nodes - nodes_before;
        if (FAST && !done) {  // rare: hand the ray to the full instantiation
            w.redo[atomicAdd(&w.redo_count[0], 1u)] = i;
            continue;
        }
        w.hit_shape[i] = h.shape;
        if (h.shape >= 0) {
            w.hit_mask[i] = h.shadow_mask;
            hit_ptr(w, 0)[i] = h.local.x;
            hit_ptr(w, 1)[i] = h.local.y;
            hit_ptr(w, 2)[i] = h.local.z;
            hit_ptr(w, 3)[i] = h.normal.x;
            hit_ptr(w, 4)[i] = h.normal.y;
            hit_ptr(w, 5)[i] = h.normal.z;
        }
        // primary-hit ids come from the un-offset pixel-centre ray (sub-sample 4 of the 3x3 SSAA grid)
        if (level == 0 && w.id_shape && i / fp.n_pix_slots == (fp.aa ? 4u : 0u)) {
            w.id_shape[pix_slot] = h.shape;
            w.id_tri[pix_slot] = h.tri;
        }
    }
    if (COUNT) {
        const uint32_t lane = threadIdx.x & 63u;
        kc.nodes = wave_sum(kc.nodes);
        kc.tris = wave_sum(kc.tris);
        kc.leaves = wave_sum(kc.leaves);
        kc.pops = wave_sum(kc.pops);
        kc.shapes = wave_sum(kc.shapes);
        kc.mesh_hits = wave_sum(kc.mesh_hits);
        kc.wave_steps = wave_sum(kc.wave_steps);
        kc.wave_tris = wave_sum(kc.wave_tris);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            uint32_t o2 = __shfl_xor(max_ray_nodes, off, 64);
            max_ray_nodes = o2 > max_ray_nodes ? o2 : max_ray_nodes;
        }
        const unsigned long long cycles = (unsigned long long)(clock64() - t_start);
        if (lane == 0 && fp.count_kd == 2) {  // light mode: only the longest wave (and its step counts), almost no atomics
            if (cycles > w.stats[ST_WAVE_CYCLES_MAX] && atomicMax(&w.stats[ST_WAVE_CYCLES_MAX], cycles) < cycles) {
                w.stats[ST_WAVE_STEPS_MAX] = kc.wave_steps;  // racy by design: diagnostics of "a" longest wave
                w.stats[ST_WAVE_TRIS_MAX] = kc.wave_tris;
                w.stats[ST_KD_MAXRAY] = max_ray_nodes;
            }
        } else if (lane == 0 && (kc.nodes | kc.shapes)) {
            atomicMax(&w.stats[ST_KD_MAXRAY], (unsigned long long)max_ray_nodes);
            atomicMax(&w.stats[ST_WAVE_STEPS_MAX], (unsigned long long)kc.wave_steps);
            atomicMax(&w.stats[ST_WAVE_CYCLES_MAX], cycles);
            atomicMax(&w.stats[ST_WAVE_TRIS_MAX], (unsigned long long)kc.wave_tris);
            atomicAdd(&w.stats[ST_WAVE_STEPS], (unsigned long long)kc.wave_steps);
            atomicAdd(&w.stats[ST_WAVE_TRIS], (unsigned long long)kc.wave_tris);
            atomicAdd(&w.stats[ST_KD_NODES], (unsigned long long)kc.nodes);
            atomicAdd(&w.stats[ST_KD_TRIS], (unsigned long long)kc.tris);
            atomicAdd(&w.stats[ST_KD_LEAVES], (unsigned long long)kc.leaves);
            atomicAdd(&w.stats[ST_KD_POPS], (unsigned long long)kc.pops);
            atomicAdd(&w.stats[ST_SHAPES], (unsigned long long)kc.shapes);
            atomicAdd(&w.stats[ST_MESH_HITS], (unsigned long long)kc.mesh_hits);
        }
    }
}

// Stage 2 of a level: material / noise / background, Lambert terms with the shadow mask, and the
// reflected rays, compacted into the next level's queue: wave ballot + prefix popcount, the four
// waves of a workgroup combined through LDS so that ONE returning atomic per workgroup reserves the
// queue range (32 K per-wave atomics on one word cost more than the shading itself).
// The Perlin tables (256 B each) are staged in LDS: their chained lookups are the latency chain of
// every sky / marble sample.
__global__ __launch_bounds__(kBlock, RT_SHADE_OCC) void k_shade(SceneView sv, FrameParams fp, Work w, uint32_t level, uint32_t n_tables) {
    uint8_t *tabs = reinterpret_cast<uint8_t *>(lds_dyn);
    __shared__ uint32_t wave_cnt[kBlock / 64];
    __shared__ uint32_t block_base;
    {
        const uint32_t words = n_tables * 64;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(sv.perlin_tables);
        for (uint32_t k = threadIdx.x; k < words; k += kBlock) lds_dyn[k] = src[k];
        __syncthreads();
    }
    const uint32_t n = (level == 0) ? fp.N0 : w.counts[level];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t shadow_total = 0, spawn_total = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        uint32_t i = base + threadIdx.x;
        ShadeOut so;
        so.spawn = false;
        so.shadow_rays = 0;
        int32_t shape = i < n ? w.hit_shape[i] : -2;
        if (shape > -2) {
            HitRec h;
            h.shape = shape;
            h.tri = -1;
            h.shadow_mask = 0;
            h.local = h.normal = mk(0, 0, 0);
            V3 d = mk(ray_ptr(w, level, 3)[i], ray_ptr(w, level, 4)[i], ray_ptr(w, level, 5)[i]);
            if (shape >= 0) {
                h.shadow_mask = w.hit_mask[i];
                h.local = mk(hit_ptr(w, 0)[i], hit_ptr(w, 1)[i], hit_ptr(w, 2)[i]);
                h.normal = mk(hit_ptr(w, 3)[i], hit_ptr(w, 4)[i], hit_ptr(w, 5)[i]);
            }
            if (fp.renderer == RTHIP_RENDER_RAYTRACER) {
                shade_stage(sv, tabs, h, d, level < fp.max_rec, so);
            } else {
                so.color = debug_shade(sv, tabs, h, d, fp.renderer - 1);
                so.refl = 0.0;
            }
            rec_ptr(w, level, 0)[i] = so.color.r;
            rec_ptr(w, level, 1)[i] = so.color.g;
            rec_ptr(w, level, 2)[i] = so.color.b;
            rec_ptr(w, level, 3)[i] = so.refl;
        }
        // compaction of the reflected rays
        unsigned long long mask = __ballot(so.spawn);
        if (lane == 0) wave_cnt[wid] = (uint32_t)__popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            block_base = total ? atomicAdd(&w.counts[level + 1], total) : 0u;
        }
        __syncthreads();
        if (so.spawn) {
            uint32_t j = block_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            for (uint32_t k = 0; k < wid; k++) j += wave_cnt[k];
            ray_ptr(w, level + 1, 0)[j] = so.child_o.x;
            ray_ptr(w, level + 1, 1)[j] = so.child_o.y;
            ray_ptr(w, level + 1, 2)[j] = so.child_o.z;
            ray_ptr(w, level + 1, 3)[j] = so.child_d.x;
            ray_ptr(w, level + 1, 4)[j] = so.child_d.y;
            ray_ptr(w, level + 1, 5)[j] = so.child_d.z;
            w.parent[(size_t)(level + 1) * w.N + j] = i;
            spawn_total++;
        }
        shadow_total += so.shadow_rays;
        __syncthreads();  // wave_cnt / block_base are reused by the next chunk
    }
    shadow_total = wave_sum(shadow_total);
    spawn_total = wave_sum(spawn_total);
    if (lane == 0) {
        if (shadow_total) atomicAdd(&w.stats[ST_SHADOW], (unsigned long long)shadow_total);
        if (spawn_total) atomicAdd(&w.stats[ST_REFLECT], (unsigned long long)spawn_total);
    }
}

// level k -> k-1:  parent.colour = directLighting (+) (child (x) reflectance)   (Raytracer.cpp:283-286)
__global__ __launch_bounds__(kBlock) void k_fold(Work w, uint32_t level) {
    const uint32_t n = w.counts[level];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        uint32_t p = w.parent[(size_t)level * w.N + i];
        Col child = Col{rec_ptr(w, level, 0)[i], rec_ptr(w, level, 1)[i], rec_ptr(w, level, 2)[i]};
        Col direct = Col{rec_ptr(w, level - 1, 0)[p], rec_ptr(w, level - 1, 1)[p], rec_ptr(w, level - 1, 2)[p]};
        Col c = fold(direct, child, rec_ptr(w, level - 1, 3)[p]);
        rec_ptr(w, level - 1, 0)[p] = c.r;
        rec_ptr(w, level - 1, 1)[p] = c.g;
        rec_ptr(w, level - 1, 2)[p] = c.b;
    }
}

// level-0 colours -> pixels (Common.cpp:275-282); SSAA: result += c * (1.0f/9.0f), clamped (Raytracer.cpp:139)
__global__ __launch_bounds__(kBlock) void k_resolve(FrameParams fp, Work w, uint32_t *pixels, int32_t *out_shape,
                                                    int32_t *out_tri) {
    uint32_t ps = blockIdx.x * kBlock + threadIdx.x;
    if (ps >= fp.n_pix_slots) return;
    uint32_t x, y;
    if (!slot_to_pixel(fp, ps, x, y)) return;
    Col c;
    if (fp.aa) {
        c = Col{0, 0, 0};
        const double k = (double)(1.0f / 9.0f);
        for (uint32_t s = 0; s < 9; s++) {
            size_t i = (size_t)s * fp.n_pix_slots + ps;
            Col r = Col{rec_ptr(w, 0, 0)[i], rec_ptr(w, 0, 1)[i], rec_ptr(w, 0, 2)[i]};
            c = c_add(c, c_scale(r, k));
        }
    } else {
        c = Col{rec_ptr(w, 0, 0)[ps], rec_ptr(w, 0, 1)[ps], rec_ptr(w, 0, 2)[ps]};
    }
    size_t idx = fp.layout ? (size_t)fp.q0 * fp.tile_w * fp.tile_h + ps : (size_t)y * fp.W + x;
    pixels[idx] = c_pack(c);
    if (out_shape) out_shape[idx] = w.id_shape[ps];
    if (out_tri) out_tri[idx] = w.id_tri[ps];
}

// multi-GPU: part_count tile-compact buffers (part after part) -> row-major frame
__global__ __launch_bounds__(kBlock) void k_assemble(FrameParams fp, const uint32_t *gathered, uint32_t part_slots,
                                                     uint32_t *frame) {
    uint32_t ps = blockIdx.x * kBlock + threadIdx.x;
    uint32_t part = blockIdx.y;
    if (ps >= part_slots) return;
    fp.part_index = part;
    fp.q0 = 0;
    uint32_t x, y;
    if (!slot_to_pixel(fp, ps, x, y)) return;
    frame[(size_t)y * fp.W + x] = gathered[(size_t)part * part_slots + ps];
}

// ---- probe kernels (KAT parity of single device functions) ----
__global__ __launch_bounds__(kBlock) void k_probe_cast(SceneView sv, uint32_t n, const double *o, const double *d,
                                                       int32_t *hit, double *point, double *local, double *normal,
                                                       int32_t *shape, int32_t *tri, uint32_t stack_cap) {
    KdStack st = make_stack(lds_dyn + threadIdx.x, kBlock, stack_cap, nullptr);
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Hit h = cast_ray(sv, mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), st, false, nullptr);
    hit[i] = h.shape >= 0;
    shape[i] = h.shape;
    tri[i] = h.tri;
    point[3 * i] = h.point.x; point[3 * i + 1] = h.point.y; point[3 * i + 2] = h.point.z;
    local[3 * i] = h.local.x; local[3 * i + 1] = h.local.y; local[3 * i + 2] = h.local.z;
    normal[3 * i] = h.normal.x; normal[3 * i + 1] = h.normal.y; normal[3 * i + 2] = h.normal.z;
}
__global__ __launch_bounds__(kBlock) void k_probe_noise(SceneView sv, int32_t noise, uint32_t n, const double *p, double *out) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    out[i] = noise_sample(sv, sv.perlin_tables, noise, mk(p[3 * i], p[3 * i + 1], p[3 * i + 2]));
}
__global__ __launch_bounds__(kBlock) void k_probe_material(SceneView sv, int32_t mat, uint32_t n, const double *p,
                                                           double *color, double *absorptance) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Col c;
    double a;
    material_eval(sv, sv.perlin_tables, mat, mk(p[3 * i], p[3 * i + 1], p[3 * i + 2]), c, a);
    color[3 * i] = c.r; color[3 * i + 1] = c.g; color[3 * i + 2] = c.b;
    absorptance[i] = a;
}
__global__ __launch_bounds__(kBlock) void k_probe_background(SceneView sv, uint32_t n, const double *d, double *color) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Col c = background_color(sv, sv.perlin_tables, mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
    color[3 * i] = c.r; color[3 * i + 1] = c.g; color[3 * i + 2] = c.b;
}
__global__ __launch_bounds__(kBlock) void k_probe_raycast(SceneView sv, uint32_t max_rec, uint32_t n, const double *o,
                                                          const double *d, double *color, uint32_t stack_cap) {
    KdStack st = make_stack(lds_dyn + threadIdx.x, kBlock, stack_cap, nullptr);
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Col c = raycast(sv, mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), max_rec, st, nullptr);
    color[3 * i] = c.r; color[3 * i + 1] = c.g; color[3 * i + 2] = c.b;
}
__global__ __launch_bounds__(kBlock) void k_probe_arith(int32_t op, uint32_t n, const double *a, const double *b, double *out) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, r = 0.0;
    switch (op) {
    case 0: r = x + y; break;
    case 1: r = x * y; break;
    case 2: r = x / y; break;
    case 3: r = sqrt(x); break;
    case 4: r = floor(x); break;
    case 5: r = (double)fmodf((float)x, (float)y); break;
    case 6: r = (double)(float)x; break;
    case 7: r = sin(x); break;
    case 8: r = (double)(int)x; break;
    case 9: r = fmax(x, y); break;
    case 10: r = fmin(x, y); break;
    }
    out[i] = r;
}

// ============================================================================ context
struct rthip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevBuf bvh_nodes, bvh_tris, bvh_index, kd_planes;
    uint32_t stack_cap_kd = 1, stack_cap_bvh = 1, stack_cap_full = 1;
    DevBuf shapes, meshes, kd_nodes, kd_bounds, kd_refs, tris, tri_normals, materials, noises, perlin, worley, lights;
    SceneView sv;
    bool has_scene = false;
    uint32_t stack_cap = 1;
    // frame workspace
    DevBuf rec, parent, rays, counts, stats, ids, pixels_tmp, ids_out, hits, hit_ids, redo, probe[8];
    uint32_t n_tables = 0;
    bool has_meshes = false;
    std::atomic<int> interrupt{0};
    std::atomic<float> progress{0.f};
    rthip_stats last{};
    int count_kd = 0;
    int profile_launches = 0;
    std::vector<hipEvent_t> ev_pool;
};

static int set_device(rthip_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

extern "C" {

const char *rthip_version(void) { return "raytracing_amd rthip 0.1 (gfx950, fp64, -ffp-contract=off)"; }
const char *rthip_last_error(void) { return g_err.c_str(); }

int rthip_create(int device, rthip_ctx **out) {
    if (!out) return fail(RTHIP_E_INVALID, "rthip_create: out is null");
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(RTHIP_E_INVALID, "rthip_create: device %d not in [0,%d)", device, ndev);
    rthip_ctx *c = new rthip_ctx;
    c->device = device;
    if (set_device(c)) { delete c; return RTHIP_E_HIP; }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        delete c;
        return fail(RTHIP_E_HIP, "rthip_create: %s", hipGetErrorString(e));
    }
    const char *env = getenv("RTHIP_COUNTERS");
    c->count_kd = env ? atoi(env) : 0;
    env = getenv("RTHIP_PROFILE_LAUNCHES");
    c->profile_launches = env && atoi(env) != 0;
    memset(&c->sv, 0, sizeof c->sv);
    *out = c;
    return 0;
}

int rthip_destroy(rthip_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *all[] = {&c->shapes, &c->meshes, &c->kd_nodes, &c->kd_bounds, &c->kd_refs, &c->tris, &c->tri_normals,
                     &c->materials, &c->noises, &c->perlin, &c->worley, &c->lights, &c->rec, &c->parent, &c->rays,
                     &c->counts, &c->stats, &c->ids, &c->pixels_tmp, &c->ids_out, &c->bvh_nodes, &c->bvh_tris, &c->bvh_index, &c->kd_planes, &c->hits, &c->hit_ids, &c->redo};
    for (DevBuf *b : all) b->release();
    for (DevBuf &b : c->probe) b.release();
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

void *rthip_stream(rthip_ctx *c) { return c ? (void *)c->stream : nullptr; }
int rthip_synchronize(rthip_ctx *c) {
    if (!c) return fail(RTHIP_E_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
float rthip_progress(rthip_ctx *c) { return c ? c->progress.load() : 0.f; }
int rthip_interrupt(rthip_ctx *c) {
    if (!c) return fail(RTHIP_E_INVALID, "null context");
    c->interrupt.store(1);
    return 0;
}
int rthip_get_stats(rthip_ctx *c, rthip_stats *out) {
    if (!c || !out) return fail(RTHIP_E_INVALID, "null argument");
    *out = c->last;
    return 0;
}

// ---------------------------------------------------------------------------- scene upload
static int validate_scene(const rthip_scene_desc *s, uint32_t *max_depth, uint32_t *max_bvh_depth) {
#define BAD(...) return fail(RTHIP_E_INVALID, "rthip_upload_scene: " __VA_ARGS__)
    if (s->n_shapes && !s->shapes) BAD("shapes is null");
    if (s->n_lights && !s->lights) BAD("lights is null");
    if (s->n_materials && !s->materials) BAD("materials is null");
    if (s->n_noises && !s->noises) BAD("noises is null");
    if (s->n_meshes && (!s->meshes || !s->kd_nodes || !s->kd_bounds || !s->tris || !s->tri_normals)) BAD("mesh arrays are null");
    if (s->n_meshes && s->n_tris && (!s->bvh_nodes || !s->bvh_tris || !s->bvh_tri_index)) BAD("bvh arrays are null");
    if (s->n_kd_refs && !s->kd_refs) BAD("kd_refs is null");
    if (s->n_perlin_tables && !s->perlin_tables) BAD("perlin_tables is null");
    if (s->n_worley_points && !s->worley_points) BAD("worley_points is null");
    *max_depth = 0;
    *max_bvh_depth = 0;
    for (uint32_t i = 0; i < s->n_noises; i++) {
        const rthip_noise &n = s->noises[i];
        if (n.type < RTHIP_NOISE_CHECKER || n.type > RTHIP_NOISE_MARBLE) BAD("noise %u: bad type %d", i, n.type);
        if (n.type == RTHIP_NOISE_PERLIN || n.type == RTHIP_NOISE_MARBLE) {
            if (n.table < 0 || (uint32_t)n.table >= s->n_perlin_tables) BAD("noise %u: perlin table %d out of range", i, n.table);
            if (n.octaves < 1 || n.octaves > 9) return fail(RTHIP_E_UNSUPPORTED, "noise %u: octaves %u outside 1..9", i, n.octaves);
        }
        if (n.type == RTHIP_NOISE_WORLEY) {
            if (n.divisions < 1 || n.divisions > 1024) BAD("noise %u: divisions %d", i, n.divisions);
            uint64_t need = (uint64_t)n.points_base + (uint64_t)n.divisions * n.divisions * n.divisions;
            if (n.points_base < 0 || need > s->n_worley_points) BAD("noise %u: worley points out of range", i);
        }
    }
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const rthip_matop &m = s->materials[i];
        if (m.op != RTHIP_MAT_UNIFORM && m.op != RTHIP_MAT_INTERP) BAD("material op %u: bad op", i);
        if (m.first_op < 0 || (uint32_t)m.first_op > i) BAD("material op %u: first_op %d", i, m.first_op);
        if (m.op == RTHIP_MAT_INTERP && (m.noise < 0 || (uint32_t)m.noise >= s->n_noises)) BAD("material op %u: noise %d", i, m.noise);
        int sp = 0, maxsp = 0;
        for (int32_t k = m.first_op; k <= (int32_t)i; k++) {
            if (s->materials[k].op == RTHIP_MAT_UNIFORM) sp++;
            else { if (sp < 2) BAD("material op %u: malformed postfix program", i); sp--; }
            if (sp > maxsp) maxsp = sp;
        }
        if (sp != 1) BAD("material op %u: postfix program leaves %d values", i, sp);
        if (maxsp > RTHIP_MAX_MAT_STACK) return fail(RTHIP_E_UNSUPPORTED, "material op %u: nesting deeper than %d", i, RTHIP_MAX_MAT_STACK);
    }
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        const rthip_mesh &m = s->meshes[i];
        if ((uint64_t)m.node_base + m.node_count > s->n_kd_nodes) BAD("mesh %u: kd nodes out of range", i);
        if ((uint64_t)m.ref_base + m.ref_count > s->n_kd_refs) BAD("mesh %u: kd refs out of range", i);
        if ((uint64_t)m.tri_base + m.tri_count > s->n_tris) BAD("mesh %u: triangles out of range", i);
        if (m.max_depth > RTHIP_KD_MAX_DEPTH) BAD("mesh %u: max_depth %u", i, m.max_depth);
        if (m.max_depth > *max_depth) *max_depth = m.max_depth;
        for (uint32_t k = 0; k < m.node_count; k++) {
            const rthip_kdnode &nd = s->kd_nodes[m.node_base + k];
            if (nd.info & RTHIP_KD_LEAF) {
                uint64_t cnt = nd.info >> RTHIP_KD_COUNT_SHIFT;
                if ((uint64_t)nd.link + cnt > m.ref_count) BAD("mesh %u node %u: leaf refs out of range", i, k);
            } else {
                if ((nd.info & RTHIP_KD_AXIS_MASK) > 2) BAD("mesh %u node %u: axis", i, k);
                if ((nd.info & RTHIP_KD_HAS_LEFT) && k + 1 >= m.node_count) BAD("mesh %u node %u: left child out of range", i, k);
                if ((nd.info & RTHIP_KD_HAS_RIGHT) && (nd.link >= m.node_count || nd.link <= k)) BAD("mesh %u node %u: right child out of range", i, k);
            }
        }
        for (uint32_t k = 0; k < m.ref_count; k++)
            if (s->kd_refs[m.ref_base + k] >= m.tri_count) BAD("mesh %u: ref %u -> triangle out of range", i, k);
        for (int a = 0; a < 3; a++)
            if ((uint64_t)m.plane_base[a] + m.plane_count[a] > s->n_kd_planes || (m.plane_count[a] && !s->kd_planes)) BAD("mesh %u: kd planes out of range", i);
        if ((uint64_t)m.bvh_node_base + m.bvh_node_count > s->n_bvh_nodes) BAD("mesh %u: bvh nodes out of range", i);
        if (m.bvh_depth > RTHIP_BVH_MAX_DEPTH) BAD("mesh %u: bvh_depth %u", i, m.bvh_depth);
        if (m.bvh_node_count && (uint64_t)m.bvh_tri_base + m.tri_count > s->n_tris) BAD("mesh %u: bvh triangles out of range", i);
        if (m.bvh_depth > *max_bvh_depth) *max_bvh_depth = m.bvh_depth;
        if (!(m.abs_max >= 0.0f)) BAD("mesh %u: abs_max", i);
        for (uint32_t k = 0; k < m.bvh_node_count; k++) {
            const rthip_bvhnode &nd = s->bvh_nodes[m.bvh_node_base + k];
            const uint32_t ch[2] = {nd.child0, nd.child1};
            for (int c = 0; c < 2; c++) {
                if (ch[c] & RTHIP_BVH_LEAF) {
                    bool absent = c == 1 ? !(nd.lo1[0] <= nd.hi1[0]) : !(nd.lo0[0] <= nd.hi0[0]);
                    uint64_t first = ch[c] & RTHIP_BVH_FIRST_MASK, cnt = ((ch[c] >> RTHIP_BVH_COUNT_SHIFT) & 7u) + 1u;
                    if (!absent && first + cnt > m.tri_count) BAD("mesh %u bvh node %u: leaf out of range", i, k);
                } else if (ch[c] >= m.bvh_node_count || ch[c] <= k) {
                    BAD("mesh %u bvh node %u: child out of range", i, k);
                }
            }
        }
        for (uint32_t k = 0; k < (m.bvh_node_count ? m.tri_count : 0); k++)
            if (s->bvh_tri_index[m.bvh_tri_base + k] >= m.tri_count) BAD("mesh %u: bvh triangle index out of range", i);
    }
    for (uint32_t i = 0; i < s->n_shapes; i++) {
        const rthip_shape &sh = s->shapes[i];
        if (sh.type < RTHIP_SHAPE_SPHERE || sh.type > RTHIP_SHAPE_MESH) return fail(RTHIP_E_UNSUPPORTED, "shape %u: unknown type %d", i, sh.type);
        if (sh.material < 0 || (uint32_t)sh.material >= s->n_materials) BAD("shape %u: material %d out of range", i, sh.material);
        if (sh.type == RTHIP_SHAPE_MESH && (sh.mesh < 0 || (uint32_t)sh.mesh >= s->n_meshes)) BAD("shape %u: mesh %d out of range", i, sh.mesh);
    }
    if (s->n_lights > 32) return fail(RTHIP_E_UNSUPPORTED, "rthip_upload_scene: %u lights (the shadow mask holds 32)", s->n_lights);
    if (s->n_perlin_tables > 128) return fail(RTHIP_E_UNSUPPORTED, "rthip_upload_scene: %u Perlin tables (LDS staging holds 128)", s->n_perlin_tables);
    for (uint32_t i = 0; i < s->n_lights; i++)
        if (s->lights[i].type < RTHIP_LIGHT_DIRECTIONAL || s->lights[i].type > RTHIP_LIGHT_POINT) BAD("light %u: bad type", i);
    const rthip_background &bg = s->background;
    if (bg.type < RTHIP_BG_NONE || bg.type > RTHIP_BG_SKYBOX) BAD("background type %d", bg.type);
    if (bg.type == RTHIP_BG_SKYBOX) {
        if (bg.table < 0 || (uint32_t)bg.table >= s->n_perlin_tables) BAD("skybox perlin table out of range");
        if (bg.sun_light >= (int32_t)s->n_lights) BAD("skybox sun light out of range");
    }
#undef BAD
    return 0;
}

static int upload(rthip_ctx *c, DevBuf &b, const void *src, size_t bytes) {
    if (b.ensure(bytes)) return RTHIP_E_HIP;
    if (bytes) HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

int rthip_upload_scene(rthip_ctx *c, const rthip_scene_desc *s) {
    if (!c || !s) return fail(RTHIP_E_INVALID, "rthip_upload_scene: null argument");
    uint32_t max_depth = 0, max_bvh_depth = 0;
    int rc = validate_scene(s, &max_depth, &max_bvh_depth);
    if (rc) return rc;
    if (set_device(c)) return RTHIP_E_HIP;
    c->has_scene = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<rthip_shape> shapes(s->shapes, s->shapes + s->n_shapes);
    for (rthip_shape &sh : shapes) roundtrip_bounds(sh);
    if ((rc = upload(c, c->shapes, shapes.data(), sizeof(rthip_shape) * s->n_shapes))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));  // `shapes` is a local
    if ((rc = upload(c, c->meshes, s->meshes, sizeof(rthip_mesh) * s->n_meshes))) return rc;
    if ((rc = upload(c, c->kd_nodes, s->kd_nodes, sizeof(rthip_kdnode) * s->n_kd_nodes))) return rc;
    if ((rc = upload(c, c->kd_bounds, s->kd_bounds, sizeof(double) * 6 * s->n_kd_nodes))) return rc;
    if ((rc = upload(c, c->kd_refs, s->kd_refs, sizeof(uint32_t) * s->n_kd_refs))) return rc;
    if ((rc = upload(c, c->tris, s->tris, sizeof(rthip_tri) * s->n_tris))) return rc;
    if ((rc = upload(c, c->tri_normals, s->tri_normals, sizeof(double) * 9 * s->n_tris))) return rc;
    if ((rc = upload(c, c->kd_planes, s->kd_planes, sizeof(double) * s->n_kd_planes))) return rc;
    if ((rc = upload(c, c->bvh_nodes, s->bvh_nodes, sizeof(rthip_bvhnode) * s->n_bvh_nodes))) return rc;
    if ((rc = upload(c, c->bvh_tris, s->bvh_tris, s->n_bvh_nodes ? sizeof(rthip_tri) * s->n_tris : 0))) return rc;
    if ((rc = upload(c, c->bvh_index, s->bvh_tri_index, s->n_bvh_nodes ? sizeof(uint32_t) * s->n_tris : 0))) return rc;
    if ((rc = upload(c, c->materials, s->materials, sizeof(rthip_matop) * s->n_materials))) return rc;
    if ((rc = upload(c, c->noises, s->noises, sizeof(rthip_noise) * s->n_noises))) return rc;
    if ((rc = upload(c, c->perlin, s->perlin_tables, (size_t)256 * s->n_perlin_tables))) return rc;
    if ((rc = upload(c, c->worley, s->worley_points, sizeof(double) * 3 * s->n_worley_points))) return rc;
    if ((rc = upload(c, c->lights, s->lights, sizeof(rthip_light) * s->n_lights))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    SceneView &v = c->sv;
    v.shapes = (const rthip_shape *)c->shapes.p;
    v.n_shapes = s->n_shapes;
    v.meshes = (const rthip_mesh *)c->meshes.p;
    v.kd_nodes = (const rthip_kdnode *)c->kd_nodes.p;
    v.kd_bounds = (const double *)c->kd_bounds.p;
    v.kd_refs = (const uint32_t *)c->kd_refs.p;
    v.tris = (const rthip_tri *)c->tris.p;
    v.tri_normals = (const double *)c->tri_normals.p;
    v.kd_planes = (const double *)c->kd_planes.p;
    v.bvh_nodes = (const rthip_bvhnode *)c->bvh_nodes.p;
    v.bvh_tris = (const rthip_tri *)c->bvh_tris.p;
    v.bvh_tri_index = (const uint32_t *)c->bvh_index.p;
    v.accel = 0;
    v.materials = (const rthip_matop *)c->materials.p;
    v.noises = (const rthip_noise *)c->noises.p;
    v.perlin_tables = (const uint8_t *)c->perlin.p;
    v.worley_points = (const double *)c->worley.p;
    v.lights = (const rthip_light *)c->lights.p;
    v.n_lights = s->n_lights;
    c->n_tables = s->n_perlin_tables;
    c->has_meshes = s->n_meshes > 0;
    v.bg = s->background;
    c->stack_cap_kd = max_depth + 2;
    c->stack_cap_bvh = max_bvh_depth + 2;
    c->stack_cap_full = (max_depth > max_bvh_depth ? max_depth : max_bvh_depth) + 2;  // degenerate rays walk the kd-tree
    c->stack_cap = c->stack_cap_full;
    c->has_scene = true;
    return 0;
}

// ---------------------------------------------------------------------------- frame
static const size_t kMaxLevelRays = (size_t)1 << 24;  // level-0 rays per chunk (bounds the workspace)

static int check_params(const rthip_camera *cam, const rthip_params *p) {
    if (!cam || !p) return fail(RTHIP_E_INVALID, "rthip_render: null argument");
    if (p->width == 0 || p->height == 0 || p->width > 65536 || p->height > 65536) return fail(RTHIP_E_INVALID, "rthip_render: bad size %ux%u", p->width, p->height);
    if (p->max_recursion > RTHIP_MAX_RECURSION) return fail(RTHIP_E_UNSUPPORTED, "rthip_render: max_recursion %u > %d", p->max_recursion, RTHIP_MAX_RECURSION);
    if (p->aa_mode != 0 && p->aa_mode != 1) return fail(RTHIP_E_INVALID, "rthip_render: aa_mode %d", p->aa_mode);
    if (p->renderer < 0 || p->renderer > 2) return fail(RTHIP_E_INVALID, "rthip_render: renderer %d", p->renderer);
    if (p->part_count == 0 || p->part_index >= p->part_count) return fail(RTHIP_E_INVALID, "rthip_render: part %u of %u", p->part_index, p->part_count);
    uint32_t tw = p->tile_w ? p->tile_w : 32, th = p->tile_h ? p->tile_h : 8;
    if (tw % 8 || th % 8 || tw > 1024 || th > 1024) return fail(RTHIP_E_INVALID, "rthip_render: tile %ux%u must be multiples of 8", tw, th);
    if (p->layout != 0 && p->layout != 1) return fail(RTHIP_E_INVALID, "rthip_render: layout %d", p->layout);
    if (p->accel != 0 && p->accel != 1) return fail(RTHIP_E_INVALID, "rthip_render: accel %d", p->accel);
    return 0;
}

// renders this context's part into d_pixels (device, W*H) and optionally device id images
static int render_frame(rthip_ctx *c, const rthip_camera *cam, const rthip_params *p, uint32_t *d_pixels,
                        int32_t *d_shape, int32_t *d_tri) {
    int rc = check_params(cam, p);
    if (rc) return rc;
    if (!c->has_scene) return fail(RTHIP_E_STATE, "rthip_render: no scene uploaded");
    if (set_device(c)) return RTHIP_E_HIP;
    c->interrupt.store(0);
    c->progress.store(0.f);

    FrameParams fp;
    memset(&fp, 0, sizeof fp);
    fp.cam = *cam;
    fp.W = p->width;
    fp.H = p->height;
    fp.max_rec = p->renderer == RTHIP_RENDER_RAYTRACER ? p->max_recursion : 0;
    fp.aa = p->aa_mode;
    fp.renderer = p->renderer;
    fp.part_index = p->part_index;
    fp.part_count = p->part_count;
    fp.tile_w = p->tile_w ? p->tile_w : 32;
    fp.tile_h = p->tile_h ? p->tile_h : 8;
    fp.tiles_x = (fp.W + fp.tile_w - 1) / fp.tile_w;
    fp.tiles_y = (fp.H + fp.tile_h - 1) / fp.tile_h;
    fp.count_kd = c->count_kd;
    fp.layout = (d_shape || d_tri) ? 0 : p->layout;
    const uint32_t n_tiles = fp.tiles_x * fp.tiles_y;
    const uint32_t my_tiles = n_tiles > fp.part_index ? (n_tiles - fp.part_index + fp.part_count - 1) / fp.part_count : 0;
    const uint32_t per_tile = fp.tile_w * fp.tile_h;
    const uint32_t spp = fp.aa ? 9 : 1;
    const uint32_t levels = fp.max_rec + 1;
    uint32_t tiles_per_chunk = (uint32_t)(kMaxLevelRays / ((size_t)per_tile * spp));
    if (tiles_per_chunk == 0) tiles_per_chunk = 1;
    if (tiles_per_chunk > my_tiles) tiles_per_chunk = my_tiles ? my_tiles : 1;
    const size_t N = (size_t)tiles_per_chunk * per_tile * spp;

    if (c->rec.ensure(sizeof(double) * 4 * N * levels)) return RTHIP_E_HIP;
    if (c->parent.ensure(sizeof(uint32_t) * N * levels)) return RTHIP_E_HIP;
    if (c->rays.ensure(sizeof(double) * 12 * N)) return RTHIP_E_HIP;
    if (c->counts.ensure(sizeof(uint32_t) * 2 * (RTHIP_MAX_RECURSION + 3))) return RTHIP_E_HIP;
    if (c->stats.ensure(sizeof(unsigned long long) * ST_COUNT)) return RTHIP_E_HIP;
    if (c->ids.ensure(sizeof(int32_t) * 2 * (size_t)tiles_per_chunk * per_tile)) return RTHIP_E_HIP;
    if (c->hits.ensure(sizeof(double) * 6 * N)) return RTHIP_E_HIP;
    if (c->hit_ids.ensure(sizeof(uint32_t) * 2 * N)) return RTHIP_E_HIP;
    if (c->redo.ensure(sizeof(uint32_t) * N)) return RTHIP_E_HIP;
    Work w;
    w.rec = (double *)c->rec.p;
    w.parent = (uint32_t *)c->parent.p;
    w.rays = (double *)c->rays.p;
    w.counts = (uint32_t *)c->counts.p;
    w.stats = (unsigned long long *)c->stats.p;
    w.id_shape = (d_shape || d_tri) ? (int32_t *)c->ids.p : nullptr;
    w.id_tri = w.id_shape ? w.id_shape + (size_t)tiles_per_chunk * per_tile : nullptr;
    w.hit = (double *)c->hits.p;
    w.hit_shape = (int32_t *)c->hit_ids.p;
    w.hit_mask = (uint32_t *)c->hit_ids.p + N;
    w.redo = (uint32_t *)c->redo.p;
    w.N = (uint32_t)N;

    c->sv.accel = p->accel;
    c->stack_cap = c->stack_cap_full;
    const size_t lds_bytes = (size_t)c->stack_cap * kBlock * sizeof(uint32_t);
    if (lds_bytes > 160 * 1024) return fail(RTHIP_E_UNSUPPORTED, "kd stack of %u entries does not fit LDS", c->stack_cap);
    if (lds_bytes > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    // hot kernel: short stack (child word + entry parameter per entry); deeper rays overflow into the redo list
    const uint32_t cap_fast = c->stack_cap_bvh < kFastStackCap ? c->stack_cap_bvh : kFastStackCap;
    const size_t lds_fast = (size_t)cap_fast * kBlock * sizeof(uint32_t) * 2;
    const size_t lds_tables = (size_t)c->n_tables * 256;

    HIP_TRY(hipMemsetAsync(w.stats, 0, sizeof(unsigned long long) * ST_COUNT, c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    size_t ev_used = 0;
    auto get_event = [&](hipEvent_t *e) -> int {
        if (ev_used == c->ev_pool.size()) {
            hipEvent_t ne;
            HIP_TRY(hipEventCreate(&ne));
            c->ev_pool.push_back(ne);
        }
        *e = c->ev_pool[ev_used++];
        return 0;
    };
    uint64_t pixels_done = 0;
    bool interrupted = false;
    for (uint32_t q0 = 0; q0 < my_tiles; q0 += tiles_per_chunk) {
        if (c->interrupt.load()) { interrupted = true; break; }
        uint32_t nq = my_tiles - q0 < tiles_per_chunk ? my_tiles - q0 : tiles_per_chunk;
        fp.q0 = q0;
        fp.n_pix_slots = nq * per_tile;
        fp.N0 = fp.n_pix_slots * spp;
        HIP_TRY(hipMemsetAsync(w.counts, 0, sizeof(uint32_t) * 2 * (RTHIP_MAX_RECURSION + 3), c->stream));
        for (uint32_t level = 0; level < levels; level++) {
            uint32_t grid = (fp.N0 + kBlock - 1) / kBlock;
            if (level > 0 && grid > kMaxGridLevel) grid = kMaxGridLevel;
            hipEvent_t ea = nullptr, eb = nullptr;
            if (c->profile_launches) {
                if (get_event(&ea) || get_event(&eb)) return RTHIP_E_HIP;
                HIP_TRY(hipEventRecord(ea, c->stream));
            }
            w.redo_count = w.counts + (RTHIP_MAX_RECURSION + 3) + level;
            if (p->accel == 0) {
                if (c->count_kd)
                    hipLaunchKernelGGL((k_trace<true, false, true>), dim3(grid), dim3(kBlock), lds_fast, c->stream, c->sv, fp, w, level, cap_fast);
                else
                    hipLaunchKernelGGL((k_trace<true, false, false>), dim3(grid), dim3(kBlock), lds_fast, c->stream, c->sv, fp, w, level, cap_fast);
                if (c->profile_launches) HIP_TRY(hipEventRecord(eb, c->stream));
                if (c->has_meshes) {  // re-trace of the (rare) rays lying exactly in a kd split plane or overflowing the short stack
                    if (c->count_kd)
                        hipLaunchKernelGGL((k_trace<false, true, true>), dim3(64), dim3(kBlock), lds_bytes, c->stream, c->sv, fp, w, level, c->stack_cap);
                    else
                        hipLaunchKernelGGL((k_trace<false, true, false>), dim3(64), dim3(kBlock), lds_bytes, c->stream, c->sv, fp, w, level, c->stack_cap);
                }
            } else {
                if (c->count_kd)
                    hipLaunchKernelGGL((k_trace<false, false, true>), dim3(grid), dim3(kBlock), lds_bytes, c->stream, c->sv, fp, w, level, c->stack_cap);
                else
                    hipLaunchKernelGGL((k_trace<false, false, false>), dim3(grid), dim3(kBlock), lds_bytes, c->stream, c->sv, fp, w, level, c->stack_cap);
                if (c->profile_launches) HIP_TRY(hipEventRecord(eb, c->stream));
            }
            hipLaunchKernelGGL(k_shade, dim3(grid), dim3(kBlock), lds_tables, c->stream, c->sv, fp, w, level, c->n_tables);
        }
        for (uint32_t level = levels - 1; level >= 1; level--) {
            uint32_t grid = (fp.N0 + kBlock - 1) / kBlock;
            if (grid > kMaxGridLevel) grid = kMaxGridLevel;
            hipLaunchKernelGGL(k_fold, dim3(grid), dim3(kBlock), 0, c->stream, w, level);
        }
        hipLaunchKernelGGL(k_resolve, dim3((fp.n_pix_slots + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, fp, w,
                           d_pixels, d_shape, d_tri);
        HIP_TRY(hipGetLastError());
        pixels_done += fp.n_pix_slots;
        if (q0 + tiles_per_chunk < my_tiles) {  // more chunks: the workspace is reused, so drain first
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->progress.store((float)(q0 + nq) / (float)my_tiles);
        }
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->last.pixels = pixels_done;
    if (interrupted) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        return fail(RTHIP_E_INTERRUPTED, "rthip_render: interrupted");
    }
    c->last.launches_trace = (uint32_t)(ev_used / 2);
    return 0;
}

static int finish_stats(rthip_ctx *c, const rthip_params *p) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long st[ST_COUNT];
    HIP_TRY(hipMemcpy(st, c->stats.p, sizeof st, hipMemcpyDeviceToHost));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    // primary rays = pixels inside the image handled by this part (x9 with SSAA)
    uint32_t tw = p->tile_w ? p->tile_w : 32, th = p->tile_h ? p->tile_h : 8;
    uint32_t tiles_x = (p->width + tw - 1) / tw, tiles_y = (p->height + th - 1) / th;
    uint64_t px = 0;
    for (uint32_t t = p->part_index; t < tiles_x * tiles_y; t += p->part_count) {
        uint32_t tx = t % tiles_x, ty = t / tiles_x;
        uint32_t w0 = tx * tw, h0 = ty * th;
        uint32_t ww = p->width - w0 < tw ? p->width - w0 : tw, hh = p->height - h0 < th ? p->height - h0 : th;
        px += (uint64_t)ww * hh;
    }
    c->last.pixels = px;
    c->last.rays_primary = px * (p->aa_mode ? 9 : 1);
    c->last.rays_shadow = st[ST_SHADOW];
    c->last.rays_reflect = st[ST_REFLECT];
    c->last.kd_nodes_visited = st[ST_KD_NODES];
    c->last.kd_tris_tested = st[ST_KD_TRIS];
    c->last.kd_leaves_visited = st[ST_KD_LEAVES];
    c->last.kd_pops = st[ST_KD_POPS];
    c->last.shape_tests = st[ST_SHAPES];
    c->last.mesh_hits = st[ST_MESH_HITS];
    c->last.kd_max_nodes_ray = st[ST_KD_MAXRAY];
    c->last.wave_steps_max = st[ST_WAVE_STEPS_MAX];
    c->last.wave_tris_max = st[ST_WAVE_TRIS_MAX];
    c->last.wave_steps = st[ST_WAVE_STEPS];
    c->last.wave_tris = st[ST_WAVE_TRIS];
    c->last.wave_cycles_max = st[ST_WAVE_CYCLES_MAX];
    uint32_t lr[RTHIP_MAX_RECURSION + 1];
    HIP_TRY(hipMemcpy(lr, c->counts.p, sizeof lr, hipMemcpyDeviceToHost));
    memset(c->last.level_ms_trace, 0, sizeof c->last.level_ms_trace);
    for (uint32_t i = 0; i <= RTHIP_MAX_RECURSION; i++) c->last.level_rays[i] = i <= p->max_recursion ? lr[i] : 0;
    c->last.level_rays[0] = (uint32_t)c->last.rays_primary;
    c->last.ms_total = ms;
    float sum = 0.f;
    for (uint32_t i = 0; i < c->last.launches_trace; i++) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, c->ev_pool[2 * i], c->ev_pool[2 * i + 1]));
        sum += t;
        if (i <= RTHIP_MAX_RECURSION) c->last.level_ms_trace[i] = t;  // single-chunk frames: launch i = level i
    }
    c->last.ms_trace = sum;
    c->progress.store(1.f);
    return 0;
}

int rthip_render_device(rthip_ctx *c, const rthip_camera *cam, const rthip_params *p, void *d_pixels, int sync) {
    if (!c || !d_pixels) return fail(RTHIP_E_INVALID, "rthip_render_device: null argument");
    int rc = render_frame(c, cam, p, (uint32_t *)d_pixels, nullptr, nullptr);
    if (rc) return rc;
    if (sync) return finish_stats(c, p);
    return 0;
}

int rthip_render(rthip_ctx *c, const rthip_camera *cam, const rthip_params *p, uint32_t *out_pixels,
                 int32_t *out_shape, int32_t *out_tri) {
    if (!c || !out_pixels) return fail(RTHIP_E_INVALID, "rthip_render: null argument");
    int rc = check_params(cam, p);
    if (rc) return rc;
    if (set_device(c)) return RTHIP_E_HIP;
    size_t npx = (size_t)p->width * p->height;
    if (c->pixels_tmp.ensure(sizeof(uint32_t) * npx)) return RTHIP_E_HIP;
    bool ids = out_shape || out_tri;
    if (ids && c->ids_out.ensure(sizeof(int32_t) * 2 * npx)) return RTHIP_E_HIP;
    uint32_t *d_px = (uint32_t *)c->pixels_tmp.p;
    int32_t *d_sh = ids ? (int32_t *)c->ids_out.p : nullptr;
    int32_t *d_tr = ids ? d_sh + npx : nullptr;
    if (p->part_count > 1) {  // pixels of other parts keep the caller's contents
        HIP_TRY(hipMemcpyAsync(d_px, out_pixels, sizeof(uint32_t) * npx, hipMemcpyHostToDevice, c->stream));
    }
    if (ids) HIP_TRY(hipMemsetAsync(d_sh, 0xff, sizeof(int32_t) * 2 * npx, c->stream));
    rthip_params pf = *p;
    pf.layout = 0;
    rc = render_frame(c, cam, &pf, d_px, d_sh, d_tr);
    if (rc) return rc;
    rc = finish_stats(c, p);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_pixels, d_px, sizeof(uint32_t) * npx, hipMemcpyDeviceToHost));
    if (out_shape) HIP_TRY(hipMemcpy(out_shape, d_sh, sizeof(int32_t) * npx, hipMemcpyDeviceToHost));
    if (out_tri) HIP_TRY(hipMemcpy(out_tri, d_tr, sizeof(int32_t) * npx, hipMemcpyDeviceToHost));
    return 0;
}

uint64_t rthip_part_pixels(const rthip_params *p) {
    if (!p || p->width == 0 || p->height == 0 || p->part_count == 0) return 0;
    uint32_t tw = p->tile_w ? p->tile_w : 32, th = p->tile_h ? p->tile_h : 8;
    uint64_t tiles = (uint64_t)((p->width + tw - 1) / tw) * ((p->height + th - 1) / th);
    return ((tiles + p->part_count - 1) / p->part_count) * tw * th;
}

int rthip_assemble(rthip_ctx *c, const rthip_params *p, const void *d_gathered, void *d_frame, int sync) {
    if (!c || !p || !d_gathered || !d_frame) return fail(RTHIP_E_INVALID, "rthip_assemble: null argument");
    rthip_camera dummy;
    memset(&dummy, 0, sizeof dummy);
    int rc = check_params(&dummy, p);
    if (rc) return rc;
    if (set_device(c)) return RTHIP_E_HIP;
    FrameParams fp;
    memset(&fp, 0, sizeof fp);
    fp.W = p->width;
    fp.H = p->height;
    fp.part_count = p->part_count;
    fp.tile_w = p->tile_w ? p->tile_w : 32;
    fp.tile_h = p->tile_h ? p->tile_h : 8;
    fp.tiles_x = (fp.W + fp.tile_w - 1) / fp.tile_w;
    fp.tiles_y = (fp.H + fp.tile_h - 1) / fp.tile_h;
    uint32_t part_slots = (uint32_t)rthip_part_pixels(p);
    hipLaunchKernelGGL(k_assemble, dim3((part_slots + kBlock - 1) / kBlock, p->part_count), dim3(kBlock), 0, c->stream, fp,
                       (const uint32_t *)d_gathered, part_slots, (uint32_t *)d_frame);
    HIP_TRY(hipGetLastError());
    if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------------------- probes
static int probe_in(rthip_ctx *c, int slot, const void *src, size_t bytes, void **out) {
    if (c->probe[slot].ensure(bytes)) return RTHIP_E_HIP;
    if (src) HIP_TRY(hipMemcpyAsync(c->probe[slot].p, src, bytes, hipMemcpyHostToDevice, c->stream));
    *out = c->probe[slot].p;
    return 0;
}
static int probe_out(rthip_ctx *c, void *dst, const void *src, size_t bytes) {
    if (dst) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return 0;
}
#define PROBE_PROLOGUE(need_scene)                                                   \
    if (!c) return fail(RTHIP_E_INVALID, "null context");                           \
    if (need_scene && !c->has_scene) return fail(RTHIP_E_STATE, "no scene uploaded"); \
    if (set_device(c)) return RTHIP_E_HIP;                                           \
    if (n == 0) return 0;                                                            \
    const uint32_t grid = (n + kBlock - 1) / kBlock;                                 \
    const size_t lds_bytes = (size_t)c->stack_cap * kBlock * sizeof(uint32_t);      \
    (void)lds_bytes;                                                                 \
    int rc;

int rthip_probe_cast(rthip_ctx *c, uint32_t n, const double *o, const double *d, int32_t *hit, double *point,
                     double *local, double *normal, int32_t *shape, int32_t *tri) {
    PROBE_PROLOGUE(true)
    if (!o || !d || !hit) return fail(RTHIP_E_INVALID, "rthip_probe_cast: null argument");
    void *po, *pd, *ph, *pp, *pl, *pn, *ps, *pt;
    size_t v = sizeof(double) * 3 * n, s = sizeof(int32_t) * n;
    if ((rc = probe_in(c, 0, o, v, &po)) || (rc = probe_in(c, 1, d, v, &pd)) || (rc = probe_in(c, 2, nullptr, s, &ph)) ||
        (rc = probe_in(c, 3, nullptr, v, &pp)) || (rc = probe_in(c, 4, nullptr, v, &pl)) || (rc = probe_in(c, 5, nullptr, v, &pn)) ||
        (rc = probe_in(c, 6, nullptr, s, &ps)) || (rc = probe_in(c, 7, nullptr, s, &pt)))
        return rc;
    if (lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)k_probe_cast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k_probe_cast, dim3(grid), dim3(kBlock), lds_bytes, c->stream, c->sv, n, (const double *)po, (const double *)pd,
                       (int32_t *)ph, (double *)pp, (double *)pl, (double *)pn, (int32_t *)ps, (int32_t *)pt, c->stack_cap);
    HIP_TRY(hipGetLastError());
    if ((rc = probe_out(c, hit, ph, s)) || (rc = probe_out(c, point, pp, v)) || (rc = probe_out(c, local, pl, v)) ||
        (rc = probe_out(c, normal, pn, v)) || (rc = probe_out(c, shape, ps, s)) || (rc = probe_out(c, tri, pt, s)))
        return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int rthip_probe_noise(rthip_ctx *c, int32_t noise, uint32_t n, const double *p, double *out) {
    PROBE_PROLOGUE(true)
    if (!p || !out) return fail(RTHIP_E_INVALID, "rthip_probe_noise: null argument");
    if (noise < 0) return fail(RTHIP_E_INVALID, "rthip_probe_noise: noise %d", noise);
    void *pp, *po;
    if ((rc = probe_in(c, 0, p, sizeof(double) * 3 * n, &pp)) || (rc = probe_in(c, 1, nullptr, sizeof(double) * n, &po))) return rc;
    hipLaunchKernelGGL(k_probe_noise, dim3(grid), dim3(kBlock), 0, c->stream, c->sv, noise, n, (const double *)pp, (double *)po);
    HIP_TRY(hipGetLastError());
    if ((rc = probe_out(c, out, po, sizeof(double) * n))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int rthip_probe_material(rthip_ctx *c, int32_t material, uint32_t n, const double *p, double *color, double *absorptance) {
    PROBE_PROLOGUE(true)
    if (!p || !color || !absorptance || material < 0) return fail(RTHIP_E_INVALID, "rthip_probe_material: bad argument");
    void *pp, *pc, *pa;
    if ((rc = probe_in(c, 0, p, sizeof(double) * 3 * n, &pp)) || (rc = probe_in(c, 1, nullptr, sizeof(double) * 3 * n, &pc)) ||
        (rc = probe_in(c, 2, nullptr, sizeof(double) * n, &pa)))
        return rc;
    hipLaunchKernelGGL(k_probe_material, dim3(grid), dim3(kBlock), 0, c->stream, c->sv, material, n, (const double *)pp, (double *)pc, (double *)pa);
    HIP_TRY(hipGetLastError());
    if ((rc = probe_out(c, color, pc, sizeof(double) * 3 * n)) || (rc = probe_out(c, absorptance, pa, sizeof(double) * n))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int rthip_probe_background(rthip_ctx *c, uint32_t n, const double *d, double *color) {
    PROBE_PROLOGUE(true)
    if (!d || !color) return fail(RTHIP_E_INVALID, "rthip_probe_background: null argument");
    void *pd, *pc;
    if ((rc = probe_in(c, 0, d, sizeof(double) * 3 * n, &pd)) || (rc = probe_in(c, 1, nullptr, sizeof(double) * 3 * n, &pc))) return rc;
    hipLaunchKernelGGL(k_probe_background, dim3(grid), dim3(kBlock), 0, c->stream, c->sv, n, (const double *)pd, (double *)pc);
    HIP_TRY(hipGetLastError());
    if ((rc = probe_out(c, color, pc, sizeof(double) * 3 * n))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int rthip_probe_raycast(rthip_ctx *c, uint32_t max_recursion, uint32_t n, const double *o, const double *d, double *color) {
    PROBE_PROLOGUE(true)
    if (!o || !d || !color) return fail(RTHIP_E_INVALID, "rthip_probe_raycast: null argument");
    if (max_recursion > RTHIP_MAX_RECURSION) return fail(RTHIP_E_UNSUPPORTED, "max_recursion %u", max_recursion);
    void *po, *pd, *pc;
    size_t v = sizeof(double) * 3 * n;
    if ((rc = probe_in(c, 0, o, v, &po)) || (rc = probe_in(c, 1, d, v, &pd)) || (rc = probe_in(c, 2, nullptr, v, &pc))) return rc;
    if (lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)k_probe_raycast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k_probe_raycast, dim3(grid), dim3(kBlock), lds_bytes, c->stream, c->sv, max_recursion, n, (const double *)po,
                       (const double *)pd, (double *)pc, c->stack_cap);
    HIP_TRY(hipGetLastError());
    if ((rc = probe_out(c, color, pc, v))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int rthip_probe_arith(rthip_ctx *c, int32_t op, uint32_t n, const double *a, const double *b, double *out) {
    PROBE_PROLOGUE(false)
    if (!a || !out || op < 0 || op > 10) return fail(RTHIP_E_INVALID, "rthip_probe_arith: bad argument");
    void *pa, *pb = nullptr, *po;
    if ((rc = probe_in(c, 0, a, sizeof(double) * n, &pa)) || (rc = probe_in(c, 2, nullptr, sizeof(double) * n, &po))) return rc;
    if (b && (rc = probe_in(c, 1, b, sizeof(double) * n, &pb))) return rc;
    hipLaunchKernelGGL(k_probe_arith, dim3(grid), dim3(kBlock), 0, c->stream, op, n, (const double *)pa, (const double *)pb, (double *)po);
    HIP_TRY(hipGetLastError());
    if ((rc = probe_out(c, out, po, sizeof(double) * n))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
