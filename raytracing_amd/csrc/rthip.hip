// rthip.hip — HIP kernels (gfx950 / CDNA4) + C-ABI of the Raycast hot path.  See include/rthip.h.
//
// Frame pipeline, per recursion level k (wavefront style; ray counts stay on the device, the host never
// synchronises between levels):
//   k_trace_easy     : one ray per lane.  Level 0 generates its rays.  Every query of the ray over the
//                      analytic shapes; meshes only through a conservative test of the top two levels of
//                      their BVH.  Rays that cannot touch a mesh are finished AND shaded here (material /
//                      noise / background, Lambert terms, reflected ray compacted into the level-(k+1) queue
//                      with a wave ballot + prefix popcount and ONE atomic per workgroup); the others are
//                      listed together with their closest hit over the analytic shapes.
//   k_trace_quad     : the listed rays, FOUR LANES PER RAY over the 4-wide BVH (namespace quad), exact fp64
//                      triangle tests; writes hit records.  16 rays per wave, everything of a ray in its quad.
//   k_trace_coop     : the same for LONG lists (>= 300 000 rays; namespace coop): up to 64 rays per
//                      wave, one ray per lane for whatever is per-ray work (transforms, hit record, shadow set-up,
//                      analytic shadow casters), the mesh walks 16 rays at a time through quad::mesh_closest.
//                      (Which of the two a level launches: render_frame, walk_mode.)
//   k_shade<true>    : shading + spawn of the listed rays from their hit records.
//   k_redo           : re-traces and shades the redo list (rays only the reference kd-tree reproduces; normally empty).
// k_trace_easy and k_shade have a LEAN instantiation for scenes whose materials need neither Worley / Marble noise nor the
// material stack (upload_scene: lean_scene).  What is the same for every ray is computed once at upload (k_light_dirs: the
// directional lights' directions in every shape's frame, the planes' world normals).
// then k_resolve: per pixel, C_k = direct_k (+) C_{k+1} (x) refl_k folded back to front along the pixel's
// chain of rays (clamping Color algebra forbids a forward throughput product), 3x3 SSAA
// sum, 0xAABBGGRR pack.  accel = 1 runs k_trace<false,.> + k_shade<false> instead (reference kd-tree,
// one ray per lane).  No MFMA: the path is divergent fp64 VALU work.
// A call traces 1..RTHIP_MAX_BATCH frames (one camera each) as ONE such sequence of launches: the rays of all
// the frames share the queues.  The sequence of a repeated call is captured into a HIP graph and replayed.
//
// Built with -ffp-contract=off: the oracle is the non-fused x86-64 build of the reference.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <future>
#include <mutex>
#include <string>
#include <vector>

#include "rt_core.hpp"
#include "rthip_bvh.hpp"
#include "rthip_kd.hpp"

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
extern "C" void rthip_set_last_error(const char *msg) { g_err = msg ? msg : ""; }  // for the library's other translation units
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(RTHIP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static std::atomic<uint64_t> g_device_allocs{0};  // hipMalloc calls of the frame / scene buffers (rthip_debug_device_allocs)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        if (n == 0) n = 256;
        g_device_allocs.fetch_add(1);
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

// Workspace of rthip_render_device_noalloc (include/rthip.h): the per-frame records that render_frame otherwise keeps in
// hipMalloc'ed DevBufs, for callers that cannot allocate.  Every GPU the library runs on holds its own copy of this array
// at its own address until the process ends.  In host code the NAME below is only a stand-in for the symbol: the device
// address is taken once per device with hipGetSymbolAddress (rthip_create, that device current) and kept in
// g_scratch_ptr[device]; host code only ever uses that pointer (kernel arguments included).
static __device__ __align__(256) unsigned char scratch[RTHIP_NOALLOC_SCRATCH_BYTES];
static const int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices];   // device address of `scratch` on device d, nullptr until a context exists there
static std::mutex g_scratch_mutex;
// minimum waves per SIMD the register allocator must leave room for (the kernels are
// bound by dependent fp64 chains, so resident waves matter more than a few spilled values)
#ifndef RT_TRACE_OCC
#define RT_TRACE_OCC 4
#endif
#ifndef RT_EASY_OCC
#define RT_EASY_OCC 4
#endif
#ifndef RT_QUAD_OCC
#define RT_QUAD_OCC 5
#endif
#ifndef RT_COOP_OCC
// k_trace_coop: 4 (121 VGPRs, nothing spilled).  At 5 (96 VGPRs) 43 registers of the per-ray phases live in scratch memory and the
// kernel waits for them (VALU issue 62 % against k_trace_quad's 75 %): 32-frame calls 12 290 - 12 390 -> 12 540 Mrays/s, 20-step
// bursts 11 720 -> 12 000; at 6 (80 VGPRs, 70 spilled) 11 920 / 11 270 (profiles/r03_coop_walk.txt)
#define RT_COOP_OCC 4
#endif
#ifndef RT_SHADE_OCC
#define RT_SHADE_OCC 4
#endif
static const uint32_t kMaxGridQuad = 16384;    // quad kernel (64 rays per workgroup, grid-stride): 8192 costs 1.3 % on a lone 4K frame, 4096 5 %; calls of 32 frames do not care (tools/sweep_grids.sh)
static const uint32_t kMaxGridCoop = 4096;     // k_trace_coop (up to 256 rays per workgroup and stride): on a short list the launch returns at once, and 16384 idle
                                               // workgroups cost a lone frame 10 us over its four levels; 32-frame calls: 2048 / 4096 / 8192 / 16384 = 12 385 / 12 467 / 12 398 / 12 270 Mrays/s
static const size_t kMaxLevelRays = (size_t)1 << 26;  // level-0 rays per chunk; RTHIP_CHUNK_RAYS overrides (tests).  Together with
static const size_t kMaxWorkspaceBytes = (size_t)32 << 30;  // ... this bound on the per-ray records it sizes the workspace for 288 GB of
                                                      // HBM: 32 frames of 1080p (or one 8K frame at MaxRecursion 8) are ONE chunk - a
                                                      // call's fixed cost (a chain of launches, each as long as its slowest wave) is
                                                      // paid once per chunk: 8-frame chunks (2^24) cost the bench 5 %
static const uint32_t kMaxGridLevel = 4096;   // k_trace_easy / k_shade / k_trace: grid-stride over a bounded grid (one workgroup per 256 rays
                                              // costs 6 % when calls overlap: LDS table staging, statistics flush and launch per workgroup)
static const uint32_t kShadeCountShapes = 64;                      // per-shape hit counters of a counting context (shapes beyond share the last)
static const uint32_t kShadeCounts = kShadeCountShapes + 1;        // + background evaluations
enum { ST_KD_NODES = 0, ST_KD_TRIS, ST_KD_LEAVES, ST_KD_POPS, ST_SHAPES, ST_MESH_HITS, ST_KD_MAXRAY, ST_WAVE_STEPS_MAX, ST_WAVE_TRIS_MAX, ST_WAVE_STEPS, ST_WAVE_TRIS, ST_WAVE_CYCLES_MAX, ST_FETCH_CYCLES, ST_SHAPES_EASY, ST_SHAPES_FULL, ST_SHAPES_EASY_FULL, ST_COUNT };

// Per-frame constants
struct FrameParams {
    const rthip_camera *cams;  // device: one per frame of the batch (a plain frame is a batch of one); written by k_begin at the head of the
                               // call's launches from the context's page-locked camera ring (a batch holds up to RTHIP_MAX_BATCH)
    uint32_t n_frames;
    uint32_t my_tiles;     // tiles of this part in ONE frame: per-part tile ordinal q = frame * my_tiles + tile
    uint32_t part_slots;   // pixel slots of one frame's tile-compact part buffer (rthip_part_pixels)
    uint32_t W, H, max_rec;
    int32_t aa, renderer;
    uint32_t part_index, part_count, tile_w, tile_h, tiles_x, tiles_y;
    uint32_t q0;           // first per-part tile ordinal (over the whole batch) of this chunk
    uint32_t n_pix_slots;  // tile slots in this chunk * tile_w * tile_h
    uint32_t N0;           // level-0 rays of this chunk = n_pix_slots * (aa ? 9 : 1)
    uint32_t count_kd;     // collect kd counters
    int32_t layout;        // 0 frame, 1 tile-compact
    rtc::FastDiv d_per_tile, d_my_tiles, d_tiles_x, d_sbx, d_pix_slots;  // divisions of slot_to_pixel / primary_ray (see rtc::fast_div)
};
// Per-frame device workspace.  Every per-ray record is ONE aligned array-of-structs entry (a 64-byte
// ray, a 64-byte hit, a 32-byte colour): a lane reads or writes whole cache lines with dwordx4
// accesses.  (Seven separate 8-byte arrays written at compacted offsets cost 0.2 ms per 1 M spawned
// rays on MI355X; the same data as one 64-byte store per lane is free.)
struct __attribute__((aligned(64))) RayRec {
    double ox, oy, oz, dx, dy, dz;
    uint32_t parent;  // index of the spawning ray in the previous level
    uint32_t pad;
};
struct __attribute__((aligned(64))) HitSlot {
    double lx, ly, lz, nx, ny, nz;  // RaycastResult.LocalPoint / .Normal
    int32_t shape;                  // shape ordinal, -1 miss, -2 no ray (slot outside the image)
    uint32_t mask;                  // shadow mask
};
struct __attribute__((aligned(32))) ColRec {
    double r, g, b, refl;  // direct colour (later: folded colour) and the reflectance applied to the child
};
static const uint32_t kNoChild = 0xffffffffu;
static const uint32_t kRedoBlocks = 64;  // workgroups of k_shade that take part in re-tracing the redo list
static const int32_t kShapeNoRay = -2, kShapeRedo = -3;  // HitSlot::shape: slot outside the image / ray on the redo list (no record yet)
// Every device counter that workgroups reserve list space from (rays of a level, hard list, redo list) sits in a
// 256-byte line of its own: returning atomics to ONE line serialise in L2 (~7 ns each), and the spawn and hard-list
// reservations of a workgroup are in flight together.
static const uint32_t kCounterStride = 64;  // in uint32
static const uint32_t kCounterSlots = 3 * (RTHIP_MAX_RECURSION + 3);
// what k_trace_easy already knows about a hard ray: the closest-hit query over the ANALYTIC shapes (the reference's
// d = |O - M*P|^2 of the winner, DBL_MAX / 0xffffffff if none), so that the quad kernel only adds the meshes
struct __attribute__((aligned(16))) PendRec {
    double bd;
    uint32_t bs;
    uint32_t pad;
};
struct Work {
    ColRec *rec;            // [levels][N]
    uint32_t *child;        // [levels][N]  : index of the reflected ray in level+1, kNoChild if none
    RayRec *rays;           // [2][N]       : ping-pong ray queues
    uint32_t *counts;       // [levels+1] x kCounterStride : rays per level (see level_count)
    unsigned long long *stats;  // [ST_COUNT] + [2][kShadeCounts]: RTHIP_COUNTERS=1 - shaded hits per shape and background evaluations, by kernel (0 = k_trace_easy, 1 = k_shade)
    int32_t *id_shape, *id_tri; // [n_pix_slots] primary-hit ids (optional)
    HitSlot *hit;               // [N] per ray of the current level
    uint32_t *redo;             // [N] rays of the current level that must be re-traced by the full kernel
    uint32_t *redo_count;       // [levels] one counter per level (zeroed with `counts`)
    uint32_t *redo_stack;       // [stack_cap][kRedoBlocks * kBlock] kd / BVH stacks of the lanes that re-trace the redo list (k_shade)
    uint32_t *hard;             // [N] rays of the current level that need a mesh walk (traced by the quad kernel)
    uint32_t *hard_count;       // [levels]
    PendRec *pend;              // [N] per entry of the hard list
    uint2 *block_stats;         // [2][levels][bs_stride] per workgroup of k_trace_easy / k_shade: (shadow queries, reflected rays) of the frame
    uint32_t bs_stride;
    uint32_t N;
    // Interrupt() / RenderStatus::Percent at kernel granularity (Raytracer.cpp:91-99, 150-154): `irq` is a word in FINE-GRAINED
    // device memory (not cached in L2, so a write made while a kernel runs is seen by that kernel) raised by the watcher
    // workgroups (irq_load); every workgroup reads it when it starts and once per grid-stride iteration and leaves when it is
    // set.  `stage` is a word in page-locked host memory: workgroup 0 of every kernel stores the index of the stage it begins.
    const uint32_t *irq;
    const uint32_t *irq_host;   // the request itself: a word in page-locked host memory that rthip_interrupt stores to
    uint32_t *stage;
    uint32_t stage_base;        // stage index of the current chunk's first kernel
    // the lengths of this call's hard lists, level by level, for the host (page-locked memory): the next call of the same shape
    // launches only the walk kernel such lists are for (render_frame: walk_mode); hist[0] = hist_tag says whose lengths they are
    uint32_t *hist;
    uint32_t hist_tag;
};
// Every 1024th workgroup of every kernel is a watcher: it reads the request word in host memory (a PCIe read of a microsecond or
// two that its whole workgroup waits for, so only a handful of workgroups do it, once per poll) and raises the device-side flag;
// everybody polls that flag, which costs an uncached read of HBM.  (Raising the device flag from the host with hipMemsetAsync /
// hipMemcpyAsync on a side stream waited for a free slot on a saturated GPU: 20 ms.)
// Not workgroup 0 alone: where a grid is several times what the GPU holds at once, workgroup 0 has finished its share after the
// first of those rounds, and a request that arrived later in the kernel's run was relayed by the NEXT kernel (round 3,
// tools/gpu_irq_probe3.py; profiles/r03_interrupt_watchers.txt).  Not every 64th either: those watchers' PCIe reads cost 6 % of the
// headline rate (same file); with every 1024th the rate is that of the single watcher and a walk kernel leaves 0.2-0.4 ms after
// Interrupt().
static const uint32_t kWatcherEvery = 1024;  // (a power of two)
__device__ __forceinline__ uint32_t irq_load(const Work &w) {
    if ((blockIdx.x & (kWatcherEvery - 1u)) == 0u && __hip_atomic_load(w.irq_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
        __hip_atomic_store(const_cast<uint32_t *>(w.irq), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __hip_atomic_load(w.irq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t irq_load_wave(const Work &w) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)irq_load(w)); }
__device__ __forceinline__ void stage_store(const Work &w, uint32_t stage) {
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(w.stage, stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static const uint32_t kIrqEvery = 8;  // grid-stride iterations between two polls of the interrupt flag (a power of two)
enum { STAGE_EASY = 0, STAGE_WALK = 1, STAGE_SHADE = 2, STAGES_PER_LEVEL = 3 };

__device__ __forceinline__ uint32_t &level_count(const Work &w, uint32_t level) { return w.counts[(size_t)level * kCounterStride]; }
__device__ __forceinline__ ColRec &rec_at(const Work &w, uint32_t level, uint32_t i) { return w.rec[(size_t)level * w.N + i]; }
__device__ __forceinline__ RayRec &ray_at(const Work &w, uint32_t level, uint32_t i) { return w.rays[(size_t)(level & 1) * w.N + i]; }
__device__ __forceinline__ void store_ray(const Work &w, uint32_t level, uint32_t i, V3 o, V3 d, uint32_t parent) {
    RayRec r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.parent = parent;
    r.pad = 0;
    ray_at(w, level, i) = r;
}
__device__ __forceinline__ void store_hit(const Work &w, uint32_t i, const HitRec &h) {
    HitSlot s;
    s.lx = h.local.x; s.ly = h.local.y; s.lz = h.local.z;
    s.nx = h.normal.x; s.ny = h.normal.y; s.nz = h.normal.z;
    s.shape = h.shape;
    s.mask = h.shadow_mask;
    w.hit[i] = s;
}
__device__ __forceinline__ void store_no_ray(const Work &w, uint32_t i) {
    HitSlot s;
    s.lx = s.ly = s.lz = s.nx = s.ny = s.nz = 0.0;
    s.shape = -2;
    s.mask = 0;
    w.hit[i] = s;
}

// slot (within chunk) -> pixel.  Tiles are tile_w x tile_h; inside a tile the pixels are ordered
// as 8x8 sub-blocks, and inside a sub-block as four 4x4 patches: 16 consecutive slots (the rays of one
// wave of the quad trace kernel) cover a compact 4x4 patch, 64 (one shade wave) an 8x8 one.
// `frame` = index within the batch, `part_slot` = slot within that frame's part (tile-compact order).
// ONE_FRAME: the caller's slots all lie in frame 0 (k_assemble: one part buffer of one frame)
template <bool ONE_FRAME = false>
__device__ __forceinline__ bool slot_to_pixel(const FrameParams &fp, uint32_t pix_slot, uint32_t &x, uint32_t &y, uint32_t &frame,
                                              uint32_t &part_slot) {
    const uint32_t per_tile = fp.tile_w * fp.tile_h;
    const uint32_t qc = rtc::fast_div(pix_slot, fp.d_per_tile);
    uint32_t q = fp.q0 + qc;
    const uint32_t r = pix_slot - qc * per_tile;
    frame = ONE_FRAME ? 0u : rtc::fast_div(q, fp.d_my_tiles);
    q -= frame * fp.my_tiles;
    part_slot = q * per_tile + r;
    const uint32_t t = fp.part_index + q * fp.part_count;
    if (t >= fp.tiles_x * fp.tiles_y || frame >= fp.n_frames) return false;
    const uint32_t ty = rtc::fast_div(t, fp.d_tiles_x), tx = t - ty * fp.tiles_x;
    const uint32_t sb = r >> 6, l = r & 63u;
    const uint32_t sbx = fp.tile_w >> 3;
    const uint32_t sby = rtc::fast_div(sb, fp.d_sbx);
    x = tx * fp.tile_w + (sb - sby * sbx) * 8 + ((l & 3u) | ((l >> 2) & 4u));
    y = ty * fp.tile_h + sby * 8 + (((l >> 2) & 3u) | ((l >> 3) & 4u));
    return x < fp.W && y < fp.H;
}

// Level-0 ray of queue slot i (R1, Raytracer.cpp:102-121; SSAA offsets :134-138: dx outer, dy inner, +-0.5f).
// False when the slot lies outside the image.  The 64 slots of a wave lie in one tile, hence in one frame of the
// batch: the camera is read with scalar loads (lane 0 is live whenever any lane of the wave is).
__device__ __forceinline__ bool primary_ray(const FrameParams &fp, uint32_t i, bool live, uint32_t &pix_slot, V3 &o, V3 &d) {
    uint32_t s = fp.aa ? rtc::fast_div(i, fp.d_pix_slots) : 0u;  // SSAA sub-sample (without SSAA the chunk holds one ray per slot)
    pix_slot = i - s * fp.n_pix_slots;
    uint32_t x = 0, y = 0, frame = 0, part_slot = 0;
    const bool inside = slot_to_pixel(fp, pix_slot, x, y, frame, part_slot);
    double fx = (double)x, fy = (double)y;
    if (fp.aa) {
        int dx = (int)(s / 3) - 1, dy = (int)(s % 3) - 1;
        fx = (double)((float)x + (float)dx * .5f);
        fy = (double)((float)(int)y + (float)dy * .5f);
    }
    frame = (uint32_t)__builtin_amdgcn_readfirstlane((int)frame);
    if (frame >= fp.n_frames) frame = 0;
    typedef const __attribute__((address_space(4))) rthip_camera uniform_camera;  // wave-uniform index: scalar loads
    rthip_camera cam;
    __builtin_memcpy(&cam, (uniform_camera *)(fp.cams + frame), sizeof cam);
    screen_ray(cam, fp.W, fp.H, fx, fy, o, d);
    return live && inside;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ============================================================================ kernels
extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];

// COUNT instantiations: wave-reduced traversal counters -> w.stats (fp.count_kd == 2: only the longest wave)
__device__ __forceinline__ void trace_counters_flush(const FrameParams &fp, const Work &w, KdCounters kc, uint32_t max_ray_nodes, long long t_start) {
        const uint32_t lane = threadIdx.x & 63u;
        kc.nodes = wave_sum(kc.nodes);
        kc.tris = wave_sum(kc.tris);
        kc.leaves = wave_sum(kc.leaves);
        kc.pops = wave_sum(kc.pops);
        kc.shapes = wave_sum(kc.shapes);
        kc.shapes_full = wave_sum(kc.shapes_full);
        kc.mesh_hits = wave_sum(kc.mesh_hits);
        kc.wave_steps = wave_sum(kc.wave_steps);
        kc.wave_tris = wave_sum(kc.wave_tris);
        kc.fetch_cycles = wave_sum(kc.fetch_cycles);
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
                w.stats[ST_FETCH_CYCLES] = kc.fetch_cycles;
                w.stats[ST_WAVE_STEPS] = kc.wave_steps;
            }
        } else if (lane == 0 && (kc.nodes | kc.shapes)) {
            atomicMax(&w.stats[ST_KD_MAXRAY], (unsigned long long)max_ray_nodes);
            atomicMax(&w.stats[ST_WAVE_STEPS_MAX], (unsigned long long)kc.wave_steps);
            atomicMax(&w.stats[ST_WAVE_CYCLES_MAX], cycles);
            atomicMax(&w.stats[ST_WAVE_TRIS_MAX], (unsigned long long)kc.wave_tris);
            atomicAdd(&w.stats[ST_WAVE_STEPS], (unsigned long long)kc.wave_steps);
            atomicAdd(&w.stats[ST_WAVE_TRIS], (unsigned long long)kc.wave_tris);
            atomicAdd(&w.stats[ST_FETCH_CYCLES], (unsigned long long)kc.fetch_cycles);
            atomicAdd(&w.stats[ST_KD_NODES], (unsigned long long)kc.nodes);
            atomicAdd(&w.stats[ST_KD_TRIS], (unsigned long long)kc.tris);
            atomicAdd(&w.stats[ST_KD_LEAVES], (unsigned long long)kc.leaves);
            atomicAdd(&w.stats[ST_KD_POPS], (unsigned long long)kc.pops);
            atomicAdd(&w.stats[ST_SHAPES], (unsigned long long)kc.shapes);
            atomicAdd(&w.stats[ST_SHAPES_FULL], (unsigned long long)kc.shapes_full);
            atomicAdd(&w.stats[ST_MESH_HITS], (unsigned long long)kc.mesh_hits);
        }
}

// One ray per lane through every scene query of the ray: the whole
// trace stage of accel = 1 (reference kd-tree everywhere), and with REDO = true the re-trace of the listed rays
// the quad kernel could not resolve (`n` and the ray indices come from the redo list).  Level 0 generates its
// rays (R1).  Output: one hit record per ray.
template <bool REDO, bool COUNT>
__global__ __launch_bounds__(kBlock, RT_TRACE_OCC) void k_trace(SceneView sv, FrameParams fp, Work w, uint32_t level,
                                                                 uint32_t stack_cap) {
    // LDS: stack_cap child words per lane (no entry parameters: nothing is culled on a pop here)
    KdStack st = make_stack(lds_dyn + threadIdx.x, kBlock, stack_cap, nullptr);
    const uint32_t n = min(w.N, REDO ? w.redo_count[0] : ((level == 0) ? fp.N0 : level_count(w, level)));  // (lists are bounded by the workspace)
    const uint32_t n_shadow = fp.renderer == RTHIP_RENDER_RAYTRACER ? sv.n_lights : 0u;  // DebugRaycaster: closest hit only
    KdCounters kc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t max_ray_nodes = 0;
    const long long t_start = COUNT ? clock64() : 0;
    if (!REDO) stage_store(w, w.stage_base + STAGES_PER_LEVEL * level + STAGE_WALK);
    // Interrupt(): polled when the level's first kernel starts and then every kIrqEvery-th grid-stride iteration (an uncached
    // read; vector loads return in order, so polling every iteration delayed each iteration's first fetch: 2.6 % on the bench)
    if (!REDO && __builtin_amdgcn_readfirstlane((int)irq_load(w))) return;
    uint32_t it = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        if ((++it & (kIrqEvery - 1u)) == 0u && __builtin_amdgcn_readfirstlane((int)irq_load(w))) break;  // wave-uniform
        const uint32_t j = base + threadIdx.x;
        const bool in_range = j < n;
        const uint32_t i = in_range ? (REDO ? w.redo[j] : j) : 0u;
        const uint32_t nodes_before = kc.nodes;
        V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
        uint32_t pix_slot = 0;
        bool live = in_range;
        if (level == 0) pix_slot = fp.aa ? i - rtc::fast_div(i, fp.d_pix_slots) * fp.n_pix_slots : i;
        if (level == 0 && !REDO) {
            if (!primary_ray(fp, i, live, pix_slot, o, d)) {
                if (live) store_no_ray(w, i);  // slot outside the image
                live = false;
            }
            if (live) store_ray(w, 0, i, o, d, 0u);  // the shade stage needs the direction (background, reflection)
        } else if (live) {  // (the redo list of level 0 holds rays k_trace_easy has stored)
            const RayRec r = ray_at(w, level, i);
            o = mk(r.ox, r.oy, r.oz);
            d = mk(r.dx, r.dy, r.dz);
        }
        // every lane of the wave walks the (wave-uniform) query and shape loops; dead lanes carry a
        // ray that is never traced (n_shapes = 0 would need a second code path)
        HitRec h;
        st.overflow = false;
        if (live) trace_stage<COUNT>(sv, o, d, n_shadow, st, h, COUNT ? &kc : nullptr);
        if (!live) continue;
        if (COUNT && kc.nodes - nodes_before > max_ray_nodes) max_ray_nodes = kc.nodes - nodes_before;
        store_hit(w, i, h);
        // primary-hit ids come from the un-offset pixel-centre ray (sub-sample 4 of the 3x3 SSAA grid)
        if (level == 0 && w.id_shape && i / fp.n_pix_slots == (fp.aa ? 4u : 0u)) {
            w.id_shape[pix_slot] = h.shape;
            w.id_tri[pix_slot] = h.tri;
        }
    }
    if (COUNT) trace_counters_flush(fp, w, kc, max_ray_nodes, t_start);
}

// ============================================================================ four lanes per ray
// The frame's trace kernel.  A ray is owned by a QUAD of four adjacent lanes (16 rays per wave):
//  * analytic shapes (spheres, planes) are spread over the four lanes, one shape per lane and pass;
//  * a mesh is walked by the four lanes together over the 4-wide BVH: each lane fetches the 32-byte
//    slot of ONE child (the quad reads one 128-byte line per step), tests its box, and the quad ranks
//    the hit children with DPP quad permutes; the nearest is entered, the others go to the quad's LDS
//    stack far-first with their entry parameters;
//  * a leaf holds up to four triangles: each lane runs the exact fp64 test of one, the quad reduces.
// Compared with one ray per lane this cuts the dependent steps of a ray (4 boxes / 4 triangles per
// step), divides the lanes whose slowest member a wave waits for by four, and turns 64 scattered
// 64-byte gathers per wave-load into 16 full lines.  Every branch that encloses a cross-lane
// operation is quad-uniform by construction (ray state is replicated bit-identically in the 4 lanes).
// A directional light's shadow rays all have the same direction, so their direction in a shape's frame - normalized(inv * d):
// two dozen multiply-adds, a square root and three divisions, ~96 instructions per (ray, shape) - is a property of the (light,
// shape) pair: k_light_dirs computes it once at upload with the functions the kernels would use (SceneView::shadow_ld), and a
// shadow query towards such a light reads its three doubles through the scalar cache.  `row`: the light's row of the table,
// or nullptr (closest-hit queries, point lights).
typedef const __attribute__((address_space(4))) double rt_uniform_double;
__device__ __forceinline__ const double *shadow_ld_row(const SceneView &sv, uint32_t light, int32_t ltype) {
    return (ltype == RTHIP_LIGHT_DIRECTIONAL && sv.shadow_ld) ? sv.shadow_ld + (size_t)light * sv.n_shapes * 3u : nullptr;
}
__device__ __forceinline__ V3 local_dir(const double *row, uint32_t s, const double *inv, V3 d) {
    if (row) {  // (wave-uniform)
        rt_uniform_double *t = (rt_uniform_double *)row + 3u * s;
        return mk(t[0], t[1], t[2]);
    }
    return normalized(xform(inv, d, 0.0));
}
__global__ void k_light_dirs(SceneView sv, double *out, double *plane_wn) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < sv.n_shapes) {  // SceneView::plane_wn: hit_details' normal of a plane, the same for every hit
        const rthip_shape &sh = sv.shapes[k];
        V3 r = mk(0, 0, 0);
        if (sh.type == RTHIP_SHAPE_PLANE) r = normalized(xform(sh.nrm, mk(sh.plane_n[0], sh.plane_n[1], sh.plane_n[2]), 0.0));
        plane_wn[3 * (size_t)k] = r.x;
        plane_wn[3 * (size_t)k + 1] = r.y;
        plane_wn[3 * (size_t)k + 2] = r.z;
    }
    if (k >= sv.n_lights * sv.n_shapes) return;
    const uint32_t l = k / sv.n_shapes, s = k % sv.n_shapes;
    const rthip_light &light = sv.lights[l];
    V3 r = mk(0, 0, 0);
    if (light.type == RTHIP_LIGHT_DIRECTIONAL) {
        double inv[12];
        for (int j = 0; j < 12; j++) inv[j] = sv.shapes[s].inv[j];
        r = normalized(xform(inv, mk(light.direction[0], light.direction[1], light.direction[2]), 0.0));
    }
    out[3 * (size_t)k] = r.x;
    out[3 * (size_t)k + 1] = r.y;
    out[3 * (size_t)k + 2] = r.z;
}

namespace quad {
constexpr int kXor1 = 0xB1, kXor2 = 0x4E, kRot1 = 0x39, kRot3 = 0x93;  // quad_perm [1,0,3,2] [2,3,0,1] [1,2,3,0] [3,0,1,2]
static const uint32_t kRaysPerBlock = kBlock / 4;

template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ double dpp(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    uint32_t lo = dpp<CTRL>((uint32_t)b), hi = dpp<CTRL>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// The quad's stack of pending children in LDS: 8-byte entries {child word, entry parameter}, sized by the
// tree (3 * (depth + 1) entries can never overflow: a step pushes at most three children and descends one
// level), plus one dummy entry that absorbs the writes of lanes with nothing to push.
struct Stack {
    uint2 *e;
    uint32_t cap;  // usable entries; e[cap] is the dummy
};

// lexicographic min of (d, id) over the quad; id = 0xffffffff means "nothing"
__device__ __forceinline__ void reduce_min(double &d, uint32_t &id) {
    {
        double od = dpp<kXor1>(d);
        uint32_t oi = dpp<kXor1>(id);
        if (od < d || (od == d && oi < id)) { d = od; id = oi; }
    }
    {
        double od = dpp<kXor2>(d);
        uint32_t oi = dpp<kXor2>(id);
        if (od < d || (od == d && oi < id)) { d = od; id = oi; }
    }
}
// the same, and `tie` is raised (in all four lanes) when two lanes hold different ids at the winning d
__device__ __forceinline__ void reduce_min_tie(double &d, uint32_t &id, bool &tie) {
    const double d0 = d;
    const uint32_t id0 = id;
    reduce_min(d, id);
    uint32_t t = (id0 != 0xffffffffu && d0 == d && id0 != id) ? 1u : 0u;  // this lane lost at equal distance
    t |= dpp<kXor1>(t);
    t |= dpp<kXor2>(t);
    tie = t != 0u;
}

// Scene.cpp:421-467 + :495-507 for one triangle: accepted? and the reference's d = |O - P|^2
__device__ __forceinline__ bool tri_eval(const rthip_tri &t, V3 o, V3 d, double &dd) {
    V3 v0 = mk(t.v0[0], t.v0[1], t.v0[2]);
    V3 nf = mk(t.face_normal[0], t.face_normal[1], t.face_normal[2]);
    double distance = dot(sub(o, v0), nf);
    if (distance < 0) return false;
    double cosine = dot(d, nf);
    if (cosine >= 0) return false;
    V3 projection = add(o, scale(d, distance / -cosine));
    V3 v2 = sub(projection, v0);
    double d20 = dot(v2, mk(t.e0[0], t.e0[1], t.e0[2]));
    double d21 = dot(v2, mk(t.e1[0], t.e1[1], t.e1[2]));
    double v = (t.d11 * d20 - t.d01 * d21) * t.inv_den;
    double w = (t.d00 * d21 - t.d01 * d20) * t.inv_den;
    double u = 1.0 - v - w;
    if (!(u >= 0 && v >= 0 && w >= 0)) return false;
    dd = sqlen(sub(o, projection));
    return true;
}

// rtc::bvh_slab with the near / far plane of every axis picked by the sign of the ray direction (a select
// instead of a min and a max) and the widening folded into the constants: near = fma(plane_near, inv, -(c + s)),
// far = fma(plane_far, inv, s - c).  Same error budget as rtc::BvhRay (one rounding fewer per parameter).
struct SlabRay {
    float ix, iy, iz;
    float nx, ny, nz;  // -(o*inv + widening)
    float fx, fy, fz;  // widening - o*inv
    bool sx, sy, sz;   // direction component negative: the near plane is `hi`
};
__device__ __forceinline__ SlabRay slab_ray(const BvhRay &r) {
    SlabRay s;
    s.ix = r.ix; s.iy = r.iy; s.iz = r.iz;
    s.nx = -(r.cx + r.sx); s.ny = -(r.cy + r.sy); s.nz = -(r.cz + r.sz);
    s.fx = r.sx - r.cx; s.fy = r.sy - r.cy; s.fz = r.sz - r.cz;
    s.sx = r.ix < 0.0f; s.sy = r.iy < 0.0f; s.sz = r.iz < 0.0f;
    return s;
}
__device__ __forceinline__ bool slab_hit(const float *lo, const float *hi, const SlabRay &s, float tmax, float &tn) {
    const float ax = __builtin_fmaf(s.sx ? hi[0] : lo[0], s.ix, s.nx), bx = __builtin_fmaf(s.sx ? lo[0] : hi[0], s.ix, s.fx);
    const float ay = __builtin_fmaf(s.sy ? hi[1] : lo[1], s.iy, s.ny), by = __builtin_fmaf(s.sy ? lo[1] : hi[1], s.iy, s.fy);
    const float az = __builtin_fmaf(s.sz ? hi[2] : lo[2], s.iz, s.nz), bz = __builtin_fmaf(s.sz ? lo[2] : hi[2], s.iz, s.fz);
    tn = fmaxf(fmaxf(ax, ay), fmaxf(az, 0.0f));
    const float tf = fminf(fminf(bx, by), fminf(bz, tmax));
    return tn <= tf;
}

template <int CTRL> __device__ __forceinline__ uint32_t min_dpp(uint32_t v) {
    const uint32_t o = dpp<CTRL>(v);
    return o < v ? o : v;
}
// The interior loop is written branch-free (selects, a dummy stack slot, a speculative read of the stack
// top): with 16 quads in different phases every branch of a step would be taken by some quad anyway, and
// a lone wave pays ~20 cycles per taken branch.  A quad whose popped entry is culled carries kSkip for
// one step (its box tests are forced to miss and it pops again).
template <bool COUNT>
__device__ __forceinline__ void mesh_closest(const SceneView &sv, const rthip_mesh &m, V3 o, V3 d, Stack qs, uint32_t q,
                                             float t_stop, TriBest &best, KdCounters &cnt) {
    best.hit = false;
    best.tie = false;
    best.ext = 0.0;
    best.d = DBL_MAX;
    best.tri = 0xffffffffu;
    best.ref = 0xffffffffu;
    if (m.bvh_node_count == 0) return;
    const char *nodes = reinterpret_cast<const char *>(sv.bvh_nodes + m.bvh_node_base) + (q << 5);
    const rthip_tri *tris = sv.bvh_tris + m.bvh_tri_base;
    const uint32_t *index = sv.bvh_tri_index + m.bvh_tri_base;
    const SlabRay r = slab_ray(bvh_ray(o, d, m.abs_max));
    const uint32_t kSkip = 0x7ffffffeu;              // "pop again": no leaf bit, never a node index
    const uint32_t kDone = RTHIP_BVH_ABSENT;         // has the leaf bit: leaves the interior loop
    const uint32_t kNone = 0xfffffffcu;
    const uint32_t qbit = 1u << q, below = qbit - 1u;
    float tmax = 3.0e38f;
    uint32_t n = 0;
    uint32_t cur = 0;
    for (;;) {
        // The descent is left - for a leaf step of the quads that wait at one - when nobody descends any more, or as soon as
        // twice as many quads wait at a leaf as still descend (waiting for the last descending quad kept most of a wave
        // idle: config 4 -1.6 %, a lone config-3 frame -2.7 %; leaving at 1 : 1 or at 3 : 1 gains less).  Wave-uniform loop,
        // the quads that are at a leaf or done skip the step.
        for (;;) {
            const bool interior = !(cur & RTHIP_BVH_LEAF);
            const unsigned long long descending = __ballot(interior);
            if (!descending) break;
            const unsigned long long waiting = __ballot(!interior && cur != kDone);
            if ((uint32_t)__popcll(waiting) >= 2u * (uint32_t)__popcll(descending)) break;
            if (interior) {  // (an `if`, not a `continue`: the compiler keeps the step's code as it was)
                const bool valid = cur != kSkip;
                const uint2 top = qs.e[(n ? n : 1u) - 1u];  // speculative: only used when no child is hit
                const uint4 *p = reinterpret_cast<const uint4 *>(nodes + ((size_t)(valid ? cur : 0u) << 7));
                const long long tm0 = COUNT ? clock64() : 0;
                const uint4 A = p[0], B = p[1];  // lo.xyz hi.x | hi.yz child pad
                if (COUNT) {
                    asm volatile("" ::"v"(A.x), "v"(B.x));  // the loads have landed
                    const uint32_t first = rt_first_active();
                    cnt.fetch_cycles += first ? (uint32_t)(clock64() - tm0) : 0u;
                    cnt.nodes += (q == 0 && valid);
                    cnt.pops += (q == 0 && !valid);
                    cnt.wave_steps += first;
                }
                const float lo[3] = {__uint_as_float(A.x), __uint_as_float(A.y), __uint_as_float(A.z)};
                const float hi[3] = {__uint_as_float(A.w), __uint_as_float(B.x), __uint_as_float(B.y)};
                const uint32_t child = B.z;
                float tn;
                const bool h = slab_hit(lo, hi, r, tmax, tn) && valid && child != RTHIP_BVH_ABSENT;
                const uint32_t key = h ? ((__float_as_uint(tn) & ~3u) | q) : (kNone | q);  // tn >= 0: bit order == value order
                uint32_t kmin = min_dpp<kXor1>(key);
                kmin = min_dpp<kXor2>(kmin);
                uint32_t hm = h ? qbit : 0u;  // which lanes of the quad hit
                hm |= dpp<kXor1>(hm);
                hm |= dpp<kXor2>(hm);
                const bool nearest = key == kmin;
                uint32_t sel = (h && nearest) ? child : 0u;  // node indices are >= 1 below the root, leaf words have bit 31
                sel |= dpp<kXor1>(sel);
                sel |= dpp<kXor2>(sel);
                // the other hit children go to the stack (lane order); lanes with nothing to push write the dummy
                const uint32_t others = hm & ~(1u << (kmin & 3u));
                const bool push = h && !nearest;
                const uint32_t slot = push ? n + (uint32_t)__popc(others & below) : qs.cap;
                qs.e[slot] = make_uint2(child, __float_as_uint(tn));
                const uint32_t nh = (uint32_t)__popc(hm);
                const bool popped_ok = __uint_as_float(top.y) <= tmax;
                const uint32_t after_pop = n ? (popped_ok ? top.x : kSkip) : kDone;
                cur = nh ? sel : after_pop;
                n = nh ? n + nh - 1u : (n ? n - 1u : 0u);
            }
        }
        if (!__ballot(cur != kDone)) break;  // (the wave leaves together: its loops are wave-uniform)
        if ((cur & RTHIP_BVH_LEAF) && cur != kDone) {
        {   // leaf: one triangle per lane
            const uint32_t first = cur & RTHIP_BVH_FIRST_MASK, count = ((cur >> RTHIP_BVH_COUNT_SHIFT) & 7u) + 1u;
            if (COUNT) {
                cnt.leaves += (q == 0);
                cnt.tris += (q == 0) ? count : 0u;
                cnt.wave_tris += rt_first_active();
            }
            double dd = DBL_MAX;
            uint32_t ti = 0xffffffffu;
            if (q < count) {
                const uint32_t id = index[first + q];  // fetched with the triangle, not after the test: one round trip per leaf step
                double e;
                if (tri_eval(tris[first + q], o, d, e)) {
                    dd = e;
                    ti = id;
                }
            }
            bool leaf_tie;
            reduce_min_tie(dd, ti, leaf_tie);
            // strict '<' of Scene.cpp:479.  Two different triangles accepted at the same distance: the lower index is kept
            // for now and the tie recorded - a closest-hit query that ends with one goes to the redo kernel, whose walk
            // settles it in the reference's order (rt_core.hpp: mesh_closest_reference_order)
            if (ti != 0xffffffffu && dd == best.d && (ti != best.tri || leaf_tie)) best.tie = true;
            if (ti != 0xffffffffu && (dd < best.d || (dd == best.d && ti < best.tri))) {
                if (dd < best.d) best.tie = leaf_tie;
                best.d = dd;
                best.tri = ti;
                best.ref = ti;
                best.hit = true;
                tmax = fminf(bvh_tmax_of(dd), t_stop);
            }
        }
        {   // next pending child, two entries at once: after a hit has pulled tmax in, the top of the stack is often culled, and a
            // culled entry popped alone costs the quad a whole node step (config 4 -0.4 %, a lone config-3 frame -0.7 %; four at
            // once: config 4 -0.8 %, 8-frame calls of config 3 +0.7 %).  What is still culled is dropped by the interior loop.
            const uint2 top = qs.e[(n ? n : 1u) - 1u];
            const uint2 under = qs.e[(n > 1u ? n : 2u) - 2u];
            const bool ok1 = __uint_as_float(top.y) <= tmax, ok2 = __uint_as_float(under.y) <= tmax;
            const uint32_t second = n > 1u ? (ok2 ? under.x : kSkip) : kDone;
            cur = n ? (ok1 ? top.x : second) : kDone;
            n = n ? (ok1 ? n - 1u : (n > 1u ? n - 2u : 0u)) : 0u;
        }
        }
    }
}

// The world-space ray a quad is casting is PARKED in LDS (behind the stacks, kParkStride bytes per quad) and read back
// where it is needed: before each mesh (world -> local), after a walk (distance of the hit), for the hit details.  The
// walk's loop then holds the mesh-local ray only; with the world ray (and the loop-carried copies a shared closest /
// shadow code path needs) in registers the kernel took 128 VGPRs = 4 waves per SIMD, and a config-4 frame is 17 % slower
// at 3 waves than at 4: it is resident waves that hide the dependent node fetches.  The stack stores of the walk may
// alias the parked words as far as the compiler knows, so it cannot keep them in registers across a walk.
static const uint32_t kParkStride = 112;  // bytes: 12 doubles, padded so that 16 quads' 16-byte reads fall in distinct banks
__host__ __device__ __forceinline__ uint32_t park_offset(uint32_t stack_cap) {  // the stacks, rounded up to 16 bytes
    return (kRaysPerBlock * (stack_cap + 1u) * 8u + 15u) & ~15u;
}
struct Park {
    double *p;  // [0..2] origin, [3..5] direction
    // all four lanes store (the same values to the same words): a lane then reads back what IT wrote, so nothing
    // here depends on another lane's store being visible (the compiler reasons per thread and may forward a lane's
    // own older value past a store it did not execute)
    __device__ __forceinline__ void put_o(V3 o) const { p[0] = o.x; p[1] = o.y; p[2] = o.z; }
    __device__ __forceinline__ void put_d(V3 d) const { p[3] = d.x; p[4] = d.y; p[5] = d.z; }
    __device__ __forceinline__ V3 o() const { return mk(p[0], p[1], p[2]); }
    __device__ __forceinline__ V3 d() const { return mk(p[3], p[4], p[5]); }
};

// Scene::CastRay (Scene.cpp:163-237) for the quad's parked ray; same contract as rtc::cast_ray_id.
// `known`: the query over the analytic shapes has been answered already (k_trace_easy, closest-hit query only).
template <bool COUNT>
__device__ __forceinline__ int32_t cast_ray(const SceneView &sv, Park park, Stack qs, uint32_t q, bool any_hit, bool active,
                                            const PendRec *known, int32_t &best_tri, bool &redo, KdCounters &cnt, const double *ld_row = nullptr) {
    double bd = known ? known->bd : DBL_MAX;
    uint32_t bs = known ? known->bs : 0xffffffffu;
    best_tri = -1;
    bool meshes = known != nullptr;  // (the quad kernel only runs on scenes with meshes)
    if (!known) {
        // analytic shapes, one per lane and pass; the earlier node wins exact distance ties (Scene.cpp:225)
        const V3 o = park.o(), d = park.d();
        for (uint32_t s0 = 0; s0 < sv.n_shapes; s0 += 4) {
            const uint32_t s = s0 + q;
            if (!active || s >= sv.n_shapes) continue;
            const rthip_shape &sh = sv.shapes[s];
            const int32_t type = sh.type;
            if (type == RTHIP_SHAPE_MESH) {
                meshes = true;
                continue;
            }
            if (COUNT) cnt.shapes++;
            if (type == RTHIP_SHAPE_SPHERE && ball_missed(sh.bound_c[0], sh.bound_c[1], sh.bound_c[2], sh.bound_r2, o, d)) continue;
            if (COUNT) cnt.shapes_full++;
            V3 lo = xform(sh.inv, o, 1.0);
            V3 ld = ld_row ? mk(ld_row[3u * s], ld_row[3u * s + 1u], ld_row[3u * s + 2u]) : normalized(xform(sh.inv, d, 0.0));  // (see local_dir; s differs per lane here)
            V3 p;
            bool hit = type == RTHIP_SHAPE_SPHERE ? sphere_intersect(lo, ld, p)
                                                  : plane_intersect(mk(sh.plane_n[0], sh.plane_n[1], sh.plane_n[2]), lo, ld, p);
            if (!hit) continue;
            V3 worldPoint = xform(sh.fwd, p, 1.0);
            double distance = sqlen(sub(o, worldPoint));
            if (!(distance >= DBL_EPSILON)) continue;
            if (!(distance < bd)) continue;
            bd = distance;
            bs = s;
        }
        reduce_min(bd, bs);
        uint32_t mm = meshes ? 1u : 0u;
        mm |= dpp<kXor1>(mm);
        mm |= dpp<kXor2>(mm);
        meshes = mm != 0;
    }
    if (!meshes) return (int32_t)bs;
    // meshes, one after the other, all four lanes together (the shape index is wave-uniform: scalar loads)
    for (uint32_t s = 0; s < sv.n_shapes; s++) {
        const rt_uniform_shape *sh = rt_uniform_cast(sv.shapes) + s;
        if (sh->type != RTHIP_SHAPE_MESH) continue;
        if (!active || redo || (any_hit && bs != 0xffffffffu)) continue;
        if (COUNT) cnt.shapes += (q == 0);
        V3 lo, ld;
        float t_stop = 3.0e38f;
        {
            const V3 o = park.o(), d = park.d();
            if (ball_missed(sh->bound_c[0], sh->bound_c[1], sh->bound_c[2], sh->bound_r2, o, d)) continue;
            if (COUNT) cnt.shapes_full += (q == 0);
            double inv[12];
#pragma unroll
            for (int k = 0; k < 12; k++) inv[k] = sh->inv[k];
            lo = xform(inv, o, 1.0);
            ld = local_dir(ld_row, s, inv, d);
        }
        const rthip_mesh m = rt_load_mesh(sv.meshes + sh->mesh);
        if (kd_degenerate_ray(sv, m, lo, ld)) {
            redo = true;
            continue;
        }
        if (any_hit) {
            double fwd[12];
#pragma unroll
            for (int k = 0; k < 12; k++) fwd[k] = sh->fwd[k];
            t_stop = anyhit_t_stop(fwd, sh->rt_lin, sh->rt_t, park.o(), ld);
        }
        TriBest tb;
        mesh_closest<COUNT>(sv, m, lo, ld, qs, q, t_stop, tb, cnt);
        if (!tb.hit) continue;
        if (tb.tie && !any_hit) {  // (a shadow query only consumes the distance)
            redo = true;
            continue;
        }
        V3 p;
        double u, v, w;
        tri_point(sv.tris[m.tri_base + tb.tri], lo, ld, p, u, v, w);
        double fwd[12];
#pragma unroll
        for (int k = 0; k < 12; k++) fwd[k] = sh->fwd[k];
        V3 worldPoint = xform(fwd, p, 1.0);
        double distance = sqlen(sub(park.o(), worldPoint));
        if (!(distance >= DBL_EPSILON)) continue;
        if (distance < bd || (distance == bd && s < bs)) {
            bd = distance;
            bs = s;
            best_tri = (int32_t)tb.tri;
        }
    }
    return (int32_t)bs;
}

// rtc::trace_stage for the quad's ray (parked by the caller).
// The hit record is stored as soon as the closest query is resolved (mask 0), so that the local point and the
// normal do not stay in registers through the shadow walks; the caller patches the shadow mask in afterwards.
template <bool COUNT>
__device__ __forceinline__ bool trace_stage(const SceneView &sv, Park park, uint32_t n_shadow_lights, Stack qs, uint32_t q,
                                            bool live, const Work &w, uint32_t i, const PendRec &analytic, int32_t &out_shape,
                                            int32_t &out_tri, uint32_t &out_mask, KdCounters &cnt) {
    bool redo = false;
    out_shape = -1;
    out_tri = -1;
    out_mask = 0u;
    {   // the closest hit
        int32_t tri;
        const int32_t shape = cast_ray<COUNT>(sv, park, qs, q, false, live, &analytic, tri, redo, cnt);
        if (shape >= 0 && !redo) {
            Hit c;
            hit_details(sv, park.o(), park.d(), shape, tri, c, (COUNT && q == 0) ? &cnt : nullptr);
            out_shape = shape;
            out_tri = tri;
            if (q == 0) {
                HitRec rec;
                rec.shape = shape;
                rec.tri = tri;
                rec.shadow_mask = 0;
                rec.local = c.local;
                rec.normal = c.normal;
                store_hit(w, i, rec);
            }
            park.put_o(c.point);  // every shadow ray starts there
        }
    }
    for (uint32_t l = 0; l < n_shadow_lights; l++) {
        const rt_uniform_light *light = rt_uniform_cast(sv.lights) + l;
        const int32_t ltype = light->type;
        if (!light->enabled || ltype == RTHIP_LIGHT_AMBIENT) continue;  // Raytracer.cpp:213, 234-237
        const bool active = live && !redo && out_shape >= 0;
        V3 rd;
        if (ltype == RTHIP_LIGHT_DIRECTIONAL)
            rd = mk(light->direction[0], light->direction[1], light->direction[2]);
        else
            rd = normalized(sub(mk(light->position[0], light->position[1], light->position[2]), park.o()));
        park.put_d(rd);
        int32_t tri;
        const int32_t shape = cast_ray<COUNT>(sv, park, qs, q, true, active, nullptr, tri, redo, cnt, shadow_ld_row(sv, l, ltype));
        if (shape >= 0) out_mask |= 1u << l;
    }
    return !redo;
}
}  // namespace quad

// ---------------------------------------------------------------------------- shading + spawn
// Stage 2 of a level for ONE chunk of a workgroup (called by every thread of the workgroup): material /
// noise / background, Lambert terms with the shadow mask (R4, R5, M*, N*, B*), and the reflected rays,
// compacted into the next level's queue: wave ballot + prefix popcount, the four waves of a workgroup
// combined through LDS so that ONE returning atomic per workgroup reserves the queue range.
// `have`: this lane holds a traced ray (index i in the level, hit record h, direction d).
// HARD (k_trace_easy): the same reservation step also appends the workgroup's hard rays to the hard list - with one
// atomic per WAVE, a 4K frame whose rays mostly need the mesh sent 130 K returning atomics to one address per level,
// ~0.7 ms of serialisation in L2.
struct ShadeShared {
    uint32_t wave_cnt[kBlock / 64];
    uint32_t block_base;
    uint32_t hard_cnt[kBlock / 64];
    uint32_t hard_base;
    uint32_t part[2][kBlock / 64];
    uint32_t irq;  // Work::irq as thread 0 read it: the whole workgroup leaves together (its iterations contain barriers)
};
template <bool HARD, bool LEAN = false>
__device__ __forceinline__ void shade_and_spawn(const SceneView &sv, const FrameParams &fp, const Work &w, const uint8_t *tabs,
                                                ShadeShared &sh, uint32_t level, bool have, uint32_t i, const HitRec &h, V3 d,
                                                uint32_t &shadow_total, uint32_t &spawn_total, bool is_hard = false,
                                                const PendRec *analytic = nullptr, V3 o = V3{0, 0, 0}) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    ShadeOut so;
    so.spawn = false;
    so.shadow_rays = 0;
    if (have) {
        if (fp.renderer == RTHIP_RENDER_RAYTRACER) {
            shade_stage<LEAN>(sv, tabs, h, d, level < fp.max_rec, so);
        } else {
            so.color = debug_shade<LEAN>(sv, tabs, h, d, fp.renderer - 1);
            so.refl = 0.0;
        }
        ColRec cr;
        cr.r = so.color.r; cr.g = so.color.g; cr.b = so.color.b; cr.refl = so.refl;
        rec_at(w, level, i) = cr;
    }
    if (fp.count_kd == 1) {  // counting context: what was shaded, for the flop accounting of this kernel (bench.py); one atomic per
                             // wave and distinct shape (a 4K frame of sky sent millions of atomics to one address otherwise)
        const uint32_t key = h.shape < 0 ? kShadeCountShapes : min((uint32_t)h.shape, kShadeCountShapes - 1u);
        unsigned long long todo = __ballot(have);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t k = (uint32_t)__shfl((int)key, leader, 64);
            const unsigned long long same = __ballot(have && key == k);
            if ((int)lane == leader) atomicAdd(&w.stats[ST_COUNT + (HARD ? 0u : kShadeCounts) + k], (unsigned long long)__popcll(same));
            todo &= ~same;
        }
    }
    const unsigned long long mask = __ballot(so.spawn);
    const unsigned long long hmask = HARD ? __ballot(is_hard) : 0ull;
    if (lane == 0) {
        sh.wave_cnt[wid] = (uint32_t)__popcll(mask);
        if (HARD) sh.hard_cnt[wid] = (uint32_t)__popcll(hmask);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = sh.wave_cnt[0] + sh.wave_cnt[1] + sh.wave_cnt[2] + sh.wave_cnt[3];
        sh.block_base = total ? atomicAdd(&level_count(w, level + 1), total) : 0u;
    }
    if (HARD && threadIdx.x == 64) {  // (another wave: the two atomics are in flight together)
        uint32_t total = sh.hard_cnt[0] + sh.hard_cnt[1] + sh.hard_cnt[2] + sh.hard_cnt[3];
        sh.hard_base = total ? atomicAdd(&w.hard_count[0], total) : 0u;
    }
    __syncthreads();
    if (HARD && is_hard) {  // listed for the quad kernel (which traces it) and k_shade (which shades it)
        uint32_t slot = sh.hard_base + (uint32_t)__popcll(hmask & ((1ull << lane) - 1ull));
        for (uint32_t k = 0; k < wid; k++) slot += sh.hard_cnt[k];
        if (slot < w.N) {  // (always, unless the list counter was not zero when the level began)
            w.hard[slot] = i;
            w.pend[slot] = *analytic;
        }
        if (level == 0) store_ray(w, 0, i, o, d, 0u);  // k_shade needs the direction (background, reflection)
    }
    uint32_t child = kNoChild;
    if (so.spawn) {
        uint32_t j = sh.block_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        for (uint32_t k = 0; k < wid; k++) j += sh.wave_cnt[k];
        if (j < w.N) {  // (always, unless the level counter was not zero when the frame began)
            store_ray(w, level + 1, j, so.child_o, so.child_d, i);
            child = j;
        }
        spawn_total++;
    }
    if (have) w.child[(size_t)level * w.N + i] = child;  // the chain k_resolve folds back to front
    shadow_total += so.shadow_rays;
    __syncthreads();  // wave_cnt / block_base are reused by the next chunk
}
// ray statistics without atomics (32 K same-address atomics cost more than the whole shading): one plain
// read-modify-write per workgroup into its own slot (`which`: 0 = k_trace_easy, 1 = k_shade), summed on
// the host on demand
__device__ __forceinline__ void shade_stats_flush(const Work &w, ShadeShared &sh, uint32_t which, uint32_t level, uint32_t levels,
                                                  uint32_t shadow_total, uint32_t spawn_total) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    shadow_total = wave_sum(shadow_total);
    spawn_total = wave_sum(spawn_total);
    if (lane == 0) {
        sh.part[0][wid] = shadow_total;
        sh.part[1][wid] = spawn_total;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint2 *slot = w.block_stats + ((size_t)which * levels + level) * w.bs_stride + blockIdx.x;
        uint2 v = *slot;
        v.x += sh.part[0][0] + sh.part[0][1] + sh.part[0][2] + sh.part[0][3];
        v.y += sh.part[1][0] + sh.part[1][1] + sh.part[1][2] + sh.part[1][3];
        *slot = v;
    }
}
// The Perlin tables of the scene in LDS (their chained lookups are the latency chain of every sky / marble sample), 256 bytes each
__device__ __forceinline__ void stage_perlin_tables(const SceneView &sv, uint32_t n_tables) {
    const uint32_t words = n_tables * 64;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sv.perlin_tables);
    for (uint32_t k = threadIdx.x; k < words; k += kBlock) lds_dyn[k] = src[k];
    __syncthreads();
}

// ---------------------------------------------------------------------------- easy rays
// Stage 1a of a level for the BVH path (accel = 0), one ray per lane: every query of the ray over the
// analytic shapes, and for each mesh only a conservative test of the ray against the four root boxes
// of its BVH.  A ray none of whose queries can touch a mesh is finished here (most of a frame: sky,
// ground plane, spheres) in a kernel with no divergent loop at all; the others are listed, compacted
// with a wave ballot + one atomic per wave, for the quad kernel, which traces them from scratch.
namespace easy {
// could the ray (local frame) hit any triangle of the mesh?  fp32, conservative (see rtc::bvh_slab)
__device__ __forceinline__ bool mesh_may_hit(const SceneView &sv, const rthip_mesh &m, V3 lo, V3 ld) {
    if (m.bvh_node_count == 0) return false;
    const BvhRay r = bvh_ray(lo, ld, m.abs_max);
    typedef const __attribute__((address_space(4))) rthip_bvhnode uniform_node;
    uniform_node *nd = (uniform_node *)(sv.bvh_nodes + m.bvh_node_base);  // the root: wave-uniform address
    bool any = false;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float blo[3] = {nd->c[c].lo[0], nd->c[c].lo[1], nd->c[c].lo[2]};
        const float bhi[3] = {nd->c[c].hi[0], nd->c[c].hi[1], nd->c[c].hi[2]};
        const uint32_t child = nd->c[c].child;
        float tn;
        bool h;
        bvh_slab(blo, bhi, r, 3.0e38f, tn, h);
        h = h && child != RTHIP_BVH_ABSENT;
        // second level: an interior child only counts if the ray also meets one of ITS four boxes (the node is the
        // same for every lane: scalar loads; skipped when no lane of the wave reaches it)
        if (!(child & RTHIP_BVH_LEAF) && __ballot(h)) {
            uniform_node *n2 = nd + child;
            bool any2 = false;
#pragma unroll
            for (int c2 = 0; c2 < 4; c2++) {
                const float lo2[3] = {n2->c[c2].lo[0], n2->c[c2].lo[1], n2->c[c2].lo[2]};
                const float hi2[3] = {n2->c[c2].hi[0], n2->c[c2].hi[1], n2->c[c2].hi[2]};
                float tn2;
                bool h2;
                bvh_slab(lo2, hi2, r, 3.0e38f, tn2, h2);
                any2 = any2 || (h2 && n2->c[c2].child != RTHIP_BVH_ABSENT);
            }
            h = h && any2;
        }
        any = any || h;
    }
    return any;
}

// rtc::cast_ray_id restricted to the analytic shapes; `hard` is set when a mesh would have to be walked
template <bool COUNT>
__device__ __forceinline__ int32_t cast_ray(const SceneView &sv, V3 o, V3 d, bool any_hit, bool active, bool &hard, double &hitDistance,
                                            KdCounters &cnt, const double *ld_row, V3 &best_local) {
    int32_t best = -1;
    hitDistance = DBL_MAX;
    bool mesh_box = false;
    for (uint32_t s = 0; s < sv.n_shapes; s++) {
        const rt_uniform_shape *sh = rt_uniform_cast(sv.shapes) + s;
        if (!active) continue;
        if (COUNT) cnt.shapes++;
        const int32_t type = sh->type;
        if (type != RTHIP_SHAPE_PLANE && ball_missed(sh->bound_c[0], sh->bound_c[1], sh->bound_c[2], sh->bound_r2, o, d)) continue;
        if (COUNT) cnt.shapes_full++;
        double inv[12];
#pragma unroll
        for (int k = 0; k < 12; k++) inv[k] = sh->inv[k];
        V3 lo = xform(inv, o, 1.0);
        V3 ld = local_dir(ld_row, s, inv, d);
        if (type == RTHIP_SHAPE_MESH) {
            if (!mesh_box) mesh_box = mesh_may_hit(sv, rt_load_mesh(sv.meshes + sh->mesh), lo, ld);
            continue;
        }
        V3 p;
        const bool hit = type == RTHIP_SHAPE_SPHERE ? sphere_intersect(lo, ld, p)
                                                    : plane_intersect(mk(sh->plane_n[0], sh->plane_n[1], sh->plane_n[2]), lo, ld, p);
        if (!hit) continue;
        double fwd[12];
#pragma unroll
        for (int k = 0; k < 12; k++) fwd[k] = sh->fwd[k];
        V3 worldPoint = xform(fwd, p, 1.0);
        double distance = sqlen(sub(o, worldPoint));
        if (!(distance >= DBL_EPSILON)) continue;
        if (!(distance < hitDistance)) continue;
        hitDistance = distance;
        best = (int32_t)s;
        best_local = p;  // (the closest query's hit record needs it: trace_stage)
    }
    // an any-hit query already answered by an analytic shape does not need the mesh (only .Hit is consumed)
    if (mesh_box && !(any_hit && best >= 0)) hard = true;
    return best;
}

template <bool COUNT>
__device__ __forceinline__ bool trace_stage(const SceneView &sv, V3 o, V3 d, uint32_t n_shadow_lights, bool live, HitRec &h,
                                            PendRec &analytic, KdCounters &cnt) {
    bool hard = false;
    analytic.bd = DBL_MAX;
    analytic.bs = 0xffffffffu;
    analytic.pad = 0;
    h.shape = -1;
    h.tri = -1;
    h.shadow_mask = 0;
    h.local = h.normal = mk(0, 0, 0);
    V3 point = mk(0, 0, 0);
    for (int32_t l = -1; l < (int32_t)n_shadow_lights; l++) {
        V3 ro = o, rd = d;
        bool active = live && !hard;
        const double *ld_row = nullptr;
        if (l >= 0) {
            const rt_uniform_light *light = rt_uniform_cast(sv.lights) + l;
            const int32_t ltype = light->type;
            if (!light->enabled || ltype == RTHIP_LIGHT_AMBIENT) continue;  // Raytracer.cpp:213, 234-237
            active = active && h.shape >= 0;
            ld_row = shadow_ld_row(sv, (uint32_t)l, ltype);
            ro = point;
            if (ltype == RTHIP_LIGHT_DIRECTIONAL)
                rd = mk(light->direction[0], light->direction[1], light->direction[2]);
            else
                rd = normalized(sub(mk(light->position[0], light->position[1], light->position[2]), point));
        }
        double distance;
        V3 local = mk(0, 0, 0);
        const int32_t shape = cast_ray<COUNT>(sv, ro, rd, l >= 0, active, hard, distance, cnt, ld_row, local);
        if (l < 0) {
            if (shape >= 0) {
                analytic.bd = distance;
                analytic.bs = (uint32_t)shape;
            }
            if (shape >= 0 && !hard) {
                // rtc::hit_details for an analytic shape, from the local hit point the query has just computed (hit_details would
                // transform and normalise the ray and intersect the shape again: the same operations on the same operands, ~155
                // instructions per hit)
                const rthip_shape &sh = sv.shapes[shape];
                h.shape = shape;
                h.local = local;
                point = xform(sh.fwd, local, 1.0);
                if (sh.type == RTHIP_SHAPE_PLANE && sv.plane_wn) {
                    const double *n = sv.plane_wn + 3 * (size_t)shape;
                    h.normal = mk(n[0], n[1], n[2]);
                } else {
                    const V3 ln = sh.type == RTHIP_SHAPE_SPHERE ? normalized(local) : mk(sh.plane_n[0], sh.plane_n[1], sh.plane_n[2]);
                    h.normal = normalized(xform(sh.nrm, ln, 0.0));
                }
            }
        } else if (shape >= 0) {
            h.shadow_mask |= 1u << l;
        }
    }
    return !hard;
}
}  // namespace easy

// Also stage 2 for the rays it finishes: they are shaded on the spot (no hit record leaves the
// registers) and their reflected rays go straight to the next level's queue.
// LEAN: the scene's materials need neither Worley / Marble noise nor the material stack (rtc::noise_sample)
template <bool COUNT, bool LEAN = false>
__global__ __launch_bounds__(kBlock, RT_EASY_OCC) void k_trace_easy(SceneView sv, FrameParams fp, Work w, uint32_t level,
                                                                      uint32_t n_tables, uint32_t levels) {
    __shared__ ShadeShared sh;
    const uint32_t n = min(w.N, (level == 0) ? fp.N0 : level_count(w, level));
    stage_store(w, w.stage_base + STAGES_PER_LEVEL * level + STAGE_EASY);
    if (blockIdx.x * kBlock >= n) return;
    const uint8_t *tabs = reinterpret_cast<const uint8_t *>(lds_dyn);
    const uint32_t irq = threadIdx.x == 0 ? irq_load(w) : 0u;  // (in flight while the tables are staged)
    uint32_t it = 0;
    stage_perlin_tables(sv, n_tables);
    const uint32_t n_shadow = fp.renderer == RTHIP_RENDER_RAYTRACER ? sv.n_lights : 0u;  // DebugRaycaster: closest hit only
    const uint32_t lane = threadIdx.x & 63u;
    KdCounters kc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t shadow_total = 0, spawn_total = 0;
    const long long t_start = COUNT ? clock64() : 0;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        // Interrupt(): thread 0 reads the flag (when the kernel starts, then every kIrqEvery-th iteration) and publishes it; the
        // workgroup leaves together (its iterations contain barriers)
        if ((it++ & (kIrqEvery - 1u)) == 0u) {
            if (threadIdx.x == 0) sh.irq = it == 1u ? irq : irq_load(w);
            __syncthreads();
            if (sh.irq) break;
        }
        const uint32_t j = base + threadIdx.x;
        bool live = j < n;
        const uint32_t i = live ? j : 0u;
        V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
        uint32_t pix_slot = 0;
        if (level == 0) {
            live = primary_ray(fp, i, live, pix_slot, o, d);  // a slot outside the image: k_resolve skips it too
        } else if (live) {
            const RayRec r = ray_at(w, level, i);
            o = mk(r.ox, r.oy, r.oz);
            d = mk(r.dx, r.dy, r.dz);
        }
        HitRec h;
        PendRec analytic;
        const bool done = easy::trace_stage<COUNT>(sv, o, d, n_shadow, live, h, analytic, kc);
        const bool is_hard = live && !done;  // -> hard list, in shade_and_spawn's reservation step
        if (live && done && level == 0 && w.id_shape && i / fp.n_pix_slots == (fp.aa ? 4u : 0u)) {
            w.id_shape[pix_slot] = h.shape;
            w.id_tri[pix_slot] = -1;
        }
        shade_and_spawn<true, LEAN>(sv, fp, w, tabs, sh, level, live && done, i, h, d, shadow_total, spawn_total, is_hard, &analytic, o);
    }
    shade_stats_flush(w, sh, 0, level, levels, shadow_total, spawn_total);
    if (COUNT) {
        kc.shapes = wave_sum(kc.shapes);
        kc.shapes_full = wave_sum(kc.shapes_full);
        if ((threadIdx.x & 63u) == 0 && kc.shapes && fp.count_kd != 2) {
            atomicAdd(&w.stats[ST_SHAPES_EASY], (unsigned long long)kc.shapes);
            atomicAdd(&w.stats[ST_SHAPES_EASY_FULL], (unsigned long long)kc.shapes_full);
        }
        (void)t_start;
    }
}

// Stage 1b of a level for the BVH path (accel = 0): the rays k_trace_easy listed as needing a mesh walk,
// four lanes per ray, see namespace quad.  Rays the quad walk cannot finish (degenerate (ray, mesh) pairs,
// stack overflow) go to the redo list and are re-traced by the redo instantiation of k_trace.
// OCC: waves per SIMD the kernel is compiled for.  5 (95 VGPRs, 2 spilled) is the default.  6 (80 VGPRs, 27 spilled outside the
// walk loop) is an opt-in (RTHIP_QUAD_OCC6=1): ONE mesh of millions of triangles, whose steps wait for HBM / MALL, gains from the sixth
// resident wave (config 4: 12.98 -> 12.68 ms); the spills are paid per (ray, mesh) entry, so five such meshes lose (config 5:
// 133.8 -> 139.3 ms), and a tree that stays in L2 has nothing to hide (config 3: +3.5 %).  profiles/r03_experiment_quad_occ6.txt
template <bool COUNT>
__device__ __forceinline__ void quad_body(const SceneView &sv, const FrameParams &fp, const Work &w, uint32_t level, uint32_t stack_cap) {
    const uint32_t q = threadIdx.x & 3u, g = threadIdx.x >> 2;
    quad::Stack qs;
    qs.cap = stack_cap;
    qs.e = reinterpret_cast<uint2 *>(lds_dyn) + g * (stack_cap + 1u);
    quad::Park park;
    park.p = reinterpret_cast<double *>(reinterpret_cast<char *>(lds_dyn) + quad::park_offset(stack_cap) + g * quad::kParkStride);
    const uint32_t n = min(w.N, w.hard_count[0]);
    const uint32_t n_shadow = fp.renderer == RTHIP_RENDER_RAYTRACER ? sv.n_lights : 0u;  // DebugRaycaster: closest hit only
    KdCounters kc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t max_ray_nodes = 0;
    const long long t_start = COUNT ? clock64() : 0;
    stage_store(w, w.stage_base + STAGES_PER_LEVEL * level + STAGE_WALK);
    uint32_t it = 0;
    for (uint32_t base = blockIdx.x * quad::kRaysPerBlock; base < n; base += gridDim.x * quad::kRaysPerBlock) {
        if ((++it & (kIrqEvery - 1u)) == 0u && __builtin_amdgcn_readfirstlane((int)irq_load(w))) break;  // Interrupt(): see k_trace
        const uint32_t j = base + g;
        bool live = j < n;
        const uint32_t i = live ? w.hard[j] : 0u;
        const uint32_t nodes_before = kc.nodes;
        uint32_t pix_slot = 0;
        if (level == 0) pix_slot = fp.aa ? i - rtc::fast_div(i, fp.d_pix_slots) * fp.n_pix_slots : i;
        {   // level 0 too: k_trace_easy has stored the primary rays it listed
            V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
            if (live) {
                const RayRec r = ray_at(w, level, i);
                o = mk(r.ox, r.oy, r.oz);
                d = mk(r.dx, r.dy, r.dz);
            }
            park.put_o(o);
            park.put_d(d);
        }
        int32_t hit_shape, hit_tri;
        uint32_t shadow_mask;
        PendRec analytic;
        analytic.bd = DBL_MAX;
        analytic.bs = 0xffffffffu;
        analytic.pad = 0;
        if (live) analytic = w.pend[j];
        const bool done = quad::trace_stage<COUNT>(sv, park, n_shadow, qs, q, live, w, i, analytic, hit_shape, hit_tri, shadow_mask, kc);
        if (!live || q != 0) continue;
        if (COUNT && kc.nodes - nodes_before > max_ray_nodes) max_ray_nodes = kc.nodes - nodes_before;
        if (!done) {  // -> redo list; its hit slot says so, so that k_shade's pass over the hard list leaves the ray to its redo pass
            const uint32_t slot = atomicAdd(&w.redo_count[0], 1u);
            if (slot < w.N) w.redo[slot] = i;
            w.hit[i].shape = kShapeRedo;
            continue;
        }
        if (hit_shape < 0) {  // a miss: the whole record
            HitRec h;
            h.shape = -1;
            h.tri = -1;
            h.shadow_mask = 0;
            h.local = h.normal = mk(0, 0, 0);
            store_hit(w, i, h);
        } else if (shadow_mask) {
            w.hit[i].mask = shadow_mask;  // the record itself was stored after the closest query
        }
        if (level == 0 && w.id_shape && i / fp.n_pix_slots == (fp.aa ? 4u : 0u)) {
            w.id_shape[pix_slot] = hit_shape;
            w.id_tri[pix_slot] = hit_tri;
        }
    }
    if (COUNT) trace_counters_flush(fp, w, kc, max_ray_nodes, t_start);
}
// n_coop: lists of this length and longer are k_trace_coop's (launched right behind this kernel when the scene's trees are
// small; 0xffffffff otherwise)
template <bool COUNT, int OCC = RT_QUAD_OCC>
__global__ __launch_bounds__(kBlock, OCC) void k_trace_quad(SceneView sv, FrameParams fp, Work w, uint32_t level,
                                                            uint32_t stack_cap, uint32_t n_coop) {
    if (min(w.N, w.hard_count[0]) >= n_coop) return;
    quad_body<COUNT>(sv, fp, w, level, stack_cap);
}

// ---------------------------------------------------------------------------- hard rays, up to 64 per wave
// k_trace_quad does everything of a hard ray in the four lanes of its quad, so whatever is per-RAY work - the world -> local
// transforms, the hit record of the winner, the shadow rays' set-up and their analytic shapes - is issued for 16 rays per wave
// with each value computed four times over: on the bunny ~1 000 of the ~2 600 VALU instructions a wave issues.  Here a wave
// takes up to 64 rays, ONE RAY PER LANE for all of that, and only the mesh walks run four lanes per ray: the lanes whose ray
// needs the walk of the (wave-uniform) mesh publish the mesh-local ray in LDS, and the wave walks them 16 at a time
// (quad::mesh_closest, unchanged: quad g takes the g-th waiting ray), the results going back to the owners' lanes.  Rays that
// do not need a walk (missed the mesh's ball, answered by an analytic shape, no shadow query at all) take no quad's place.
// Rays per wave: 64 when the list is long enough to fill the GPU that way, 32 for shorter ones; lists below 300 000 rays stay
// with k_trace_quad (a wave's rounds of walks are serial: a short list dealt 64 at a time would end with a few waves still
// walking while the SIMDs idle).
namespace coop {
static const uint32_t kRecWords = 13;  // lo, ld (12 words) + t_stop: an odd stride, so 64 lanes' word k fall in 64 different banks
__host__ __device__ __forceinline__ uint32_t wave_bytes(uint32_t stack_cap) {  // 16 stacks, 64 records, 16 source lanes
    return (16u * (stack_cap + 1u) * 8u + 64u * kRecWords * 4u + 16u * 4u + 15u) & ~15u;
}
struct Lds {
    uint2 *stack;     // this lane's quad's stack
    uint32_t *rec;    // the wave's records
    uint32_t *order;  // lanes of the rays the quads walk in this sub-round
    uint32_t cap;
};
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t lo32(double v) { return (uint32_t)(unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ uint32_t hi32(double v) { return (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ double mkd(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }

// quad::mesh_closest of mesh m for every lane that `want`s it (its ray in the mesh's frame: lo, ld), 16 rays at a time
template <bool COUNT>
__device__ __forceinline__ void walk(const SceneView &sv, const rthip_mesh &m, V3 lo, V3 ld, float t_stop, bool want, const Lds &L,
                                     uint32_t lane, TriBest &out, KdCounters &cnt, uint32_t &max_walk_nodes) {
    out.hit = false;
    out.tie = false;
    out.ext = 0.0;
    out.d = DBL_MAX;
    out.tri = 0xffffffffu;
    out.ref = 0xffffffffu;
    unsigned long long todo = __ballot(want);
    if (!todo) return;
    const uint32_t q = lane & 3u, g = lane >> 2;
    if (want) {
        uint32_t *r = L.rec + lane * kRecWords;
        r[0] = lo32(lo.x); r[1] = hi32(lo.x); r[2] = lo32(lo.y); r[3] = hi32(lo.y); r[4] = lo32(lo.z); r[5] = hi32(lo.z);
        r[6] = lo32(ld.x); r[7] = hi32(ld.x); r[8] = lo32(ld.y); r[9] = hi32(ld.y); r[10] = lo32(ld.z); r[11] = hi32(ld.z);
        r[12] = __float_as_uint(t_stop);
    }
    quad::Stack qs;
    qs.e = L.stack;
    qs.cap = L.cap;
    const unsigned long long below = (1ull << lane) - 1ull;
    while (todo) {  // (wave-uniform)
        const uint32_t rank = (uint32_t)__popcll(todo & below);
        const bool mine = ((todo >> lane) & 1ull) && rank < 16u;
        if (mine) L.order[rank] = lane;
        const uint32_t waiting = (uint32_t)__popcll(todo);
        const bool act = g < (waiting < 16u ? waiting : 16u);
        lds_fence();
        const uint32_t *r = L.rec + (act ? L.order[g] : lane) * kRecWords;
        const V3 o = mk(mkd(r[0], r[1]), mkd(r[2], r[3]), mkd(r[4], r[5]));
        const V3 d = mk(mkd(r[6], r[7]), mkd(r[8], r[9]), mkd(r[10], r[11]));
        const float ts = __uint_as_float(r[12]);
        TriBest tb;
        tb.hit = false;
        tb.tie = false;
        tb.d = DBL_MAX;
        tb.tri = 0xffffffffu;
        if (act) {
            const uint32_t before = cnt.nodes;
            quad::mesh_closest<COUNT>(sv, m, o, d, qs, q, ts, tb, cnt);
            if (COUNT && cnt.nodes - before > max_walk_nodes) max_walk_nodes = cnt.nodes - before;
        }
        // the quads' results -> the owners' lanes (rank r was walked by quad r: lanes 4 r .. 4 r + 3 hold it)
        const int from = (int)((rank & 15u) << 2);
        const uint32_t rdlo = (uint32_t)__shfl((int)lo32(tb.d), from, 64), rdhi = (uint32_t)__shfl((int)hi32(tb.d), from, 64);
        const uint32_t rtri = (uint32_t)__shfl((int)tb.tri, from, 64);
        const uint32_t rfl = (uint32_t)__shfl((int)((tb.hit ? 1u : 0u) | (tb.tie ? 2u : 0u)), from, 64);
        if (mine) {
            out.d = mkd(rdlo, rdhi);
            out.tri = rtri;
            out.ref = rtri;
            out.hit = (rfl & 1u) != 0u;
            out.tie = (rfl & 2u) != 0u;
        }
        todo &= ~__ballot(mine);
    }
}

// the mesh part of Scene::CastRay (Scene.cpp:163-237) for the lane's ray; quad::cast_ray's second loop, one ray per lane.
// The ray is not passed by value: `ray(o, d)` fetches it where it is needed (closest query: the level's ray record in global
// memory; shadow query: the hit point + the light's direction), and the lane's own mesh-local ray is read back from its LDS
// record after the walk - nothing of it stays in registers across a walk, whose loop needs them all (5 waves per SIMD).
template <bool COUNT, class RAY>
__device__ __forceinline__ void cast_meshes(const SceneView &sv, const RAY &ray, bool any_hit, bool active, double &bd, uint32_t &bs,
                                            int32_t &best_tri, bool &redo, const Lds &L, uint32_t lane, KdCounters &cnt,
                                            uint32_t &max_walk_nodes, const double *ld_row = nullptr) {
    for (uint32_t s = 0; s < sv.n_shapes; s++) {
        const rt_uniform_shape *sh = rt_uniform_cast(sv.shapes) + s;
        if (sh->type != RTHIP_SHAPE_MESH) continue;
        bool want = active && !redo && !(any_hit && bs != 0xffffffffu);
        if (COUNT && want) cnt.shapes++;
        const rthip_mesh m = rt_load_mesh(sv.meshes + sh->mesh);
        TriBest tb;
        {
            V3 o, d;
            ray(o, d);
            if (want && ball_missed(sh->bound_c[0], sh->bound_c[1], sh->bound_c[2], sh->bound_r2, o, d)) want = false;
            if (!__ballot(want)) continue;  // (wave-uniform: nobody is near this mesh)
            if (COUNT && want) cnt.shapes_full++;
            double inv[12];
#pragma unroll
            for (int k = 0; k < 12; k++) inv[k] = sh->inv[k];
            const V3 lo = xform(inv, o, 1.0);
            const V3 ld = local_dir(ld_row, s, inv, d);
            if (want && kd_degenerate_ray(sv, m, lo, ld)) {
                redo = true;
                want = false;
            }
            float t_stop = 3.0e38f;
            if (any_hit) {
                double fwd[12];
#pragma unroll
                for (int k = 0; k < 12; k++) fwd[k] = sh->fwd[k];
                t_stop = anyhit_t_stop(fwd, sh->rt_lin, sh->rt_t, o, ld);
            }
            walk<COUNT>(sv, m, lo, ld, t_stop, want, L, lane, tb, cnt, max_walk_nodes);
        }
        lds_fence();  // (also: nothing computed before the walk is reused below - it is fetched again)
        if (!want || !tb.hit) continue;
        if (tb.tie && !any_hit) {  // (a shadow query only consumes the distance)
            redo = true;
            continue;
        }
        const uint32_t *r = L.rec + lane * kRecWords;  // this lane's own record: its mesh-local ray
        const V3 lo = mk(mkd(r[0], r[1]), mkd(r[2], r[3]), mkd(r[4], r[5]));
        const V3 ld = mk(mkd(r[6], r[7]), mkd(r[8], r[9]), mkd(r[10], r[11]));
        V3 p;
        double u, v, ww;
        tri_point(sv.tris[m.tri_base + tb.tri], lo, ld, p, u, v, ww);
        double fwd[12];
#pragma unroll
        for (int k = 0; k < 12; k++) fwd[k] = sh->fwd[k];
        const V3 worldPoint = xform(fwd, p, 1.0);
        V3 o, d;
        ray(o, d);
        const double distance = sqlen(sub(o, worldPoint));
        if (!(distance >= DBL_EPSILON)) continue;
        if (distance < bd || (distance == bd && s < bs)) {
            bd = distance;
            bs = s;
            best_tri = (int32_t)tb.tri;
        }
    }
}

// the analytic shapes of a shadow query, one ray per lane (quad::cast_ray's first loop; only .Hit is consumed, so the
// first accepted shape ends the lane's search)
template <bool COUNT>
__device__ __forceinline__ void cast_analytic_any(const SceneView &sv, V3 o, V3 d, bool active, double &bd, uint32_t &bs, KdCounters &cnt,
                                                  const double *ld_row) {
    for (uint32_t s = 0; s < sv.n_shapes; s++) {
        const rt_uniform_shape *sh = rt_uniform_cast(sv.shapes) + s;
        const int32_t type = sh->type;
        if (type == RTHIP_SHAPE_MESH) continue;
        if (!active || bs != 0xffffffffu) continue;
        if (COUNT) cnt.shapes++;
        if (type == RTHIP_SHAPE_SPHERE && ball_missed(sh->bound_c[0], sh->bound_c[1], sh->bound_c[2], sh->bound_r2, o, d)) continue;
        if (COUNT) cnt.shapes_full++;
        double inv[12];
#pragma unroll
        for (int k = 0; k < 12; k++) inv[k] = sh->inv[k];
        const V3 lo = xform(inv, o, 1.0);
        const V3 ld = local_dir(ld_row, s, inv, d);
        V3 p;
        const bool hit = type == RTHIP_SHAPE_SPHERE ? sphere_intersect(lo, ld, p)
                                                    : plane_intersect(mk(sh->plane_n[0], sh->plane_n[1], sh->plane_n[2]), lo, ld, p);
        if (!hit) continue;
        double fwd[12];
#pragma unroll
        for (int k = 0; k < 12; k++) fwd[k] = sh->fwd[k];
        const V3 worldPoint = xform(fwd, p, 1.0);
        const double distance = sqlen(sub(o, worldPoint));
        if (!(distance >= DBL_EPSILON)) continue;
        bd = distance;
        bs = s;
    }
}
}  // namespace coop

// Lists shorter than thr32 are k_trace_quad's (launched just before: one of the two kernels returns at once).  Both paths in ONE
// kernel had the quad code's walk loops spill (their registers are allocated together): a lone frame 2.6 % slower.
template <bool COUNT, int OCC = RT_COOP_OCC>
__global__ __launch_bounds__(kBlock, OCC) void k_trace_coop(SceneView sv, FrameParams fp, Work w, uint32_t level, uint32_t stack_cap,
                                                            uint32_t thr32, uint32_t thr64) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t n = min(w.N, w.hard_count[0]);
    if (n < thr32) return;
    char *wave_lds = reinterpret_cast<char *>(lds_dyn) + wid * coop::wave_bytes(stack_cap);
    coop::Lds L;
    L.cap = stack_cap;
    L.stack = reinterpret_cast<uint2 *>(wave_lds) + (lane >> 2) * (stack_cap + 1u);
    L.rec = reinterpret_cast<uint32_t *>(wave_lds + 16u * (stack_cap + 1u) * 8u);
    L.order = L.rec + 64u * coop::kRecWords;
    const uint32_t n_shadow = fp.renderer == RTHIP_RENDER_RAYTRACER ? sv.n_lights : 0u;  // DebugRaycaster: closest hit only
    KdCounters kc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t max_walk_nodes = 0;
    const long long t_start = COUNT ? clock64() : 0;
    stage_store(w, w.stage_base + STAGES_PER_LEVEL * level + STAGE_WALK);
    // Rays per wave: 64 when the list is long enough to fill the GPU that way, 32 for a shorter one (a wave's rounds are serial:
    // a short list dealt 64 at a time would leave SIMDs idle while a few waves walk their four rounds).  Waves stride over the
    // list.  (Claiming the rays with one atomic per wave on a counter cost more than the walks - ~15 ns per claim, serialised -,
    // and dealing the list in the order the waves start, 64 m consecutive rays each and 16 at a time for the last ones, was
    // slower than the stride on long lists: profiles/r03_coop_walk.txt.)
    const uint32_t R = n >= thr64 ? 64u : 32u;
    const uint32_t gw = blockIdx.x * (kBlock / 64u) + wid, n_waves = gridDim.x * (kBlock / 64u);
    uint32_t it = 0;
    for (uint32_t base = gw * R; base < n; base += n_waves * R) {  // (wave-uniform)
        // Interrupt(): see k_trace.  EVERY iteration but the first: one of this kernel's iterations is 64 rays through two queries each
        // (0.1 - 0.7 ms), and a wave makes only a few of them per level - polled every eighth (or second) iteration most waves of
        // a 4096 x 4096 frame never looked again, and an Interrupt() took 10 ms of a 26 ms frame to be noticed
        if (it++ && __builtin_amdgcn_readfirstlane((int)irq_load(w))) break;
        const uint32_t j = base + lane;
        const bool live = lane < R && j < n;
        const uint32_t i = live ? w.hard[j] : 0u;
        double bd = DBL_MAX;
        uint32_t bs = 0xffffffffu;
        if (live) {
            const PendRec known = w.pend[j];  // the closest hit over the analytic shapes (k_trace_easy)
            bd = known.bd;
            bs = known.bs;
        }
        // the level's ray of this lane, fetched from its record wherever it is needed (level 0 too: k_trace_easy has stored the
        // primary rays it listed); the asm keeps the compiler from holding the first fetch in registers instead
        const RayRec *my_ray = &ray_at(w, level, i);
        auto level_ray = [&](V3 &o, V3 &d) {
            asm volatile("" ::: "memory");
            o = mk(0, 0, 0);
            d = mk(0, 0, 1);
            if (live) {
                const RayRec r = *my_ray;
                o = mk(r.ox, r.oy, r.oz);
                d = mk(r.dx, r.dy, r.dz);
            }
        };
        bool redo = false;
        int32_t hit_shape = -1, hit_tri = -1;
        uint32_t shadow_mask = 0u;
        V3 point = mk(0, 0, 0);
        {   // the closest hit
            int32_t tri = -1;
            coop::cast_meshes<COUNT>(sv, level_ray, false, live, bd, bs, tri, redo, L, lane, kc, max_walk_nodes);
            if (live && !redo && bs != 0xffffffffu) {
                Hit c;
                V3 o, d;
                level_ray(o, d);
                hit_details(sv, o, d, (int32_t)bs, tri, c, COUNT ? &kc : nullptr);
                hit_shape = (int32_t)bs;
                hit_tri = tri;
                HitRec rec;
                rec.shape = hit_shape;
                rec.tri = tri;
                rec.shadow_mask = 0;
                rec.local = c.local;
                rec.normal = c.normal;
                store_hit(w, i, rec);  // (the shadow mask is patched in below)
                point = c.point;       // every shadow ray starts there
            }
        }
        for (uint32_t l = 0; l < n_shadow; l++) {
            const rt_uniform_light *light = rt_uniform_cast(sv.lights) + l;
            const int32_t ltype = light->type;
            if (!light->enabled || ltype == RTHIP_LIGHT_AMBIENT) continue;  // Raytracer.cpp:213, 234-237
            const bool active = live && !redo && hit_shape >= 0;
            auto shadow_ray = [&](V3 &o, V3 &d) {  // (a directional light's direction is wave-uniform: scalar registers)
                o = point;
                if (ltype == RTHIP_LIGHT_DIRECTIONAL)
                    d = mk(light->direction[0], light->direction[1], light->direction[2]);
                else
                    d = normalized(sub(mk(light->position[0], light->position[1], light->position[2]), point));
            };
            double sd = DBL_MAX;
            uint32_t ss = 0xffffffffu;
            int32_t tri = -1;
            const double *ld_row = shadow_ld_row(sv, l, ltype);
            {
                V3 ro, rd;
                shadow_ray(ro, rd);
                coop::cast_analytic_any<COUNT>(sv, ro, rd, active, sd, ss, kc, ld_row);
            }
            coop::cast_meshes<COUNT>(sv, shadow_ray, true, active, sd, ss, tri, redo, L, lane, kc, max_walk_nodes, ld_row);
            if (ss != 0xffffffffu) shadow_mask |= 1u << l;
        }
        if (!live) continue;
        if (redo) {  // -> redo list; its hit slot says so, so that k_shade's pass over the hard list leaves the ray to its redo pass
            const uint32_t slot = atomicAdd(&w.redo_count[0], 1u);
            if (slot < w.N) w.redo[slot] = i;
            w.hit[i].shape = kShapeRedo;
            continue;
        }
        if (hit_shape < 0) {  // a miss: the whole record
            HitRec h;
            h.shape = -1;
            h.tri = -1;
            h.shadow_mask = 0;
            h.local = h.normal = mk(0, 0, 0);
            store_hit(w, i, h);
        } else if (shadow_mask) {
            w.hit[i].mask = shadow_mask;  // the record itself was stored after the closest query
        }
        if (level == 0 && w.id_shape && i / fp.n_pix_slots == (fp.aa ? 4u : 0u)) {
            const uint32_t pix_slot = fp.aa ? i - rtc::fast_div(i, fp.d_pix_slots) * fp.n_pix_slots : i;
            w.id_shape[pix_slot] = hit_shape;
            w.id_tri[pix_slot] = hit_tri;
        }
    }
    if (COUNT) trace_counters_flush(fp, w, kc, max_walk_nodes, t_start);
}
// The redo list (rays only the reference kd-tree reproduces: kd-degenerate (ray, mesh) pairs and exact distance ties; normally
// empty): re-traced one ray per lane through every query (rtc::trace_stage) and shaded on the spot by `busy` workgroups (all of
// which call this).  Stacks in global memory (w.redo_stack): the rare path takes no LDS from the common one.
__device__ __forceinline__ void redo_pass(const SceneView &sv, const FrameParams &fp, const Work &w, const uint8_t *tabs, ShadeShared &sh,
                                          uint32_t level, uint32_t stack_cap, uint32_t n_redo, uint32_t busy, uint32_t &shadow_total,
                                          uint32_t &spawn_total) {
    KdStack st = make_stack(w.redo_stack + (size_t)blockIdx.x * kBlock + threadIdx.x, kRedoBlocks * kBlock, stack_cap, nullptr);
    const uint32_t n_shadow = fp.renderer == RTHIP_RENDER_RAYTRACER ? sv.n_lights : 0u;
    for (uint32_t base = blockIdx.x * kBlock; base < n_redo; base += busy * kBlock) {
        {   // Interrupt(): EVERY iteration - one of them is 256 rays through the reference's kd walk (on a height field seen along
            // its grid lines the pass takes milliseconds on its 64 workgroups; polled every eighth iteration it never looked again)
            if (threadIdx.x == 0) sh.irq = irq_load(w);
            __syncthreads();
            if (sh.irq) break;
        }
        const uint32_t j = base + threadIdx.x;
        const bool live = j < n_redo;
        const uint32_t i = live ? w.redo[j] : 0u;
        V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
        if (live) {  // (level 0 too: k_trace_easy stored the primary rays it listed)
            const RayRec r = ray_at(w, level, i);
            o = mk(r.ox, r.oy, r.oz);
            d = mk(r.dx, r.dy, r.dz);
        }
        HitRec h;
        h.shape = -1;
        h.tri = -1;
        h.shadow_mask = 0;
        h.local = h.normal = mk(0, 0, 0);
        st.overflow = false;
        if (live) trace_stage<false>(sv, o, d, n_shadow, st, h, nullptr);
        // primary-hit ids come from the un-offset pixel-centre ray (sub-sample 4 of the 3x3 SSAA grid)
        if (live && level == 0 && w.id_shape && i / fp.n_pix_slots == (fp.aa ? 4u : 0u)) {
            w.id_shape[i % fp.n_pix_slots] = h.shape;
            w.id_tri[i % fp.n_pix_slots] = h.tri;
        }
        shade_and_spawn<false>(sv, fp, w, tabs, sh, level, live, i, h, d, shadow_total, spawn_total);
    }
}

// Stage 2 of a level for the rays that went through a hit record: LIST = true, the hard rays of the BVH
// path (traced by k_trace_quad / k_trace_coop); LIST = false, every ray of the level (accel = 1, where
// k_trace does all the tracing).  The Perlin tables (256 B each) are staged in LDS: their chained lookups
// are the latency chain of every sky / marble sample.
// The redo list of the BVH path is re-traced and shaded by k_redo, launched behind this kernel.  (For most of round 3 the first
// workgroups of k_shade<true> did that themselves, to save four launches on a frame's chain for a list that is almost always empty;
// but the pass's kd walk then shares this kernel's register allocation - 128 VGPRs: 191 spilled instead of 35 - and the common path
// paid for it: 32-frame calls 1.2 - 2.3 %, while a lone frame and a rank's eighth of a burst do not notice the launches:
// profiles/r03_redo_split.txt.)
template <bool LIST, bool LEAN = false>
__global__ __launch_bounds__(kBlock, RT_SHADE_OCC) void k_shade(SceneView sv, FrameParams fp, Work w, uint32_t level, uint32_t n_tables,
                                                                 uint32_t levels, uint32_t stack_cap) {
    __shared__ ShadeShared sh;
    const uint32_t n = min(w.N, LIST ? w.hard_count[0] : ((level == 0) ? fp.N0 : level_count(w, level)));
    stage_store(w, w.stage_base + STAGES_PER_LEVEL * level + STAGE_SHADE);
    if (blockIdx.x * kBlock >= n) return;
    const uint8_t *tabs = reinterpret_cast<const uint8_t *>(lds_dyn);
    stage_perlin_tables(sv, n_tables);
    uint32_t shadow_total = 0, spawn_total = 0;
    uint32_t it = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        if ((++it & (kIrqEvery - 1u)) == 0u) {  // Interrupt(): see k_trace_easy (the level's first kernel polled at its start)
            if (threadIdx.x == 0) sh.irq = irq_load(w);
            __syncthreads();
            if (sh.irq) break;
        }
        const uint32_t j = base + threadIdx.x;
        uint32_t i = 0;
        HitSlot hs;
        hs.shape = kShapeNoRay;
        if (j < n) {
            i = LIST ? w.hard[j] : j;
            hs = w.hit[i];
        }
        HitRec h;
        h.shape = hs.shape;
        h.tri = -1;
        h.shadow_mask = hs.mask;
        h.local = mk(hs.lx, hs.ly, hs.lz);
        h.normal = mk(hs.nx, hs.ny, hs.nz);
        V3 d = mk(0, 0, 1);
        const bool have = hs.shape >= -1;  // (not: no ray in the slot, or a ray the redo pass traces and shades)
        if (have) {
            const RayRec rr = ray_at(w, level, i);
            d = mk(rr.dx, rr.dy, rr.dz);
        }
        shade_and_spawn<false, LEAN>(sv, fp, w, tabs, sh, level, have, i, h, d, shadow_total, spawn_total);
    }
    shade_stats_flush(w, sh, 1, level, levels, shadow_total, spawn_total);
}
// the redo pass, behind k_shade<true> (at most kRedoBlocks workgroups; their statistics go to the slots k_shade's workgroups of the
// same index have just written)
__global__ __launch_bounds__(kBlock) void k_redo(SceneView sv, FrameParams fp, Work w, uint32_t level, uint32_t n_tables, uint32_t levels,
                                                 uint32_t stack_cap) {
    __shared__ ShadeShared sh;
    const uint32_t n_redo = min(w.N, w.redo_count[0]);
    if (!n_redo) return;
    const uint8_t *tabs = reinterpret_cast<const uint8_t *>(lds_dyn);
    stage_perlin_tables(sv, n_tables);
    uint32_t shadow_total = 0, spawn_total = 0;
    redo_pass(sv, fp, w, tabs, sh, level, stack_cap, n_redo, gridDim.x, shadow_total, spawn_total);
    shade_stats_flush(w, sh, 1, level, levels, shadow_total, spawn_total);
}

// One pixel sample: its chain of rays (level 0 -> reflected ray -> ...) folded back to front,
// C_k = directLighting_k (+) C_{k+1} (x) reflectance_k with the reference's clamping Color algebra
// (Raytracer.cpp:283-286; the clamps forbid a forward throughput product).  Every ray has at most one
// reflected ray, so the chain is a list: no per-level fold kernels, no global synchronisation.
__device__ __forceinline__ Col resolve_chain(const Work &w, uint32_t i0, uint32_t max_level) {
    uint32_t idx[RTHIP_MAX_RECURSION + 1];
    uint32_t level = 0, i = i0;
    for (;;) {
        idx[level] = i;
        if (level == max_level) break;
        const uint32_t c = w.child[(size_t)level * w.N + i];
        if (c >= w.N) break;  // kNoChild; (also: a link an interrupted frame never wrote cannot lead outside the workspace)
        i = c;
        level++;
    }
    const ColRec q = rec_at(w, level, i);
    Col c = Col{q.r, q.g, q.b};
    while (level > 0) {
        level--;
        const ColRec pr = rec_at(w, level, idx[level]);
        c = fold(Col{pr.r, pr.g, pr.b}, c, pr.refl);
    }
    return c;
}

// Everything the frame path clears or fills on the device goes through these two small KERNELS, never through
// hipMemsetAsync.  With hipMemsetAsync for the per-frame counters (memset nodes once the launches are captured into a graph),
// ROCm 7.2 keeps stream order as long as calls are synchronous or a single context queues them (tools/micro/graph_memset.hip and
// tools/gpu_graph_memset.py find nothing wrong there), but with several contexts queueing asynchronous calls of two shapes - the
// bench's regime - the counters are not zero when a frame starts (round 1 saw it as garbage ray statistics): list reservations
// then run past the workspace and the GPU takes a memory fault (seen once in round 2 with the memsets switched back in).  A
// kernel node or launch is ordered like every other kernel of the stream.  The list writers below also bound their indices by
// the workspace size, so that a counter that is wrong for whatever reason can cost a frame but not the GPU.
__global__ __launch_bounds__(kBlock) void k_clear(uint32_t *a, uint32_t na, uint32_t *b, uint32_t nb, uint32_t *c, uint32_t nc) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < na) a[i] = 0u;
    if (i < nb) b[i] = 0u;
    if (i < nc) c[i] = 0u;
}
// The cameras of a call travel through a RING of slots in page-locked host memory (rthip_ctx::h_cams), not through kernel
// arguments: a captured graph then holds nothing of the camera, and a caller whose camera moves every frame (the Sandbox
// preview, Sandbox/Sandbox.cpp:315-359) replays the same graph.  Which slot a call reads is decided on the device: the calls of
// a context run in stream order, so the k-th k_begin that executes belongs to the k-th call the host enqueued, and k_begin
// counts them in `seq` (device memory).  The host writes the cameras of call k to slot k % kCamSlots before it enqueues the
// call (after call k - kCamSlots has finished: an event per slot).  Plain launches pass k explicitly and re-set the counter -
// should host and device ever disagree, the next plain launch heals it; a replayed graph passes kSeqAuto.
static const uint32_t kCamSlots = 16;
static const uint32_t kSeqAuto = 0xffffffffu;
// the head of a call's launches in ONE kernel: statistics, per-workgroup statistics and list counters cleared, the cameras
// fetched (every launch on a frame's chain costs ~4 us of latency, whatever it does)
__global__ __launch_bounds__(kBlock) void k_begin(uint32_t *a, uint32_t na, uint32_t *b, uint32_t nb, uint32_t *c, uint32_t nc, rthip_camera *dst,
                                                  uint32_t count, const rthip_camera *ring, uint32_t *seq, uint32_t seq_explicit) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < na) a[i] = 0u;
    if (i < nb) b[i] = 0u;
    if (i < nc) c[i] = 0u;
    if (blockIdx.x == 0) {
        const uint32_t k = seq_explicit != kSeqAuto ? seq_explicit : *seq;  // (every thread reads it before thread 0 advances it)
        __syncthreads();
        const uint32_t words = count * (uint32_t)(sizeof(rthip_camera) / 4);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(ring + (size_t)(k % kCamSlots) * RTHIP_MAX_BATCH);
        for (uint32_t w = threadIdx.x; w < words; w += kBlock)
            reinterpret_cast<uint32_t *>(dst)[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) *seq = k + 1u;
    }
}
__global__ __launch_bounds__(kBlock) void k_fill(uint32_t *a, size_t n, uint32_t value) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) a[i] = value;
}

// chains -> pixels (Common.cpp:275-282); SSAA: result += c * (1.0f/9.0f), clamped (Raytracer.cpp:139)
__global__ __launch_bounds__(kBlock) void k_resolve(FrameParams fp, Work w, uint32_t *pixels, int32_t *out_shape,
                                                    int32_t *out_tri) {
    uint32_t ps = blockIdx.x * kBlock + threadIdx.x;
    stage_store(w, w.stage_base + STAGES_PER_LEVEL * (fp.max_rec + 1u));
    if (blockIdx.x == 0 && threadIdx.x <= fp.max_rec && w.hist) {
        const uint32_t listed = w.counts[(size_t)(2 * (RTHIP_MAX_RECURSION + 3) + threadIdx.x) * kCounterStride];  // (the level's hard_count)
        __hip_atomic_store(w.hist + 1 + threadIdx.x, listed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) __hip_atomic_store(w.hist, w.hist_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (ps >= fp.n_pix_slots) return;
    if (irq_load(w)) return;  // Interrupt(): the chains are incomplete, the frame is abandoned (Raytracer.cpp:153)
    uint32_t x, y, frame, part_slot;
    if (!slot_to_pixel(fp, ps, x, y, frame, part_slot)) return;
    Col c;
    if (fp.aa) {
        c = Col{0, 0, 0};
        const double k = (double)(1.0f / 9.0f);
        for (uint32_t s = 0; s < 9; s++) {
            Col r = resolve_chain(w, s * fp.n_pix_slots + ps, fp.max_rec);
            c = c_add(c, c_scale(r, k));
        }
    } else {
        c = resolve_chain(w, ps, fp.max_rec);
    }
    size_t idx = fp.layout ? (size_t)frame * fp.part_slots + part_slot : ((size_t)frame * fp.H + y) * fp.W + x;
    pixels[idx] = c_pack(c);
    if (out_shape) out_shape[idx] = w.id_shape[ps];
    if (out_tri) out_tri[idx] = w.id_tri[ps];
}

// multi-GPU: part_count tile-compact buffers (part after part) -> row-major frame
__global__ __launch_bounds__(kBlock) void k_assemble(FrameParams fp, const uint32_t *gathered, uint32_t part_slots,
                                                     uint32_t *frame) {
    uint32_t ps = blockIdx.x * kBlock + threadIdx.x;
    uint32_t part = blockIdx.y, b = blockIdx.z;
    if (ps >= part_slots) return;
    fp.part_index = part;
    fp.q0 = 0;
    fp.n_frames = 1;
    const uint32_t n_tiles = fp.tiles_x * fp.tiles_y;
    if (part >= n_tiles) return;
    fp.my_tiles = (n_tiles - part + fp.part_count - 1) / fp.part_count;
    uint32_t x, y, f0, slot;
    if (!slot_to_pixel<true>(fp, ps, x, y, f0, slot)) return;
    frame[((size_t)b * fp.H + y) * fp.W + x] = gathered[((size_t)part * gridDim.z + b) * part_slots + ps];
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
    uint32_t stack_cap_kd = 1, stack_cap_bvh = 1, stack_cap_full = 1, bvh_depth = 0;
    DevBuf shapes, meshes, kd_nodes, kd_bounds, kd_refs, tris, tri_normals, materials, noises, perlin, worley, lights;
    SceneView sv;
    bool has_scene = false;
    rthip_scene_desc uploaded{};             // the last uploaded description (its counts; the pointers are the caller's)
    std::vector<rthip_mesh> mesh_table;      // host copies for rthip_update_scene: the table the DEVICE holds (bvh_* fields rewritten by the upload) ...
    std::vector<rthip_mesh> mesh_table_in;   // ... and the table as the caller described it (what a later rthip_update_scene brings again)
    std::vector<rthip_bvhnode> mesh_roots;
    std::vector<uint8_t> mesh_device_bvh;    // mesh i's BVH was built on the device (its bvh_* fields in mesh_table are the library's)
    std::vector<uint8_t> mesh_device_kd;     // mesh i's kd-tree was built on the device (its node_* / ref_* / plane_* fields likewise)
    rthip_upload_info upload_info{};
    uint32_t stack_cap = 1;
    // frame workspace
    DevBuf rec, child, rays, counts, stats, block_stats, ids, pixels_tmp, ids_out, hits, redo, redo_stack, pend, probe[8];
    DevBuf shadow_ld;  // SceneView::shadow_ld
    bool lean_scene = false;  // every shape's material is uniform or one checker / Perlin interpolation of two uniform ones: the LEAN kernels
    uint32_t bs_stride = 0, bs_levels = 0;
    uint32_t n_tables = 0;
    bool has_meshes = false;
    bool walk_coop = false;  // long hard lists through k_trace_coop (per-ray work one ray per lane) instead of k_trace_quad
    int coop_forced = -1;    // RTHIP_QUAD_COOP
    size_t coop_max_bytes = 0;
    uint32_t coop_thr32 = 300000u, coop_thr64 = 600000u;
    int mesh_without_bvh = -1;  // first mesh that has triangles but no BVH (a caller that supplied only the kd-tree), -1 = none
    std::atomic<int> interrupt{0};
    std::atomic<float> progress{0.f};
    // device-visible side of Interrupt() / Percent (see Work::irq): the flag the kernels poll, and in page-locked host memory
    // the stage word the kernels store to (h_stage[0]) and the request word rthip_interrupt stores to (h_stage[32])
    rthip_camera *d_cams = nullptr;    // RTHIP_MAX_BATCH cameras of the call in flight (all calls of a context run on one stream, in order)
    rthip_camera *h_cams = nullptr;    // page-locked ring: kCamSlots x RTHIP_MAX_BATCH cameras, slot = call sequence number % kCamSlots (see k_begin)
    uint32_t *d_cam_seq = nullptr;     // device: calls begun on this context so far
    uint32_t cam_seq = 0;              // host: calls enqueued on this context so far
    hipEvent_t cam_ev[16] = {};        // [kCamSlots] recorded behind the call that used the slot
    uint32_t graphs_captured = 0, graph_replays = 0;
    uint32_t *d_irq = nullptr;
    bool irq_finegrained = false;
    bool irq_dirty = false;            // d_irq may be non-zero (guarded by irq_mutex)
    std::mutex irq_mutex;
    volatile uint32_t *h_stage = nullptr;
    std::atomic<uint32_t> stages_total{0};
    rthip_stats last{};
    int count_kd = 0;
    uint32_t last_frames = 1;  // frames of the last render call (statistics are per call)
    size_t chunk_rays = kMaxLevelRays;
    bool chunk_rays_forced = false;  // RTHIP_CHUNK_RAYS given: taken as it is
    // one frame = ~20 dependent launches; replaying them as a HIP graph cuts the host cost of a frame from ~80 us to
    // one launch (what bounds small frames and the per-GPU share of a multi-GPU frame).  The graph is re-captured
    // when anything baked into it changes (params, frames per call, output pointers, scene, workspace) - but only once the
    // same key comes twice in a row (a capture + instantiation costs ~1 ms).  The CAMERAS are not baked in (see k_begin):
    // a camera that moves every frame (the Sandbox preview) replays one graph.
    bool graphs_enabled = true;
    struct CachedGraph {
        std::vector<unsigned char> key;
        hipGraphExec_t exec;
    };
    std::vector<CachedGraph> graphs;  // most recently captured last; at most kMaxGraphs (a caller alternating between a few frame shapes keeps them all)
    static constexpr size_t kMaxGraphs = 4;
    std::vector<unsigned char> seen_key;  // key of the previous call: a graph is only captured for a key seen twice in a row
    uint64_t scene_generation = 0;
    int profile_launches = 0;
    std::vector<hipEvent_t> ev_pool;
};

static int set_device(rthip_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

extern "C" {

uint64_t rthip_debug_device_allocs(void) { return g_device_allocs.load(); }
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
    if (device < kMaxDevices) {  // once per device, with that device current: where this GPU's copy of `scratch` lives
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            void *sp = nullptr;
            hipError_t se = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
            if (se != hipSuccess || !sp) {
                delete c;
                return fail(RTHIP_E_HIP, "rthip_create: hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(se));
            }
            g_scratch_ptr[device] = sp;
        }
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) {  // fine-grained device memory is not cached in L2: a running kernel sees the write of rthip_interrupt
        if (hipExtMallocWithFlags((void **)&c->d_irq, 256, hipDeviceMallocFinegrained) == hipSuccess && c->d_irq) {
            c->irq_finegrained = true;
        } else {  // (then an interrupt is only seen by the kernels launched after it)
            (void)hipGetLastError();
            c->d_irq = nullptr;
            e = hipMalloc((void **)&c->d_irq, 256);
        }
    }
    if (e == hipSuccess) e = hipMemset(c->d_irq, 0, 256);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_cams, sizeof(rthip_camera) * RTHIP_MAX_BATCH);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_cam_seq, 256);
    if (e == hipSuccess) e = hipMemset(c->d_cam_seq, 0, 256);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_cams, sizeof(rthip_camera) * RTHIP_MAX_BATCH * kCamSlots, hipHostMallocMapped);
    static_assert(sizeof(((rthip_ctx *)nullptr)->cam_ev) / sizeof(hipEvent_t) == kCamSlots, "one event per camera slot");
    for (uint32_t k = 0; k < kCamSlots && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->cam_ev[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_stage, 256, hipHostMallocMapped);
    if (e == hipSuccess) for (int k = 0; k < 64; k++) c->h_stage[k] = 0;  // (0: stage, 8..: the last call's list lengths, 32: interrupt request)
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        rthip_destroy(c);  // releases whichever of the three exist
        return fail(RTHIP_E_HIP, "rthip_create: %s", hipGetErrorString(e));
    }
    const char *env = getenv("RTHIP_COUNTERS");
    c->count_kd = env ? atoi(env) : 0;
    env = getenv("RTHIP_GRAPH");
    if (env && atoi(env) == 0) c->graphs_enabled = false;
    env = getenv("RTHIP_CHUNK_RAYS");
    if (env && atoll(env) > 0) { c->chunk_rays = (size_t)atoll(env); c->chunk_rays_forced = true; }
    env = getenv("RTHIP_PROFILE_LAUNCHES");
    c->profile_launches = env && atoi(env) != 0;
    // the walk kernel (k_trace_coop / k_trace_quad; the default follows the scene, see upload_scene): 1 / 0 force it, and the list
    // lengths from which a wave of k_trace_coop takes 32 / 64 rays (read per context: tests and A/B runs hold several)
    env = getenv("RTHIP_QUAD_COOP");
    c->coop_forced = env ? (atoi(env) != 0 ? 1 : 0) : -1;
    env = getenv("RTHIP_COOP_MAX_TREE_MB");
    c->coop_max_bytes = (size_t)(env && atoi(env) > 0 ? atoi(env) : 0) << 20;  // (0: no limit)
    env = getenv("RTHIP_COOP_T32");
    c->coop_thr32 = env && atoi(env) > 0 ? (uint32_t)atoi(env) : 300000u;
    env = getenv("RTHIP_COOP_T64");
    c->coop_thr64 = env && atoi(env) > 0 ? (uint32_t)atoi(env) : 600000u;
    memset(&c->sv, 0, sizeof c->sv);
    *out = c;
    return 0;
}

int rthip_destroy(rthip_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *all[] = {&c->shapes, &c->meshes, &c->kd_nodes, &c->kd_bounds, &c->kd_refs, &c->tris, &c->tri_normals,
                     &c->materials, &c->noises, &c->perlin, &c->worley, &c->lights, &c->rec, &c->child, &c->rays,
                     &c->counts, &c->stats, &c->ids, &c->pixels_tmp, &c->ids_out, &c->bvh_nodes, &c->bvh_tris, &c->bvh_index, &c->kd_planes, &c->hits, &c->redo, &c->redo_stack, &c->block_stats, &c->pend, &c->shadow_ld};
    for (DevBuf *b : all) b->release();
    for (DevBuf &b : c->probe) b.release();
    for (auto &g : c->graphs) (void)hipGraphExecDestroy(g.exec);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->d_irq) (void)hipFree(c->d_irq);
    if (c->d_cams) (void)hipFree(c->d_cams);
    if (c->d_cam_seq) (void)hipFree(c->d_cam_seq);
    if (c->h_cams) (void)hipHostFree((void *)c->h_cams);
    for (hipEvent_t ev : c->cam_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->h_stage) (void)hipHostFree((void *)c->h_stage);
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
float rthip_progress(rthip_ctx *c) {
    if (!c) return 0.f;
    const float done = c->progress.load();  // chunks finished (1 once the frame is complete)
    const uint32_t total = c->stages_total.load();
    if (done >= 1.f || !total || !c->h_stage) return done;
    const float staged = (float)c->h_stage[0] / (float)total;  // the stage the device last began, stored by its kernels
    const float p = staged > done ? staged : done;
    return p < 0.999f ? p : 0.999f;
}
int rthip_interrupt(rthip_ctx *c) {
    if (!c) return fail(RTHIP_E_INVALID, "null context");
    std::lock_guard<std::mutex> lock(c->irq_mutex);
    c->interrupt.store(1);
    // tell the kernels that are running or queued: a store to the page-locked word their watcher workgroup reads
    if (c->h_stage) {
        c->irq_dirty = true;
        __atomic_store_n(const_cast<uint32_t *>(c->h_stage + 32), 1u, __ATOMIC_RELEASE);
    }
    return 0;
}
int rthip_get_stats(rthip_ctx *c, rthip_stats *out) {
    if (!c || !out) return fail(RTHIP_E_INVALID, "null argument");
    c->last.graphs_captured = c->graphs_captured;
    c->last.graph_replays = c->graph_replays;
    *out = c->last;
    return 0;
}

// ---------------------------------------------------------------------------- scene upload
// keep_geometry: the mesh arrays are the ones already on the device (rthip_update_scene): only their counts are used
static int validate_scene(const rthip_scene_desc *s, uint32_t *max_depth, uint32_t *max_bvh_depth, bool keep_geometry, bool build_kd) {
#define BAD(...) return fail(RTHIP_E_INVALID, "rthip_upload_scene: " __VA_ARGS__)
    if (s->n_shapes && !s->shapes) BAD("shapes is null");
    if (s->n_lights && !s->lights) BAD("lights is null");
    if (s->n_materials && !s->materials) BAD("materials is null");
    if (s->n_noises && !s->noises) BAD("noises is null");
    // (a mesh without triangles is legal: Mesh::Compile builds a one-leaf kd-tree over nothing and no BVH)
    if (s->n_meshes && !s->meshes) BAD("meshes is null");
    if (!keep_geometry) {
        if (s->n_kd_nodes && (!s->kd_nodes || !s->kd_bounds)) BAD("kd node arrays are null");
        if (s->n_kd_refs && !s->kd_refs) BAD("kd_refs is null");
        if (s->n_tris && (!s->tris || !s->tri_normals)) BAD("triangle arrays are null");
        if (s->n_bvh_nodes && (!s->bvh_nodes || !s->bvh_tris || !s->bvh_tri_index)) BAD("bvh arrays are null");
    }
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
    // entries of bvh_tris / bvh_tri_index: n_tris when every mesh brings a BVH; otherwise only the meshes that do have records there
    uint64_t bvh_tris_total = 0;
    bool every_mesh_has_bvh = true;
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        if (s->meshes[i].bvh_node_count) bvh_tris_total += s->meshes[i].tri_count;
        else if (s->meshes[i].tri_count) every_mesh_has_bvh = false;
    }
    if (every_mesh_has_bvh) bvh_tris_total = s->n_tris;
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        const rthip_mesh &m = s->meshes[i];
        if ((uint64_t)m.node_base + m.node_count > s->n_kd_nodes) BAD("mesh %u: kd nodes out of range", i);
        if ((uint64_t)m.ref_base + m.ref_count > s->n_kd_refs) BAD("mesh %u: kd refs out of range", i);
        if ((uint64_t)m.tri_base + m.tri_count > s->n_tris) BAD("mesh %u: triangles out of range", i);
        if (m.max_depth > RTHIP_KD_MAX_DEPTH) BAD("mesh %u: max_depth %u", i, m.max_depth);
        if (m.max_depth > *max_depth) *max_depth = m.max_depth;
        if (m.bvh_depth > *max_bvh_depth) *max_bvh_depth = m.bvh_depth;
        if (keep_geometry) continue;  // (the table itself is compared with the uploaded one by the caller)
        if (m.tri_count && !m.node_count) {  // no kd-tree came with the mesh: the device builds it, if asked to
            if (!build_kd) BAD("mesh %u has %u triangles and no kd nodes (upload with RTHIP_UPLOAD_BUILD_KD, or supply the tree)", i, m.tri_count);
            if (!s->tri_verts || !s->mesh_bounds) BAD("mesh %u: RTHIP_UPLOAD_BUILD_KD needs tri_verts and mesh_bounds", i);
        }
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
        if (m.bvh_node_count && (uint64_t)m.bvh_tri_base + m.tri_count > bvh_tris_total) BAD("mesh %u: bvh triangles out of range", i);
        if (m.bvh_depth > *max_bvh_depth) *max_bvh_depth = m.bvh_depth;
        if (!(m.abs_max >= 0.0f)) BAD("mesh %u: abs_max", i);
        for (uint32_t k = 0; k < m.bvh_node_count; k++) {
            const rthip_bvhnode &nd = s->bvh_nodes[m.bvh_node_base + k];
            for (int c = 0; c < 4; c++) {
                const uint32_t w = nd.c[c].child;
                if (w == RTHIP_BVH_ABSENT) continue;
                if (w & RTHIP_BVH_LEAF) {
                    uint64_t first = w & RTHIP_BVH_FIRST_MASK, cnt = ((w >> RTHIP_BVH_COUNT_SHIFT) & 7u) + 1u;
                    if (cnt > RTHIP_BVH_MAX_LEAF || first + cnt > m.tri_count) BAD("mesh %u bvh node %u: leaf out of range", i, k);
                } else if (w >= m.bvh_node_count || w <= k) {
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
        if (bg.sun_light < -1 || bg.sun_light >= (int32_t)s->n_lights) BAD("skybox sun light %d out of range (-1 = none)", bg.sun_light);
    }
#undef BAD
    return 0;
}

static int upload(rthip_ctx *c, DevBuf &b, const void *src, size_t bytes) {
    if (b.ensure(bytes)) return RTHIP_E_HIP;
    if (bytes) HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

static int upload_scene(rthip_ctx *c, const rthip_scene_desc *s, bool keep_geometry, bool build_bvh = false, bool build_kd = false);

int rthip_upload_scene(rthip_ctx *c, const rthip_scene_desc *s) {
    if (!c || !s) return fail(RTHIP_E_INVALID, "rthip_upload_scene: null argument");
    return upload_scene(c, s, false);
}

int rthip_upload_scene_ex(rthip_ctx *c, const rthip_scene_desc *s, uint32_t flags) {
    if (!c || !s) return fail(RTHIP_E_INVALID, "rthip_upload_scene_ex: null argument");
    if (flags & ~(uint32_t)(RTHIP_UPLOAD_BUILD_BVH | RTHIP_UPLOAD_KEEP_GEOMETRY | RTHIP_UPLOAD_BUILD_KD)) return fail(RTHIP_E_INVALID, "rthip_upload_scene_ex: unknown flags 0x%x", flags);
    if (flags & RTHIP_UPLOAD_KEEP_GEOMETRY) return rthip_update_scene(c, s);
    return upload_scene(c, s, false, (flags & RTHIP_UPLOAD_BUILD_BVH) != 0, (flags & RTHIP_UPLOAD_BUILD_KD) != 0);
}

int rthip_update_scene(rthip_ctx *c, const rthip_scene_desc *s) {
    if (!c || !s) return fail(RTHIP_E_INVALID, "rthip_update_scene: null argument");
    if (!c->has_scene) return fail(RTHIP_E_STATE, "rthip_update_scene: no scene uploaded");
    const rthip_scene_desc &u = c->uploaded;
    bool same = s->n_meshes == u.n_meshes && s->n_kd_nodes == u.n_kd_nodes && s->n_kd_refs == u.n_kd_refs && s->n_kd_planes == u.n_kd_planes &&
                s->n_tris == u.n_tris && s->n_bvh_nodes == u.n_bvh_nodes && !(s->n_meshes && !s->meshes);
    for (uint32_t i = 0; same && i < s->n_meshes; i++) {  // everything but normal_mode (Mesh::NormalMode is read live, Scene.cpp:456) must match
        // ... the table the caller uploaded (a mixed upload rewrites bvh_tri_base / the bvh_* fields of the meshes whose BVH the
        // device built: the caller's next description still carries ITS values), or the one the device holds (a caller that
        // read it back with rthip_probe_bvh)
        rthip_mesh a = s->meshes[i];
        a.normal_mode = 0;
        rthip_mesh in = c->mesh_table_in[i], dev = c->mesh_table[i];
        in.normal_mode = dev.normal_mode = 0;
        same = (memcmp(&a, &in, sizeof a) == 0 || memcmp(&a, &dev, sizeof a) == 0) && (s->meshes[i].normal_mode == 0 || s->meshes[i].normal_mode == 1);
    }
    if (!same) return fail(RTHIP_E_INVALID, "rthip_update_scene: the mesh table / geometry counts differ from the uploaded scene");
    return upload_scene(c, s, true);
}

static int upload_scene_impl(rthip_ctx *c, const rthip_scene_desc *s, bool keep_geometry, bool build_bvh, bool build_kd);
static int upload_scene(rthip_ctx *c, const rthip_scene_desc *s, bool keep_geometry, bool build_bvh, bool build_kd) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!keep_geometry) c->upload_info = rthip_upload_info{};
    const int rc = upload_scene_impl(c, s, keep_geometry, build_bvh, build_kd);
    c->upload_info.upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// The kd-trees of the meshes that came without one, built on the device (rthip_kd.hip), appended behind the caller's trees in
// the scene-wide kd arrays; their mesh-table entries are filled in.  Called with the caller's arrays not yet uploaded.
// `stream`: the context's own, or - any_kd - a stream of the caller's second thread (the build is one long dependent chain per
// node that keeps a handful of waves busy: the BVH build and the triangle upload run beside it on the context's stream).
static int upload_kd_arrays(rthip_ctx *c, const rthip_scene_desc *s, bool any_kd, hipStream_t stream) {
    int rc;
    if (!any_kd) {
        if ((rc = upload(c, c->kd_nodes, s->kd_nodes, sizeof(rthip_kdnode) * s->n_kd_nodes))) return rc;
        if ((rc = upload(c, c->kd_bounds, s->kd_bounds, sizeof(double) * 6 * s->n_kd_nodes))) return rc;
        if ((rc = upload(c, c->kd_refs, s->kd_refs, sizeof(uint32_t) * s->n_kd_refs))) return rc;
        if ((rc = upload(c, c->kd_planes, s->kd_planes, sizeof(double) * s->n_kd_planes))) return rc;
        return 0;
    }
    struct Built { rthip_kdnode *nodes = nullptr; double *bounds = nullptr; uint32_t *refs = nullptr; double *planes = nullptr; rthip_device_kd_info info{}; };
    std::vector<Built> built(s->n_meshes);
    DevBuf verts;
    auto release = [&]() {
        for (Built &b : built) {
            if (b.nodes) (void)hipFree(b.nodes);
            if (b.bounds) (void)hipFree(b.bounds);
            if (b.refs) (void)hipFree(b.refs);
            if (b.planes) (void)hipFree(b.planes);
        }
        verts.release();
    };
    uint64_t n_nodes = s->n_kd_nodes, n_refs = s->n_kd_refs, n_planes = s->n_kd_planes;
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        if (!c->mesh_device_kd[i]) continue;
        const rthip_mesh &m = c->mesh_table[i];
        if (verts.ensure(sizeof(double) * 9 * (size_t)m.tri_count)) { release(); return RTHIP_E_HIP; }
        hipError_t e = hipMemcpyAsync(verts.p, s->tri_verts + (size_t)m.tri_base * 9, sizeof(double) * 9 * (size_t)m.tri_count, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { release(); return fail(RTHIP_E_HIP, "rthip_upload_scene: %s", hipGetErrorString(e)); }
        char msg[256] = "";
        rc = rthip_kd_build_device(stream, (const double *)verts.p, m.tri_count, s->mesh_bounds + (size_t)i * 6, &built[i].nodes, &built[i].bounds,
                                   &built[i].refs, &built[i].planes, &built[i].info, msg, sizeof msg);
        if (rc) { release(); return fail(rc, "rthip_upload_scene: mesh %u: %s", i, msg); }
        const rthip_device_kd_info &k = built[i].info;
        n_nodes += k.n_nodes;
        n_refs += k.n_refs;
        n_planes += (uint64_t)k.plane_count[0] + k.plane_count[1] + k.plane_count[2];
        c->upload_info.kd_meshes++;
        c->upload_info.kd_nodes += k.n_nodes;
        c->upload_info.kd_refs += k.n_refs;
        c->upload_info.kd_device_ms += k.device_ms;
        if (k.levels > c->upload_info.kd_levels) c->upload_info.kd_levels = k.levels;
    }
    if (n_nodes > 0xfffffff0ull || n_refs > 0xfffffff0ull || n_planes > 0xfffffff0ull) { release(); return fail(RTHIP_E_UNSUPPORTED, "rthip_upload_scene: more than 2^32 kd nodes / references"); }
    if (c->kd_nodes.ensure(sizeof(rthip_kdnode) * n_nodes) || c->kd_bounds.ensure(sizeof(double) * 6 * n_nodes) || c->kd_refs.ensure(sizeof(uint32_t) * n_refs) ||
        c->kd_planes.ensure(sizeof(double) * n_planes)) { release(); return RTHIP_E_HIP; }
    hipError_t e = hipSuccess;
    auto copy = [&](void *dst, const void *src, size_t bytes, hipMemcpyKind kind) { if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, kind, stream); };
    copy(c->kd_nodes.p, s->kd_nodes, sizeof(rthip_kdnode) * s->n_kd_nodes, hipMemcpyHostToDevice);
    copy(c->kd_bounds.p, s->kd_bounds, sizeof(double) * 6 * s->n_kd_nodes, hipMemcpyHostToDevice);
    copy(c->kd_refs.p, s->kd_refs, sizeof(uint32_t) * s->n_kd_refs, hipMemcpyHostToDevice);
    copy(c->kd_planes.p, s->kd_planes, sizeof(double) * s->n_kd_planes, hipMemcpyHostToDevice);
    uint64_t nb = s->n_kd_nodes, rb = s->n_kd_refs, pb = s->n_kd_planes;
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        if (!c->mesh_device_kd[i]) continue;
        rthip_mesh &m = c->mesh_table[i];
        const rthip_device_kd_info &k = built[i].info;
        copy((rthip_kdnode *)c->kd_nodes.p + nb, built[i].nodes, sizeof(rthip_kdnode) * k.n_nodes, hipMemcpyDeviceToDevice);
        copy((double *)c->kd_bounds.p + 6 * nb, built[i].bounds, sizeof(double) * 6 * k.n_nodes, hipMemcpyDeviceToDevice);
        copy((uint32_t *)c->kd_refs.p + rb, built[i].refs, sizeof(uint32_t) * k.n_refs, hipMemcpyDeviceToDevice);
        const uint64_t np = (uint64_t)k.plane_count[0] + k.plane_count[1] + k.plane_count[2];
        copy((double *)c->kd_planes.p + pb, built[i].planes, sizeof(double) * np, hipMemcpyDeviceToDevice);
        m.node_base = (uint32_t)nb; m.node_count = k.n_nodes;
        m.ref_base = (uint32_t)rb; m.ref_count = k.n_refs;
        m.max_depth = k.max_depth;
        uint64_t at = pb;
        for (int a = 0; a < 3; a++) { m.plane_base[a] = (uint32_t)at; m.plane_count[a] = k.plane_count[a]; at += k.plane_count[a]; }
        nb += k.n_nodes; rb += k.n_refs; pb += np;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    release();
    if (e != hipSuccess) return fail(RTHIP_E_HIP, "rthip_upload_scene: %s", hipGetErrorString(e));
    return 0;
}

static int upload_scene_impl(rthip_ctx *c, const rthip_scene_desc *s, bool keep_geometry, bool build_bvh, bool build_kd) {
    uint32_t max_depth = 0, max_bvh_depth = 0;
    int rc = validate_scene(s, &max_depth, &max_bvh_depth, keep_geometry, build_kd);
    if (rc) return rc;
    if (set_device(c)) return RTHIP_E_HIP;
    c->has_scene = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (keep_geometry) {  // the mesh table goes up again: its normal_mode fields may have changed
        for (uint32_t i = 0; i < s->n_meshes; i++) c->mesh_table[i].normal_mode = s->meshes[i].normal_mode;
        if ((rc = upload(c, c->meshes, c->mesh_table.data(), sizeof(rthip_mesh) * s->n_meshes))) return rc;
    }
    if (!keep_geometry) {
        // host copies of what an update needs later: the mesh table and every mesh's BVH root
        c->mesh_table.assign(s->meshes, s->meshes + s->n_meshes);
        c->mesh_table_in = c->mesh_table;
        c->mesh_roots.assign(s->n_meshes, rthip_bvhnode{});
        c->mesh_device_bvh.assign(s->n_meshes, 0);
        c->mesh_device_kd.assign(s->n_meshes, 0);
        bool any_kd = false;
        for (uint32_t i = 0; i < s->n_meshes; i++)
            if (build_kd && s->meshes[i].tri_count && !s->meshes[i].node_count) any_kd = c->mesh_device_kd[i] = 1;
        // any_device: some mesh gets its BVH built here.  `mixed`: some meshes bring a BVH and others (with triangles) do not -
        // the caller's bvh_tris / bvh_tri_index then hold only the records of the former (at their bvh_tri_base), so they are
        // copied mesh by mesh too, never as n_tris records in one piece
        bool any_device = false, mixed = false;
        for (uint32_t i = 0; i < s->n_meshes; i++) {
            if (s->meshes[i].bvh_node_count) c->mesh_roots[i] = s->bvh_nodes[s->meshes[i].bvh_node_base];
            else if (build_bvh && s->meshes[i].tri_count) any_device = c->mesh_device_bvh[i] = 1;
            else if (s->meshes[i].tri_count && s->n_bvh_nodes) mixed = true;
        }
        // the kd build runs on a second thread + stream while this one uploads the triangles and builds the BVHs; joined
        // before the mesh table goes up (a std::async future also joins when an early return destroys it)
        std::future<int> kd_job;
        std::string kd_err;
        hipStream_t kd_stream = nullptr;
        struct StreamGuard { hipStream_t &s; ~StreamGuard() { if (s) (void)hipStreamDestroy(s); } } kd_stream_guard{kd_stream};
        if (any_kd && hipStreamCreateWithFlags(&kd_stream, hipStreamNonBlocking) == hipSuccess) {
            kd_job = std::async(std::launch::async, [&]() {
                int r = hipSetDevice(c->device) == hipSuccess ? upload_kd_arrays(c, s, true, kd_stream) : RTHIP_E_HIP;
                if (r) kd_err = rthip_last_error();
                return r;
            });
        } else {
            (void)hipGetLastError();
            kd_stream = nullptr;
            if ((rc = upload_kd_arrays(c, s, any_kd, c->stream))) return rc;
        }
        struct JoinGuard { std::future<int> &f; ~JoinGuard() { if (f.valid()) f.wait(); } } kd_join_guard{kd_job};  // (before the stream goes)
        if ((rc = upload(c, c->tris, s->tris, sizeof(rthip_tri) * s->n_tris))) return rc;
        if ((rc = upload(c, c->tri_normals, s->tri_normals, sizeof(double) * 9 * s->n_tris))) return rc;
        if (!any_device && !mixed) {
            if ((rc = upload(c, c->bvh_nodes, s->bvh_nodes, sizeof(rthip_bvhnode) * s->n_bvh_nodes))) return rc;
            if ((rc = upload(c, c->bvh_tris, s->bvh_tris, s->n_bvh_nodes ? sizeof(rthip_tri) * s->n_tris : 0))) return rc;
            if ((rc = upload(c, c->bvh_index, s->bvh_tri_index, s->n_bvh_nodes ? sizeof(uint32_t) * s->n_tris : 0))) return rc;
        } else {
            // RTHIP_UPLOAD_BUILD_BVH: the meshes that came without a BVH get theirs built here, on the device, from the
            // triangle records just uploaded (rthip_bvh.hip).  On the device every mesh's leaf-order arrays sit at its
            // tri_base; the nodes of the caller's BVHs come first, the library's are appended.
            if (c->bvh_tris.ensure(sizeof(rthip_tri) * s->n_tris)) return RTHIP_E_HIP;
            if (c->bvh_index.ensure(sizeof(uint32_t) * s->n_tris)) return RTHIP_E_HIP;
            struct Built { rthip_bvhnode *nodes; rthip_device_bvh_info info; };
            std::vector<Built> built(s->n_meshes, Built{nullptr, {}});
            auto release = [&]() { for (Built &b : built) if (b.nodes) (void)hipFree(b.nodes); };
            uint64_t total = s->n_bvh_nodes;
            for (uint32_t i = 0; i < s->n_meshes; i++) {
                rthip_mesh &m = c->mesh_table[i];
                rthip_tri *leaf_tris = (rthip_tri *)c->bvh_tris.p + m.tri_base;
                uint32_t *leaf_index = (uint32_t *)c->bvh_index.p + m.tri_base;
                if (!c->mesh_device_bvh[i]) {
                    if (m.bvh_node_count) {
                        hipError_t e = hipMemcpyAsync(leaf_tris, s->bvh_tris + m.bvh_tri_base, sizeof(rthip_tri) * m.tri_count, hipMemcpyHostToDevice, c->stream);
                        if (e == hipSuccess) e = hipMemcpyAsync(leaf_index, s->bvh_tri_index + m.bvh_tri_base, sizeof(uint32_t) * m.tri_count, hipMemcpyHostToDevice, c->stream);
                        if (e != hipSuccess) { release(); return fail(RTHIP_E_HIP, "rthip_upload_scene: %s", hipGetErrorString(e)); }
                        m.bvh_tri_base = m.tri_base;
                    }
                    continue;
                }
                char msg[256] = "";
                rc = rthip_bvh_build_device(c->stream, (const rthip_tri *)c->tris.p + m.tri_base, m.tri_count, leaf_tris, leaf_index, &built[i].nodes,
                                            &built[i].info, msg, sizeof msg);
                if (rc) { release(); return fail(rc, "rthip_upload_scene: mesh %u: %s", i, msg); }
                total += built[i].info.n_nodes;
                c->upload_info.bvh_meshes++;
                c->upload_info.bvh_nodes += built[i].info.n_nodes;
            }
            if (total > 0xfffffff0ull) { release(); return fail(RTHIP_E_UNSUPPORTED, "rthip_upload_scene: more than 2^32 BVH nodes"); }
            if (c->bvh_nodes.ensure(sizeof(rthip_bvhnode) * total)) { release(); return RTHIP_E_HIP; }
            hipError_t e = hipSuccess;
            if (s->n_bvh_nodes) e = hipMemcpyAsync(c->bvh_nodes.p, s->bvh_nodes, sizeof(rthip_bvhnode) * s->n_bvh_nodes, hipMemcpyHostToDevice, c->stream);
            uint64_t base = s->n_bvh_nodes;
            for (uint32_t i = 0; i < s->n_meshes && e == hipSuccess; i++) {
                if (!c->mesh_device_bvh[i]) continue;
                rthip_mesh &m = c->mesh_table[i];
                const rthip_device_bvh_info &info = built[i].info;
                e = hipMemcpyAsync((rthip_bvhnode *)c->bvh_nodes.p + base, built[i].nodes, sizeof(rthip_bvhnode) * info.n_nodes, hipMemcpyDeviceToDevice, c->stream);
                m.bvh_node_base = (uint32_t)base;
                m.bvh_node_count = info.n_nodes;
                m.bvh_tri_base = m.tri_base;
                m.bvh_depth = info.depth;
                m.abs_max = info.abs_max;
                c->mesh_roots[i] = info.root;
                base += info.n_nodes;
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            release();
            if (e != hipSuccess) return fail(RTHIP_E_HIP, "rthip_upload_scene: %s", hipGetErrorString(e));
        }
        if (kd_job.valid() && (rc = kd_job.get())) return fail(rc, "%s", kd_err.c_str());
        if ((rc = upload(c, c->meshes, c->mesh_table.data(), sizeof(rthip_mesh) * s->n_meshes))) return rc;
    }
    max_bvh_depth = 0;  // from the table the device holds (it carries the depths of the trees built here)
    max_depth = 0;
    for (const rthip_mesh &m : c->mesh_table) {
        max_bvh_depth = m.bvh_depth > max_bvh_depth ? m.bvh_depth : max_bvh_depth;
        max_depth = m.max_depth > max_depth ? m.max_depth : max_depth;
    }
    std::vector<rthip_shape> shapes(s->shapes, s->shapes + s->n_shapes);
    for (rthip_shape &sh : shapes) roundtrip_bounds(sh, c->mesh_table.data(), c->mesh_roots.data());
    if ((rc = upload(c, c->shapes, shapes.data(), sizeof(rthip_shape) * s->n_shapes))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));  // `shapes` is a local
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
    v.shadow_ld = v.plane_wn = nullptr;
    {   // what is the same for every ray, once: the directional lights' directions in every shape's frame and the planes' world
        // normals (k_light_dirs); RTHIP_SHADOW_LD=0: every lane computes its own
        const char *env = getenv("RTHIP_SHADOW_LD");  // (read at every upload: tests compare the two)
        const bool enabled = !(env && atoi(env) == 0);
        const size_t pairs = (size_t)s->n_lights * s->n_shapes, threads = pairs > s->n_shapes ? pairs : s->n_shapes;
        if (enabled && threads) {
            if (c->shadow_ld.ensure(sizeof(double) * 3 * (pairs + s->n_shapes))) return RTHIP_E_HIP;
            double *ld = (double *)c->shadow_ld.p, *wn = ld + 3 * pairs;
            hipLaunchKernelGGL(k_light_dirs, dim3((uint32_t)((threads + 63) / 64)), dim3(64), 0, c->stream, v, ld, wn);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(c->stream));
            v.shadow_ld = pairs ? ld : nullptr;
            v.plane_wn = wn;
        }
    }
    c->n_tables = s->n_perlin_tables;
    {   // the LEAN instantiations of k_trace_easy / k_shade (rtc::noise_sample); RTHIP_LEAN=0: never
        const char *env = getenv("RTHIP_LEAN");  // (read at every upload: tests compare the two)
        bool lean = !(env && atoi(env) == 0);
        for (uint32_t i = 0; lean && i < s->n_shapes; i++) {
            const int32_t root = s->shapes[i].material;
            if (root < 0 || (uint32_t)root >= s->n_materials) { lean = false; break; }
            const rthip_matop &r = s->materials[root];
            if (r.op == RTHIP_MAT_UNIFORM) continue;
            lean = root - r.first_op == 2 && root >= 2 && s->materials[root - 2].op == RTHIP_MAT_UNIFORM && s->materials[root - 1].op == RTHIP_MAT_UNIFORM &&
                   r.noise >= 0 && (uint32_t)r.noise < s->n_noises &&
                   (s->noises[r.noise].type == RTHIP_NOISE_CHECKER || s->noises[r.noise].type == RTHIP_NOISE_PERLIN);
        }
        c->lean_scene = lean;
    }
    c->has_meshes = s->n_meshes > 0;
    c->mesh_without_bvh = -1;
    for (uint32_t i = 0; i < s->n_meshes; i++)
        if (c->mesh_table[i].tri_count && !c->mesh_table[i].bvh_node_count && c->mesh_without_bvh < 0) c->mesh_without_bvh = (int)i;
    {   // Which walk kernel serves this scene's LONG lists: k_trace_coop, for every scene with meshes.  Measured: the bunny +8 % (its
        // short walks leave the per-ray work a 40 % share), config 5's five meshes -6.4 % of a frame (a transform per mesh and query),
        // config 4's one 2 M-triangle mesh +-0.3 % (walks that wait for HBM).  RTHIP_QUAD_COOP=0 / 1 forces it; RTHIP_COOP_MAX_TREE_MB=<n>
        // restricts it to scenes whose BVH nodes + leaf-order triangles take at most n MB (0 = no limit, the default).
        size_t bytes = 0;
        for (const rthip_mesh &m : c->mesh_table) bytes += (size_t)m.bvh_node_count * sizeof(rthip_bvhnode) + (size_t)m.tri_count * (sizeof(rthip_tri) + 4u);
        c->walk_coop = c->coop_forced >= 0 ? c->coop_forced == 1 : (c->has_meshes && (c->coop_max_bytes == 0 || bytes <= c->coop_max_bytes));
    }
    v.bg = s->background;
    c->stack_cap_kd = max_depth + 2;
    c->bvh_depth = max_bvh_depth;
    c->stack_cap_bvh = 3 * (max_bvh_depth + 1) + 1;  // one ray per lane over the 4-wide BVH: up to three pending children per level, the root's included
    c->stack_cap_full = (max_depth + 2 > c->stack_cap_bvh ? max_depth + 2 : c->stack_cap_bvh);  // degenerate rays walk the kd-tree
    c->stack_cap = c->stack_cap_full;
    c->uploaded = *s;  // counts only are read later (rthip_update_scene)
    c->has_scene = true;
    c->scene_generation++;
    return 0;
}

// ---------------------------------------------------------------------------- frame

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

// renders this context's part of n_frames frames (one camera each) into d_pixels (device; frame after frame, each
// W*H or, tile-compact, rthip_part_pixels) and optionally device id images (single frames only)
// noalloc: the workspace is carved out of this device's `scratch` array (g_scratch_ptr) instead of the context's DevBufs;
// no HIP graph, no event pool, no chunking - nothing on that path allocates (rthip_render_device_noalloc).
static int render_frame(rthip_ctx *c, const rthip_camera *cam, uint32_t n_frames, const rthip_params *p, uint32_t *d_pixels,
                        int32_t *d_shape, int32_t *d_tri, bool noalloc = false) {
    int rc = check_params(cam, p);
    if (rc) return rc;
    if (n_frames == 0 || n_frames > RTHIP_MAX_BATCH) return fail(RTHIP_E_INVALID, "rthip_render: %u frames per call, limit %d", n_frames, RTHIP_MAX_BATCH);
    if (n_frames > 1 && (d_shape || d_tri)) return fail(RTHIP_E_INVALID, "rthip_render: id images are for single frames");
    if (!c->has_scene) return fail(RTHIP_E_STATE, "rthip_render: no scene uploaded");
    if (p->accel == 0 && c->mesh_without_bvh >= 0)
        return fail(RTHIP_E_INVALID, "rthip_render: mesh %d has triangles but no BVH (bvh_node_count = 0): accel = 0 would leave it out of the image; "
                    "supply bvh_nodes / bvh_tris / bvh_tri_index or render with accel = 1 (kd-tree)", c->mesh_without_bvh);
    if (set_device(c)) return RTHIP_E_HIP;
    {   // a new frame starts uninterrupted (m_Status = { false, false, 0, m_Pixels }, Raytracer.cpp:36); an Interrupt() that
        // arrives from here on belongs to this frame
        std::lock_guard<std::mutex> lock(c->irq_mutex);
        if (c->irq_dirty) {  // the flag of an earlier frame is still up: let that frame's kernels drain, then lower it
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->h_stage[32] = 0;
            HIP_TRY(hipMemset(c->d_irq, 0, sizeof(uint32_t)));
            c->irq_dirty = false;
        }
        c->interrupt.store(0);
    }
    c->progress.store(0.f);
    c->stages_total.store(0);
    if (c->h_stage) c->h_stage[0] = 0;

    FrameParams fp;
    memset(&fp, 0, sizeof fp);
    fp.cams = c->d_cams;
    fp.n_frames = n_frames;
    fp.part_slots = (uint32_t)rthip_part_pixels(p);
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
    const uint32_t frame_tiles = n_tiles > fp.part_index ? (n_tiles - fp.part_index + fp.part_count - 1) / fp.part_count : 0;
    fp.my_tiles = frame_tiles ? frame_tiles : 1;
    fp.d_per_tile = rtc::make_fastdiv(fp.tile_w * fp.tile_h);
    fp.d_my_tiles = rtc::make_fastdiv(fp.my_tiles);
    fp.d_tiles_x = rtc::make_fastdiv(fp.tiles_x);
    fp.d_sbx = rtc::make_fastdiv(fp.tile_w >> 3);
    const uint32_t my_tiles = frame_tiles * n_frames;  // tile ordinals of the whole batch
    const uint32_t per_tile = fp.tile_w * fp.tile_h;
    const uint32_t spp = fp.aa ? 9 : 1;
    const uint32_t levels = fp.max_rec + 1;
    size_t chunk_rays = c->chunk_rays;
    if (!c->chunk_rays_forced) chunk_rays = std::min(chunk_rays, kMaxWorkspaceBytes / (36 * (size_t)levels + 216));
    uint32_t tiles_per_chunk = (uint32_t)(chunk_rays / ((size_t)per_tile * spp));
    if (tiles_per_chunk == 0) tiles_per_chunk = 1;
    if (tiles_per_chunk > my_tiles) tiles_per_chunk = my_tiles ? my_tiles : 1;
    const size_t N = (size_t)tiles_per_chunk * per_tile * spp;

    Work w;
    uint32_t bs_stride = (uint32_t)((N + kBlock - 1) / kBlock);
    if (!noalloc) {
        if (c->rec.ensure(sizeof(ColRec) * N * levels)) return RTHIP_E_HIP;
        if (c->child.ensure(sizeof(uint32_t) * N * levels)) return RTHIP_E_HIP;
        if (c->rays.ensure(sizeof(RayRec) * 2 * N)) return RTHIP_E_HIP;
        if (c->counts.ensure(sizeof(uint32_t) * kCounterSlots * kCounterStride)) return RTHIP_E_HIP;
        if (c->stats.ensure(sizeof(unsigned long long) * (ST_COUNT + 2 * kShadeCounts))) return RTHIP_E_HIP;
        c->bs_stride = bs_stride;
        c->bs_levels = levels;
        if (c->block_stats.ensure(sizeof(uint2) * 2 * (size_t)c->bs_stride * levels)) return RTHIP_E_HIP;
        if (c->ids.ensure(sizeof(int32_t) * 2 * (size_t)tiles_per_chunk * per_tile)) return RTHIP_E_HIP;
        if (c->hits.ensure(sizeof(HitSlot) * N)) return RTHIP_E_HIP;
        if (c->redo.ensure(sizeof(uint32_t) * 2 * N)) return RTHIP_E_HIP;
        if (c->pend.ensure(sizeof(PendRec) * N)) return RTHIP_E_HIP;
        if (c->redo_stack.ensure(sizeof(uint32_t) * kRedoBlocks * kBlock * (size_t)c->stack_cap_full)) return RTHIP_E_HIP;
        w.redo_stack = (uint32_t *)c->redo_stack.p;
        w.rec = (ColRec *)c->rec.p;
        w.child = (uint32_t *)c->child.p;
        w.rays = (RayRec *)c->rays.p;
        w.counts = (uint32_t *)c->counts.p;
        w.stats = (unsigned long long *)c->stats.p;
        w.id_shape = (d_shape || d_tri) ? (int32_t *)c->ids.p : nullptr;
        w.id_tri = w.id_shape ? w.id_shape + (size_t)tiles_per_chunk * per_tile : nullptr;
        w.hit = (HitSlot *)c->hits.p;
        w.redo = (uint32_t *)c->redo.p;
        w.pend = (PendRec *)c->pend.p;
        w.block_stats = (uint2 *)c->block_stats.p;
    } else {
        // the same records, one after the other in this GPU's `scratch` array (256-byte aligned each)
        if (tiles_per_chunk < my_tiles)
            return fail(RTHIP_E_UNSUPPORTED, "rthip_render_device_noalloc: %ux%u (x%u frames%s) is %zu level-0 rays, more than one workspace chunk (%zu)",
                        p->width, p->height, n_frames, fp.aa ? ", SSAA" : "", (size_t)my_tiles * per_tile * spp, chunk_rays);
        unsigned char *base = c->device < kMaxDevices ? (unsigned char *)g_scratch_ptr[c->device] : nullptr;
        if (!base) return fail(RTHIP_E_STATE, "rthip_render_device_noalloc: no scratch address for device %d", c->device);
        size_t off = 0;
        auto take = [&](size_t bytes) {
            unsigned char *q = base + off;
            off += (bytes + 255) & ~(size_t)255;
            return q;
        };
        w.rec = (ColRec *)take(sizeof(ColRec) * N * levels);
        w.child = (uint32_t *)take(sizeof(uint32_t) * N * levels);
        w.rays = (RayRec *)take(sizeof(RayRec) * 2 * N);
        w.counts = (uint32_t *)take(sizeof(uint32_t) * kCounterSlots * kCounterStride);
        w.stats = (unsigned long long *)take(sizeof(unsigned long long) * (ST_COUNT + 2 * kShadeCounts));
        w.block_stats = (uint2 *)take(sizeof(uint2) * 2 * (size_t)bs_stride * levels);
        w.hit = (HitSlot *)take(sizeof(HitSlot) * N);
        w.redo = (uint32_t *)take(sizeof(uint32_t) * 2 * N);
        w.pend = (PendRec *)take(sizeof(PendRec) * N);
        w.redo_stack = (uint32_t *)take(sizeof(uint32_t) * kRedoBlocks * kBlock * (size_t)c->stack_cap_full);
        w.id_shape = w.id_tri = nullptr;
        if (off > RTHIP_NOALLOC_SCRATCH_BYTES)
            return fail(RTHIP_E_UNSUPPORTED, "rthip_render_device_noalloc: %ux%u, max_recursion %u%s needs a %zu-byte workspace, the static scratch "
                        "holds %u (about %zu level-0 rays at this recursion depth)", p->width, p->height, p->max_recursion, fp.aa ? ", SSAA" : "",
                        off, RTHIP_NOALLOC_SCRATCH_BYTES, (size_t)RTHIP_NOALLOC_SCRATCH_BYTES / (36 * (size_t)levels + 216));
    }
    w.hard = w.redo + N;
    w.bs_stride = bs_stride;
    w.N = (uint32_t)N;
    w.irq = c->d_irq;
    w.irq_host = (const uint32_t *)(c->h_stage + 32);
    w.stage = (uint32_t *)c->h_stage;
    w.hist = nullptr;
    w.hist_tag = 0;
    w.stage_base = 0;
    const uint32_t stages_per_chunk = STAGES_PER_LEVEL * levels + 1;  // + k_resolve
    c->stages_total.store(((my_tiles + tiles_per_chunk - 1) / tiles_per_chunk) * stages_per_chunk);

    c->sv.accel = p->accel;
    c->stack_cap = c->stack_cap_full;
    const size_t lds_bytes = (size_t)c->stack_cap * kBlock * sizeof(uint32_t);
    if (lds_bytes > 160 * 1024) return fail(RTHIP_E_UNSUPPORTED, "kd stack of %u entries does not fit LDS", c->stack_cap);
    if (lds_bytes > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    // hot kernel: one short stack per quad (child word + entry parameter per entry); deeper rays overflow into the redo list
    // cannot overflow, see quad::Stack: the deepest node is bvh_depth levels below the root, so a path makes bvh_depth + 1 steps
    // of at most three pushes (3 * depth + 1 was one level short: a mesh whose BVH is a lone root overflowed with two pushes)
    const uint32_t quad_cap = 3 * (c->bvh_depth + 1);
    // the quad kernel's occupancy variant (opt-in, RTHIP_QUAD_OCC6=1): see k_trace_quad
    static const bool quad_occ6 = [] { const char *e = getenv("RTHIP_QUAD_OCC6"); return e && atoi(e) != 0; }();
    static const uint32_t max_grid_level = [] { const char *e = getenv("RTHIP_GRID_LEVEL"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : kMaxGridLevel; }();  // (tuning)
    static const uint32_t max_grid_quad = [] { const char *e = getenv("RTHIP_GRID_QUAD"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : kMaxGridQuad; }();
    const size_t lds_quad = (size_t)quad::park_offset(quad_cap) + (size_t)quad::kRaysPerBlock * quad::kParkStride;
    if (lds_quad > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace_quad<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_quad));
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace_quad<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_quad));
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace_quad<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_quad));
    }
    // the walk kernel with one ray per lane outside the walks (k_trace_coop) for the long lists (upload_scene: walk_coop)
    const bool quad_coop = c->walk_coop && !quad_occ6;
    // Which of the two walk kernels a level launches.  The length of its list is on the device, so a scene with small trees gets
    // both and one returns at once (~2.5 us of a lone frame's chain per level) - unless the last call of this very shape (frames,
    // size, depth, part) has reported its lists (k_resolve -> page-locked words) and the level's was clearly on one side of the
    // threshold: then only that kernel is launched, with no threshold of its own.  A wrong guess costs speed, never a ray.
    enum { WALK_BOTH = 0, WALK_QUAD = 1, WALK_COOP = 2 };
    uint8_t walk_mode[RTHIP_MAX_RECURSION + 1];
    for (uint32_t l = 0; l <= RTHIP_MAX_RECURSION; l++) walk_mode[l] = quad_coop ? WALK_BOTH : WALK_QUAD;
    if (quad_coop && c->h_stage && !c->count_kd) {
        uint32_t tag = 2166136261u;
        auto mix = [&](const void *ptr, size_t nb) { for (size_t k = 0; k < nb; k++) tag = (tag ^ ((const unsigned char *)ptr)[k]) * 16777619u; };
        mix(&n_frames, sizeof n_frames);
        mix(p, sizeof *p);
        mix(&c->scene_generation, sizeof c->scene_generation);
        if (!tag) tag = 1;
        w.hist = (uint32_t *)c->h_stage + 8;  // words 8 .. 8 + levels of the stage block (0: stage, 32: interrupt request)
        w.hist_tag = tag;
        if (__atomic_load_n(c->h_stage + 8, __ATOMIC_ACQUIRE) == tag)
            for (uint32_t l = 0; l < levels; l++) {
                const uint64_t listed = c->h_stage[9 + l];
                walk_mode[l] = listed * 10u < (uint64_t)c->coop_thr32 * 7u ? WALK_QUAD : (listed * 10u > (uint64_t)c->coop_thr32 * 14u ? WALK_COOP : WALK_BOTH);
            }
    }
    static const uint32_t max_grid_coop = [] { const char *e = getenv("RTHIP_GRID_COOP"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : kMaxGridCoop; }();
    const uint32_t coop_thr32 = c->coop_thr32, coop_thr64 = c->coop_thr64;
    const size_t lds_coop = (size_t)(kBlock / 64u) * coop::wave_bytes(quad_cap);
    if (quad_coop && lds_coop > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace_coop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_coop));
        HIP_TRY(hipFuncSetAttribute((const void *)k_trace_coop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_coop));
    }
    static const bool debug_occ = [] { const char *e = getenv("RTHIP_DEBUG_OCC"); return e && atoi(e) != 0; }();
    if (debug_occ) {  // resident workgroups per CU of the two walk kernels at this scene's LDS sizes (tuning aid)
        int bq = 0, bc = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&bq, (const void *)k_trace_quad<false>, (int)kBlock, lds_quad);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&bc, (const void *)k_trace_coop<false>, (int)kBlock, lds_coop);
        fprintf(stderr, "rthip: quad_cap %u: k_trace_quad %zu B LDS -> %d workgroups / CU, k_trace_coop %zu B -> %d\n", quad_cap, lds_quad, bq, lds_coop, bc);
    }
    const size_t lds_tables = (size_t)c->n_tables * 256;

    // ---- the call's cameras -> this call's slot of the page-locked ring (k_begin fetches them; see kCamSlots)
    auto stage_cams = [&]() -> int {
        const uint32_t slot = c->cam_seq % kCamSlots;
        if (c->cam_seq >= kCamSlots) HIP_TRY(hipEventSynchronize(c->cam_ev[slot]));  // the call that last used the slot has finished
        memcpy(c->h_cams + (size_t)slot * RTHIP_MAX_BATCH, cam, sizeof(rthip_camera) * n_frames);
        return 0;
    };
    auto cams_enqueued = [&]() -> int {  // behind the call's last launch
        HIP_TRY(hipEventRecord(c->cam_ev[c->cam_seq % kCamSlots], c->stream));
        c->cam_seq++;
        return 0;
    };
    bool capturing = false;

    // ---- what a cached graph depends on
    const bool single_chunk = tiles_per_chunk >= my_tiles;
    const bool profile = c->profile_launches && !noalloc;
    const bool use_graph = c->graphs_enabled && single_chunk && !profile && !noalloc;
    std::vector<unsigned char> key;
    if (use_graph) {
        auto put = [&](const void *ptr, size_t n) { key.insert(key.end(), (const unsigned char *)ptr, (const unsigned char *)ptr + n); };
        const void *ptrs[] = {d_pixels, d_shape, d_tri, c->rec.p, c->child.p, c->rays.p, c->counts.p, c->stats.p, c->block_stats.p,
                              c->ids.p, c->hits.p, c->redo.p, c->pend.p, c->redo_stack.p};
        put(&n_frames, sizeof n_frames);  // (NOT the cameras: they reach the kernels through the camera ring, see k_begin)
        put(p, sizeof *p);
        put(ptrs, sizeof ptrs);
        put(&c->scene_generation, sizeof c->scene_generation);
        put(&c->count_kd, sizeof c->count_kd);
        put(walk_mode, sizeof walk_mode);
        for (const auto &g : c->graphs) {
            if (g.key != key) continue;
            if ((rc = stage_cams())) return rc;
            HIP_TRY(hipEventRecord(c->ev0, c->stream));
            HIP_TRY(hipGraphLaunch(g.exec, c->stream));
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
            if ((rc = cams_enqueued())) return rc;
            c->graph_replays++;
            c->last.pixels = (uint64_t)my_tiles * per_tile;
            c->last.launches_trace = 0;
            c->last_frames = n_frames;
            c->seen_key = key;
            return 0;
        }
    }
    size_t ev_used = 0;
    uint64_t pixels_done = 0;
    bool interrupted = false;
    auto enqueue = [&]() -> int {
    {   // statistics, list counters of the first chunk, the call's cameras: one launch (k_begin)
        const uint32_t na = 2 * (ST_COUNT + 2 * kShadeCounts), nb = (uint32_t)(2 * 2 * (size_t)bs_stride * levels), nc = kCounterSlots * kCounterStride;
        const uint32_t most = na > nb ? (na > nc ? na : nc) : (nb > nc ? nb : nc);
        hipLaunchKernelGGL(k_begin, dim3((most + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, (uint32_t *)w.stats, na,
                           (uint32_t *)w.block_stats, nb, w.counts, nc, c->d_cams, n_frames, (const rthip_camera *)c->h_cams, c->d_cam_seq,
                           capturing ? kSeqAuto : c->cam_seq);
    }
    auto get_event = [&](hipEvent_t *e) -> int {
        if (ev_used == c->ev_pool.size()) {
            hipEvent_t ne;
            HIP_TRY(hipEventCreate(&ne));
            c->ev_pool.push_back(ne);
        }
        *e = c->ev_pool[ev_used++];
        return 0;
    };
    for (uint32_t q0 = 0; q0 < my_tiles; q0 += tiles_per_chunk) {
        if (c->interrupt.load()) { interrupted = true; break; }
        uint32_t nq = my_tiles - q0 < tiles_per_chunk ? my_tiles - q0 : tiles_per_chunk;
        fp.q0 = q0;
        fp.n_pix_slots = nq * per_tile;
        fp.d_pix_slots = rtc::make_fastdiv(fp.n_pix_slots);
        fp.N0 = fp.n_pix_slots * spp;
        w.stage_base = (q0 / tiles_per_chunk) * stages_per_chunk + 1;  // (stage 0 = nothing begun yet)
        if (q0)  // (the first chunk's were cleared with the statistics)
            hipLaunchKernelGGL(k_clear, dim3((kCounterSlots * kCounterStride + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, w.counts,
                               kCounterSlots * kCounterStride, (uint32_t *)nullptr, 0u, (uint32_t *)nullptr, 0u);
        for (uint32_t level = 0; level < levels; level++) {
            uint32_t grid = (fp.N0 + kBlock - 1) / kBlock;
            if (grid > max_grid_level) grid = max_grid_level;  // every level: the kernels stride over their rays (see kMaxGridLevel)
            hipEvent_t ea = nullptr, eq = nullptr, eb = nullptr, ec = nullptr;
            if (profile) {
                if (get_event(&ea) || get_event(&eq) || get_event(&eb) || get_event(&ec)) return RTHIP_E_HIP;
                HIP_TRY(hipEventRecord(ea, c->stream));
            }
            w.redo_count = w.counts + (size_t)((RTHIP_MAX_RECURSION + 3) + level) * kCounterStride;
            w.hard_count = w.counts + (size_t)(2 * (RTHIP_MAX_RECURSION + 3) + level) * kCounterStride;
            if (p->accel == 0) {
                if (c->count_kd)
                    hipLaunchKernelGGL((k_trace_easy<true>), dim3(grid), dim3(kBlock), lds_tables, c->stream, c->sv, fp, w, level, c->n_tables, levels);
                else if (c->lean_scene)
                    hipLaunchKernelGGL((k_trace_easy<false, true>), dim3(grid), dim3(kBlock), lds_tables, c->stream, c->sv, fp, w, level, c->n_tables, levels);
                else
                    hipLaunchKernelGGL((k_trace_easy<false>), dim3(grid), dim3(kBlock), lds_tables, c->stream, c->sv, fp, w, level, c->n_tables, levels);
                if (profile) HIP_TRY(hipEventRecord(eq, c->stream));
                if (c->has_meshes) {  // the rays that need a mesh walk (count on the device: the grid is an upper bound, strided)
                    uint32_t qgrid = (fp.N0 + quad::kRaysPerBlock - 1) / quad::kRaysPerBlock;
                    if (qgrid > max_grid_quad) qgrid = max_grid_quad;
                    // short lists through k_trace_quad, long ones of a scene with small trees through k_trace_coop: the length is on
                    // the device, so both are launched and one of them returns at once
                    const uint32_t mode = walk_mode[level];
                    const uint32_t n_coop = mode == WALK_BOTH ? coop_thr32 : 0xffffffffu;  // lists from this length on are k_trace_coop's
                    const uint32_t thr32 = mode == WALK_BOTH ? coop_thr32 : 0u;
                    if (mode != WALK_COOP) {
                        if (c->count_kd)
                            hipLaunchKernelGGL((k_trace_quad<true>), dim3(qgrid), dim3(kBlock), lds_quad, c->stream, c->sv, fp, w, level, quad_cap, n_coop);
                        else if (quad_occ6)
                            hipLaunchKernelGGL((k_trace_quad<false, 6>), dim3(qgrid), dim3(kBlock), lds_quad, c->stream, c->sv, fp, w, level, quad_cap, n_coop);
                        else
                            hipLaunchKernelGGL((k_trace_quad<false>), dim3(qgrid), dim3(kBlock), lds_quad, c->stream, c->sv, fp, w, level, quad_cap, n_coop);
                    }
                    if (mode != WALK_QUAD) {
                        const uint32_t cgrid = qgrid < max_grid_coop ? qgrid : max_grid_coop;
                        if (c->count_kd)
                            hipLaunchKernelGGL((k_trace_coop<true>), dim3(cgrid), dim3(kBlock), lds_coop, c->stream, c->sv, fp, w, level, quad_cap, thr32, coop_thr64);
                        else
                            hipLaunchKernelGGL((k_trace_coop<false>), dim3(cgrid), dim3(kBlock), lds_coop, c->stream, c->sv, fp, w, level, quad_cap, thr32, coop_thr64);
                    }
                }
                if (profile) HIP_TRY(hipEventRecord(eb, c->stream));
                // (the redo list - rays lying exactly in a kd split plane, exact distance ties - is re-traced inside k_shade)
            } else {
                if (profile) HIP_TRY(hipEventRecord(eq, c->stream));
                if (c->count_kd)
                    hipLaunchKernelGGL((k_trace<false, true>), dim3(grid), dim3(kBlock), lds_bytes, c->stream, c->sv, fp, w, level, c->stack_cap);
                else
                    hipLaunchKernelGGL((k_trace<false, false>), dim3(grid), dim3(kBlock), lds_bytes, c->stream, c->sv, fp, w, level, c->stack_cap);
                if (profile) HIP_TRY(hipEventRecord(eb, c->stream));
            }
            if (p->accel == 0) {
                if (c->has_meshes) {
                    if (c->lean_scene && !c->count_kd)
                        hipLaunchKernelGGL((k_shade<true, true>), dim3(grid), dim3(kBlock), lds_tables, c->stream, c->sv, fp, w, level, c->n_tables, levels, c->stack_cap);
                    else
                        hipLaunchKernelGGL((k_shade<true>), dim3(grid), dim3(kBlock), lds_tables, c->stream, c->sv, fp, w, level, c->n_tables, levels, c->stack_cap);
                    hipLaunchKernelGGL(k_redo, dim3(grid < kRedoBlocks ? grid : kRedoBlocks), dim3(kBlock), lds_tables, c->stream, c->sv, fp, w, level, c->n_tables, levels, c->stack_cap);
                }
            } else {
                hipLaunchKernelGGL((k_shade<false>), dim3(grid), dim3(kBlock), lds_tables, c->stream, c->sv, fp, w, level, c->n_tables, levels, c->stack_cap);
            }
            if (profile) HIP_TRY(hipEventRecord(ec, c->stream));
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
        return 0;
    };
    const bool capture = use_graph && key == c->seen_key;
    if (use_graph) c->seen_key = key;
    if (capture) {  // capture once, then replay
        if (c->graphs.size() >= rthip_ctx::kMaxGraphs) {
            (void)hipGraphExecDestroy(c->graphs.front().exec);
            c->graphs.erase(c->graphs.begin());
        }
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if ((rc = stage_cams())) return rc;
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        int erc = 0;
        if (e == hipSuccess) {
            capturing = true;
            erc = enqueue();
            capturing = false;
            hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
            if (e2 != hipSuccess) e = e2;
        }
        if (e == hipSuccess && erc == 0 && interrupted) {  // rthip_interrupt arrived while capturing: the graph is incomplete, never cache it
            if (graph) (void)hipGraphDestroy(graph);
            c->seen_key.clear();
            c->last.pixels = 0;
            return fail(RTHIP_E_INTERRUPTED, "rthip_render: interrupted");
        }
        if (e == hipSuccess && erc == 0 && graph) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e == hipSuccess && erc == 0 && exec) {
            c->graphs.push_back({key, exec});
            c->graphs_captured++;
            c->last_frames = n_frames;
            HIP_TRY(hipEventRecord(c->ev0, c->stream));
            HIP_TRY(hipGraphLaunch(exec, c->stream));
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
            if ((rc = cams_enqueued())) return rc;
            c->last.pixels = pixels_done;
            c->last.launches_trace = 0;
            return 0;
        }
        (void)hipGetLastError();  // capture is not available here: plain launches from now on
        c->graphs_enabled = false;
        ev_used = 0;
        pixels_done = 0;
    }
    c->last_frames = n_frames;
    if ((rc = stage_cams())) return rc;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    rc = enqueue();
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    if ((rc = cams_enqueued())) return rc;
    c->last.pixels = pixels_done;
    if (interrupted) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        return fail(RTHIP_E_INTERRUPTED, "rthip_render: interrupted");
    }
    c->last.launches_trace = (uint32_t)(ev_used / 4);
    return 0;
}

static int finish_stats(rthip_ctx *c, const rthip_params *p) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long st[ST_COUNT + 2 * kShadeCounts];
    HIP_TRY(hipMemcpy(st, c->stats.p, sizeof st, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < 2; k++) {
        for (uint32_t i = 0; i < kShadeCountShapes; i++) c->last.shape_hits[k][i] = st[ST_COUNT + k * kShadeCounts + i];
        c->last.background_evals[k] = st[ST_COUNT + k * kShadeCounts + kShadeCountShapes];
    }
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
    px *= c->last_frames;
    c->last.pixels = px;
    c->last.rays_primary = px * (p->aa_mode ? 9 : 1);
    {
        std::vector<uint2> parts(2 * (size_t)c->bs_stride * c->bs_levels);
        if (!parts.empty()) HIP_TRY(hipMemcpy(parts.data(), c->block_stats.p, sizeof(uint2) * parts.size(), hipMemcpyDeviceToHost));
        uint64_t shadow[2] = {0, 0}, reflect[2] = {0, 0};  // [0] from k_trace_easy's workgroups, [1] from k_shade's
        const size_t half = parts.size() / 2;
        for (size_t i = 0; i < parts.size(); i++) { shadow[i >= half] += parts[i].x; reflect[i >= half] += parts[i].y; }
        c->last.rays_shadow = shadow[0] + shadow[1];
        c->last.rays_reflect = reflect[0] + reflect[1];
        for (int k = 0; k < 2; k++) { c->last.rays_shadow_by_kernel[k] = shadow[k]; c->last.rays_reflect_by_kernel[k] = reflect[k]; }
    }
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
    c->last.fetch_cycles = st[ST_FETCH_CYCLES];
    c->last.shape_tests_easy = st[ST_SHAPES_EASY];
    c->last.shape_tests_full = st[ST_SHAPES_FULL];
    c->last.shape_tests_easy_full = st[ST_SHAPES_EASY_FULL];
    {
        std::vector<uint32_t> all(kCounterSlots * kCounterStride);
        HIP_TRY(hipMemcpy(all.data(), c->counts.p, sizeof(uint32_t) * all.size(), hipMemcpyDeviceToHost));
        uint64_t hard = 0;
        for (uint32_t i = 0; i <= p->max_recursion && i <= RTHIP_MAX_RECURSION; i++) hard += all[(size_t)(2 * (RTHIP_MAX_RECURSION + 3) + i) * kCounterStride];
        for (uint32_t i = 0; i <= RTHIP_MAX_RECURSION; i++) c->last.level_rays[i] = i <= p->max_recursion ? all[(size_t)i * kCounterStride] : 0;
        c->last.rays_hard = p->accel == 0 ? hard : 0;  // last chunk only (frames of <= 2^24 rays per level are one chunk)
        uint64_t redo = 0;
        for (uint32_t i = 0; i <= p->max_recursion && i <= RTHIP_MAX_RECURSION; i++) redo += all[(size_t)((RTHIP_MAX_RECURSION + 3) + i) * kCounterStride];
        c->last.rays_redo = p->accel == 0 ? (uint32_t)std::min<uint64_t>(redo, 0xffffffffu) : 0;
    }
    memset(c->last.level_ms_trace, 0, sizeof c->last.level_ms_trace);
    memset(c->last.level_ms_shade, 0, sizeof c->last.level_ms_shade);
    c->last.level_rays[0] = (uint32_t)c->last.rays_primary;
    c->last.ms_total = ms;
    float sum = 0.f, sum_easy = 0.f;
    memset(c->last.level_ms_easy, 0, sizeof c->last.level_ms_easy);
    for (uint32_t i = 0; i < c->last.launches_trace; i++) {
        float e = 0.f, t = 0.f, u = 0.f;
        HIP_TRY(hipEventElapsedTime(&e, c->ev_pool[4 * i], c->ev_pool[4 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&t, c->ev_pool[4 * i + 1], c->ev_pool[4 * i + 2]));
        HIP_TRY(hipEventElapsedTime(&u, c->ev_pool[4 * i + 2], c->ev_pool[4 * i + 3]));
        sum += t;
        sum_easy += e;
        if (i <= RTHIP_MAX_RECURSION) {  // single-chunk frames: launch i = level i
            c->last.level_ms_easy[i] = e;
            c->last.level_ms_trace[i] = t;
            c->last.level_ms_shade[i] = u;  // includes the (normally empty) redo kernel
        }
    }
    c->last.ms_trace = sum;
    c->last.ms_easy = sum_easy;
    c->progress.store(1.f);
    return 0;
}

int rthip_collect_stats(rthip_ctx *c, const rthip_params *p, rthip_stats *out) {
    if (!c || !p) return fail(RTHIP_E_INVALID, "rthip_collect_stats: null argument");
    if (set_device(c)) return RTHIP_E_HIP;
    int rc = finish_stats(c, p);
    if (rc) return rc;
    c->last.graphs_captured = c->graphs_captured;
    c->last.graph_replays = c->graph_replays;
    if (out) *out = c->last;
    if (c->interrupt.load()) return fail(RTHIP_E_INTERRUPTED, "rthip_render_device: interrupted");
    return 0;
}

int rthip_render_device(rthip_ctx *c, const rthip_camera *cam, const rthip_params *p, void *d_pixels, int sync) {
    return rthip_render_batch_device(c, cam, 1, p, d_pixels, sync);
}

int rthip_render_batch_device(rthip_ctx *c, const rthip_camera *cams, uint32_t n_frames, const rthip_params *p, void *d_pixels, int sync) {
    if (!c || !d_pixels) return fail(RTHIP_E_INVALID, "rthip_render_device: null argument");
    int rc = render_frame(c, cams, n_frames, p, (uint32_t *)d_pixels, nullptr, nullptr);
    if (rc) return rc;
    if (!sync) return 0;
    rc = finish_stats(c, p);
    if (rc) return rc;
    if (c->interrupt.load()) return fail(RTHIP_E_INTERRUPTED, "rthip_render_device: interrupted");
    return 0;
}

int rthip_render_device_noalloc(rthip_ctx *c, const rthip_camera *cam, const rthip_params *p, void *d_pixels, int sync) {
    if (!c || !d_pixels) return fail(RTHIP_E_INVALID, "rthip_render_device_noalloc: null argument");
    int rc = render_frame(c, cam, 1, p, (uint32_t *)d_pixels, nullptr, nullptr, true);
    if (rc) return rc;
    if (sync) {
        HIP_TRY(hipStreamSynchronize(c->stream));  // (no statistics: reading them back would stage through the runtime)
        if (c->interrupt.load()) return fail(RTHIP_E_INTERRUPTED, "rthip_render_device_noalloc: interrupted");
    }
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
    if (ids) hipLaunchKernelGGL(k_fill, dim3(1024), dim3(kBlock), 0, c->stream, (uint32_t *)d_sh, 2 * npx, 0xffffffffu);  // -1 = no hit
    rthip_params pf = *p;
    pf.layout = 0;
    rc = render_frame(c, cam, 1, &pf, d_px, d_sh, d_tr);
    if (rc) return rc;
    rc = finish_stats(c, p);
    if (rc) return rc;
    // Interrupt() during the frame: its kernels left early and the frame is abandoned (Raytracer.cpp:153); out_pixels untouched
    if (c->interrupt.load()) return fail(RTHIP_E_INTERRUPTED, "rthip_render: interrupted");
    HIP_TRY(hipMemcpy(out_pixels, d_px, sizeof(uint32_t) * npx, hipMemcpyDeviceToHost));
    if (out_shape) HIP_TRY(hipMemcpy(out_shape, d_sh, sizeof(int32_t) * npx, hipMemcpyDeviceToHost));
    if (out_tri) HIP_TRY(hipMemcpy(out_tri, d_tr, sizeof(int32_t) * npx, hipMemcpyDeviceToHost));
    return 0;
}

int rthip_render_batch(rthip_ctx *c, const rthip_camera *cams, uint32_t n_frames, const rthip_params *p, uint32_t *out_pixels) {
    if (!c || !out_pixels) return fail(RTHIP_E_INVALID, "rthip_render_batch: null argument");
    int rc = check_params(cams, p);
    if (rc) return rc;
    if (n_frames == 0 || n_frames > RTHIP_MAX_BATCH) return fail(RTHIP_E_INVALID, "rthip_render_batch: %u frames per call, limit %d", n_frames, RTHIP_MAX_BATCH);
    if (p->part_count != 1) return fail(RTHIP_E_INVALID, "rthip_render_batch: whole frames only (part_count = 1)");
    if (set_device(c)) return RTHIP_E_HIP;
    const size_t npx = (size_t)p->width * p->height * n_frames;
    if (c->pixels_tmp.ensure(sizeof(uint32_t) * npx)) return RTHIP_E_HIP;
    rthip_params pf = *p;
    pf.layout = 0;
    rc = render_frame(c, cams, n_frames, &pf, (uint32_t *)c->pixels_tmp.p, nullptr, nullptr);
    if (rc) return rc;
    rc = finish_stats(c, p);
    if (rc) return rc;
    if (c->interrupt.load()) return fail(RTHIP_E_INTERRUPTED, "rthip_render_batch: interrupted");
    HIP_TRY(hipMemcpy(out_pixels, c->pixels_tmp.p, sizeof(uint32_t) * npx, hipMemcpyDeviceToHost));
    return 0;
}

int rthip_host_alloc(rthip_ctx *c, size_t bytes, void **out) {
    if (!c || !out || !bytes) return fail(RTHIP_E_INVALID, "rthip_host_alloc: null argument");
    *out = nullptr;
    if (set_device(c)) return RTHIP_E_HIP;
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    memset(*out, 0, bytes);
    return 0;
}
int rthip_host_free(rthip_ctx *c, void *ptr) {
    if (!c) return fail(RTHIP_E_INVALID, "rthip_host_free: null context");
    if (!ptr) return 0;
    if (set_device(c)) return RTHIP_E_HIP;
    HIP_TRY(hipHostFree(ptr));
    return 0;
}

uint64_t rthip_part_pixels(const rthip_params *p) {
    if (!p || p->width == 0 || p->height == 0 || p->part_count == 0) return 0;
    uint32_t tw = p->tile_w ? p->tile_w : 32, th = p->tile_h ? p->tile_h : 8;
    uint64_t tiles = (uint64_t)((p->width + tw - 1) / tw) * ((p->height + th - 1) / th);
    return ((tiles + p->part_count - 1) / p->part_count) * tw * th;
}

int rthip_assemble(rthip_ctx *c, const rthip_params *p, const void *d_gathered, void *d_frame, int sync) {
    return rthip_assemble_batch(c, p, 1, d_gathered, d_frame, sync);
}

int rthip_assemble_batch(rthip_ctx *c, const rthip_params *p, uint32_t n_frames, const void *d_gathered, void *d_frame, int sync) {
    if (!c || !p || !d_gathered || !d_frame) return fail(RTHIP_E_INVALID, "rthip_assemble: null argument");
    if (n_frames == 0 || n_frames > RTHIP_MAX_BATCH) return fail(RTHIP_E_INVALID, "rthip_assemble: %u frames per call, limit %d", n_frames, RTHIP_MAX_BATCH);
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
    fp.d_per_tile = rtc::make_fastdiv(fp.tile_w * fp.tile_h);
    fp.d_my_tiles = rtc::make_fastdiv(1);  // (k_assemble: one frame per part buffer, the division is skipped)
    fp.d_tiles_x = rtc::make_fastdiv(fp.tiles_x);
    fp.d_sbx = rtc::make_fastdiv(fp.tile_w >> 3);
    uint32_t part_slots = (uint32_t)rthip_part_pixels(p);
    hipLaunchKernelGGL(k_assemble, dim3((part_slots + kBlock - 1) / kBlock, p->part_count, n_frames), dim3(kBlock), 0, c->stream, fp,
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
    if (need_scene && c->mesh_without_bvh >= 0) c->sv.accel = 1; /* a kd-only mesh: the BVH walk would miss it (render_frame refuses accel = 0 there) */ \
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
    if (noise < 0 || (uint32_t)noise >= c->uploaded.n_noises) return fail(RTHIP_E_INVALID, "rthip_probe_noise: noise %d not in [0,%u)", noise, c->uploaded.n_noises);
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
    if (!p || !color || !absorptance) return fail(RTHIP_E_INVALID, "rthip_probe_material: null argument");
    if (material < 0 || (uint32_t)material >= c->uploaded.n_materials) return fail(RTHIP_E_INVALID, "rthip_probe_material: material %d not in [0,%u)", material, c->uploaded.n_materials);
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
int rthip_probe_bvh(rthip_ctx *c, uint32_t mesh, rthip_mesh *out_mesh, rthip_bvhnode *nodes, uint64_t cap_nodes, uint32_t *tri_index, uint64_t cap_tris) {
    if (!c) return fail(RTHIP_E_INVALID, "null context");
    if (!c->has_scene) return fail(RTHIP_E_STATE, "no scene uploaded");
    if (mesh >= c->mesh_table.size()) return fail(RTHIP_E_INVALID, "rthip_probe_bvh: mesh %u not in [0,%zu)", mesh, c->mesh_table.size());
    if (set_device(c)) return RTHIP_E_HIP;
    const rthip_mesh &m = c->mesh_table[mesh];
    if (out_mesh) *out_mesh = m;
    if (nodes) {
        if (cap_nodes < m.bvh_node_count) return fail(RTHIP_E_INVALID, "rthip_probe_bvh: %u nodes, room for %llu", m.bvh_node_count, (unsigned long long)cap_nodes);
        if (m.bvh_node_count) HIP_TRY(hipMemcpy(nodes, (const rthip_bvhnode *)c->bvh_nodes.p + m.bvh_node_base, sizeof(rthip_bvhnode) * m.bvh_node_count, hipMemcpyDeviceToHost));
    }
    if (tri_index) {
        if (cap_tris < m.tri_count) return fail(RTHIP_E_INVALID, "rthip_probe_bvh: %u triangles, room for %llu", m.tri_count, (unsigned long long)cap_tris);
        if (m.tri_count && m.bvh_node_count) HIP_TRY(hipMemcpy(tri_index, (const uint32_t *)c->bvh_index.p + m.bvh_tri_base, sizeof(uint32_t) * m.tri_count, hipMemcpyDeviceToHost));
    }
    return 0;
}
int rthip_probe_kd(rthip_ctx *c, uint32_t mesh, rthip_mesh *out_mesh, rthip_kdnode *nodes, double *bounds, uint64_t cap_nodes, uint32_t *refs, uint64_t cap_refs,
                   double *planes, uint64_t cap_planes) {
    if (!c) return fail(RTHIP_E_INVALID, "null context");
    if (!c->has_scene) return fail(RTHIP_E_STATE, "no scene uploaded");
    if (mesh >= c->mesh_table.size()) return fail(RTHIP_E_INVALID, "rthip_probe_kd: mesh %u not in [0,%zu)", mesh, c->mesh_table.size());
    if (set_device(c)) return RTHIP_E_HIP;
    const rthip_mesh &m = c->mesh_table[mesh];
    if (out_mesh) *out_mesh = m;
    if ((nodes || bounds) && cap_nodes < m.node_count) return fail(RTHIP_E_INVALID, "rthip_probe_kd: %u nodes, room for %llu", m.node_count, (unsigned long long)cap_nodes);
    if (nodes && m.node_count) HIP_TRY(hipMemcpy(nodes, (const rthip_kdnode *)c->kd_nodes.p + m.node_base, sizeof(rthip_kdnode) * m.node_count, hipMemcpyDeviceToHost));
    if (bounds && m.node_count) HIP_TRY(hipMemcpy(bounds, (const double *)c->kd_bounds.p + (size_t)6 * m.node_base, sizeof(double) * 6 * m.node_count, hipMemcpyDeviceToHost));
    if (refs) {
        if (cap_refs < m.ref_count) return fail(RTHIP_E_INVALID, "rthip_probe_kd: %u references, room for %llu", m.ref_count, (unsigned long long)cap_refs);
        if (m.ref_count) HIP_TRY(hipMemcpy(refs, (const uint32_t *)c->kd_refs.p + m.ref_base, sizeof(uint32_t) * m.ref_count, hipMemcpyDeviceToHost));
    }
    if (planes) {
        const uint64_t np = (uint64_t)m.plane_count[0] + m.plane_count[1] + m.plane_count[2];
        if (cap_planes < np) return fail(RTHIP_E_INVALID, "rthip_probe_kd: %llu planes, room for %llu", (unsigned long long)np, (unsigned long long)cap_planes);
        uint64_t at = 0;
        for (int a = 0; a < 3; a++) {
            if (m.plane_count[a]) HIP_TRY(hipMemcpy(planes + at, (const double *)c->kd_planes.p + m.plane_base[a], sizeof(double) * m.plane_count[a], hipMemcpyDeviceToHost));
            at += m.plane_count[a];
        }
    }
    return 0;
}
int rthip_get_upload_info(rthip_ctx *c, rthip_upload_info *out) {
    if (!c || !out) return fail(RTHIP_E_INVALID, "null argument");
    *out = c->upload_info;
    out->walk_coop = c->walk_coop ? 1u : 0u;
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
