// rthip_bvh.hip — the 4-wide BVH of a mesh built ON THE DEVICE (acceleration-structure build at
// 2-10 M triangles; replaces the single-threaded recursive by-value build of Mesh::Compile, Raytracer/Scene.cpp:386-414 +
// :18-95, for the structure the frame kernels walk).  The BVH takes no part in the arithmetic of a result: Mesh::Intersect
// returns the strictly closest accepted triangle of the whole mesh (Scene.cpp:469-493), so any conservative culling
// structure gives the same image; only the kd-tree (accel = 1, degenerate rays) has to be the reference's, and that one
// stays on the host (its split is a sequential mean, Scene.cpp:36-47).
//
// Input: the mesh's triangle records already in HBM (rthip_tri, mesh order).  Pipeline, all on one stream:
//   k_prims     : per triangle the fp32 box (rounded OUTWARD like the host builder's Down / Up) and the centroid; bounds of
//                 the centroids and max |coordinate| by wave reduction + atomics on order-preserving integer keys
//   k_morton    : 63-bit Morton code of the centroid (21 bits per axis)
//   rocprim     : radix sort of (code, triangle index)
//   k_radix_tree: binary radix tree over the sorted codes (Karras 2012; ties broken by position), one thread per internal node
//   k_refit     : bottom-up boxes, the second thread to arrive at a node continues (atomic arrival counter)
//   k_collapse  : breadth-first, one launch per level: a wide node takes a binary node, opens the child with the largest box
//                 until it has four children; a subtree of <= 4 triangles becomes a leaf (one triangle per lane of the quad
//                 that walks it); node indices grow with depth (children follow their parent, as the host builder's)
//   k_gather    : triangle records and indices in leaf order
// Output layout = include/rthip.h (rthip_bvhnode, 128 B; leaf words; RTHIP_BVH_ABSENT).
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>
#include <vector>

#include "rthip.h"
#include "rthip_bvh.hpp"

namespace {
const uint32_t kB = 256;

// fp32 boxes rounded outward: the value below / above x, then one more ulp (same margin as BvhBuilder::Down / Up)
__device__ __forceinline__ float next_down(float f) {
    if (f == 0.0f) return -FLT_MIN * FLT_EPSILON;  // -denorm_min
    uint32_t b = __float_as_uint(f);
    b = (f > 0.0f) ? b - 1u : b + 1u;
    return __uint_as_float(b);
}
__device__ __forceinline__ float next_up(float f) {
    if (f == 0.0f) return FLT_MIN * FLT_EPSILON;
    uint32_t b = __float_as_uint(f);
    b = (f > 0.0f) ? b + 1u : b - 1u;
    return __uint_as_float(b);
}
__device__ __forceinline__ float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = next_down(f);
    return next_down(f);
}
__device__ __forceinline__ float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = next_up(f);
    return next_up(f);
}
// order-preserving float <-> uint (for atomicMin / atomicMax)
__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float key2f(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

struct Box {
    float lo[3], hi[3];
};
__device__ __forceinline__ Box merge(const Box &a, const Box &b) {
    Box r;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r.lo[k] = fminf(a.lo[k], b.lo[k]);
        r.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    }
    return r;
}
__device__ __forceinline__ float area(const Box &b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dx * dz + dy * dz;
}

// reduction cell: [0..2] min centroid keys, [3..5] max centroid keys, [6] max |coordinate| key
__global__ __launch_bounds__(kB) void k_prims(const rthip_tri *tris, uint32_t n, Box *prim_box, float *centroid, uint32_t *red) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    float c[3] = {0, 0, 0}, am = 0.0f;
    const bool live = i < n;
    if (live) {
        const rthip_tri t = tris[i];
        Box b;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // the vertices as the triangle test sees them: v0, v0 + e0, v0 + e1 (fp64; the outward fp32 rounding plus one ulp
            // swallows the rounding of the two sums, 2^-53 relative)
            const double v0 = t.v0[k], v1 = t.v0[k] + t.e0[k], v2 = t.v0[k] + t.e1[k];
            const double l = fmin(v0, fmin(v1, v2)), h = fmax(v0, fmax(v1, v2));
            b.lo[k] = round_down(l);
            b.hi[k] = round_up(h);
            c[k] = (float)(0.5 * (l + h));
            am = fmaxf(am, fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])));
        }
        prim_box[i] = b;
        centroid[3 * (size_t)i] = c[0];
        centroid[3 * (size_t)i + 1] = c[1];
        centroid[3 * (size_t)i + 2] = c[2];
    }
    uint32_t kmin[3], kmax[3], kam = f2key(am);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        kmin[k] = live ? f2key(c[k]) : 0xffffffffu;
        kmax[k] = live ? f2key(c[k]) : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            kmin[k] = min(kmin[k], (uint32_t)__shfl_xor((int)kmin[k], off, 64));
            kmax[k] = max(kmax[k], (uint32_t)__shfl_xor((int)kmax[k], off, 64));
        }
        kam = max(kam, (uint32_t)__shfl_xor((int)kam, off, 64));
    }
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            atomicMin(&red[k], kmin[k]);
            atomicMax(&red[3 + k], kmax[k]);
        }
        atomicMax(&red[6], kam);
    }
}

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {  // 21 bits -> every third bit of 63
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ __launch_bounds__(kB) void k_morton(const float *centroid, uint32_t n, const uint32_t *red, unsigned long long *code, uint32_t *index) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    unsigned long long m = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = key2f(red[k]), hi = key2f(red[3 + k]);
        const float ext = hi - lo;
        float u = ext > 0.0f ? (centroid[3 * (size_t)i + k] - lo) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        uint32_t q = (uint32_t)(u * 2097151.0f);
        q = q > 2097151u ? 2097151u : q;
        m |= spread21(q) << k;
    }
    code[i] = m;
    index[i] = i;
}

// Binary radix tree over n sorted codes: internal nodes 0 .. n-2, leaves 0 .. n-1.  Child / parent references: bit 31 set = leaf.
struct Bin {
    uint32_t left, right;  // bit 31: leaf (sorted position), else internal node
    uint32_t first, last;  // sorted positions covered
};
__device__ __forceinline__ int delta(const unsigned long long *code, uint32_t n, int i, int j) {
    if (j < 0 || j >= (int)n) return -1;
    const unsigned long long a = code[i], b = code[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));  // equal codes: the positions decide
}
__global__ __launch_bounds__(kB) void k_radix_tree(const unsigned long long *code, uint32_t n, Bin *bin, uint32_t *parent_internal, uint32_t *parent_leaf) {
    const int i = (int)(blockIdx.x * kB + threadIdx.x);
    if (i >= (int)n - 1) return;
    const int d = delta(code, n, i, i + 1) - delta(code, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(code, n, i, i - d);
    int lmax = 2;
    while (delta(code, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(code, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(code, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(code, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    Bin b;
    b.first = (uint32_t)lo;
    b.last = (uint32_t)hi;
    if (lo == gamma) {
        b.left = 0x80000000u | (uint32_t)gamma;
        parent_leaf[gamma] = (uint32_t)i;
    } else {
        b.left = (uint32_t)gamma;
        parent_internal[gamma] = (uint32_t)i;
    }
    if (hi == gamma + 1) {
        b.right = 0x80000000u | (uint32_t)(gamma + 1);
        parent_leaf[gamma + 1] = (uint32_t)i;
    } else {
        b.right = (uint32_t)(gamma + 1);
        parent_internal[gamma + 1] = (uint32_t)i;
    }
    bin[i] = b;
    if (i == 0) parent_internal[0] = 0xffffffffu;
}

__device__ __forceinline__ Box box_of(uint32_t ref, const Box *node_box, const Box *prim_box, const uint32_t *index) {
    return (ref & 0x80000000u) ? prim_box[index[ref & 0x7fffffffu]] : node_box[ref];
}
__global__ __launch_bounds__(kB) void k_refit(const Bin *bin, uint32_t n, const uint32_t *parent_internal, const uint32_t *parent_leaf,
                                              const Box *prim_box, const uint32_t *index, Box *node_box, uint32_t *arrived) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    uint32_t node = parent_leaf[i];
    while (node != 0xffffffffu) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;  // the first to arrive leaves; the second has both children ready
        __threadfence();
        const Bin b = bin[node];
        const Box l = box_of(b.left, node_box, prim_box, index), r = box_of(b.right, node_box, prim_box, index);
        const Box m = merge(l, r);
        // (plain stores + the fences above: the sibling that continues from the parent reads them after its own atomic)
        node_box[node] = m;
        node = parent_internal[node];
    }
}

// number of triangles under a reference
__device__ __forceinline__ uint32_t span_of(uint32_t ref, const Bin *bin) { return (ref & 0x80000000u) ? 1u : bin[ref].last - bin[ref].first + 1u; }

struct Item {
    uint32_t bin_node;   // binary node this wide node is made from (covers > 4 triangles)
    uint32_t wide_node;  // its index in the output
};
__global__ __launch_bounds__(kB) void k_collapse(const Item *items, uint32_t n_items, const Bin *bin, const Box *node_box, const Box *prim_box,
                                                 const uint32_t *index, rthip_bvhnode *out, uint32_t out_cap, uint32_t *counters /* [0] nodes, [1] next items, [2] overflow */,
                                                 Item *next) {
    const uint32_t t = blockIdx.x * kB + threadIdx.x;
    if (t >= n_items) return;
    const Item it = items[t];
    uint32_t ref[4];
    Box box[4];
    const Bin root = bin[it.bin_node];
    ref[0] = root.left;
    ref[1] = root.right;
    box[0] = box_of(ref[0], node_box, prim_box, index);
    box[1] = box_of(ref[1], node_box, prim_box, index);
    int cnt = 2;
    while (cnt < 4) {  // open the child with the largest box that is still a subtree of more than four triangles
        int pick = -1;
        float best = -1.0f;
        for (int k = 0; k < cnt; k++) {
            if (span_of(ref[k], bin) <= RTHIP_BVH_MAX_LEAF) continue;
            const float a = area(box[k]);
            if (a > best) { best = a; pick = k; }
        }
        if (pick < 0) break;
        const Bin o = bin[ref[pick]];
        ref[pick] = o.left;
        box[pick] = box_of(o.left, node_box, prim_box, index);
        ref[cnt] = o.right;
        box[cnt] = box_of(o.right, node_box, prim_box, index);
        cnt++;
    }
    rthip_bvhnode nd;
    for (int k = 0; k < 4; k++) {
        if (k >= cnt) {
            for (int a = 0; a < 3; a++) { nd.c[k].lo[a] = INFINITY; nd.c[k].hi[a] = -INFINITY; }
            nd.c[k].child = RTHIP_BVH_ABSENT;
            nd.c[k].pad = 0;
            continue;
        }
        for (int a = 0; a < 3; a++) { nd.c[k].lo[a] = box[k].lo[a]; nd.c[k].hi[a] = box[k].hi[a]; }
        nd.c[k].pad = 0;
        const uint32_t span = span_of(ref[k], bin);
        if (span <= RTHIP_BVH_MAX_LEAF) {
            const uint32_t first = (ref[k] & 0x80000000u) ? (ref[k] & 0x7fffffffu) : bin[ref[k]].first;
            nd.c[k].child = RTHIP_BVH_LEAF | ((span - 1u) << RTHIP_BVH_COUNT_SHIFT) | first;
        } else {
            const uint32_t w = atomicAdd(&counters[0], 1u);
            if (w >= out_cap) {  // cannot happen with out_cap = number of binary nodes of more than four triangles; never write past it
                atomicExch(&counters[2], 1u);
                nd.c[k].child = RTHIP_BVH_ABSENT;
                continue;
            }
            nd.c[k].child = w;
            const uint32_t q = atomicAdd(&counters[1], 1u);
            next[q] = Item{ref[k], w};
        }
    }
    out[it.wide_node] = nd;
}

// a mesh of 1 .. 4 triangles: a root whose only child is the single leaf (as the host builder)
__global__ void k_tiny(const Box *prim_box, uint32_t n, rthip_bvhnode *out) {
    if (threadIdx.x || blockIdx.x) return;
    Box b = prim_box[0];
    for (uint32_t i = 1; i < n; i++) b = merge(b, prim_box[i]);
    rthip_bvhnode nd;
    for (int k = 0; k < 4; k++) {
        for (int a = 0; a < 3; a++) {
            nd.c[k].lo[a] = k == 0 ? b.lo[a] : INFINITY;
            nd.c[k].hi[a] = k == 0 ? b.hi[a] : -INFINITY;
        }
        nd.c[k].child = k == 0 ? (RTHIP_BVH_LEAF | ((n - 1u) << RTHIP_BVH_COUNT_SHIFT)) : RTHIP_BVH_ABSENT;
        nd.c[k].pad = 0;
    }
    out[0] = nd;
}

__global__ __launch_bounds__(kB) void k_count_big(const Bin *bin, uint32_t n_internal, uint32_t *count) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    const bool big = i < n_internal && bin[i].last - bin[i].first + 1u > RTHIP_BVH_MAX_LEAF;
    const unsigned long long m = __ballot(big);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(count, (uint32_t)__popcll(m));
}

__global__ __launch_bounds__(kB) void k_gather(const rthip_tri *tris, const uint32_t *index, uint32_t n, rthip_tri *bvh_tris, uint32_t *bvh_index) {
    // 8 lanes per 128-byte record (uint4 each): whole lines in and out
    const size_t g = (size_t)blockIdx.x * kB + threadIdx.x;
    const size_t rec = g >> 3, part = g & 7u;
    if (rec >= n) return;
    const uint32_t src = index[rec];
    reinterpret_cast<uint4 *>(bvh_tris + rec)[part] = reinterpret_cast<const uint4 *>(tris + src)[part];
    if (part == 0) bvh_index[rec] = src;
}

struct Tmp {
    void *p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 256); }
    template <class T> T *as() { return (T *)p; }
};
}  // namespace

#define BHIP(expr)                                                             \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            snprintf(err, err_len, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            return RTHIP_E_HIP;                                                \
        }                                                                      \
    } while (0)

int rthip_bvh_build_device(hipStream_t stream, const rthip_tri *d_tris, uint32_t n, rthip_tri *d_bvh_tris, uint32_t *d_bvh_index,
                           rthip_bvhnode **d_nodes_out, rthip_device_bvh_info *info, char *err, size_t err_len) {
    memset(info, 0, sizeof *info);
    *d_nodes_out = nullptr;
    if (n == 0) return 0;
    if (n > RTHIP_BVH_FIRST_MASK) {
        snprintf(err, err_len, "mesh of %u triangles is too large for the BVH leaf encoding", n);
        return RTHIP_E_UNSUPPORTED;
    }
    const uint32_t grid_n = (n + kB - 1) / kB;
    Tmp prim_box, centroid, red, code_in, code_out, idx_in, idx_out, sort_tmp, bin, par_int, par_leaf, node_box, arrived, counters, items_a, items_b;
    BHIP(prim_box.alloc(sizeof(Box) * (size_t)n));
    BHIP(centroid.alloc(sizeof(float) * 3 * (size_t)n));
    BHIP(red.alloc(256));
    {
        uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
        BHIP(hipMemcpyAsync(red.p, init, sizeof init, hipMemcpyHostToDevice, stream));
        BHIP(hipStreamSynchronize(stream));  // `init` is a local
    }
    hipLaunchKernelGGL(k_prims, dim3(grid_n), dim3(kB), 0, stream, d_tris, n, prim_box.as<Box>(), centroid.as<float>(), red.as<uint32_t>());
    uint32_t red_h[8];
    BHIP(hipMemcpyAsync(red_h, red.p, sizeof red_h, hipMemcpyDeviceToHost, stream));
    BHIP(hipStreamSynchronize(stream));
    info->abs_max = key2f(red_h[6]);
    info->abs_max = std::nextafterf(info->abs_max, INFINITY);

    rthip_bvhnode *nodes = nullptr;
    if (n <= RTHIP_BVH_MAX_LEAF) {  // one leaf under a root
        BHIP(hipMalloc((void **)&nodes, sizeof(rthip_bvhnode)));
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, stream, prim_box.as<Box>(), n, nodes);
        BHIP(idx_out.alloc(sizeof(uint32_t) * n));
        std::vector<uint32_t> iota(n);
        for (uint32_t i = 0; i < n; i++) iota[i] = i;
        BHIP(hipMemcpyAsync(idx_out.p, iota.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, stream));
        BHIP(hipStreamSynchronize(stream));
        info->n_nodes = 1;
        info->depth = 0;
    } else {
        BHIP(code_in.alloc(sizeof(unsigned long long) * (size_t)n));
        BHIP(code_out.alloc(sizeof(unsigned long long) * (size_t)n));
        BHIP(idx_in.alloc(sizeof(uint32_t) * (size_t)n));
        BHIP(idx_out.alloc(sizeof(uint32_t) * (size_t)n));
        hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(kB), 0, stream, centroid.as<float>(), n, red.as<uint32_t>(), code_in.as<unsigned long long>(), idx_in.as<uint32_t>());
        size_t sort_bytes = 0;
        BHIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, code_in.as<unsigned long long>(), code_out.as<unsigned long long>(), idx_in.as<uint32_t>(),
                                       idx_out.as<uint32_t>(), (size_t)n, 0u, 63u, stream));
        BHIP(sort_tmp.alloc(sort_bytes));
        BHIP(rocprim::radix_sort_pairs(sort_tmp.p, sort_bytes, code_in.as<unsigned long long>(), code_out.as<unsigned long long>(), idx_in.as<uint32_t>(),
                                       idx_out.as<uint32_t>(), (size_t)n, 0u, 63u, stream));
        const uint32_t n_int = n - 1;
        BHIP(bin.alloc(sizeof(Bin) * (size_t)n_int));
        BHIP(par_int.alloc(sizeof(uint32_t) * (size_t)n_int));
        BHIP(par_leaf.alloc(sizeof(uint32_t) * (size_t)n));
        BHIP(node_box.alloc(sizeof(Box) * (size_t)n_int));
        BHIP(arrived.alloc(sizeof(uint32_t) * (size_t)n_int));
        BHIP(counters.alloc(256));
        BHIP(hipMemsetAsync(arrived.p, 0, sizeof(uint32_t) * (size_t)n_int, stream));
        BHIP(hipMemsetAsync(counters.p, 0, 256, stream));
        BHIP(hipStreamSynchronize(stream));  // (the memsets are not relied upon to keep stream order, see rthip.hip: k_clear)
        hipLaunchKernelGGL(k_radix_tree, dim3((n_int + kB - 1) / kB), dim3(kB), 0, stream, code_out.as<unsigned long long>(), n, bin.as<Bin>(), par_int.as<uint32_t>(),
                           par_leaf.as<uint32_t>());
        hipLaunchKernelGGL(k_refit, dim3(grid_n), dim3(kB), 0, stream, bin.as<Bin>(), n, par_int.as<uint32_t>(), par_leaf.as<uint32_t>(), prim_box.as<Box>(),
                           idx_out.as<uint32_t>(), node_box.as<Box>(), arrived.as<uint32_t>());
        // upper bound of the wide nodes: every wide node is made from a different binary node of more than four triangles
        uint32_t *cnt = counters.as<uint32_t>();
        hipLaunchKernelGGL(k_count_big, dim3((n_int + kB - 1) / kB), dim3(kB), 0, stream, bin.as<Bin>(), n_int, cnt + 3);
        uint32_t cap = 0;
        BHIP(hipMemcpyAsync(&cap, cnt + 3, sizeof cap, hipMemcpyDeviceToHost, stream));
        BHIP(hipStreamSynchronize(stream));
        if (cap == 0) cap = 1;
        BHIP(hipMalloc((void **)&nodes, sizeof(rthip_bvhnode) * (size_t)cap));
        BHIP(items_a.alloc(sizeof(Item) * (size_t)cap));
        BHIP(items_b.alloc(sizeof(Item) * (size_t)cap));
        // breadth-first collapse: level 0 = the root (binary node 0 -> wide node 0)
        const Item root = {0u, 0u};
        const uint32_t one = 1u;
        hipError_t e = hipMemcpyAsync(items_a.p, &root, sizeof root, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(cnt, &one, sizeof one, hipMemcpyHostToDevice, stream);  // wide nodes allocated so far
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {
            (void)hipFree(nodes);
            snprintf(err, err_len, "BVH build: %s", hipGetErrorString(e));
            return RTHIP_E_HIP;
        }
        uint32_t n_items = 1, depth = 0;
        Item *cur = items_a.as<Item>(), *nxt = items_b.as<Item>();
        for (;;) {
            hipLaunchKernelGGL(k_collapse, dim3((n_items + kB - 1) / kB), dim3(kB), 0, stream, cur, n_items, bin.as<Bin>(), node_box.as<Box>(), prim_box.as<Box>(),
                               idx_out.as<uint32_t>(), nodes, cap, cnt, nxt);
            uint32_t c3[3];
            e = hipMemcpyAsync(c3, cnt, sizeof c3, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess || c3[2]) {
                (void)hipFree(nodes);
                snprintf(err, err_len, "BVH build: %s", e != hipSuccess ? hipGetErrorString(e) : "node buffer overflow");
                return RTHIP_E_HIP;
            }
            info->n_nodes = c3[0];
            n_items = c3[1];
            if (n_items == 0) break;
            depth++;
            if (depth > RTHIP_BVH_MAX_DEPTH) {
                (void)hipFree(nodes);
                snprintf(err, err_len, "device-built BVH is deeper than %d levels", RTHIP_BVH_MAX_DEPTH);
                return RTHIP_E_UNSUPPORTED;
            }
            const uint32_t zero = 0u;
            e = hipMemcpyAsync(cnt + 1, &zero, sizeof zero, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                (void)hipFree(nodes);
                snprintf(err, err_len, "BVH build: %s", hipGetErrorString(e));
                return RTHIP_E_HIP;
            }
            Item *sw = cur; cur = nxt; nxt = sw;
        }
        info->depth = depth;
    }
    {
        const size_t threads = (size_t)n * 8;
        hipLaunchKernelGGL(k_gather, dim3((uint32_t)((threads + kB - 1) / kB)), dim3(kB), 0, stream, d_tris, idx_out.as<uint32_t>(), n, d_bvh_tris, d_bvh_index);
    }
    hipError_t e = hipMemcpyAsync(&info->root, nodes, sizeof(rthip_bvhnode), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipFree(nodes);
        snprintf(err, err_len, "BVH build: %s", hipGetErrorString(e));
        return RTHIP_E_HIP;
    }
    *d_nodes_out = nodes;
    return 0;
}
