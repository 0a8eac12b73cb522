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
//   k_ploc_*    : binary tree by parallel locally-ordered clustering (Meister & Bittner 2018): every cluster looks 16 places
//                 to either side in Morton order for the neighbour with the smallest union box, mutual nearest neighbours
//                 merge, the array is compacted (rocprim scan), repeat until one cluster is left (a few dozen passes)
//   k_collapse  : breadth-first, one launch per level: a wide node takes a binary node, opens the child with the largest box
//                 until it has four children; a subtree of <= 4 triangles becomes a leaf (one triangle per lane of the quad
//                 that walks it); leaf order is assigned on the way down; node indices grow with depth (children follow
//                 their parent, as the host builder's)
//   k_gather    : triangle records in leaf order
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
    // ONE scale for the three axes (the largest extent of the centroids): a flat mesh - a height field whose heights jitter
    // from vertex to vertex - must not have that jitter stretched over the full code range, where it would decide the top
    // splits and leave boxes that overlap everywhere in the other two axes (3x the frame time on config 4)
    float ext = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) ext = fmaxf(ext, key2f(red[3 + k]) - key2f(red[k]));
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = key2f(red[k]);
        float u = ext > 0.0f ? (centroid[3 * (size_t)i + k] - lo) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        uint32_t q = (uint32_t)(u * 2097151.0f);
        q = q > 2097151u ? 2097151u : q;
        m |= spread21(q) << k;
    }
    code[i] = m;
    index[i] = i;
}

// Binary tree by PLOC (parallel locally-ordered clustering, Meister & Bittner 2018): the clusters - at first the triangles in
// Morton order - each look `radius` (16) positions to either side for the neighbour whose union box is smallest; mutual nearest
// neighbours merge into a new node, the array is compacted (order kept) and the search repeats until one cluster is left.
// Bottom-up agglomeration by box area gives trees of about the quality of a top-down SAH build (a plain Morton radix tree
// walked 34 % slower on config 4), in a few dozen cheap passes.  References: bit 31 set = triangle (sorted position).
struct PNode {
    uint32_t left, right;  // children
    uint32_t size;         // triangles below
    uint32_t pad;
};
const int kRadiusMax = 64;
__global__ __launch_bounds__(kB) void k_ploc_init(const Box *prim_box, const uint32_t *index, uint32_t n, Box *cbox, uint32_t *cref, uint32_t *csize) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    cbox[i] = prim_box[index[i]];
    cref[i] = 0x80000000u | i;
    csize[i] = 1u;
}
// max_size > 0: only unions of at most max_size triangles are considered (the leaf-forming passes)
__global__ __launch_bounds__(kB) void k_ploc_nn(const Box *cbox, const uint32_t *csize, uint32_t n, uint32_t *nn, int kRadius, uint32_t max_size) {
    __shared__ Box sh[kB + 2 * kRadiusMax];
    __shared__ uint32_t shs[kB + 2 * kRadiusMax];
    const int base = (int)(blockIdx.x * kB) - kRadius;
    for (int k = threadIdx.x; k < (int)kB + 2 * kRadius; k += kB) {
        const int j = base + k;
        if (j >= 0 && j < (int)n) {
            sh[k] = cbox[j];
            shs[k] = csize[j];
        }
    }
    __syncthreads();
    const int i = (int)(blockIdx.x * kB + threadIdx.x);
    if (i >= (int)n) return;
    const Box me = sh[threadIdx.x + kRadius];
    const uint32_t room = max_size ? max_size - min(max_size, shs[threadIdx.x + kRadius]) : 0xffffffffu;  // triangles a partner may bring
    // Ties go to the partner i ^ 1, then to the lower position: equal boxes (duplicated triangles) then pair up two by two
    // instead of forming one chain of one-sided choices that merges a single pair per pass.
    float best = FLT_MAX;
    int who = -1;
    {
        const int j = i ^ 1;
        if (j < (int)n && shs[threadIdx.x + kRadius + (j - i)] <= room) {
            best = area(merge(me, sh[threadIdx.x + kRadius + (j - i)]));
            who = j;
        }
    }
    for (int d = -kRadius; d <= kRadius; d++) {
        const int j = i + d;
        if (d == 0 || j < 0 || j >= (int)n || shs[threadIdx.x + kRadius + d] > room) continue;
        const float a = area(merge(me, sh[threadIdx.x + kRadius + d]));
        if (a < best) { best = a; who = j; }
    }
    nn[i] = (uint32_t)who;  // 0xffffffff: nobody
}
__global__ __launch_bounds__(kB) void k_ploc_merge(const uint32_t *nn, const Box *cbox, const uint32_t *cref, const uint32_t *csize, uint32_t n, PNode *nodes,
                                                   Box *node_box, uint32_t *node_count, uint32_t *valid, Box *obox, uint32_t *oref, uint32_t *osize) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = nn[i];
    uint32_t keep = 1u;
    Box b = cbox[i];
    uint32_t ref = cref[i], size = csize[i];
    if (j < n && nn[j] == i) {  // mutual nearest neighbours: the lower position carries the new node, the other one goes
        if (i < j) {
            const uint32_t id = atomicAdd(node_count, 1u);
            b = merge(b, cbox[j]);
            size += csize[j];
            nodes[id] = PNode{ref, cref[j], size, 0u};
            node_box[id] = b;
            ref = id;
        } else {
            keep = 0u;
        }
    }
    valid[i] = keep;
    obox[i] = b;
    oref[i] = ref;
    osize[i] = size;
}
__global__ __launch_bounds__(kB) void k_ploc_compact(const uint32_t *valid, const uint32_t *pos, const Box *ibox, const uint32_t *iref, const uint32_t *isize, uint32_t n,
                                                     Box *obox, uint32_t *oref, uint32_t *osize, uint32_t *n_out) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    if (valid[i]) {
        const uint32_t p = pos[i];
        obox[p] = ibox[i];
        oref[p] = iref[i];
        osize[p] = isize[i];
    }
    if (i == n - 1) *n_out = pos[i] + valid[i];
}

__device__ __forceinline__ Box box_of(uint32_t ref, const Box *node_box, const Box *prim_box, const uint32_t *index) {
    return (ref & 0x80000000u) ? prim_box[index[ref & 0x7fffffffu]] : node_box[ref];
}
// number of triangles under a reference
__device__ __forceinline__ uint32_t span_of(uint32_t ref, const PNode *nodes) { return (ref & 0x80000000u) ? 1u : nodes[ref].size; }

// The optimal collapse of the binary tree to 4-wide nodes under the surface-area cost (the host builder's, mesh_build.hpp;
// Ylitie, Karras, Laine 2017, section 3.1): c[i-1] = cheapest way to stand for the subtree as at most i wide sub-trees.
// A greedy "open the largest child, a subtree of <= 4 triangles is a leaf" collapse left 3.07 children per node and one
// more level than the host tree at the bottom of a 2 M-triangle height field.
struct Plan {
    float c[3];
    uint8_t k[3];  // k[0]: slots of an interior node's four that the left child takes; k[i-1], i = 2, 3: left share of the forest, 0 = no better than i - 1
    uint8_t leaf;  // as ONE tree the subtree is a leaf (<= 4 triangles)
};
static const float kLeafCost = 1.5f;  // a triangle step (<= 4 fp64 tests) against a node step (4 fp32 slab tests)
__device__ __forceinline__ float cost_of(uint32_t ref, int i, const Plan *plan, const Box *prim_box, const uint32_t *index) {
    return (ref & 0x80000000u) ? area(prim_box[index[ref & 0x7fffffffu]]) * kLeafCost : plan[ref].c[i - 1];
}
// binary nodes [first, last): their children were made by earlier clustering passes, so their plans are ready
__global__ __launch_bounds__(kB) void k_plan(uint32_t first, uint32_t last, const PNode *bin, const Box *node_box, const Box *prim_box, const uint32_t *index,
                                             Plan *plan) {
    const uint32_t id = first + blockIdx.x * kB + threadIdx.x;
    if (id >= last) return;
    const PNode nd = bin[id];
    const float a_me = area(node_box[id]);
    float dist[5];
    uint8_t at[5];
    for (int j = 2; j <= 4; j++) {
        dist[j] = 3.0e38f;
        at[j] = 1;
        for (int k = 1; k < j; k++) {
            const float c = cost_of(nd.left, k < 3 ? k : 3, plan, prim_box, index) + cost_of(nd.right, j - k < 3 ? j - k : 3, plan, prim_box, index);
            if (c < dist[j]) { dist[j] = c; at[j] = (uint8_t)k; }
        }
    }
    Plan p;
    const float interior = dist[4] + a_me, as_leaf = nd.size <= RTHIP_BVH_MAX_LEAF ? a_me * kLeafCost : 3.0e38f;
    p.leaf = as_leaf <= interior;
    p.c[0] = p.leaf ? as_leaf : interior;
    p.k[0] = at[4];
    for (int i = 2; i <= 3; i++) {
        const bool better = dist[i] < p.c[i - 2];
        p.c[i - 1] = better ? dist[i] : p.c[i - 2];
        p.k[i - 1] = better ? at[i] : 0;
    }
    plan[id] = p;
}
__global__ __launch_bounds__(kB) void k_count_interior(const Plan *plan, uint32_t n_internal, uint32_t *count) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    const bool big = i < n_internal && !plan[i].leaf;
    const unsigned long long m = __ballot(big);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(count, (uint32_t)__popcll(m));
}

struct Item {
    uint32_t bin_node;   // binary node this wide node is made from (covers > 4 triangles)
    uint32_t wide_node;  // its index in the output
    uint32_t first;      // leaf-order position of its first triangle
    uint32_t pad;
};
// Breadth-first collapse, one launch per level.  Leaf order is assigned on the way down: a node's triangles take the positions
// [first, first + size), its children one after the other, so that a leaf's triangles are contiguous (leaf word = first, count).
__global__ __launch_bounds__(kB) void k_collapse(const Item *items, uint32_t n_items, const PNode *bin, const Box *node_box, const Box *prim_box,
                                                 const uint32_t *index, const Plan *plan, rthip_bvhnode *out, uint32_t out_cap,
                                                 uint32_t *counters /* [0] nodes, [1] next items, [2] overflow */, Item *next, uint32_t *leaf_index) {
    const uint32_t t = blockIdx.x * kB + threadIdx.x;
    if (t >= n_items) return;
    const Item it = items[t];
    uint32_t ref[4];
    Box box[4];
    int cnt = 0;
    {   // the planned children, left to right: the roots of the forests of the two binary children
        const PNode root = bin[it.bin_node];
        const uint32_t k0 = plan[it.bin_node].k[0];
        uint32_t st_ref[4], st_i[4];
        int sp = 0;
        st_ref[sp] = root.right; st_i[sp++] = 4u - k0;
        st_ref[sp] = root.left; st_i[sp++] = k0;
        while (sp && cnt < 4) {
            const uint32_t r = st_ref[--sp];
            uint32_t i = st_i[sp];
            bool open = false;
            if (!(r & 0x80000000u)) {
                const Plan p = plan[r];
                if (i > 3u) i = 3u;
                while (i > 1u && p.k[i - 1u] == 0) i--;
                if (i > 1u && sp + 2 <= 4) {
                    const PNode o = bin[r];
                    st_ref[sp] = o.right; st_i[sp++] = i - p.k[i - 1u];
                    st_ref[sp] = o.left; st_i[sp++] = p.k[i - 1u];
                    open = true;
                }
            }
            if (!open) {
                ref[cnt] = r;
                box[cnt] = box_of(r, node_box, prim_box, index);
                cnt++;
            }
        }
    }
    rthip_bvhnode nd;
    uint32_t first = it.first;
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
        if ((ref[k] & 0x80000000u) || plan[ref[k]].leaf) {
            nd.c[k].child = RTHIP_BVH_LEAF | ((span - 1u) << RTHIP_BVH_COUNT_SHIFT) | first;
            uint32_t stack[RTHIP_BVH_MAX_LEAF], sp = 0, w = first;  // the (at most four) triangles of the subtree, left to right
            stack[sp++] = ref[k];
            while (sp) {
                const uint32_t r = stack[--sp];
                if (r & 0x80000000u) {
                    leaf_index[w++] = index[r & 0x7fffffffu];
                } else {
                    const PNode o = bin[r];
                    if (sp + 2 <= RTHIP_BVH_MAX_LEAF) {
                        stack[sp++] = o.right;
                        stack[sp++] = o.left;
                    }
                }
            }
        } else {
            const uint32_t w = atomicAdd(&counters[0], 1u);
            if (w >= out_cap) {  // cannot happen with out_cap = number of binary nodes of more than four triangles; never write past it
                atomicExch(&counters[2], 1u);
                nd.c[k].child = RTHIP_BVH_ABSENT;
            } else {
                nd.c[k].child = w;
                const uint32_t q = atomicAdd(&counters[1], 1u);
                next[q] = Item{ref[k], w, first, 0u};
            }
        }
        first += span;
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

// bvh_tris[k] = tris[bvh_index[k]]
__global__ __launch_bounds__(kB) void k_gather(const rthip_tri *tris, const uint32_t *bvh_index, uint32_t n, rthip_tri *bvh_tris) {
    // 8 lanes per 128-byte record (uint4 each): whole lines in and out
    const size_t g = (size_t)blockIdx.x * kB + threadIdx.x;
    const size_t rec = g >> 3, part = g & 7u;
    if (rec >= n) return;
    const uint32_t src = bvh_index[rec];
    reinterpret_cast<uint4 *>(bvh_tris + rec)[part] = reinterpret_cast<const uint4 *>(tris + src)[part];
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

static int build_nodes(hipStream_t stream, const rthip_tri *d_tris, uint32_t n, uint32_t *d_bvh_index, rthip_bvhnode *&nodes,
                       rthip_device_bvh_info *info, char *err, size_t err_len) {
    const uint32_t grid_n = (n + kB - 1) / kB;
    Tmp prim_box, centroid, red;
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
    info->abs_max = std::nextafterf(key2f(red_h[6]), INFINITY);

    if (n <= RTHIP_BVH_MAX_LEAF) {  // one leaf under a root
        BHIP(hipMalloc((void **)&nodes, sizeof(rthip_bvhnode)));
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, stream, prim_box.as<Box>(), n, nodes);
        uint32_t iota[RTHIP_BVH_MAX_LEAF];
        for (uint32_t i = 0; i < n; i++) iota[i] = i;
        BHIP(hipMemcpyAsync(d_bvh_index, iota, sizeof(uint32_t) * n, hipMemcpyHostToDevice, stream));
        BHIP(hipStreamSynchronize(stream));
        info->n_nodes = 1;
        info->depth = 0;
        return 0;
    }
    // Morton order
    Tmp code_in, code_out, idx_in, idx_out, sort_tmp;
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
    const uint32_t *index = idx_out.as<uint32_t>();  // sorted position -> triangle

    // PLOC: cluster arrays ping-pong between A (current) and B (merged, before compaction)
    const uint32_t n_int = n - 1;
    Tmp bin, node_box, counters, boxA, refA, sizeA, boxB, refB, sizeB, nn, valid, pos, scan_tmp;
    BHIP(bin.alloc(sizeof(PNode) * (size_t)n_int));
    BHIP(node_box.alloc(sizeof(Box) * (size_t)n_int));
    BHIP(counters.alloc(256));
    BHIP(boxA.alloc(sizeof(Box) * (size_t)n));
    BHIP(boxB.alloc(sizeof(Box) * (size_t)n));
    BHIP(refA.alloc(sizeof(uint32_t) * (size_t)n));
    BHIP(refB.alloc(sizeof(uint32_t) * (size_t)n));
    BHIP(sizeA.alloc(sizeof(uint32_t) * (size_t)n));
    BHIP(sizeB.alloc(sizeof(uint32_t) * (size_t)n));
    BHIP(nn.alloc(sizeof(uint32_t) * (size_t)n));
    BHIP(valid.alloc(sizeof(uint32_t) * (size_t)n));
    BHIP(pos.alloc(sizeof(uint32_t) * (size_t)n));
    size_t scan_bytes = 0;
    BHIP(rocprim::exclusive_scan(nullptr, scan_bytes, valid.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    BHIP(scan_tmp.alloc(scan_bytes));
    uint32_t *cnt = counters.as<uint32_t>();  // [0] wide nodes, [1] next items, [2] overflow, [3] big binary nodes, [4] binary nodes, [5] clusters left
    {
        const uint32_t zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        BHIP(hipMemcpyAsync(cnt, zeros, sizeof zeros, hipMemcpyHostToDevice, stream));
        BHIP(hipStreamSynchronize(stream));
    }
    hipLaunchKernelGGL(k_ploc_init, dim3(grid_n), dim3(kB), 0, stream, prim_box.as<Box>(), index, n, boxA.as<Box>(), refA.as<uint32_t>(), sizeA.as<uint32_t>());
    uint32_t nc = n;
    int radius = 16;
    if (const char *e = getenv("RTHIP_PLOC_RADIUS")) radius = atoi(e) < 1 ? 1 : (atoi(e) > kRadiusMax ? kRadiusMax : atoi(e));
    // (Passes that first form leaves of at most four triangles - leaf_limit = 4 - fill the leaves better, 3.6 instead of 3.1
    // triangles, but the greedy groups have worse boxes: 4 % more traversal steps and 8 % slower frames on config 4.  Off.)
    uint32_t leaf_limit = 0;
    std::vector<uint32_t> made;  // binary nodes in existence after every clustering pass
    for (int pass = 0; nc > 1; pass++) {
        if (pass > 4096) {
            snprintf(err, err_len, "BVH build: clustering did not converge (%u clusters left)", nc);
            return RTHIP_E_UNSUPPORTED;  // (the caller may build this BVH on the host instead)
        }
        const uint32_t g = (nc + kB - 1) / kB;
        hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(kB), 0, stream, boxA.as<Box>(), sizeA.as<uint32_t>(), nc, nn.as<uint32_t>(), radius, leaf_limit);
        hipLaunchKernelGGL(k_ploc_merge, dim3(g), dim3(kB), 0, stream, nn.as<uint32_t>(), boxA.as<Box>(), refA.as<uint32_t>(), sizeA.as<uint32_t>(), nc, bin.as<PNode>(),
                           node_box.as<Box>(), cnt + 4, valid.as<uint32_t>(), boxB.as<Box>(), refB.as<uint32_t>(), sizeB.as<uint32_t>());
        BHIP(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, valid.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)nc, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_ploc_compact, dim3(g), dim3(kB), 0, stream, valid.as<uint32_t>(), pos.as<uint32_t>(), boxB.as<Box>(), refB.as<uint32_t>(), sizeB.as<uint32_t>(), nc,
                           boxA.as<Box>(), refA.as<uint32_t>(), sizeA.as<uint32_t>(), cnt + 5);
        uint32_t left = 0, made_now = 0;
        BHIP(hipMemcpyAsync(&left, cnt + 5, sizeof left, hipMemcpyDeviceToHost, stream));
        BHIP(hipMemcpyAsync(&made_now, cnt + 4, sizeof made_now, hipMemcpyDeviceToHost, stream));
        BHIP(hipStreamSynchronize(stream));
        made.push_back(made_now);
        if (leaf_limit && left == nc) {  // no pair of at most four triangles left: on to the tree above the leaves
            leaf_limit = 0;
            continue;
        }
        if (left == 0 || left >= nc) {  // (every unrestricted pass merges at least the globally closest pair)
            snprintf(err, err_len, "BVH build: clustering pass %d went from %u to %u clusters", pass, nc, left);
            return RTHIP_E_HIP;
        }
        nc = left;
    }
    uint32_t root_ref = 0;
    BHIP(hipMemcpyAsync(&root_ref, refA.p, sizeof root_ref, hipMemcpyDeviceToHost, stream));
    BHIP(hipStreamSynchronize(stream));
    if (root_ref & 0x80000000u) {
        snprintf(err, err_len, "BVH build: the clustering left a triangle as the root");
        return RTHIP_E_HIP;
    }
    // the collapse plan, bottom-up in the order the clustering made the nodes (a pass only merges clusters of earlier passes)
    Tmp plan;
    BHIP(plan.alloc(sizeof(Plan) * (size_t)n_int));
    {
        uint32_t from = 0;
        for (uint32_t upto : made) {
            if (upto > from)
                hipLaunchKernelGGL(k_plan, dim3((upto - from + kB - 1) / kB), dim3(kB), 0, stream, from, upto, bin.as<PNode>(), node_box.as<Box>(), prim_box.as<Box>(),
                                   index, plan.as<Plan>());
            from = upto;
        }
        if (from != n_int) {
            snprintf(err, err_len, "BVH build: %u binary nodes made, %u expected", from, n_int);
            return RTHIP_E_HIP;
        }
    }
    // upper bound of the wide nodes: every wide node is made from a different binary node that is not planned as a leaf
    hipLaunchKernelGGL(k_count_interior, dim3((n_int + kB - 1) / kB), dim3(kB), 0, stream, plan.as<Plan>(), n_int, cnt + 3);
    uint32_t cap = 0;
    BHIP(hipMemcpyAsync(&cap, cnt + 3, sizeof cap, hipMemcpyDeviceToHost, stream));
    BHIP(hipStreamSynchronize(stream));
    if (cap == 0) cap = 1;
    Tmp items_a, items_b;
    BHIP(hipMalloc((void **)&nodes, sizeof(rthip_bvhnode) * (size_t)cap));
    BHIP(items_a.alloc(sizeof(Item) * (size_t)cap));
    BHIP(items_b.alloc(sizeof(Item) * (size_t)cap));
    // breadth-first collapse: level 0 = the root (-> wide node 0, leaf-order positions from 0)
    const Item root = {root_ref, 0u, 0u, 0u};
    const uint32_t one = 1u;
    BHIP(hipMemcpyAsync(items_a.p, &root, sizeof root, hipMemcpyHostToDevice, stream));
    BHIP(hipMemcpyAsync(cnt, &one, sizeof one, hipMemcpyHostToDevice, stream));  // wide nodes allocated so far
    BHIP(hipStreamSynchronize(stream));
    uint32_t n_items = 1, depth = 0;
    Item *cur = items_a.as<Item>(), *nxt = items_b.as<Item>();
    for (;;) {
        hipLaunchKernelGGL(k_collapse, dim3((n_items + kB - 1) / kB), dim3(kB), 0, stream, cur, n_items, bin.as<PNode>(), node_box.as<Box>(), prim_box.as<Box>(), index,
                           plan.as<Plan>(), nodes, cap, cnt, nxt, d_bvh_index);
        uint32_t c3[3];
        BHIP(hipMemcpyAsync(c3, cnt, sizeof c3, hipMemcpyDeviceToHost, stream));
        BHIP(hipStreamSynchronize(stream));
        if (c3[2]) {
            snprintf(err, err_len, "BVH build: node buffer overflow");
            return RTHIP_E_HIP;
        }
        info->n_nodes = c3[0];
        n_items = c3[1];
        if (n_items == 0) break;
        depth++;
        if (depth > RTHIP_BVH_MAX_DEPTH) {
            snprintf(err, err_len, "device-built BVH is deeper than %d levels", RTHIP_BVH_MAX_DEPTH);
            return RTHIP_E_UNSUPPORTED;
        }
        const uint32_t zero = 0u;
        BHIP(hipMemcpyAsync(cnt + 1, &zero, sizeof zero, hipMemcpyHostToDevice, stream));
        BHIP(hipStreamSynchronize(stream));
        Item *sw = cur; cur = nxt; nxt = sw;
    }
    info->depth = depth;
    return 0;
}

int rthip_bvh_build_device(hipStream_t stream, const rthip_tri *d_tris, uint32_t n, rthip_tri *d_bvh_tris, uint32_t *d_bvh_index,
                           rthip_bvhnode **d_nodes_out, rthip_device_bvh_info *info, char *err, size_t err_len) {
    memset(info, 0, sizeof *info);
    *d_nodes_out = nullptr;
    if (n == 0) return 0;
    if (n > RTHIP_BVH_FIRST_MASK) {
        snprintf(err, err_len, "mesh of %u triangles is too large for the BVH leaf encoding", n);
        return RTHIP_E_UNSUPPORTED;
    }
    rthip_bvhnode *nodes = nullptr;
    int rc = build_nodes(stream, d_tris, n, d_bvh_index, nodes, info, err, err_len);
    if (rc == 0) {
        const size_t threads = (size_t)n * 8;
        hipLaunchKernelGGL(k_gather, dim3((uint32_t)((threads + kB - 1) / kB)), dim3(kB), 0, stream, d_tris, d_bvh_index, n, d_bvh_tris);
        hipError_t e = hipMemcpyAsync(&info->root, nodes, sizeof(rthip_bvhnode), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            snprintf(err, err_len, "BVH build: %s", hipGetErrorString(e));
            rc = RTHIP_E_HIP;
        }
    }
    if (rc) {
        (void)hipStreamSynchronize(stream);
        if (nodes) (void)hipFree(nodes);
        return rc;
    }
    *d_nodes_out = nodes;
    return 0;
}
