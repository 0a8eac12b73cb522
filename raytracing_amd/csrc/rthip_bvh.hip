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
//   binary tree : (default) k_sah_*: top-down binned SAH, the host builder's algorithm (16 bins per axis over the centroid
//                 bounds, cost = area x ceil(count / 4) per side, down to single triangles) - level-synchronous for nodes of
//                 more than 64 triangles (bounds, bins, split, partition: one launch each per level, LDS accumulation for
//                 the chunks that lie inside one node), one wave per subtree of <= 64 triangles (full sweep, a lane per
//                 triangle); every link of the result is checked (k_sah_validate) before anything walks it.
//                 (RTHIP_BVH_BUILDER=ploc) k_morton + rocprim radix sort + k_ploc_*: parallel locally-ordered clustering
//                 (Meister & Bittner 2018) in Morton order - as fast to build, 22 % slower to walk on config 4
//   k_plan      : the optimal collapse of the binary tree to 4-wide nodes and leaves of <= 4 triangles under the surface-area
//                 cost (bottom-up: by clustering pass / by depth)
//   k_collapse  : breadth-first, one launch per level, children as planned; leaf order is assigned on the way down; node
//                 indices grow with depth (children follow their parent, as the host builder's)
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
    // (a bounded grid that strides over the triangles: the seven atomics of every wave go to ONE line, and 32 K waves' worth of
    // them took 2.3 ms of a 2 M-triangle build)
    float am = 0.0f;
    uint32_t kmin[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, kmax[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB) {
        const rthip_tri t = tris[i];
        Box b;
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // the vertices as the triangle test sees them: v0, v0 + e0, v0 + e1 (fp64; the outward fp32 rounding plus one ulp
            // swallows the rounding of the two sums, 2^-53 relative)
            const double v0 = t.v0[k], v1 = t.v0[k] + t.e0[k], v2 = t.v0[k] + t.e1[k];
            const double l = fmin(v0, fmin(v1, v2)), h = fmax(v0, fmax(v1, v2));
            b.lo[k] = round_down(l);
            b.hi[k] = round_up(h);
            c[k] = (float)(0.5 * (l + h));
            if (!(fabsf(c[k]) < 3.0e38f)) c[k] = 0.0f;  // (the builders order triangles by centroid: keep it comparable)
            am = fmaxf(am, fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])));
            kmin[k] = min(kmin[k], f2key(c[k]));
            kmax[k] = max(kmax[k], f2key(c[k]));
        }
        prim_box[i] = b;
        centroid[3 * (size_t)i] = c[0];
        centroid[3 * (size_t)i + 1] = c[1];
        centroid[3 * (size_t)i + 2] = c[2];
    }
    uint32_t kam = f2key(am);
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
__device__ __forceinline__ void plan_node(uint32_t id, const PNode *bin, const Box *node_box, const Box *prim_box, const uint32_t *index, Plan *plan) {
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
    p.leaf = nd.size <= RTHIP_BVH_MAX_LEAF && as_leaf <= interior;  // (never a leaf of more than four, whatever the costs)
    p.c[0] = p.leaf ? as_leaf : interior;
    p.k[0] = at[4];
    for (int i = 2; i <= 3; i++) {
        const bool better = dist[i] < p.c[i - 2];
        p.c[i - 1] = better ? dist[i] : p.c[i - 2];
        p.k[i - 1] = better ? at[i] : 0;
    }
    plan[id] = p;
}
// binary nodes [first, last): their children were made by earlier clustering passes, so their plans are ready
__global__ __launch_bounds__(kB) void k_plan(uint32_t first, uint32_t last, const PNode *bin, const Box *node_box, const Box *prim_box, const uint32_t *index,
                                             Plan *plan) {
    const uint32_t id = first + blockIdx.x * kB + threadIdx.x;
    if (id >= last) return;
    plan_node(id, bin, node_box, prim_box, index, plan);
}
// the binary nodes at depth d of a tree built top-down (deepest first: the children's plans are ready)
__global__ __launch_bounds__(kB) void k_plan_depth(uint32_t n_nodes, uint32_t d, const uint32_t *node_depth, const PNode *bin, const Box *node_box,
                                                   const Box *prim_box, const uint32_t *index, Plan *plan) {
    const uint32_t id = blockIdx.x * kB + threadIdx.x;
    if (id >= n_nodes || node_depth[id] != d) return;
    plan_node(id, bin, node_box, prim_box, index, plan);
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


// ---------------------------------------------------------------------------------------------------------------------
// Binary tree by top-down binned SAH on the device - the host builder's algorithm (mesh_build.hpp: 16 bins on each axis over
// the centroid bounds, cost = area * ceil(count / 4) per side, down to single triangles), level-synchronous for the nodes of
// more than kSmall triangles (bounds -> bins -> split -> partition, one launch each per level; a wave's chunk of positions that
// lies inside one node accumulates in registers / LDS and sends one set of atomics) and one WAVE per subtree of <= kSmall triangles (one triangle per lane,
// every lane's centroid a candidate plane on each axis: a full sweep).  Positions: `order` holds triangle ids, a node owns a
// contiguous range; a leaf reference is 0x80000000 | position, as in the clustering builder.
const uint32_t kSmall = 64, kNone = 0xffffffffu, kSahBins = 16;
const uint32_t kNbWords = 12, kBinWords = 3 * kSahBins * 7;
struct ANode { uint32_t lo, hi, parent, side; };                   // parent: binary node whose child link this node fills in (kNone: the root)
struct Job { uint32_t lo, hi, parent, side, depth, pad[3]; };
struct SplitRec { int32_t axis; uint32_t bin, mid, slot[2]; };     // axis < 0: split by position; slot: next level's node of each side, kNone if finished
enum { SC_NODES = 0, SC_NEXT = 1, SC_JOBS = 2, SC_ROOT = 3, SC_DEPTH = 4 };

__global__ __launch_bounds__(kB) void k_sah_clear(uint32_t *nb, uint32_t *bins, uint32_t n_active) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i < n_active * kNbWords) {
        const uint32_t w = i % kNbWords;
        nb[i] = (w < 3 || (w >= 6 && w < 9)) ? 0xffffffffu : 0u;
    }
    if (i < n_active * kBinWords) {
        const uint32_t w = i % 7u;
        bins[i] = w == 0 ? 0u : (w < 4 ? 0xffffffffu : 0u);
    }
}
// A WAVE takes `wchunk` consecutive positions (512 where the nodes are large, 64 at the deep levels).  A node's range is
// contiguous, so a wave-chunk whose first and last position belong to the same node lies inside it: the wave then accumulates
// in registers / in its own LDS table and sends ONE set of atomics for the chunk; a mixed chunk sends them per element.
// (Finished positions - kNone - are not one range: such a chunk is always treated as mixed.)  No workgroup barrier anywhere:
// the four waves of a workgroup take their own paths.
struct WaveChunk {
    uint32_t w0, w1, node;
    bool uniform;
};
__device__ __forceinline__ WaveChunk wave_chunk(const uint32_t *seg, uint32_t n, uint32_t wchunk) {
    WaveChunk c;
    const unsigned long long first = (unsigned long long)((blockIdx.x * kB + threadIdx.x) >> 6) * wchunk;
    c.w0 = first < n ? (uint32_t)first : n;
    c.w1 = (first + wchunk < n) ? (uint32_t)(first + wchunk) : n;
    c.node = kNone;
    c.uniform = false;
    if (c.w0 < c.w1) {
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg[c.w0]), s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg[c.w1 - 1]);
        c.node = s0;
        c.uniform = s0 == s1 && s0 != kNone;
    }
    return c;
}
__device__ __forceinline__ void wave_lds_sync() {  // this wave's LDS accesses so far are done before the ones that follow (other lanes' included)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// per node: [0..2] / [3..5] min / max keys of the triangle boxes, [6..8] / [9..11] of the centroids
__global__ __launch_bounds__(kB) void k_sah_bounds(const uint32_t *order, const uint32_t *seg, uint32_t n, uint32_t wchunk, const Box *prim_box,
                                                   const float *centroid, uint32_t *nb) {
    const WaveChunk c = wave_chunk(seg, n, wchunk);
    if (c.w0 >= c.w1) return;
    const uint32_t lane = threadIdx.x & 63u;
    if (c.uniform) {
        uint32_t k[kNbWords];
#pragma unroll
        for (int w = 0; w < (int)kNbWords; w++) k[w] = (w < 3 || (w >= 6 && w < 9)) ? 0xffffffffu : 0u;
        for (uint32_t p = c.w0 + lane; p < c.w1; p += 64u) {
            const uint32_t t = order[p];
            const Box b = prim_box[t];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const uint32_t ck = f2key(centroid[3 * (size_t)t + a]);
                k[a] = min(k[a], f2key(b.lo[a]));
                k[3 + a] = max(k[3 + a], f2key(b.hi[a]));
                k[6 + a] = min(k[6 + a], ck);
                k[9 + a] = max(k[9 + a], ck);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int w = 0; w < (int)kNbWords; w++) {
                const uint32_t o = (uint32_t)__shfl_xor((int)k[w], off, 64);
                k[w] = (w < 3 || (w >= 6 && w < 9)) ? min(k[w], o) : max(k[w], o);
            }
        }
        if (lane == 0) {
            uint32_t *dst = nb + (size_t)c.node * kNbWords;
#pragma unroll
            for (int w = 0; w < (int)kNbWords; w++) {
                if (w < 3 || (w >= 6 && w < 9)) atomicMin(dst + w, k[w]);
                else atomicMax(dst + w, k[w]);
            }
        }
        return;
    }
    for (uint32_t p = c.w0 + lane; p < c.w1; p += 64u) {
        const uint32_t a_ = seg[p];
        if (a_ == kNone) continue;
        const uint32_t t = order[p];
        const Box b = prim_box[t];
        uint32_t *dst = nb + (size_t)a_ * kNbWords;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const uint32_t ck = f2key(centroid[3 * (size_t)t + a]);
            atomicMin(dst + a, f2key(b.lo[a]));
            atomicMax(dst + 3 + a, f2key(b.hi[a]));
            atomicMin(dst + 6 + a, ck);
            atomicMax(dst + 9 + a, ck);
        }
    }
}
__global__ __launch_bounds__(kB) void k_iota(uint32_t *a, uint32_t n) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i < n) a[i] = i;
}
// bin of a centroid coordinate (the same expression in k_sah_bin and k_sah_partition)
__device__ __forceinline__ uint32_t sah_bin(float c, float base, float k) {
    int b = (int)((c - base) * k);
    return (uint32_t)(b < 0 ? 0 : (b >= (int)kSahBins ? (int)kSahBins - 1 : b));
}
__device__ __forceinline__ void sah_axis(const uint32_t *nbn, int a, float &base, float &k) {
    const float lo = key2f(nbn[6 + a]), hi = key2f(nbn[9 + a]);
    base = lo;
    k = hi > lo ? (float)kSahBins / (hi - lo) : 0.0f;
}
// per node, axis, bin: count, min keys (3), max keys (3) of the triangle boxes whose centroid falls in the bin
__global__ __launch_bounds__(kB) void k_sah_bin(const uint32_t *order, const uint32_t *seg, uint32_t n, uint32_t wchunk, const Box *prim_box,
                                                const float *centroid, const uint32_t *nb, uint32_t *bins) {
    __shared__ uint32_t sh_all[kB / 64][kBinWords];
    const WaveChunk c = wave_chunk(seg, n, wchunk);
    if (c.w0 >= c.w1) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *sh = sh_all[threadIdx.x >> 6];
    if (c.uniform) {
        for (uint32_t w = lane; w < kBinWords; w += 64u) sh[w] = (w % 7u) == 0 ? 0u : ((w % 7u) < 4 ? 0xffffffffu : 0u);
        wave_lds_sync();
    }
    for (uint32_t p = c.w0 + lane; p < c.w1; p += 64u) {
        const uint32_t nd = seg[p];
        if (nd == kNone) continue;
        const uint32_t t = order[p];
        const Box b = prim_box[t];
        const uint32_t *nbn = nb + (size_t)nd * kNbWords;
        uint32_t *dst = c.uniform ? sh : bins + (size_t)nd * kBinWords;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float base, k;
            sah_axis(nbn, a, base, k);
            if (k == 0.0f) continue;  // no extent on this axis: not a candidate
            uint32_t *cell = dst + ((uint32_t)a * kSahBins + sah_bin(centroid[3 * (size_t)t + a], base, k)) * 7u;
            atomicAdd(cell, 1u);
#pragma unroll
            for (int x = 0; x < 3; x++) {
                atomicMin(cell + 1 + x, f2key(b.lo[x]));
                atomicMax(cell + 4 + x, f2key(b.hi[x]));
            }
        }
    }
    if (c.uniform) {
        wave_lds_sync();
        uint32_t *dst = bins + (size_t)c.node * kBinWords;
        for (uint32_t w = lane; w < kBinWords; w += 64u) {
            const uint32_t v = sh[w], f = w % 7u;
            if (f == 0) { if (v) atomicAdd(dst + w, v); }
            else if (f < 4) { if (v != 0xffffffffu) atomicMin(dst + w, v); }
            else if (v) atomicMax(dst + w, v);
        }
    }
}
__device__ __forceinline__ Box key_box(const uint32_t *lo, const uint32_t *hi) {
    Box b;
#pragma unroll
    for (int a = 0; a < 3; a++) { b.lo[a] = key2f(lo[a]); b.hi[a] = key2f(hi[a]); }
    return b;
}
__device__ __forceinline__ Box empty_box() {
    Box b;
#pragma unroll
    for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
    return b;
}
// one thread per node of the level: the split, the binary node, the children (next level / small-subtree job / leaf)
__global__ __launch_bounds__(kB) void k_sah_split(const ANode *active, uint32_t n_active, uint32_t level, const uint32_t *nb, const uint32_t *bins,
                                                  PNode *bin, Box *node_box, uint32_t *node_depth, uint32_t *counters, SplitRec *split, uint32_t *cursors,
                                                  ANode *next, Job *jobs) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n_active) return;
    const ANode nd = active[i];
    const uint32_t n = nd.hi - nd.lo;
    const uint32_t *nbn = nb + (size_t)i * kNbWords;
    const uint32_t *bn = bins + (size_t)i * kBinWords;
    int best_axis = -1;
    uint32_t best_bin = 0, best_left = 0;
    float best_cost = 3.0e38f;
    if (level < 40) {
        for (int a = 0; a < 3; a++) {
            float base, k;
            sah_axis(nbn, a, base, k);
            if (k == 0.0f) continue;
            float right_area[kSahBins];
            uint32_t right_cnt[kSahBins];
            Box acc = empty_box();
            uint32_t c = 0;
            for (int b = (int)kSahBins - 1; b > 0; b--) {
                const uint32_t *cell = bn + ((uint32_t)a * kSahBins + (uint32_t)b) * 7u;
                if (cell[0]) acc = merge(acc, key_box(cell + 1, cell + 4));
                c += cell[0];
                right_area[b] = c ? area(acc) : 0.0f;
                right_cnt[b] = c;
            }
            acc = empty_box();
            c = 0;
            for (int b = 0; b < (int)kSahBins - 1; b++) {
                const uint32_t *cell = bn + ((uint32_t)a * kSahBins + (uint32_t)b) * 7u;
                if (cell[0]) acc = merge(acc, key_box(cell + 1, cell + 4));
                c += cell[0];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                const float cost = area(acc) * (float)((c + 3u) / 4u) + right_area[b + 1] * (float)((right_cnt[b + 1] + 3u) / 4u);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = (uint32_t)b; best_left = c; }
            }
        }
    }
    SplitRec sr;
    sr.axis = best_axis;
    sr.bin = best_bin;
    sr.mid = best_axis >= 0 ? nd.lo + best_left : nd.lo + n / 2;  // coincident centroids or depth guard: halves by position
    const uint32_t id = atomicAdd(&counters[SC_NODES], 1u);
    node_box[id] = key_box(nbn, nbn + 3);
    node_depth[id] = level;
    atomicMax(&counters[SC_DEPTH], level);
    bin[id].size = n;
    if (nd.parent == kNone) counters[SC_ROOT] = id;
    else (&bin[nd.parent].left)[nd.side] = id;
    for (uint32_t sd = 0; sd < 2; sd++) {
        const uint32_t lo = sd ? sr.mid : nd.lo, hi = sd ? nd.hi : sr.mid, cnt = hi - lo;
        sr.slot[sd] = kNone;
        if (cnt == 1) {
            (&bin[id].left)[sd] = 0x80000000u | lo;
        } else if (cnt <= kSmall) {
            const uint32_t j = atomicAdd(&counters[SC_JOBS], 1u);
            jobs[j] = Job{lo, hi, id, sd, level + 1u, {0u, 0u, 0u}};
        } else {
            const uint32_t j = atomicAdd(&counters[SC_NEXT], 1u);
            next[j] = ANode{lo, hi, id, sd};
            sr.slot[sd] = j;
        }
    }
    split[i] = sr;
    cursors[2 * i] = nd.lo;
    cursors[2 * i + 1] = sr.mid;
}
__global__ __launch_bounds__(kB) void k_sah_partition(const uint32_t *order, const uint32_t *seg, uint32_t n, uint32_t wchunk, const float *centroid,
                                                      const uint32_t *nb, const SplitRec *split, uint32_t *cursors, uint32_t *order_out, uint32_t *seg_out) {
    const WaveChunk c = wave_chunk(seg, n, wchunk);
    if (c.w0 >= c.w1) return;
    const uint32_t lane = threadIdx.x & 63u;
    if (c.uniform && split[c.node].axis >= 0) {  // one node: count the sides (ballots), reserve both runs with ONE atomic each, keep the order
        const SplitRec sr = split[c.node];
        float base, k;
        sah_axis(nb + (size_t)c.node * kNbWords, sr.axis, base, k);
        uint32_t cnt[2] = {0u, 0u};
        for (uint32_t p0 = c.w0; p0 < c.w1; p0 += 64u) {
            const uint32_t p = p0 + lane;
            const bool live = p < c.w1;
            const bool left = live && sah_bin(centroid[3 * (size_t)order[live ? p : c.w0] + sr.axis], base, k) <= sr.bin;
            const unsigned long long ml = __ballot(left), mr = __ballot(live && !left);
            cnt[0] += (uint32_t)__popcll(ml);
            cnt[1] += (uint32_t)__popcll(mr);
        }
        uint32_t at[2] = {0u, 0u};
        if (lane < 2u && cnt[lane]) at[lane] = atomicAdd(&cursors[2 * c.node + lane], cnt[lane]);
        at[0] = (uint32_t)__builtin_amdgcn_readlane((int)at[0], 0);
        at[1] = (uint32_t)__builtin_amdgcn_readlane((int)at[1], 1);
        const unsigned long long below = (1ull << lane) - 1ull;
        for (uint32_t p0 = c.w0; p0 < c.w1; p0 += 64u) {
            const uint32_t p = p0 + lane;
            const bool live = p < c.w1;
            const uint32_t t = order[live ? p : c.w0];
            const bool left = live && sah_bin(centroid[3 * (size_t)t + sr.axis], base, k) <= sr.bin;
            const unsigned long long ml = __ballot(left), mr = __ballot(live && !left);
            if (live) {
                const uint32_t q = left ? at[0] + (uint32_t)__popcll(ml & below) : at[1] + (uint32_t)__popcll(mr & below);
                order_out[q] = t;
                seg_out[q] = sr.slot[left ? 0 : 1];
            }
            at[0] += (uint32_t)__popcll(ml);
            at[1] += (uint32_t)__popcll(mr);
        }
        return;
    }
    for (uint32_t p = c.w0 + lane; p < c.w1; p += 64u) {
        const uint32_t nd = seg[p];
        const uint32_t t = order[p];
        if (nd == kNone) { order_out[p] = t; seg_out[p] = kNone; continue; }
        const SplitRec sr = split[nd];
        uint32_t sd, q;
        if (sr.axis < 0) {
            sd = p < sr.mid ? 0u : 1u;
            q = p;
        } else {
            float base, k;
            sah_axis(nb + (size_t)nd * kNbWords, sr.axis, base, k);
            sd = sah_bin(centroid[3 * (size_t)t + sr.axis], base, k) <= sr.bin ? 0u : 1u;
            q = atomicAdd(&cursors[2 * nd + sd], 1u);
        }
        order_out[q] = t;
        seg_out[q] = sr.slot[sd];
    }
}
// One wave (= one 64-thread workgroup) per subtree of 2 .. 64 triangles, one triangle per lane.  Every round splits every segment
// of two or more lanes: lane i proposes, on each axis, the plane behind its own centroid (left = the lanes of its segment that
// come before it in (centroid, lane) order, itself included), the segment takes the cheapest proposal, its lanes are
// rearranged left | right.  The binary node of a split at lane boundary m is base + m - 1 (every boundary is cut exactly once).
struct SmallLane {
    uint32_t t, s, e, par, side, dep;
    Box b;
    float c[3];
};
__device__ __forceinline__ float lane_f(float v, uint32_t j) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)j)); }
__global__ __launch_bounds__(64) void k_sah_small(const Job *jobs, uint32_t n_jobs, const Box *prim_box, const float *centroid, uint32_t *order, PNode *bin,
                                                  Box *node_box, uint32_t *node_depth, uint32_t *counters) {
    __shared__ SmallLane sh[64];
    if (blockIdx.x >= n_jobs) return;
    const Job jb = jobs[blockIdx.x];
    const uint32_t n = jb.hi - jb.lo, lane = threadIdx.x;
    SmallLane me;
    me.t = 0; me.s = lane; me.e = lane + 1u; me.par = kNone; me.side = 0; me.dep = jb.depth;
    me.b = empty_box();
    me.c[0] = me.c[1] = me.c[2] = 0.0f;
    if (lane < n) {
        me.t = order[jb.lo + lane];
        me.b = prim_box[me.t];
        for (int a = 0; a < 3; a++) me.c[a] = centroid[3 * (size_t)me.t + a];
        me.s = 0; me.e = n; me.par = jb.parent; me.side = jb.side;
    }
    {   // The lanes in the order of their triangle ids: the partition above fills a range in the order its atomics land, and every
        // tie below (equal centroids, equal costs) goes to the lower lane - with the lanes in a canonical order the subtree is a
        // function of the SET of triangles, the same from run to run (the splits keep the lanes' relative order).
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) rank += ((uint32_t)__builtin_amdgcn_readlane((int)me.t, (int)j) < me.t) ? 1u : 0u;
        sh[lane < n ? rank : lane] = me;
        __syncthreads();
        me = sh[lane];
        __syncthreads();
    }
    uint32_t base = lane == 0 ? atomicAdd(&counters[SC_NODES], n - 1u) : 0u;
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    for (uint32_t round = 0; round < 64; round++) {
        const bool splitting = me.e - me.s >= 2u;
        if (!__ballot(splitting)) break;
        float best_cost = 3.0e38f;
        int best_axis = 0;
        uint32_t best_left = 0;
        Box seg_box = empty_box();
        {   // one sweep over the lanes for the three axes (the box of lane j is broadcast once)
            Box L[3], R[3];
            uint32_t cl[3] = {0u, 0u, 0u}, cr[3] = {0u, 0u, 0u};
#pragma unroll
            for (int a = 0; a < 3; a++) { L[a] = empty_box(); R[a] = empty_box(); }
            for (uint32_t j = 0; j < n; j++) {
                const float cj[3] = {lane_f(me.c[0], j), lane_f(me.c[1], j), lane_f(me.c[2], j)};
                Box bj;
#pragma unroll
                for (int x = 0; x < 3; x++) { bj.lo[x] = lane_f(me.b.lo[x], j); bj.hi[x] = lane_f(me.b.hi[x], j); }
                const bool in_seg = j >= me.s && j < me.e;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const bool left = cj[a] < me.c[a] || (cj[a] == me.c[a] && j <= lane);
                    if (in_seg && left) { L[a] = merge(L[a], bj); cl[a]++; }
                    if (in_seg && !left) { R[a] = merge(R[a], bj); cr[a]++; }
                }
            }
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float cost = cr[a] ? area(L[a]) * (float)((cl[a] + 3u) / 4u) + area(R[a]) * (float)((cr[a] + 3u) / 4u) : 3.0e38f;
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_left = cl[a]; }
            }
            seg_box = merge(L[0], R[0]);  // (whatever the plane: the whole segment)
        }
        // the segment's cheapest proposal (lowest lane among equals)
        float wcost = best_cost;
        uint32_t w = lane;
        for (uint32_t j = 0; j < n; j++) {
            const float cj = lane_f(best_cost, j);
            if (j >= me.s && j < me.e && (cj < wcost || (cj == wcost && j < w))) { wcost = cj; w = j; }
        }
        const bool none = !(wcost < 3.0e38f);  // (boxes whose area overflows: halves by lane)
        if (none) w = me.s;
        const int wa = __shfl(best_axis, (int)w, 64);
        const uint32_t wl = none ? (me.e - me.s) / 2u : (uint32_t)__shfl((int)best_left, (int)w, 64);
        const float wc0 = __shfl(me.c[0], (int)w, 64), wc1 = __shfl(me.c[1], (int)w, 64), wc2 = __shfl(me.c[2], (int)w, 64);
        const float wc = wa == 0 ? wc0 : (wa == 1 ? wc1 : wc2), ck = wa == 0 ? me.c[0] : (wa == 1 ? me.c[1] : me.c[2]);
        const bool left = none ? lane < me.s + wl : (ck < wc || (ck == wc && lane <= w));
        const uint32_t len = me.e - me.s;
        const unsigned long long seg_mask = (len >= 64u ? ~0ull : ((1ull << len) - 1ull)) << me.s;
        const unsigned long long lm = __ballot(left) & seg_mask, below = (1ull << lane) - 1ull;
        const uint32_t m = me.s + wl;  // the boundary this split cuts
        uint32_t pos = lane;
        if (splitting) pos = left ? me.s + (uint32_t)__popcll(lm & below) : m + (uint32_t)__popcll(~lm & seg_mask & below);
        if (splitting && lane == w) {  // the proposer writes the node
            const uint32_t id = base + m - 1u;
            node_box[id] = seg_box;
            node_depth[id] = me.dep;
            atomicMax(&counters[SC_DEPTH], me.dep);
            bin[id].size = len;
            if (me.par == kNone) counters[SC_ROOT] = id;
            else (&bin[me.par].left)[me.side] = id;
            if (wl == 1u) bin[id].left = 0x80000000u | (jb.lo + me.s);
            if (me.e - m == 1u) bin[id].right = 0x80000000u | (jb.lo + me.e - 1u);
        }
        if (splitting) {
            me.par = base + m - 1u;
            me.side = left ? 0u : 1u;
            me.dep++;
            if (left) me.e = m; else me.s = m;
        }
        sh[pos] = me;
        __syncthreads();
        me = sh[lane];
        __syncthreads();
    }
    if (lane < n) order[jb.lo + lane] = me.t;
}

// Every link of the finished binary tree is checked before anything walks it (a wild reference would be a GPU fault in the
// kernels that follow): references in range, sizes adding up, depths increasing by one, boxes finite and containing their children.
// bad[0..5]: left / right reference out of range, size, depth, box containment, box not finite.
__global__ __launch_bounds__(kB) void k_sah_validate(const PNode *bin, const Box *node_box, const uint32_t *node_depth, uint32_t n_int, uint32_t n,
                                                     const Box *prim_box, const uint32_t *index, uint32_t root, uint32_t *bad) {
    const uint32_t id = blockIdx.x * kB + threadIdx.x;
    if (id >= n_int) return;
    const PNode nd = bin[id];
    const Box me = node_box[id];
    bool finite = true;
    for (int a = 0; a < 3; a++) finite = finite && fabsf(me.lo[a]) < 3.0e38f && fabsf(me.hi[a]) < 3.0e38f && me.lo[a] <= me.hi[a];
    if (!finite) atomicAdd(&bad[5], 1u);
    uint32_t total = 0;
    for (int sd = 0; sd < 2; sd++) {
        const uint32_t ref = sd ? nd.right : nd.left;
        Box cb;
        if (ref & 0x80000000u) {
            const uint32_t pos = ref & 0x7fffffffu;
            if (pos >= n || index[pos] >= n) { atomicAdd(&bad[sd], 1u); continue; }
            cb = prim_box[index[pos]];
            total += 1u;
        } else {
            if (ref >= n_int || ref == id || ref == root) { atomicAdd(&bad[sd], 1u); continue; }
            cb = node_box[ref];
            total += bin[ref].size;
            if (node_depth && node_depth[ref] != node_depth[id] + 1u) atomicAdd(&bad[3], 1u);  // (the clustering builder keeps no depths)
        }
        bool inside = true;
        for (int a = 0; a < 3; a++) inside = inside && cb.lo[a] >= me.lo[a] && cb.hi[a] <= me.hi[a];
        if (!inside) atomicAdd(&bad[4], 1u);
    }
    if (total != nd.size) atomicAdd(&bad[2], 1u);
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
    hipLaunchKernelGGL(k_prims, dim3(grid_n < 2048u ? grid_n : 2048u), dim3(kB), 0, stream, d_tris, n, prim_box.as<Box>(), centroid.as<float>(), red.as<uint32_t>());
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
    // the binary tree: top-down binned SAH (the host builder's algorithm; default) or PLOC clustering (RTHIP_BVH_BUILDER=ploc)
    bool sah = true;
    if (const char *e = getenv("RTHIP_BVH_BUILDER")) sah = strcmp(e, "ploc") != 0;
    const uint32_t n_int = n - 1;
    Tmp bin, node_box, counters, plan, idx_out;
    BHIP(bin.alloc(sizeof(PNode) * (size_t)n_int));
    BHIP(node_box.alloc(sizeof(Box) * (size_t)n_int));
    BHIP(plan.alloc(sizeof(Plan) * (size_t)n_int));
    BHIP(idx_out.alloc(sizeof(uint32_t) * (size_t)n));
    BHIP(counters.alloc(256));
    uint32_t *cnt = counters.as<uint32_t>();  // [0] wide nodes, [1] next items, [2] overflow, [3] big binary nodes, [4] binary nodes, [5] clusters left
    {
        const uint32_t zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        BHIP(hipMemcpyAsync(cnt, zeros, sizeof zeros, hipMemcpyHostToDevice, stream));
        BHIP(hipStreamSynchronize(stream));
    }
    const uint32_t *index = nullptr;
    uint32_t root_ref = 0;
    if (!sah) {
        // Morton order
        Tmp code_in, code_out, idx_in, sort_tmp;
        BHIP(code_in.alloc(sizeof(unsigned long long) * (size_t)n));
        BHIP(code_out.alloc(sizeof(unsigned long long) * (size_t)n));
        BHIP(idx_in.alloc(sizeof(uint32_t) * (size_t)n));
        hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(kB), 0, stream, centroid.as<float>(), n, red.as<uint32_t>(), code_in.as<unsigned long long>(), idx_in.as<uint32_t>());
        size_t sort_bytes = 0;
        BHIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, code_in.as<unsigned long long>(), code_out.as<unsigned long long>(), idx_in.as<uint32_t>(),
                                       idx_out.as<uint32_t>(), (size_t)n, 0u, 63u, stream));
        BHIP(sort_tmp.alloc(sort_bytes));
        BHIP(rocprim::radix_sort_pairs(sort_tmp.p, sort_bytes, code_in.as<unsigned long long>(), code_out.as<unsigned long long>(), idx_in.as<uint32_t>(),
                                       idx_out.as<uint32_t>(), (size_t)n, 0u, 63u, stream));
        index = idx_out.as<uint32_t>();  // sorted position -> triangle

        // PLOC: cluster arrays ping-pong between A (current) and B (merged, before compaction)
        Tmp boxA, refA, sizeA, boxB, refB, sizeB, nn, valid, pos, scan_tmp;
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
        BHIP(hipMemcpyAsync(&root_ref, refA.p, sizeof root_ref, hipMemcpyDeviceToHost, stream));
        BHIP(hipStreamSynchronize(stream));
        if (root_ref & 0x80000000u) {
            snprintf(err, err_len, "BVH build: the clustering left a triangle as the root");
            return RTHIP_E_HIP;
        }
        {   // the same check of every link as for the top-down tree, before the plan and collapse kernels walk it
            uint32_t *bad_d = cnt + 8;
            BHIP(hipMemsetAsync(bad_d, 0, sizeof(uint32_t) * 8, stream));
            hipLaunchKernelGGL(k_sah_validate, dim3((n_int + kB - 1) / kB), dim3(kB), 0, stream, bin.as<PNode>(), node_box.as<Box>(), (const uint32_t *)nullptr,
                               n_int, n, prim_box.as<Box>(), index, root_ref, bad_d);
            uint32_t bad[8];
            BHIP(hipMemcpyAsync(bad, bad_d, sizeof bad, hipMemcpyDeviceToHost, stream));
            BHIP(hipStreamSynchronize(stream));
            if (root_ref >= n_int || bad[0] || bad[1] || bad[2] || bad[4] || bad[5]) {
                snprintf(err, err_len, "BVH build: the clustered tree failed its check (root %u of %u; bad left %u right %u size %u box %u finite %u)",
                         root_ref, n_int, bad[0], bad[1], bad[2], bad[4], bad[5]);
                return RTHIP_E_UNSUPPORTED;  // (the caller may build this BVH on the host instead)
            }
        }
        // the collapse plan, bottom-up in the order the clustering made the nodes (a pass only merges clusters of earlier passes)
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
    } else {
        // top-down binned SAH (see k_sah_*): level-synchronous above kSmall triangles, one wave per subtree below
        const uint32_t a_max = n / (kSmall + 1u) + 2u, j_max = n / 2u + 2u;
        Tmp order_b, seg_a, seg_b, nb, bins, split, cursors, act_a, act_b, jobs, node_depth, sc;
        BHIP(order_b.alloc(sizeof(uint32_t) * (size_t)n));
        BHIP(seg_a.alloc(sizeof(uint32_t) * (size_t)n));
        BHIP(seg_b.alloc(sizeof(uint32_t) * (size_t)n));
        BHIP(nb.alloc(sizeof(uint32_t) * kNbWords * (size_t)a_max));
        BHIP(bins.alloc(sizeof(uint32_t) * kBinWords * (size_t)a_max));
        BHIP(split.alloc(sizeof(SplitRec) * (size_t)a_max));
        BHIP(cursors.alloc(sizeof(uint32_t) * 2 * (size_t)a_max));
        BHIP(act_a.alloc(sizeof(ANode) * (size_t)a_max));
        BHIP(act_b.alloc(sizeof(ANode) * (size_t)a_max));
        BHIP(jobs.alloc(sizeof(Job) * (size_t)j_max));
        BHIP(node_depth.alloc(sizeof(uint32_t) * (size_t)n_int));
        BHIP(sc.alloc(256));
        uint32_t *scp = sc.as<uint32_t>();
        uint32_t *ord_cur = idx_out.as<uint32_t>(), *ord_nxt = order_b.as<uint32_t>();
        uint32_t *seg_cur = seg_a.as<uint32_t>(), *seg_nxt = seg_b.as<uint32_t>();
        ANode *act_cur = act_a.as<ANode>(), *act_nxt = act_b.as<ANode>();
        hipLaunchKernelGGL(k_iota, dim3(grid_n), dim3(kB), 0, stream, ord_cur, n);
        uint32_t n_active = 0, n_jobs = 0;
        {
            const uint32_t zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            BHIP(hipMemcpyAsync(scp, zeros, sizeof zeros, hipMemcpyHostToDevice, stream));
            if (n > kSmall) {
                const ANode root = {0u, n, kNone, 0u};
                BHIP(hipMemcpyAsync(act_cur, &root, sizeof root, hipMemcpyHostToDevice, stream));
                BHIP(hipMemsetAsync(seg_cur, 0, sizeof(uint32_t) * (size_t)n, stream));
                n_active = 1;
            } else {
                const Job root = {0u, n, kNone, 0u, 0u, {0u, 0u, 0u}};
                BHIP(hipMemcpyAsync(jobs.p, &root, sizeof root, hipMemcpyHostToDevice, stream));
                const uint32_t one = 1u;
                BHIP(hipMemcpyAsync(scp + SC_JOBS, &one, sizeof one, hipMemcpyHostToDevice, stream));
            }
            BHIP(hipStreamSynchronize(stream));  // (locals)
        }
        for (uint32_t level = 0; n_active; level++) {
            if (level > 4096) {
                snprintf(err, err_len, "BVH build: the top-down build did not end (%u nodes at level %u)", n_active, level);
                return RTHIP_E_UNSUPPORTED;
            }
            const uint32_t words = n_active * kBinWords;
            // positions per wave: 512 while the nodes are large, down to 64 so that the chunks of the deep levels still lie inside one node
            uint32_t wchunk = 512;
            while (wchunk > 64 && (unsigned long long)wchunk * 8ull * n_active > n) wchunk >>= 1;
            const uint32_t grid_c = (uint32_t)((((unsigned long long)n + wchunk - 1) / wchunk + kB / 64 - 1) / (kB / 64));
            hipLaunchKernelGGL(k_sah_clear, dim3((words + kB - 1) / kB), dim3(kB), 0, stream, nb.as<uint32_t>(), bins.as<uint32_t>(), n_active);
            hipLaunchKernelGGL(k_sah_bounds, dim3(grid_c), dim3(kB), 0, stream, ord_cur, seg_cur, n, wchunk, prim_box.as<Box>(), centroid.as<float>(), nb.as<uint32_t>());
            hipLaunchKernelGGL(k_sah_bin, dim3(grid_c), dim3(kB), 0, stream, ord_cur, seg_cur, n, wchunk, prim_box.as<Box>(), centroid.as<float>(), nb.as<uint32_t>(),
                               bins.as<uint32_t>());
            hipLaunchKernelGGL(k_sah_split, dim3((n_active + kB - 1) / kB), dim3(kB), 0, stream, act_cur, n_active, level, nb.as<uint32_t>(), bins.as<uint32_t>(),
                               bin.as<PNode>(), node_box.as<Box>(), node_depth.as<uint32_t>(), scp, split.as<SplitRec>(), cursors.as<uint32_t>(), act_nxt,
                               jobs.as<Job>());
            hipLaunchKernelGGL(k_sah_partition, dim3(grid_c), dim3(kB), 0, stream, ord_cur, seg_cur, n, wchunk, centroid.as<float>(), nb.as<uint32_t>(),
                               split.as<SplitRec>(), cursors.as<uint32_t>(), ord_nxt, seg_nxt);
            uint32_t next_active = 0;
            BHIP(hipMemcpyAsync(&next_active, scp + SC_NEXT, sizeof next_active, hipMemcpyDeviceToHost, stream));
            BHIP(hipMemsetAsync(scp + SC_NEXT, 0, sizeof(uint32_t), stream));
            BHIP(hipStreamSynchronize(stream));
            if (next_active > a_max) {
                snprintf(err, err_len, "BVH build: %u nodes at level %u, room for %u", next_active, level + 1, a_max);
                return RTHIP_E_HIP;
            }
            n_active = next_active;
            std::swap(ord_cur, ord_nxt);
            std::swap(seg_cur, seg_nxt);
            std::swap(act_cur, act_nxt);
        }
        BHIP(hipMemcpyAsync(&n_jobs, scp + SC_JOBS, sizeof n_jobs, hipMemcpyDeviceToHost, stream));
        BHIP(hipStreamSynchronize(stream));
        if (n_jobs > j_max) {
            snprintf(err, err_len, "BVH build: %u small subtrees, room for %u", n_jobs, j_max);
            return RTHIP_E_HIP;
        }
        if (n_jobs)
            hipLaunchKernelGGL(k_sah_small, dim3(n_jobs), dim3(64), 0, stream, jobs.as<Job>(), n_jobs, prim_box.as<Box>(), centroid.as<float>(), ord_cur,
                               bin.as<PNode>(), node_box.as<Box>(), node_depth.as<uint32_t>(), scp);
        if (ord_cur != idx_out.as<uint32_t>())
            BHIP(hipMemcpyAsync(idx_out.p, ord_cur, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        index = idx_out.as<uint32_t>();  // position -> triangle
        uint32_t sc_h[8];
        BHIP(hipMemcpyAsync(sc_h, scp, sizeof sc_h, hipMemcpyDeviceToHost, stream));
        BHIP(hipStreamSynchronize(stream));
        if (sc_h[SC_NODES] != n_int) {
            snprintf(err, err_len, "BVH build: %u binary nodes made, %u expected", sc_h[SC_NODES], n_int);
            return RTHIP_E_HIP;
        }
        root_ref = sc_h[SC_ROOT];
        {
            BHIP(hipMemsetAsync(scp + 8, 0, sizeof(uint32_t) * 8, stream));
            if (root_ref < n_int)
                hipLaunchKernelGGL(k_sah_validate, dim3((n_int + kB - 1) / kB), dim3(kB), 0, stream, bin.as<PNode>(), node_box.as<Box>(), node_depth.as<uint32_t>(),
                                   n_int, n, prim_box.as<Box>(), index, root_ref, scp + 8);
            uint32_t bad[8];
            BHIP(hipMemcpyAsync(bad, scp + 8, sizeof bad, hipMemcpyDeviceToHost, stream));
            BHIP(hipStreamSynchronize(stream));
            if (root_ref >= n_int || bad[0] || bad[1] || bad[2] || bad[3] || bad[4] || bad[5]) {
                snprintf(err, err_len, "BVH build: the binary tree failed its check (root %u of %u; bad left %u right %u size %u depth %u box %u finite %u)",
                         root_ref, n_int, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
                return RTHIP_E_UNSUPPORTED;  // (the caller may build this BVH on the host instead)
            }
        }
        for (uint32_t d = sc_h[SC_DEPTH] + 1u; d-- > 0;)  // the collapse plan, deepest nodes first
            hipLaunchKernelGGL(k_plan_depth, dim3((n_int + kB - 1) / kB), dim3(kB), 0, stream, n_int, d, node_depth.as<uint32_t>(), bin.as<PNode>(),
                               node_box.as<Box>(), prim_box.as<Box>(), index, plan.as<Plan>());
        BHIP(hipStreamSynchronize(stream));  // (node_depth and the other temporaries go out of scope)
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
