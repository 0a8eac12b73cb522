// rthip_kd.hip — the REFERENCE kd-tree of a mesh built ON THE DEVICE ("GPU-side KD build, same splits";
// replaces the recursive by-value build of KDTreeTriangle's constructor, Raytracer/Scene.cpp:18-95, called from Mesh::Compile,
// Scene.cpp:386-414).  The tree takes part in results (accel = 1 walks it; kd-degenerate rays and exact distance ties are
// settled in its order), so every decision is the reference's, bit for bit:
//   * leaf if n <= 1 or depth == 100 (Scene.cpp:25); axis = depth % 3;
//   * split = the mean of all 3n vertex coordinates on that axis, SUMMED SEQUENTIALLY in triangle-list order and divided by
//     3n (Scene.cpp:36-47).  A parallel reduction rounds differently, so the sum of a node is ONE dependent chain of fp64 adds
//     here as well - but the chains of all the nodes of a level run side by side (a wave per node: its 64 lanes fetch the next
//     coordinates into LDS while the chain consumes the previous ones);
//   * child boxes = the node's box cut at the split (BoundingBox::Split, Common.cpp:313-338); Left takes the triangles with
//     any vertex <= split, Right those with any vertex >= split, both in list order (Scene.cpp:60-77): a stable partition,
//     done with one prefix sum over the level (rocPRIM) of (goes left, goes right) pairs;
//   * if |L| + |R| > 1.5 n the node stays a leaf holding all n (Scene.cpp:79-83); children exist only if non-empty.
// The build is level-synchronous (all nodes of a depth in the same launches, one small read-back per level for the sizes of
// the next one); the finished tree is renumbered in PRE-ORDER (left subtree at me + 1, then the right one: sizes bottom-up,
// indices top-down) and emitted as the arrays of include/rthip.h: kd_nodes / kd_bounds / kd_refs, and the sorted unique plane
// coordinates per axis (kd_planes) the flattener derives from them.  Byte-identical to rtbuild::KdBuilder + flatten.cpp
// (tests/test_device_kd.py).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "rthip.h"
#include "rthip_kd.hpp"

namespace {
const uint32_t kB = 256;
const uint32_t kNone = 0xffffffffu;
const uint32_t kChunk = 512;  // triangles a wave stages in LDS per round of the sequential sum (12 KB)
enum : uint32_t { F_LEAF = 1u, F_RIGHT = 2u };

// one node of the tree while it is built (level order: the nodes of a depth are contiguous)
struct KNode {
    double lo[3], hi[3];      // KDTreeTriangle::Bounds
    double split;             // the mean (kept for leaves made by the 1.5 n rule too; emitted as 0 for leaves like the host builder's)
    unsigned long long s0;    // prefix-sum value in front of the node's list (interior nodes)
    uint32_t start, n;        // the node's list: a range of the level's triangle-id array
    uint32_t parent;
    uint32_t left, right;     // global ids of the children, kNone if absent (between k_kd_decide and k_kd_children: |L|, |R|)
    uint32_t child_start[2];  // where the children's lists begin in the next level's array
    uint32_t flags, depth;
    uint32_t pool_off;        // leaves: first entry in the leaf pool
    uint32_t sub_nodes, sub_refs;  // subtree totals
    uint32_t pre, ref_base;        // pre-order index, first leaf reference of the subtree
    // the exact-sum shortcut (k_kd_grid_a / _b): smallest exponent of a set bit among the node's coordinates on its axis, the
    // largest |coordinate| (as bits), and the integer sum of coordinate / 2^emin
    int32_t emin;
    unsigned long long amax_bits;
    unsigned long long isum;
};

__global__ __launch_bounds__(kB) void k_kd_iota(uint32_t *list, uint32_t *owner, uint32_t n) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i < n) { list[i] = i; owner[i] = 0u; }
}

// The exact-sum shortcut.  The reference's mean is a SEQUENTIAL fp64 sum, and a sequential sum only differs from the exact sum
// through its roundings: if every coordinate of a node is a multiple of 2^e and 3n * max|coordinate| < 2^(53 + e), every partial
// sum (a multiple of 2^e below 2^(53 + e)) is representable, no addition rounds, and the sequential sum IS the exact sum -
// whatever the order.  Coordinates on a grid (a height field's x and z, anything modelled in 1/64ths or millimetres) qualify;
// then the sum is an integer reduction over the whole level instead of one dependent chain per node.  Pass a: per node the
// smallest set-bit exponent `emin` and the largest magnitude; pass b (nodes that qualify): the integer sum of coordinate / 2^emin.
// Nodes that do not qualify (arbitrary doubles) take the chain in k_kd_median.
const int32_t kEminNone = 0x7fffffff;
__device__ __forceinline__ bool kd_exact_ok(uint32_t n, int32_t emin, unsigned long long amax_bits) {
    if (emin == kEminNone) return true;                      // every coordinate is zero: the sum is +0.0
    if ((amax_bits >> 52) >= 0x7ffull) return false;         // an infinity or a NaN takes the chain (and its IEEE rules)
    const int32_t ea = (int32_t)(amax_bits >> 52) ? (int32_t)(amax_bits >> 52) - 1023 : -1022;  // |x| < 2^(ea + 1)
    const int32_t bits_n = 64 - __clzll(3ull * (unsigned long long)n);                        // 3n < 2^bits_n
    return bits_n + ea + 1 <= 53 + emin;
}
__device__ __forceinline__ int32_t kd_low_exponent(double x) {  // exponent of the lowest set bit of x (x finite, non-zero)
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const uint32_t e = (uint32_t)(b >> 52) & 0x7ffu;
    const unsigned long long mant = (b & 0xfffffffffffffull) | (e ? (1ull << 52) : 0ull);
    return (int32_t)(e ? e : 1u) - 1075 + (int32_t)(__ffsll((long long)mant) - 1);
}
// A wave takes kGridChunk consecutive entries of the level's list.  The entries of a node are contiguous and the owner ids grow
// along the list, so the chunk lies inside ONE node iff its first and last entries have the same owner: then the lanes combine
// and the wave sends one set of atomics (a level of 2 M entries in two nodes would otherwise send millions to two addresses).
const uint32_t kGridChunk = 512;
__global__ __launch_bounds__(kB) void k_kd_grid_a(uint32_t m, const uint32_t *list, const uint32_t *owner, KNode *nodes, const double *verts) {
    const uint32_t lane = threadIdx.x & 63u, base = (blockIdx.x * (kB / 64) + (threadIdx.x >> 6)) * kGridChunk;
    if (base >= m) return;
    const uint32_t last = base + kGridChunk - 1u < m ? base + kGridChunk - 1u : m - 1u;
    const bool uniform = owner[base] == owner[last];
    int32_t emin = kEminNone;
    unsigned long long amax = 0ull;
    for (uint32_t e = base + lane; e <= last; e += 64u) {
        const uint32_t id = owner[e];
        const KNode &nd = nodes[id];
        if (nd.n <= 1u || nd.depth == 100u) continue;
        const double *v = verts + (size_t)list[e] * 9u + nd.depth % 3u;
        int32_t le_min = kEminNone;
        unsigned long long a_max = 0ull;
        for (int k = 0; k < 3; k++) {
            const double x = v[3 * k];
            const unsigned long long a = (unsigned long long)__double_as_longlong(x) & 0x7fffffffffffffffull;
            a_max = a > a_max ? a : a_max;
            if (a != 0ull && (a >> 52) < 0x7ffull) { const int32_t le = kd_low_exponent(x); le_min = le < le_min ? le : le_min; }
        }
        if (uniform) {
            emin = le_min < emin ? le_min : emin;
            amax = a_max > amax ? a_max : amax;
        } else {
            if (le_min != kEminNone) atomicMin(&nodes[id].emin, le_min);
            if (a_max) atomicMax(&nodes[id].amax_bits, a_max);
        }
    }
    if (!uniform) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t oe = __shfl_xor(emin, off, 64);
        const unsigned long long oa = (unsigned long long)__shfl_xor((long long)amax, off, 64);
        emin = oe < emin ? oe : emin;
        amax = oa > amax ? oa : amax;
    }
    if (lane == 0) {
        const uint32_t id = owner[base];
        if (emin != kEminNone) atomicMin(&nodes[id].emin, emin);
        if (amax) atomicMax(&nodes[id].amax_bits, amax);
    }
}
__global__ __launch_bounds__(kB) void k_kd_grid_b(uint32_t m, const uint32_t *list, const uint32_t *owner, KNode *nodes, const double *verts) {
    const uint32_t lane = threadIdx.x & 63u, base = (blockIdx.x * (kB / 64) + (threadIdx.x >> 6)) * kGridChunk;
    if (base >= m) return;
    const uint32_t last = base + kGridChunk - 1u < m ? base + kGridChunk - 1u : m - 1u;
    const bool uniform = owner[base] == owner[last];
    long long total = 0;
    for (uint32_t e = base + lane; e <= last; e += 64u) {
        const uint32_t id = owner[e];
        const KNode &nd = nodes[id];
        if (nd.n <= 1u || nd.depth == 100u || nd.emin == kEminNone || !kd_exact_ok(nd.n, nd.emin, nd.amax_bits)) continue;
        const double *v = verts + (size_t)list[e] * 9u + nd.depth % 3u;
        long long part = 0;
        for (int k = 0; k < 3; k++) part += (long long)ldexp(v[3 * k], -nd.emin);  // an integer below 2^53: exact
        if (uniform) total += part;
        else if (part) atomicAdd(&nodes[id].isum, (unsigned long long)part);
    }
    if (!uniform) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    if (lane == 0 && total) atomicAdd(&nodes[owner[base]].isum, (unsigned long long)total);
}

// Scene.cpp:25 (stop condition) and :36-47 (the mean): one wave per node of the level
__global__ __launch_bounds__(64) void k_kd_median(KNode *nodes, uint32_t first, const double *verts, const uint32_t *list, int use_exact) {
    __shared__ __attribute__((aligned(16))) double buf[kChunk * 3 + 48];
    KNode &nd = nodes[first + blockIdx.x];
    const uint32_t lane = threadIdx.x, n = nd.n, depth = nd.depth;
    if (n <= 1u || depth == 100u) {
        if (lane == 0) nd.flags |= F_LEAF;
        return;
    }
    if (use_exact && kd_exact_ok(n, nd.emin, nd.amax_bits)) {  // no addition of the sequential sum rounds: it is the integer sum
        const double sum = nd.emin == kEminNone ? 0.0 : ldexp((double)(long long)nd.isum, nd.emin);
        if (lane == 0) nd.split = sum / (double)(3ull * (unsigned long long)n);
        return;
    }
    const uint32_t axis = depth % 3u;
    const uint32_t *ids = list + nd.start;
    constexpr uint32_t kU = kChunk / 64;
    double r[kU][3];
    auto fetch = [&](uint32_t c0) {  // the chunk's coordinates -> registers (in flight while the previous chunk is summed)
#pragma unroll
        for (uint32_t u = 0; u < kU; u++) {
            const uint32_t j = c0 + u * 64u + lane;
            if (j < n) {
                const double *v = verts + (size_t)ids[j] * 9u + axis;
                r[u][0] = v[0]; r[u][1] = v[3]; r[u][2] = v[6];
            } else {
                r[u][0] = r[u][1] = r[u][2] = 0.0;
            }
        }
    };
    double sum = 0.0;
    fetch(0);
    for (uint32_t c0 = 0; c0 < n; c0 += kChunk) {
#pragma unroll
        for (uint32_t u = 0; u < kU; u++) {
            double *d = buf + (size_t)(u * 64u + lane) * 3u;
            d[0] = r[u][0]; d[1] = r[u][1]; d[2] = r[u][2];
        }
        __syncthreads();
        if (c0 + kChunk < n) fetch(c0 + kChunk);
        const uint32_t cnt3 = (n - c0 < kChunk ? n - c0 : kChunk) * 3u;
        // The chain: every lane runs it on the same (broadcast) LDS words; vertices 0, 1, 2 of a triangle, triangles in list
        // order.  Blocks of 16 words, the next block read from LDS while the current one is added (the buffer is padded: the
        // read ahead of the last block stays inside it), so that the chain waits for its own additions only.
        constexpr uint32_t kW = 16;
        double a[kW], b[kW];  // two register blocks used in turn (copying one to the other cost as much as the additions)
#pragma unroll
        for (uint32_t k = 0; k < kW; k++) a[k] = buf[k];
        uint32_t j = 0;
        for (; j + 2u * kW <= cnt3; j += 2u * kW) {
#pragma unroll
            for (uint32_t k = 0; k < kW; k++) b[k] = buf[j + kW + k];
#pragma unroll
            for (uint32_t k = 0; k < kW; k++) sum += a[k];
#pragma unroll
            for (uint32_t k = 0; k < kW; k++) a[k] = buf[j + 2u * kW + k];
#pragma unroll
            for (uint32_t k = 0; k < kW; k++) sum += b[k];
        }
        // at most 2 kW - 1 words left: a[] holds buf[j .. j + kW)
#pragma unroll
        for (uint32_t k = 0; k < kW; k++)
            if (j + k < cnt3) sum += a[k];
        if (j + kW < cnt3) {
#pragma unroll
            for (uint32_t k = 0; k < kW; k++) b[k] = buf[j + kW + k];
#pragma unroll
            for (uint32_t k = 0; k < kW; k++)
                if (j + kW + k < cnt3) sum += b[k];
        }
        __syncthreads();
    }
    const double median = sum / (double)(3ull * (unsigned long long)n);  // `median /= 3 * n` (size_t -> double)
    if (lane == 0) nd.split = median;
}

// Scene.cpp:60-77: which side(s) every triangle of the level goes to, as (left, right) packed in 64 bits for ONE prefix sum
__global__ __launch_bounds__(kB) void k_kd_classify(uint32_t m, const uint32_t *list, const uint32_t *owner, const KNode *nodes, const double *verts,
                                                    unsigned long long *key) {
    const uint32_t e = blockIdx.x * kB + threadIdx.x;
    if (e >= m) return;
    const KNode &nd = nodes[owner[e]];
    if (nd.flags & F_LEAF) { key[e] = 0ull; return; }
    const double med = nd.split;
    const double *v = verts + (size_t)list[e] * 9u + nd.depth % 3u;
    const double a = v[0], b = v[3], c = v[6];
    const unsigned long long l = (a <= med || b <= med || c <= med) ? 1ull : 0ull;
    const unsigned long long r = (a >= med || b >= med || c >= med) ? 1ull : 0ull;
    key[e] = l | (r << 32);
}

// Scene.cpp:79-83 (the 1.5 n rule) and the sizes of the next level.  pack = next-level list entries | next-level nodes << 32.
__global__ __launch_bounds__(kB) void k_kd_decide(KNode *nodes, uint32_t first, uint32_t nn, const unsigned long long *S, unsigned long long *pack,
                                                  uint32_t *leaf_refs, uint32_t *max_leaf) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= nn) return;
    KNode &nd = nodes[first + i];
    const uint32_t n = nd.n;
    if (!(nd.flags & F_LEAF)) {
        const unsigned long long s0 = nd.start ? S[nd.start - 1u] : 0ull, s1 = S[nd.start + n - 1u];
        const uint32_t lc = (uint32_t)(s1 & 0xffffffffull) - (uint32_t)(s0 & 0xffffffffull), rc = (uint32_t)(s1 >> 32) - (uint32_t)(s0 >> 32);
        if ((double)((unsigned long long)lc + rc) > 1.5 * (double)(unsigned long long)n) {
            nd.flags |= F_LEAF;
        } else {
            nd.s0 = s0;
            nd.left = lc;
            nd.right = rc;
            pack[i] = (unsigned long long)(lc + rc) | ((unsigned long long)((lc ? 1u : 0u) + (rc ? 1u : 0u)) << 32);
            leaf_refs[i] = 0u;
            return;
        }
    }
    pack[i] = 0ull;
    leaf_refs[i] = n;
    atomicMax(max_leaf, n);
}

// the children (Scene.cpp:85-93; boxes: BoundingBox::Split) and the leaves' places in the pool
__global__ __launch_bounds__(kB) void k_kd_children(KNode *nodes, uint32_t first, uint32_t nn, uint32_t next_first, const unsigned long long *packS,
                                                    const unsigned long long *pack, const uint32_t *leafS, const uint32_t *leaf_refs, uint32_t pool_base) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= nn) return;
    const uint32_t id = first + i;
    KNode nd = nodes[id];
    if (nd.flags & F_LEAF) {
        nd.pool_off = pool_base + (leafS[i] - leaf_refs[i]);
        nd.left = nd.right = kNone;
    } else {
        const unsigned long long off = packS[i] - pack[i];
        const uint32_t lc = nd.left, rc = nd.right, axis = nd.depth % 3u;
        uint32_t eo = (uint32_t)(off & 0xffffffffull), child = next_first + (uint32_t)(off >> 32);
        nd.child_start[0] = eo;
        nd.child_start[1] = eo + lc;
        nd.left = nd.right = kNone;
        for (uint32_t side = 0; side < 2; side++) {
            const uint32_t cn = side ? rc : lc;
            if (!cn) continue;
            KNode ch;
            memset(&ch, 0, sizeof ch);
            for (int a = 0; a < 3; a++) { ch.lo[a] = nd.lo[a]; ch.hi[a] = nd.hi[a]; }
            if (side) ch.lo[axis] = nd.split; else ch.hi[axis] = nd.split;
            ch.start = nd.child_start[side];
            ch.n = cn;
            ch.parent = id;
            ch.left = ch.right = kNone;
            ch.flags = side ? F_RIGHT : 0u;
            ch.depth = nd.depth + 1u;
            ch.emin = kEminNone;
            nodes[child] = ch;
            (side ? nd.right : nd.left) = child;
            child++;
        }
    }
    nodes[id] = nd;
}

// the stable partition: position in the child's list = rank among the node's triangles that go to that side
__global__ __launch_bounds__(kB) void k_kd_scatter(uint32_t m, const uint32_t *list, const uint32_t *owner, const unsigned long long *key,
                                                   const unsigned long long *S, const KNode *nodes, uint32_t *next, uint32_t *owner_next, uint32_t *pool,
                                                   uint32_t *pool_owner) {
    const uint32_t e = blockIdx.x * kB + threadIdx.x;
    if (e >= m) return;
    const uint32_t id = owner[e], ti = list[e];
    const KNode &nd = nodes[id];
    if (nd.flags & F_LEAF) {
        const uint32_t p = nd.pool_off + (e - nd.start);
        pool[p] = ti;
        pool_owner[p] = id;
        return;
    }
    const unsigned long long k = key[e], s = S[e], s0 = nd.s0;
    if (k & 1ull) {
        const uint32_t pos = nd.child_start[0] + ((uint32_t)(s & 0xffffffffull) - (uint32_t)(s0 & 0xffffffffull) - 1u);
        next[pos] = ti;
        owner_next[pos] = nd.left;
    }
    if (k >> 32) {
        const uint32_t pos = nd.child_start[1] + ((uint32_t)(s >> 32) - (uint32_t)(s0 >> 32) - 1u);
        next[pos] = ti;
        owner_next[pos] = nd.right;
    }
}

// pre-order renumbering: subtree sizes bottom-up ...
__global__ __launch_bounds__(kB) void k_kd_up(KNode *nodes, uint32_t first, uint32_t nn) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= nn) return;
    KNode &nd = nodes[first + i];
    uint32_t sn = 1u, sr = (nd.flags & F_LEAF) ? nd.n : 0u;
    if (nd.left != kNone) { sn += nodes[nd.left].sub_nodes; sr += nodes[nd.left].sub_refs; }
    if (nd.right != kNone) { sn += nodes[nd.right].sub_nodes; sr += nodes[nd.right].sub_refs; }
    nd.sub_nodes = sn;
    nd.sub_refs = sr;
}
// ... indices top-down (the left subtree lands at me + 1, the right one behind it: KdBuilder::Build), and the node's record
__global__ __launch_bounds__(kB) void k_kd_down(KNode *nodes, uint32_t first, uint32_t nn, rthip_kdnode *out_nodes, double *out_bounds) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= nn) return;
    const KNode &nd = nodes[first + i];
    uint32_t after_left = nd.pre + 1u, refs_after_left = nd.ref_base;
    if (nd.left != kNone) {
        KNode &l = nodes[nd.left];
        l.pre = nd.pre + 1u;
        l.ref_base = nd.ref_base;
        after_left += l.sub_nodes;
        refs_after_left += l.sub_refs;
    }
    uint32_t link = 0u;
    if (nd.right != kNone) {
        KNode &r = nodes[nd.right];
        r.pre = after_left;
        r.ref_base = refs_after_left;
        link = after_left;
    }
    rthip_kdnode o;
    if (nd.flags & F_LEAF) {
        o.split = 0.0;
        o.link = nd.ref_base;
        o.info = RTHIP_KD_LEAF | (nd.n << RTHIP_KD_COUNT_SHIFT);
    } else {
        o.split = nd.split;
        o.link = link;
        o.info = (nd.depth % 3u) | (nd.left != kNone ? RTHIP_KD_HAS_LEFT : 0u) | (nd.right != kNone ? RTHIP_KD_HAS_RIGHT : 0u);
    }
    out_nodes[nd.pre] = o;
    double *b = out_bounds + (size_t)nd.pre * 6u;
    b[0] = nd.lo[0]; b[1] = nd.lo[1]; b[2] = nd.lo[2];
    b[3] = nd.hi[0]; b[4] = nd.hi[1]; b[5] = nd.hi[2];
}
__global__ __launch_bounds__(kB) void k_kd_refs(uint32_t total, const uint32_t *pool, const uint32_t *pool_owner, const KNode *nodes, uint32_t *out_refs) {
    const uint32_t p = blockIdx.x * kB + threadIdx.x;
    if (p >= total) return;
    const KNode &nd = nodes[pool_owner[p]];
    out_refs[nd.ref_base + (p - nd.pool_off)] = pool[p];
}

// kd_planes (flatten.cpp): per axis the root bounds and the split of every interior node of that axis; +inf where a node has none
__global__ __launch_bounds__(kB) void k_kd_plane_keys(const KNode *nodes, uint32_t total, double *keys, uint32_t stride, uint32_t *counts) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    const KNode *nd = i < total ? nodes + i : nullptr;
    for (uint32_t a = 0; a < 3; a++) {
        const bool has = nd && !(nd->flags & F_LEAF) && nd->depth % 3u == a;
        if (nd) keys[(size_t)a * stride + 2u + i] = has ? nd->split : INFINITY;
        const unsigned long long mask = __ballot(has);
        if ((threadIdx.x & 63u) == 0 && mask) atomicAdd(&counts[a], (uint32_t)__popcll(mask));
        if (i == 0) {
            keys[(size_t)a * stride] = nd->lo[a];
            keys[(size_t)a * stride + 1u] = nd->hi[a];
        }
    }
}
__global__ __launch_bounds__(kB) void k_kd_uniq_flag(const double *sorted, uint32_t n, uint32_t *flag) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || sorted[i] != sorted[i - 1u]) ? 1u : 0u;  // std::unique's ==
}
__global__ __launch_bounds__(kB) void k_kd_uniq_write(const double *sorted, uint32_t n, const uint32_t *flag, const uint32_t *pos, double *out) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i < n && flag[i]) out[pos[i] - 1u] = sorted[i];
}

struct Buf {
    void *p = nullptr;
    size_t bytes = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    template <class T> T *as() { return (T *)p; }
    // at least `need` bytes; the first `keep` bytes survive a reallocation
    hipError_t reserve(size_t need, size_t keep, hipStream_t stream) {
        if (need <= bytes && p) return hipSuccess;
        size_t nb = need > bytes + bytes / 2 ? need : bytes + bytes / 2;
        if (nb < 256) nb = 256;
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, nb);
        if (e != hipSuccess) return e;
        if (p && keep) e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (p) (void)hipFree(p);
        p = q;
        bytes = nb;
        return e;
    }
};
}  // namespace

#define KHIP(expr)                                                             \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            snprintf(err, err_len, "kd build: %s failed: %s", #expr, hipGetErrorString(e_)); \
            return RTHIP_E_HIP;                                                \
        }                                                                      \
    } while (0)

static int build(hipStream_t stream, const double *d_verts, uint32_t n, const double root_bounds[6], rthip_kdnode *&out_nodes, double *&out_bounds,
                 uint32_t *&out_refs, double *&out_planes, rthip_device_kd_info *info, char *err, size_t err_len) {
    auto grid = [](size_t items) { return dim3((uint32_t)((items + kB - 1) / kB)); };
    Buf nodes, list_a, list_b, owner_a, owner_b, key, S, pack, packS, leaf_refs, leafS, pool, pool_owner, scan_tmp, misc;
    KHIP(misc.reserve(256, 0, stream));
    KHIP(hipMemsetAsync(misc.p, 0, 256, stream));
    uint32_t *max_leaf = misc.as<uint32_t>(), *plane_counts = misc.as<uint32_t>() + 4;
    size_t node_cap = 1024;
    KHIP(nodes.reserve(sizeof(KNode) * node_cap, 0, stream));
    KHIP(list_a.reserve(sizeof(uint32_t) * (size_t)(n ? n : 1), 0, stream));
    KHIP(owner_a.reserve(sizeof(uint32_t) * (size_t)(n ? n : 1), 0, stream));
    {
        KNode root;
        memset(&root, 0, sizeof root);
        for (int a = 0; a < 3; a++) { root.lo[a] = root_bounds[a]; root.hi[a] = root_bounds[3 + a]; }
        root.n = n;
        root.parent = root.left = root.right = kNone;
        root.emin = kEminNone;
        KHIP(hipMemcpyAsync(nodes.p, &root, sizeof root, hipMemcpyHostToDevice, stream));
        KHIP(hipStreamSynchronize(stream));  // (`root` is a local)
    }
    if (n) hipLaunchKernelGGL(k_kd_iota, grid(n), dim3(kB), 0, stream, list_a.as<uint32_t>(), owner_a.as<uint32_t>(), n);
    Buf *cur = &list_a, *nxt = &list_b, *own = &owner_a, *own_nxt = &owner_b;
    int use_exact = 1;  // RTHIP_KD_EXACT_SUMS=0: every node takes the sequential chain (tests: both routes give the same tree)
    if (const char *e = getenv("RTHIP_KD_EXACT_SUMS")) use_exact = atoi(e) != 0;
    std::vector<uint32_t> level_first = {0u};
    uint32_t total_nodes = 1, first = 0, nn = 1, m = n, pool_total = 0;
    auto scan_u64 = [&](const unsigned long long *in, unsigned long long *out, size_t count) -> hipError_t {
        size_t bytes = 0;
        hipError_t e = rocprim::inclusive_scan(nullptr, bytes, in, out, count, rocprim::plus<unsigned long long>(), stream);
        if (e == hipSuccess) e = scan_tmp.reserve(bytes, 0, stream);
        if (e == hipSuccess) e = rocprim::inclusive_scan(scan_tmp.p, bytes, in, out, count, rocprim::plus<unsigned long long>(), stream);
        return e;
    };
    auto scan_u32 = [&](const uint32_t *in, uint32_t *out, size_t count) -> hipError_t {
        size_t bytes = 0;
        hipError_t e = rocprim::inclusive_scan(nullptr, bytes, in, out, count, rocprim::plus<uint32_t>(), stream);
        if (e == hipSuccess) e = scan_tmp.reserve(bytes, 0, stream);
        if (e == hipSuccess) e = rocprim::inclusive_scan(scan_tmp.p, bytes, in, out, count, rocprim::plus<uint32_t>(), stream);
        return e;
    };
    while (nn) {
        if (level_first.size() > 102) {  // (depth == 100 is a leaf by construction: at most 101 levels)
            snprintf(err, err_len, "kd build: more than 101 levels");
            return RTHIP_E_HIP;
        }
        if (use_exact && m) {
            const dim3 gchunks((uint32_t)(((size_t)m + kGridChunk * (kB / 64) - 1) / (kGridChunk * (kB / 64))));
            hipLaunchKernelGGL(k_kd_grid_a, gchunks, dim3(kB), 0, stream, m, cur->as<uint32_t>(), own->as<uint32_t>(), nodes.as<KNode>(), d_verts);
            hipLaunchKernelGGL(k_kd_grid_b, gchunks, dim3(kB), 0, stream, m, cur->as<uint32_t>(), own->as<uint32_t>(), nodes.as<KNode>(), d_verts);
        }
        hipLaunchKernelGGL(k_kd_median, dim3(nn), dim3(64), 0, stream, nodes.as<KNode>(), first, d_verts, cur->as<uint32_t>(), use_exact);
        if (m) {
            KHIP(key.reserve(sizeof(unsigned long long) * (size_t)m, 0, stream));
            KHIP(S.reserve(sizeof(unsigned long long) * (size_t)m, 0, stream));
            hipLaunchKernelGGL(k_kd_classify, grid(m), dim3(kB), 0, stream, m, cur->as<uint32_t>(), own->as<uint32_t>(), nodes.as<KNode>(), d_verts,
                               key.as<unsigned long long>());
            KHIP(scan_u64(key.as<unsigned long long>(), S.as<unsigned long long>(), m));
        }
        KHIP(pack.reserve(sizeof(unsigned long long) * (size_t)nn, 0, stream));
        KHIP(packS.reserve(sizeof(unsigned long long) * (size_t)nn, 0, stream));
        KHIP(leaf_refs.reserve(sizeof(uint32_t) * (size_t)nn, 0, stream));
        KHIP(leafS.reserve(sizeof(uint32_t) * (size_t)nn, 0, stream));
        hipLaunchKernelGGL(k_kd_decide, grid(nn), dim3(kB), 0, stream, nodes.as<KNode>(), first, nn, S.as<unsigned long long>(), pack.as<unsigned long long>(),
                           leaf_refs.as<uint32_t>(), max_leaf);
        KHIP(scan_u64(pack.as<unsigned long long>(), packS.as<unsigned long long>(), nn));
        KHIP(scan_u32(leaf_refs.as<uint32_t>(), leafS.as<uint32_t>(), nn));
        unsigned long long pack_total = 0;
        uint32_t leaf_total = 0;
        KHIP(hipMemcpyAsync(&pack_total, packS.as<unsigned long long>() + (nn - 1), sizeof pack_total, hipMemcpyDeviceToHost, stream));
        KHIP(hipMemcpyAsync(&leaf_total, leafS.as<uint32_t>() + (nn - 1), sizeof leaf_total, hipMemcpyDeviceToHost, stream));
        KHIP(hipStreamSynchronize(stream));
        const uint32_t next_m = (uint32_t)(pack_total & 0xffffffffull), next_nn = (uint32_t)(pack_total >> 32);
        // (sums of 32-bit counts in 32 bits: refuse what could have wrapped)
        if ((unsigned long long)pool_total + leaf_total > 0xfffffff0ull || (unsigned long long)total_nodes + next_nn > 0xfffffff0ull ||
            (unsigned long long)next_m > 0xfffffff0ull || (double)next_m > 1.5 * (double)m + 1.0) {
            snprintf(err, err_len, "kd build: the tree exceeds 2^32 nodes / references");
            return RTHIP_E_UNSUPPORTED;
        }
        KHIP(nodes.reserve(sizeof(KNode) * ((size_t)total_nodes + next_nn), sizeof(KNode) * (size_t)total_nodes, stream));
        KHIP(pool.reserve(sizeof(uint32_t) * ((size_t)pool_total + leaf_total), sizeof(uint32_t) * (size_t)pool_total, stream));
        KHIP(pool_owner.reserve(sizeof(uint32_t) * ((size_t)pool_total + leaf_total), sizeof(uint32_t) * (size_t)pool_total, stream));
        KHIP(nxt->reserve(sizeof(uint32_t) * (size_t)next_m, 0, stream));
        KHIP(own_nxt->reserve(sizeof(uint32_t) * (size_t)next_m, 0, stream));
        hipLaunchKernelGGL(k_kd_children, grid(nn), dim3(kB), 0, stream, nodes.as<KNode>(), first, nn, total_nodes, packS.as<unsigned long long>(),
                           pack.as<unsigned long long>(), leafS.as<uint32_t>(), leaf_refs.as<uint32_t>(), pool_total);
        if (m)
            hipLaunchKernelGGL(k_kd_scatter, grid(m), dim3(kB), 0, stream, m, cur->as<uint32_t>(), own->as<uint32_t>(), key.as<unsigned long long>(),
                               S.as<unsigned long long>(), nodes.as<KNode>(), nxt->as<uint32_t>(), own_nxt->as<uint32_t>(), pool.as<uint32_t>(),
                               pool_owner.as<uint32_t>());
        pool_total += leaf_total;
        first = total_nodes;
        level_first.push_back(total_nodes);
        total_nodes += next_nn;
        nn = next_nn;
        m = next_m;
        std::swap(cur, nxt);
        std::swap(own, own_nxt);
    }
    // level_first[L] .. level_first[L + 1] = the nodes of depth L (the loop pushed the end of every level it finished)
    const uint32_t levels = (uint32_t)level_first.size() - 1u;
    uint32_t max_leaf_h = 0;
    KHIP(hipMemcpyAsync(&max_leaf_h, max_leaf, sizeof max_leaf_h, hipMemcpyDeviceToHost, stream));
    KHIP(hipStreamSynchronize(stream));
    if (max_leaf_h >= (1u << 24) - 1u) {
        snprintf(err, err_len, "kd leaf with more than 2^24 triangles");
        return RTHIP_E_UNSUPPORTED;
    }
    KHIP(hipMalloc((void **)&out_nodes, sizeof(rthip_kdnode) * (size_t)total_nodes));
    KHIP(hipMalloc((void **)&out_bounds, sizeof(double) * 6 * (size_t)total_nodes));
    KHIP(hipMalloc((void **)&out_refs, sizeof(uint32_t) * (size_t)(pool_total ? pool_total : 1)));
    for (uint32_t L = levels; L-- > 0;)
        hipLaunchKernelGGL(k_kd_up, grid(level_first[L + 1] - level_first[L]), dim3(kB), 0, stream, nodes.as<KNode>(), level_first[L], level_first[L + 1] - level_first[L]);
    for (uint32_t L = 0; L < levels; L++)
        hipLaunchKernelGGL(k_kd_down, grid(level_first[L + 1] - level_first[L]), dim3(kB), 0, stream, nodes.as<KNode>(), level_first[L], level_first[L + 1] - level_first[L],
                           out_nodes, out_bounds);
    if (pool_total)
        hipLaunchKernelGGL(k_kd_refs, grid(pool_total), dim3(kB), 0, stream, pool_total, pool.as<uint32_t>(), pool_owner.as<uint32_t>(), nodes.as<KNode>(), out_refs);
    // planes: per axis sort (+inf = no plane, sorts last), unique
    {
        const uint32_t stride = total_nodes + 2u;
        Buf keys, sorted, flag, pos, sort_tmp;
        KHIP(keys.reserve(sizeof(double) * 3 * (size_t)stride, 0, stream));
        KHIP(sorted.reserve(sizeof(double) * 3 * (size_t)stride, 0, stream));
        KHIP(flag.reserve(sizeof(uint32_t) * (size_t)stride, 0, stream));
        KHIP(pos.reserve(sizeof(uint32_t) * (size_t)stride, 0, stream));
        hipLaunchKernelGGL(k_kd_plane_keys, grid(total_nodes), dim3(kB), 0, stream, nodes.as<KNode>(), total_nodes, keys.as<double>(), stride, plane_counts);
        size_t sort_bytes = 0;
        KHIP(rocprim::radix_sort_keys(nullptr, sort_bytes, keys.as<double>(), sorted.as<double>(), (size_t)stride, 0u, 64u, stream));
        KHIP(sort_tmp.reserve(sort_bytes, 0, stream));
        for (uint32_t a = 0; a < 3; a++)
            KHIP(rocprim::radix_sort_keys(sort_tmp.p, sort_bytes, keys.as<double>() + (size_t)a * stride, sorted.as<double>() + (size_t)a * stride, (size_t)stride, 0u, 64u, stream));
        uint32_t counts[3];
        KHIP(hipMemcpyAsync(counts, plane_counts, sizeof counts, hipMemcpyDeviceToHost, stream));
        KHIP(hipStreamSynchronize(stream));
        // the unique runs of the three axes, one after the other: sizes first (a flag + prefix sum per axis), then the writes
        Buf uniq;  // worst case: nothing repeats
        KHIP(uniq.reserve(sizeof(double) * ((size_t)counts[0] + counts[1] + counts[2] + 6), 0, stream));
        uint32_t at = 0;
        for (uint32_t a = 0; a < 3; a++) {
            const uint32_t cnt = counts[a] + 2u;  // + the root's two planes
            const double *src = sorted.as<double>() + (size_t)a * stride;
            hipLaunchKernelGGL(k_kd_uniq_flag, grid(cnt), dim3(kB), 0, stream, src, cnt, flag.as<uint32_t>());
            KHIP(scan_u32(flag.as<uint32_t>(), pos.as<uint32_t>(), cnt));
            hipLaunchKernelGGL(k_kd_uniq_write, grid(cnt), dim3(kB), 0, stream, src, cnt, flag.as<uint32_t>(), pos.as<uint32_t>(), uniq.as<double>() + at);
            uint32_t kept = 0;
            KHIP(hipMemcpyAsync(&kept, pos.as<uint32_t>() + (cnt - 1u), sizeof kept, hipMemcpyDeviceToHost, stream));
            KHIP(hipStreamSynchronize(stream));
            info->plane_count[a] = kept;
            at += kept;
        }
        KHIP(hipMalloc((void **)&out_planes, sizeof(double) * (size_t)at));
        KHIP(hipMemcpyAsync(out_planes, uniq.p, sizeof(double) * (size_t)at, hipMemcpyDeviceToDevice, stream));
        KHIP(hipStreamSynchronize(stream));
    }
    KHIP(hipGetLastError());
    info->n_nodes = total_nodes;
    info->n_refs = pool_total;
    info->max_depth = levels - 1u;
    info->levels = levels;
    return 0;
}

int rthip_kd_build_device(hipStream_t stream, const double *d_verts, uint32_t n, const double root_bounds[6], rthip_kdnode **d_nodes, double **d_bounds,
                          uint32_t **d_refs, double **d_planes, rthip_device_kd_info *info, char *err, size_t err_len) {
    memset(info, 0, sizeof *info);
    *d_nodes = nullptr; *d_bounds = nullptr; *d_refs = nullptr; *d_planes = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, stream);
    rthip_kdnode *nodes = nullptr;
    double *bounds = nullptr, *planes = nullptr;
    uint32_t *refs = nullptr;
    int rc = build(stream, d_verts, n, root_bounds, nodes, bounds, refs, planes, info, err, err_len);
    if (rc == 0 && e0 && e1 && hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&info->device_ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) {
        (void)hipStreamSynchronize(stream);
        (void)hipGetLastError();
        if (nodes) (void)hipFree(nodes);
        if (bounds) (void)hipFree(bounds);
        if (refs) (void)hipFree(refs);
        if (planes) (void)hipFree(planes);
        return rc;
    }
    *d_nodes = nodes; *d_bounds = bounds; *d_refs = refs; *d_planes = planes;
    return 0;
}
