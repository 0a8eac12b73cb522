// mesh_build.hpp — acceleration-structure builders of one triangle mesh, independent of any scene class: they read the
// triangles through a view (`double V(tri, vertex, axis)`, `size()`) and write the flat arrays of include/rthip.h.
// Two callers compile this same code: the re:: mirror's Mesh::Compile (scene.cpp) and the binding written against the
// GENUINE reference headers (oracle/ref_binding.cpp, container-only), whose flattened scenes must be byte-identical.
//   KdBuilder  : the reference's kd-tree, same decisions as KDTreeTriangle's constructor (Scene.cpp:18-95)
//   BvhBuilder : the device-oriented 4-wide BVH over the same triangles (no reference counterpart)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <thread>
#include <vector>

#include "rthip.h"

namespace rtbuild {

// Levels of the tree whose two sub-trees are built on separate threads (2^levels tasks at most): enough for the
// host's hardware threads, at most 16 of them; RE_BUILD_THREADS=1 builds single-threaded.
inline int BuildForkDepth() {
    unsigned t = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("RE_BUILD_THREADS")) t = (unsigned)std::max(1, atoi(e));
    t = std::min(t ? t : 1u, 16u);
    int d = 0;
    while ((1u << d) < t) d++;
    return d;
}

// fn(begin, end) over [0, n) in contiguous slices, one per thread
template <typename F> inline void ParallelSlices(size_t n, F fn) {
    const size_t threads = std::min<size_t>((size_t)1 << BuildForkDepth(), n / 65536 + 1);
    if (threads <= 1) { fn((size_t)0, n); return; }
    std::vector<std::future<void>> jobs;
    for (size_t t = 0; t < threads; t++) {
        const size_t a = n * t / threads, b = n * (t + 1) / threads;
        jobs.push_back(std::async(std::launch::async, [=] { fn(a, b); }));
    }
    for (auto &j : jobs) j.get();
}

struct Box3 {
    double lo[3], hi[3];
};

// Flattened kd-tree of one mesh, built with the reference's splitting rules (Scene.cpp:18-95) but
// emitted directly as the compact pre-order arrays the device traverses (include/rthip.h).
struct KdTreeFlat {
    struct Node { double split; uint32_t link; uint32_t info; };
    std::vector<Node> nodes;
    std::vector<double> bounds;   // 6 per node
    std::vector<uint32_t> refs;   // leaf triangle indices, pre-order
    std::vector<uint32_t> depth;  // per node
    uint32_t max_depth = 0;
};

// Device-oriented BVH over the same triangles (no reference counterpart; see include/rthip.h).
struct BvhFlat {
    struct Child { float lo[3], hi[3]; uint32_t child, pad; };
    struct Node { Child c[4]; };  // == rthip_bvhnode
    std::vector<Node> nodes;
    std::vector<uint32_t> order;  // leaf-order position -> triangle index
    uint32_t depth = 0;
    float abs_max = 0.f;
};

// Builder: same decisions as KDTreeTriangle's constructor (Scene.cpp:18-95) on triangle INDEX lists
// (the reference copies 272-byte triangles by value; the order of each list is the same).
//
// Sub-trees are independent, so the two children of a big node are built on two threads, each into a flat tree of
// its own, and appended in pre-order afterwards (left at me + 1, then right) with their links rebased.  The result
// is the array a single-threaded build writes: every decision of a node depends on its own triangle list only.
template <class TriView> struct KdBuilder {
    const TriView &tris;
    KdTreeFlat &out;
    static constexpr size_t kTaskMin = 1u << 15;  // triangles below which a sub-tree is not worth a thread

    // appends the flat tree `sub` (indices local to it) to `out`; returns the index its root gets
    uint32_t Append(const KdTreeFlat &sub) {
        const uint32_t nb = (uint32_t)out.nodes.size(), rb = (uint32_t)out.refs.size();
        for (const auto &n : sub.nodes) {
            auto m = n;
            if (m.info & RTHIP_KD_LEAF) m.link += rb;
            else if (m.info & RTHIP_KD_HAS_RIGHT) m.link += nb;
            out.nodes.push_back(m);
        }
        out.depth.insert(out.depth.end(), sub.depth.begin(), sub.depth.end());
        out.bounds.insert(out.bounds.end(), sub.bounds.begin(), sub.bounds.end());
        out.refs.insert(out.refs.end(), sub.refs.begin(), sub.refs.end());
        out.max_depth = std::max(out.max_depth, sub.max_depth);
        return nb;
    }

    void Leaf(uint32_t me, const std::vector<uint32_t> &list) {
        out.nodes[me].link = (uint32_t)out.refs.size();
        out.nodes[me].info = RTHIP_KD_LEAF | ((uint32_t)list.size() << RTHIP_KD_COUNT_SHIFT);
        out.refs.insert(out.refs.end(), list.begin(), list.end());
    }
    uint32_t Build(const std::vector<uint32_t> &list, const Box3 &bounds, unsigned depth, int fork = 0) {
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.push_back({0.0, 0, 0});
        out.depth.push_back(depth);
        const double b[6] = {bounds.lo[0], bounds.lo[1], bounds.lo[2], bounds.hi[0], bounds.hi[1], bounds.hi[2]};
        out.bounds.insert(out.bounds.end(), b, b + 6);
        out.max_depth = std::max(out.max_depth, depth);
        const size_t n = list.size();
        if (n <= 1 || depth == 100) {  // stop condition, Scene.cpp:25
            Leaf(me, list);
            return me;
        }
        const unsigned axis = depth % 3;  // round-robin
        double median = 0;                  // mean of all vertex coordinates, summed in list order
        for (uint32_t ti : list) {
            median += tris.V(ti, 0, axis);
            median += tris.V(ti, 1, axis);
            median += tris.V(ti, 2, axis);
        }
        median /= 3 * n;
        Box3 lb = bounds, rb = bounds;  // BoundingBox::Split, Common.cpp:313-338
        lb.hi[axis] = median;
        rb.lo[axis] = median;
        std::vector<uint32_t> left, right;
        for (uint32_t ti : list) {
            const double a = tris.V(ti, 0, axis), bb = tris.V(ti, 1, axis), c = tris.V(ti, 2, axis);
            if (a <= median || bb <= median || c <= median) left.push_back(ti);
            if (a >= median || bb >= median || c >= median) right.push_back(ti);
        }
        if (left.size() + right.size() > 1.5 * n) {  // > 50 % duplicated: stop subdividing
            Leaf(me, list);
            return me;
        }
        out.nodes[me].split = median;
        uint32_t info = axis;
        if (fork > 0 && left.size() >= kTaskMin && right.size() >= kTaskMin) {
            KdTreeFlat lsub, rsub;
            auto lf = std::async(std::launch::async, [&] { KdBuilder<TriView>{tris, lsub}.Build(left, lb, depth + 1, fork - 1); });
            KdBuilder<TriView>{tris, rsub}.Build(right, rb, depth + 1, fork - 1);
            lf.get();
            Append(lsub);  // lands at me + 1
            out.nodes[me].link = Append(rsub);
            out.nodes[me].info = info | RTHIP_KD_HAS_LEFT | RTHIP_KD_HAS_RIGHT;
            return me;
        }
        if (!left.empty()) {
            info |= RTHIP_KD_HAS_LEFT;
            Build(left, lb, depth + 1);  // lands at me + 1
        }
        std::vector<uint32_t>().swap(left);
        uint32_t link = 0;
        if (!right.empty()) {
            info |= RTHIP_KD_HAS_RIGHT;
            link = Build(right, rb, depth + 1);
        }
        out.nodes[me].link = link;
        out.nodes[me].info = info;
        return me;
    }
};
// Binned-SAH binary BVH (16 bins, 3 axes) down to single triangles, then collapsed to 4-wide nodes with leaves of
// 1..4 triangles (the four lanes that trace a ray test a leaf in one step).  The collapse is the optimal one for the
// binary topology under the surface-area cost (dynamic programme over "this subtree as a forest of at most i wide
// sub-trees", i = 1..3, Ylitie, Karras, Laine 2017, section 3.1): a greedy "open the largest child" collapse over
// leaves fixed at <= 4 triangles gave 2.8 children per node on a 2M-triangle height field.  Boxes are exact fp64 while
// building and rounded OUTWARD (plus one ulp) to fp32 when stored, so the device's fp32 slab test stays conservative.
template <class TriView> struct BvhBuilder {
    static constexpr int kBins = 16;
    static constexpr uint32_t kMaxLeaf = RTHIP_BVH_MAX_LEAF;
    struct Box {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        void Grow(const double *l, const double *h) {
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
        }
        void Grow(const Box &b) { Grow(b.lo, b.hi); }
        double Area() const {
            double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            return dx < 0 ? 0.0 : 2 * (dx * dy + dx * dz + dy * dz);
        }
    };
    struct Node2 { Box b[2]; uint32_t c[2]; uint32_t first, count; };  // children: boxes and words (a leaf word holds ONE triangle); the node's range of `order`
    struct Plan { double c[3]; uint8_t k[3]; uint8_t leaf; };  // c[i-1]: cost as a forest of <= i trees; k[0]: how many of an interior node's 4 slots the left child takes;
                                                               // k[i-1], i = 2, 3: left share of the forest, 0 = no better than i - 1 trees; leaf: as ONE tree it is a leaf
    const TriView &tris;
    BvhFlat &out;
    std::vector<double> tlo, thi, cen;  // 3 per triangle
    std::vector<Node2> bin;             // the binary tree
    std::vector<Plan> plan;
    double leaf_cost = 1.5;  // a triangle step (<= 4 fp64 tests, 128 B each) against a node step (4 fp32 slab tests, 128 B)
    static constexpr uint32_t kTaskMin = 1u << 15;  // triangles below which a sub-tree is not worth a thread

    static float Down(double x) {
        float f = (float)x;
        if ((double)f > x) f = std::nextafterf(f, -INFINITY);
        return std::nextafterf(f, -INFINITY);
    }
    static float Up(double x) {
        float f = (float)x;
        if ((double)f < x) f = std::nextafterf(f, INFINITY);
        return std::nextafterf(f, INFINITY);
    }
    void Prepare() {
        const size_t n = tris.size();
        tlo.resize(3 * n); thi.resize(3 * n); cen.resize(3 * n);
        out.order.resize(n);
        std::mutex mu;
        double amax = 0;
        ParallelSlices(n, [&](size_t i0, size_t i1) {
            double am = 0;
            for (size_t i = i0; i < i1; i++) {
                out.order[i] = (uint32_t)i;
                for (int a = 0; a < 3; a++) {
                    double v0 = tris.V((uint32_t)i, 0, a), v1 = tris.V((uint32_t)i, 1, a), v2 = tris.V((uint32_t)i, 2, a);
                    double l = std::min(v0, std::min(v1, v2)), h = std::max(v0, std::max(v1, v2));
                    tlo[3 * i + a] = l; thi[3 * i + a] = h; cen[3 * i + a] = 0.5 * (l + h);
                    am = std::max(am, std::max(std::fabs(l), std::fabs(h)));
                }
            }
            std::lock_guard<std::mutex> g(mu);
            amax = std::max(amax, am);
        });
        out.abs_max = Up(amax);
    }
    Box BoundsOf(uint32_t lo, uint32_t hi) const {
        Box b;
        for (uint32_t i = lo; i < hi; i++) b.Grow(&tlo[3 * out.order[i]], &thi[3 * out.order[i]]);
        return b;
    }
    // builds the binary subtree over order[lo,hi) into `bin`; returns the child word and its box.  fork > 0: the two
    // halves of a big range (disjoint slices of `order`) are built on two threads, the left one into a node array of
    // its own that is appended afterwards with its links rebased; the tree does not depend on the thread count.
    uint32_t Build(std::vector<Node2> &bin, uint32_t lo, uint32_t hi, unsigned depth, Box &box, int fork) {
        box = BoundsOf(lo, hi);
        const uint32_t n = hi - lo;
        auto leaf = [&]() { return RTHIP_BVH_LEAF | ((n - 1) << RTHIP_BVH_COUNT_SHIFT) | lo; };
        if (n <= 1) return leaf();
        Box cb;  // centroid bounds
        for (uint32_t i = lo; i < hi; i++) cb.Grow(&cen[3 * out.order[i]], &cen[3 * out.order[i]]);
        int best_axis = -1, best_bin = 0;
        double best_cost = 1e300;
        if (depth < 40) {
            for (int a = 0; a < 3; a++) {
                const double ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0)) continue;
                Box bb[kBins];
                uint32_t cnt[kBins] = {0};
                const double k = kBins / ext;
                auto bin_slice = [&](size_t i0, size_t i1, Box *sb, uint32_t *sc) {
                    for (size_t i = i0; i < i1; i++) {
                        const uint32_t t = out.order[i];
                        int b = (int)((cen[3 * t + a] - cb.lo[a]) * k);
                        b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                        sc[b]++;
                        sb[b].Grow(&tlo[3 * t], &thi[3 * t]);
                    }
                };
                if (fork > 0 && n >= 8 * kTaskMin) {  // the top of the tree: few nodes, long loops (counts and min / max merge exactly)
                    std::mutex mu;
                    ParallelSlices(n, [&](size_t i0, size_t i1) {
                        Box sb[kBins];
                        uint32_t sc[kBins] = {0};
                        bin_slice(lo + i0, lo + i1, sb, sc);
                        std::lock_guard<std::mutex> g(mu);
                        for (int b = 0; b < kBins; b++) { bb[b].Grow(sb[b]); cnt[b] += sc[b]; }
                    });
                } else {
                    bin_slice(lo, hi, bb, cnt);
                }
                double right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                uint32_t c = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    acc.Grow(bb[b]); c += cnt[b];
                    right_area[b] = acc.Area(); right_cnt[b] = c;
                }
                acc = Box(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.Grow(bb[b]); c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    // leaves are tested four triangles at a time: cost in units of ceil(n / 4)
                    const double cost = acc.Area() * ((c + 3) / 4) + right_area[b + 1] * ((right_cnt[b + 1] + 3) / 4);
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
        }
        uint32_t mid = lo;
        if (best_axis >= 0) {
            const int a = best_axis;
            const double k = kBins / (cb.hi[a] - cb.lo[a]), base = cb.lo[a];
            auto first_right = std::partition(out.order.begin() + lo, out.order.begin() + hi, [&](uint32_t t) {
                int b = (int)((cen[3 * t + a] - base) * k);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            });
            mid = (uint32_t)(first_right - out.order.begin());
        }
        if (mid == lo || mid == hi || depth >= 40) {  // coincident centroids or depth guard: median split on the widest axis
            int a = 0;
            for (int k2 = 1; k2 < 3; k2++)
                if (cb.hi[k2] - cb.lo[k2] > cb.hi[a] - cb.lo[a]) a = k2;
            mid = lo + n / 2;
            std::nth_element(out.order.begin() + lo, out.order.begin() + mid, out.order.begin() + hi,
                             [&](uint32_t x, uint32_t y) { return cen[3 * x + a] < cen[3 * y + a]; });
        }
        const uint32_t me = (uint32_t)bin.size();
        bin.emplace_back();
        Box b0, b1;
        uint32_t c0, c1;
        if (fork > 0 && mid - lo >= kTaskMin && hi - mid >= kTaskMin) {
            std::vector<Node2> lbin;
            auto lf = std::async(std::launch::async, [&] { return Build(lbin, lo, mid, depth + 1, b0, fork - 1); });
            c1 = Build(bin, mid, hi, depth + 1, b1, fork - 1);
            c0 = lf.get();
            const uint32_t base = (uint32_t)bin.size();
            for (Node2 &n : lbin) {
                for (uint32_t &c : n.c)
                    if (!(c & RTHIP_BVH_LEAF)) c += base;
                bin.push_back(n);
            }
            if (!(c0 & RTHIP_BVH_LEAF)) c0 += base;
        } else {
            c0 = Build(bin, lo, mid, depth + 1, b0, 0);
            c1 = Build(bin, mid, hi, depth + 1, b1, 0);
        }
        bin[me].b[0] = b0; bin[me].b[1] = b1;
        bin[me].c[0] = c0; bin[me].c[1] = c1;
        bin[me].first = lo; bin[me].count = n;
        return me;
    }
    double CostOf(uint32_t word, const Box &b, int i) const {
        return (word & RTHIP_BVH_LEAF) ? b.Area() * leaf_cost : plan[word].c[i - 1];
    }
    // bottom-up: the cheapest way to stand for the binary subtree n2 as at most i wide sub-trees
    void Solve(uint32_t n2) {
        const Node2 &o = bin[n2];
        for (int s = 0; s < 2; s++)
            if (!(o.c[s] & RTHIP_BVH_LEAF)) Solve(o.c[s]);
        Box me = o.b[0];
        me.Grow(o.b[1]);
        double dist[5];
        uint8_t at[5];
        for (int j = 2; j <= 4; j++) {
            dist[j] = 1e300;
            at[j] = 1;
            for (int k = 1; k < j; k++) {
                const double c = CostOf(o.c[0], o.b[0], std::min(k, 3)) + CostOf(o.c[1], o.b[1], std::min(j - k, 3));
                if (c < dist[j]) { dist[j] = c; at[j] = (uint8_t)k; }
            }
        }
        Plan p;
        const double interior = dist[4] + me.Area(), as_leaf = o.count <= kMaxLeaf ? me.Area() * leaf_cost : 1e300;
        p.leaf = as_leaf <= interior;
        p.c[0] = p.leaf ? as_leaf : interior;
        p.k[0] = at[4];
        for (int i = 2; i <= 3; i++) {
            const bool better = dist[i] < p.c[i - 2];
            p.c[i - 1] = better ? dist[i] : p.c[i - 2];
            p.k[i - 1] = better ? at[i] : 0;
        }
        plan[n2] = p;
    }
    struct Item { uint32_t c; Box b; };
    // the roots of the planned forest of at most i trees for `word`, appended to items
    void Forest(uint32_t word, const Box &box, int i, Item *items, int &n) const {
        if (!(word & RTHIP_BVH_LEAF)) {
            const Plan &p = plan[word];
            i = std::min(i, 3);
            while (i > 1 && p.k[i - 1] == 0) i--;
            if (i > 1) {
                Forest(bin[word].c[0], bin[word].b[0], p.k[i - 1], items, n);
                Forest(bin[word].c[1], bin[word].b[1], i - p.k[i - 1], items, n);
                return;
            }
        }
        items[n].c = word;
        items[n++].b = box;
    }
    // 4-wide node for the binary node n2 (planned as an interior node)
    uint32_t Collapse(uint32_t n2, unsigned depth) {
        Item items[4];
        int n = 0;
        Forest(bin[n2].c[0], bin[n2].b[0], plan[n2].k[0], items, n);
        Forest(bin[n2].c[1], bin[n2].b[1], 4 - plan[n2].k[0], items, n);
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        out.depth = std::max(out.depth, depth);
        for (int i = 0; i < 4; i++) {
            BvhFlat::Child ch;
            std::memset(&ch, 0, sizeof ch);
            if (i < n) {
                for (int a = 0; a < 3; a++) { ch.lo[a] = Down(items[i].b.lo[a]); ch.hi[a] = Up(items[i].b.hi[a]); }
                const uint32_t c = items[i].c;
                if (c & RTHIP_BVH_LEAF) ch.child = c;
                else if (plan[c].leaf) ch.child = RTHIP_BVH_LEAF | ((bin[c].count - 1) << RTHIP_BVH_COUNT_SHIFT) | bin[c].first;
                else ch.child = Collapse(c, depth + 1);
            } else {
                for (int a = 0; a < 3; a++) { ch.lo[a] = INFINITY; ch.hi[a] = -INFINITY; }
                ch.child = RTHIP_BVH_ABSENT;
            }
            out.nodes[me].c[i] = ch;
        }
        return me;
    }
    void Run() {
        out = BvhFlat();
        if (tris.empty()) return;
        if (const char *e = std::getenv("RE_BVH_LEAF_COST")) leaf_cost = std::atof(e);  // tuning experiments
        Prepare();
        Box root;
        uint32_t w = Build(bin, 0, (uint32_t)tris.size(), 0, root, BuildForkDepth());
        if (!(w & RTHIP_BVH_LEAF)) {
            plan.resize(bin.size());
            Solve(w);
            if (plan[w].leaf) w = RTHIP_BVH_LEAF | ((bin[w].count - 1) << RTHIP_BVH_COUNT_SHIFT) | bin[w].first;
        }
        if (w & RTHIP_BVH_LEAF) {  // tiny mesh: a root node whose only child is the single leaf
            out.nodes.emplace_back();
            for (int i = 0; i < 4; i++) {
                BvhFlat::Child ch;
                std::memset(&ch, 0, sizeof ch);
                for (int a = 0; a < 3; a++) {
                    ch.lo[a] = i == 0 ? Down(root.lo[a]) : INFINITY;
                    ch.hi[a] = i == 0 ? Up(root.hi[a]) : -INFINITY;
                }
                ch.child = i == 0 ? w : RTHIP_BVH_ABSENT;
                out.nodes[0].c[i] = ch;
            }
        } else {
            Collapse(w, 0);
        }
        std::vector<Node2>().swap(bin);
        std::vector<Plan>().swap(plan);
    }
};
}  // namespace rtbuild
