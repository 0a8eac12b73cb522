// Host threads for the scene build (Mesh::Compile, the flattener): std::async tasks, no pool.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <future>
#include <thread>
#include <vector>

namespace re {
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

}  // namespace re
