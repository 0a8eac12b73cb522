// Does a kernel's L2 content survive into the next kernel on MI355X?  Pointer chase over a small buffer,
// launched several times; prints cycles per dependent load for each launch, and for a second pass inside one launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const uint32_t *next, uint32_t steps, unsigned long long *out, uint32_t stride_lanes) {
    uint32_t p = (threadIdx.x % 64) * stride_lanes;  // every lane its own chain (divergent gather like the BVH walk)
    unsigned long long t0 = clock64();
    for (uint32_t i = 0; i < steps; i++) p = next[p];
    unsigned long long t1 = clock64();
    for (uint32_t i = 0; i < steps; i++) p = next[p];
    unsigned long long t2 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = p; }
}
int main() {
    for (size_t bytes : {size_t(256) << 10, size_t(2) << 20, size_t(16) << 20}) {
        size_t n = bytes / 128;  // one element per 128-byte line
        std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> next(n * 32, 0);
        for (size_t i = 0; i < n; i++) next[perm[i] * 32] = perm[(i + 1) % n] * 32;
        uint32_t *d; unsigned long long *o;
        hipMalloc(&d, next.size() * 4); hipMalloc(&o, 64);
        hipMemcpy(d, next.data(), next.size() * 4, hipMemcpyHostToDevice);
        uint32_t steps = (uint32_t)std::min<size_t>(n / 64, 2000);
        for (int launch = 0; launch < 4; launch++) {
            hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, steps, o, 32u * (uint32_t)(n / 64));
            unsigned long long h[3]; hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
            printf("buffer %6zu KB launch %d: first pass %6.0f cyc/step, second pass %6.0f cyc/step (steps %u)\n", bytes >> 10, launch,
                   (double)h[0] / steps, (double)h[1] / steps, steps);
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
