// Does a RUNNING kernel see (a) a word the host stores to page-locked memory, (b) a device flag another workgroup of the same kernel
// raises?  The interrupt path of rthip.hip (irq_load) relies on both.  Variants: host word in coherent / default mapped memory, device
// flag in fine-grained / ordinary device memory.  build: hipcc --offload-arch=gfx950 -O2 -o irq_visibility irq_visibility.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(const uint32_t *host_word, uint32_t *flag, long long *seen, long long budget) {
    const long long t0 = wall_clock64();
    long long when = -1;
    while (wall_clock64() - t0 < budget) {
        if (blockIdx.x == 0 && __hip_atomic_load(host_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
            __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { when = wall_clock64() - t0; break; }
        __builtin_amdgcn_s_sleep(32);
    }
    if (threadIdx.x == 0) seen[blockIdx.x] = when;
}

int main() {
    const int blocks = 2048;
    for (int host_coherent = 0; host_coherent < 2; host_coherent++)
        for (int fine = 0; fine < 2; fine++) {
            uint32_t *h = nullptr, *flag = nullptr;
            long long *seen = nullptr;
            CK(hipHostMalloc((void **)&h, 256, hipHostMallocMapped | (host_coherent ? hipHostMallocCoherent : 0)));
            if (fine) CK(hipExtMallocWithFlags((void **)&flag, 256, hipDeviceMallocFinegrained)); else CK(hipMalloc((void **)&flag, 256));
            CK(hipMalloc((void **)&seen, sizeof(long long) * blocks));
            CK(hipMemset(flag, 0, 256));
            h[0] = 0;
            CK(hipDeviceSynchronize());
            const long long budget = 100000000ll / 1000 * 20;  // 20 ms at 100 MHz
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, 0, h, flag, seen, budget);
            std::this_thread::sleep_for(std::chrono::milliseconds(3));
            __atomic_store_n(h, 1u, __ATOMIC_RELEASE);
            CK(hipDeviceSynchronize());
            std::vector<long long> s(blocks);
            CK(hipMemcpy(s.data(), seen, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
            int n_seen = 0; long long first = -1, last = -1;
            for (long long v : s) if (v >= 0) { n_seen++; if (first < 0 || v < first) first = v; if (v > last) last = v; }
            printf("host word %s, device flag %s: %d of %d workgroups saw the flag while running; first after %.2f ms, last after %.2f ms of their run (host stored at ~3 ms; budget 20 ms); block 0: %.2f ms\n",
                   host_coherent ? "coherent" : "mapped (default coherency)", fine ? "fine-grained" : "hipMalloc", n_seen, blocks, first / 1e5, last / 1e5, s[0] / 1e5);
            (void)hipFree(flag); (void)hipFree(seen); (void)hipHostFree(h);
        }
    return 0;
}
