// graph_memset.hip — reproducer for the anomaly DESIGN.md §4 records: does a memset NODE of a replayed HIP graph keep clearing
// its buffer after a SECOND graph has been captured and instantiated on the same stream?  Each graph = memset(buf, 0) followed
// by a kernel that adds 1 to every word of buf; after any replay every word must be exactly 1.
// build: hipcc --offload-arch=gfx950 -O2 -o graph_memset graph_memset.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_inc(unsigned *p, unsigned n) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1u; }
static int capture(hipStream_t s, unsigned *buf, unsigned n, size_t memset_bytes, hipGraphExec_t *exec) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(buf, 0, memset_bytes, s));
    hipLaunchKernelGGL(k_inc, dim3((n + 255) / 256), dim3(256), 0, s, buf, n);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(exec, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return 0;
}
static int check(const char *what, hipStream_t s, unsigned *buf, unsigned n) {
    std::vector<unsigned> h(n);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), buf, n * sizeof(unsigned), hipMemcpyDeviceToHost));
    unsigned bad = 0, mx = 0;
    for (unsigned v : h) { bad += v != 1u; mx = v > mx ? v : mx; }
    printf("%-58s %s (words != 1: %u, max %u)\n", what, bad ? "NOT CLEARED" : "ok", bad, mx);
    return bad ? 1 : 0;
}
int main() {
    int rc = 0;
    for (unsigned n : {28u, 3648u, 1u << 20}) {  // the sizes of the library's statistics / counter buffers, and a large one
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        unsigned *a, *b;
        CK(hipMalloc(&a, n * sizeof(unsigned)));
        CK(hipMalloc(&b, n * sizeof(unsigned)));
        CK(hipMemset(a, 0, n * sizeof(unsigned)));
        CK(hipMemset(b, 0, n * sizeof(unsigned)));
        printf("-- %u words\n", n);
        hipGraphExec_t ga, gb;
        if (capture(s, a, n, n * sizeof(unsigned), &ga)) return 2;
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ga, s));
        rc |= check("graph A replayed 3x, no other graph", s, a, n);
        if (capture(s, b, n, n * sizeof(unsigned), &gb)) return 2;
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ga, s));
        rc |= check("graph A replayed 3x after graph B was instantiated", s, a, n);
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(gb, s));
        rc |= check("graph B replayed 3x", s, b, n);
        for (int i = 0; i < 2; i++) { CK(hipGraphLaunch(ga, s)); CK(hipGraphLaunch(gb, s)); }
        rc |= check("A, B alternating: A", s, a, n);
        rc |= check("A, B alternating: B", s, b, n);
        CK(hipGraphExecDestroy(gb));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ga, s));
        rc |= check("graph A replayed after graph B was destroyed", s, a, n);
        CK(hipGraphExecDestroy(ga));
        CK(hipFree(a)); CK(hipFree(b)); CK(hipStreamDestroy(s));
    }
    // The library's case: both graphs of a context clear the SAME buffers (its statistics / per-workgroup / counter arrays), the
    // second one a shorter prefix of them (a shorter run of frames), and each graph holds three memset nodes.
    for (unsigned n : {28u, 3648u, 65536u}) {
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        unsigned *buf[3];
        for (auto &b : buf) { CK(hipMalloc(&b, n * sizeof(unsigned))); CK(hipMemset(b, 0, n * sizeof(unsigned))); }
        printf("-- shared buffers, %u words, graph B clears the first half\n", n);
        hipGraphExec_t ex[2];
        for (int g = 0; g < 2; g++) {
            const unsigned m = g == 0 ? n : n / 2;
            hipGraph_t gr;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (auto &b : buf) CK(hipMemsetAsync(b, 0, m * sizeof(unsigned), s));
            for (auto &b : buf) hipLaunchKernelGGL(k_inc, dim3((m + 255) / 256), dim3(256), 0, s, b, m);
            CK(hipStreamEndCapture(s, &gr));
            CK(hipGraphInstantiate(&ex[g], gr, nullptr, nullptr, 0));
            CK(hipGraphDestroy(gr));
            if (g == 0) {
                for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ex[0], s));
                for (auto &b : buf) rc |= check("graph A replayed 3x, alone", s, b, n);
            }
        }
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ex[0], s));
        for (auto &b : buf) rc |= check("graph A replayed 3x after graph B (same buffers) exists", s, b, n);
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ex[1], s));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ex[0], s));
        for (auto &b : buf) rc |= check("B 3x then A 3x", s, b, n);
        for (auto &e : ex) CK(hipGraphExecDestroy(e));
        for (auto &b : buf) CK(hipFree(b));
        CK(hipStreamDestroy(s));
    }
    printf(rc ? "ANOMALY REPRODUCED\n" : "no anomaly\n");
    return 0;
}
