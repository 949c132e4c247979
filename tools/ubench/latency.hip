// micro-benchmark: dependent-load latency (pointer chase) for working sets from L2-size to HBM-size, one lane
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void k_chase(const unsigned* next, int steps, unsigned* out) {
    unsigned i = 0;
    for (int s = 0; s < steps; ++s) i = next[i];
    out[0] = i;
}
__global__ void k_touch(unsigned* p, size_t n) {          // rewrite the table from ALL CUs (as a producer kernel would)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i];
}
int main() {
    for (size_t mb : {1, 4, 32, 128, 512}) {
        const size_t n = mb * 1024 * 1024 / 4, stride = 64;            // one entry per 256 B
        std::vector<unsigned> h(n, 0);
        const size_t slots = n / stride;
        std::vector<unsigned> perm(slots);
        for (size_t i = 0; i < slots; ++i) perm[i] = (unsigned)i;
        srand(1);
        for (size_t i = slots - 1; i > 0; --i) { size_t j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
        for (size_t i = 0; i < slots; ++i) h[(size_t)perm[i] * stride] = perm[(i + 1) % slots] * (unsigned)stride;
        unsigned *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 4);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int steps = 20000;
        for (int variant = 0; variant < 2; ++variant) {
            if (variant == 1) hipLaunchKernelGGL(k_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, n);
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, 100, o);
            hipDeviceSynchronize();
            if (variant == 1) hipLaunchKernelGGL(k_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, n);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, steps, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%4zu MiB table, %s: %.0f ns per dependent load\n", mb, variant ? "just rewritten by all CUs" : "idle", ms * 1e6 / steps);
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
