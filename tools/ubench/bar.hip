// micro-benchmarks: cost of workgroup barriers / LDS round trips / fp64 chains with 16 waves on one CU
#include <hip/hip_runtime.h>
#include <stdio.h>
#define BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__global__ __launch_bounds__(1024) void k_bar(double* out, int n) {
    for (int i = 0; i < n; ++i) BAR();
    if (threadIdx.x == 0) out[0] = 1.0;
}
__global__ __launch_bounds__(1024) void k_bar_lds(double* out, int n) {
    __shared__ double s[2048];
    double v = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        s[(threadIdx.x * 7 + i) & 1023] = v;
        BAR();
        v = s[(threadIdx.x * 13 + i) & 1023] + 1.0;
        BAR();
    }
    out[threadIdx.x] = v;
}
__global__ __launch_bounds__(1024) void k_chain(double* out, int n) {       // dependent fp64 fma chain, all waves
    double v = threadIdx.x * 1e-3, a = 1.0000001, b = 1e-9;
    for (int i = 0; i < n; ++i) { v = fma(v, a, b); v = fma(v, a, b); v = fma(v, a, b); v = fma(v, a, b); }
    out[threadIdx.x] = v;
}
__global__ __launch_bounds__(1024) void k_one_wave_work(double* out, int n) {  // wave 0 does 100 dependent fmas, everyone barriers
    double v = threadIdx.x * 1e-3, a = 1.0000001, b = 1e-9;
    for (int i = 0; i < n; ++i) {
        if (threadIdx.x < 16) {
#pragma unroll
            for (int j = 0; j < 50; ++j) v = fma(v, a, b);
        }
        BAR();
    }
    out[threadIdx.x] = v;
}
int main() {
    double* d; hipMalloc(&d, 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    auto run = [&](const char* name, void (*kern)(double*, int), int threads) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, n); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s threads %4d: %.1f ns per iteration\n", name, threads, ms * 1e6 / n);
    };
    for (int th : {1024, 256, 64}) {
        run("barrier only", k_bar, th);
        run("lds write,bar,read,bar", k_bar_lds, th);
        run("4 dependent fp64 fma", k_chain, th);
        run("50 fma on 16 lanes + bar", k_one_wave_work, th);
    }
    return 0;
}
