// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on one CU (1, 2, 4 waves per SIMD; 4 or 8 independent accumulators)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void k_mfma(double* out, int n) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[1024] = (double)(t1 - t0);
}
int main() {
    double* d; hipMalloc(&d, 1025 * 8);
    const int n = 2000;
    for (int th : {256, 512, 1024}) {
        for (int nacc : {4, 8}) {
            if (nacc == 4) hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(th), 0, 0, d, n);
            else hipLaunchKernelGGL(k_mfma<8>, dim3(1), dim3(th), 0, 0, d, n);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int nl = 200000;
            hipEventRecord(e0);
            if (nacc == 4) hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(th), 0, 0, d, nl);
            else hipLaunchKernelGGL(k_mfma<8>, dim3(1), dim3(th), 0, 0, d, nl);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double cyc2; hipMemcpy(&cyc2, d + 1024, 8, hipMemcpyDeviceToHost);
            printf("   wall %.3f ms for %d iterations: %.2f TFLOP/s on this ONE CU; s_memtime ticks %.0f -> %.1f MHz\n", ms, nl,
                   (double)nl * nacc * (th / 64) * 2048.0 / (ms * 1e-3) / 1e12, cyc2, cyc2 / (ms * 1e-3) / 1e6);
            hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(th), 0, 0, d, n); hipDeviceSynchronize();
            if (nacc == 8) { hipLaunchKernelGGL(k_mfma<8>, dim3(1), dim3(th), 0, 0, d, n); hipDeviceSynchronize(); }
            double cyc; hipMemcpy(&cyc, d + 1024, 8, hipMemcpyDeviceToHost);
            const double mfmas_per_simd = (double)n * nacc * (th / 64) / 4.0;
            printf("threads %4d (waves/SIMD %d) acc %d: %.1f cycles per MFMA per SIMD -> %.1f TFLOP/s on 256 CUs at 2.4 GHz\n", th, th / 256, nacc,
                   cyc / mfmas_per_simd, 2048.0 / (cyc / mfmas_per_simd) * 4 * 256 * 2.4e9 / 1e12);
        }
    }
    return 0;
}
