// micro-benchmark: cost of a DEPENDENT kernel boundary on one stream (chain of N launches, (t_chain - N * t_kernel) / N),
// bisecting the suspects named in VERDICT r1 item 4: host-coherent pinned memory mapped into the process / touched by the
// kernels, large by-value kernargs, 1024-thread workgroups with large dynamic LDS, graph replay vs eager launches, a blocking
// vs a non-blocking stream, and stores left dirty by the predecessor.
// build: hipcc --offload-arch=gfx950 -O3 -o boundary boundary.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Big { double pad[212]; };          // 1.7 KB by-value argument (VgGemmBatch is 1.68 KB)

__global__ void k_small(double* p, int spin) {                      // trivial: 1 element per thread
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double v = p[i];
    for (int s = 0; s < spin; ++s) v = fma(v, 1.0000001, 1e-9);
    p[i] = v;
}
__global__ void k_big(double* p, int spin, Big b) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double v = p[i] + b.pad[threadIdx.x & 127];
    for (int s = 0; s < spin; ++s) v = fma(v, 1.0000001, 1e-9);
    p[i] = v;
}
__global__ void k_host_rw(double* p, int spin, const double* hin, double* hout) {    // first block reads 5 doubles from pinned host, last writes 16
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double v = p[i];
    if (blockIdx.x == 0 && threadIdx.x < 5) v += hin[threadIdx.x];
    for (int s = 0; s < spin; ++s) v = fma(v, 1.0000001, 1e-9);
    p[i] = v;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 16) hout[threadIdx.x] = v;
}
__global__ __launch_bounds__(1024) void k_lds(double* p, int spin) {
    extern __shared__ double s[];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    s[threadIdx.x] = p[i];
    __syncthreads();
    double v = s[(threadIdx.x + 1) & 1023];
    for (int q = 0; q < spin; ++q) v = fma(v, 1.0000001, 1e-9);
    p[i] = v;
}

enum Kind { SMALL, BIGARG, HOSTRW, LDS };

int main(int argc, char** argv) {
    const int N = 200, REP = 20;
    double* d;
    CK(hipMalloc(&d, 64 << 20));
    CK(hipMemset(d, 0, 64 << 20));
    hipStream_t sb, snb;
    CK(hipStreamCreate(&sb));
    CK(hipStreamCreateWithFlags(&snb, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 139264));
    Big big;
    memset(&big, 0, sizeof(big));
    double *hin = nullptr, *hout = nullptr, *dhin = nullptr, *dhout = nullptr;

    auto launch = [&](Kind k, int grid, int threads, int spin, hipStream_t st) {
        switch (k) {
            case SMALL: hipLaunchKernelGGL(k_small, dim3(grid), dim3(threads), 0, st, d, spin); break;
            case BIGARG: hipLaunchKernelGGL(k_big, dim3(grid), dim3(threads), 0, st, d, spin, big); break;
            case HOSTRW: hipLaunchKernelGGL(k_host_rw, dim3(grid), dim3(threads), 0, st, d, spin, dhin, dhout); break;
            case LDS: hipLaunchKernelGGL(k_lds, dim3(grid), dim3(1024), 139264, st, d, spin); break;
        }
    };
    auto chain_us = [&](Kind k, int grid, int threads, int spin, hipStream_t st, bool graph, int n) -> double {
        hipGraphExec_t ge = nullptr;
        if (graph) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < n; ++i) launch(k, grid, threads, spin, st);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
        }
        double best = 1e30;
        for (int r = 0; r < REP + 2; ++r) {
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            if (graph) CK(hipGraphLaunch(ge, st));
            else for (int i = 0; i < n; ++i) launch(k, grid, threads, spin, st);
            CK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (r >= 2 && us < best) best = us;
        }
        if (ge) CK(hipGraphExecDestroy(ge));
        return best;
    };
    // per-boundary cost = slope between chains of N and N/2 launches (removes the fixed launch/sync cost); the kernel's own
    // duration is inside the slope, so every variant is reported next to its own spin-length twin
    auto report = [&](const char* name, Kind k, int grid, int threads, int spin, hipStream_t st, bool graph) {
        const double a = chain_us(k, grid, threads, spin, st, graph, N), b = chain_us(k, grid, threads, spin, st, graph, N / 2);
        printf("%-58s %s  %7.2f us per launch+boundary (chain %d: %.0f us)\n", name, graph ? "graph" : "eager", (a - b) / (N - N / 2), N, a);
        fflush(stdout);
    };

    for (int phase = 0; phase < 2; ++phase) {
        if (phase == 1) {
            // suspect: fine-grained host-coherent memory mapped into the process (as api.hip:71-74 does)
            CK(hipHostMalloc((void**)&hin, 4096, hipHostMallocDefault));
            CK(hipHostMalloc((void**)&hout, 4096, hipHostMallocDefault));
            CK(hipHostGetDevicePointer((void**)&dhin, hin, 0));
            CK(hipHostGetDevicePointer((void**)&dhout, hout, 0));
            memset(hin, 0, 4096);
            printf("---- after hipHostMalloc of two pinned host blocks (mapped, not necessarily used) ----\n");
        } else {
            printf("---- no pinned host memory in the process ----\n");
        }
        for (int graph = 0; graph < 2; ++graph) {
            report("trivial 256 WG x 256 thr, blocking stream", SMALL, 256, 256, 0, sb, graph);
            report("trivial 256 WG x 256 thr, non-blocking stream", SMALL, 256, 256, 0, snb, graph);
            report("trivial 2 WG x 256 thr", SMALL, 2, 256, 0, sb, graph);
            report("trivial 16 WG x 256 thr, ~5 us of fma spin", SMALL, 16, 256, 3000, sb, graph);
            report("1.7 KB by-value kernarg, 256 WG", BIGARG, 256, 256, 0, sb, graph);
            report("1.7 KB by-value kernarg, 16 WG, ~5 us spin", BIGARG, 16, 256, 3000, sb, graph);
            report("1024-thr WG + 136 KB dynamic LDS, 2 WG", LDS, 2, 1024, 0, sb, graph);
            report("1024-thr WG + 136 KB dynamic LDS, 34 WG", LDS, 34, 1024, 0, sb, graph);
            report("16384 WG x 256 thr (32 MB rewritten: dirty lines)", SMALL, 16384, 256, 0, sb, graph);
            if (phase == 1) {
                report("reads 5 + writes 16 doubles of pinned host, 256 WG", HOSTRW, 256, 256, 0, sb, graph);
                report("reads 5 + writes 16 doubles of pinned host, 16 WG, spin", HOSTRW, 16, 256, 3000, sb, graph);
            }
        }
    }
    return 0;
}
