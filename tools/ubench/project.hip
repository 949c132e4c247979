// micro-benchmark of the step's N-proportional launch  S = [B2;V2] Y  (M = 256, N = K = 1024, split-K 4 -> 256 workgroups):
// the library kernel (gemm.hip, 64 x 64 x 16 k-tiles, two barriers per k-tile) against a deep-stage candidate
// (64 x 64 x BK tiles with BK = 64: all 16-byte loads of a stage in flight at once, 4 stages and 8 barriers per workgroup),
// with in-kernel s_memrealtime stamps (100 MHz) of the phases.
// build: hipcc --offload-arch=gfx950 -O3 -I../../variational_gridded_gaussian_processes_amd/csrc -o project project.hip
#include "../../variational_gridded_gaussian_processes_amd/csrc/gemm.hip"

#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double vg_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long rt() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// A [M][lda] K-contiguous, B [K][ldb] N-contiguous, C slabs [ks][M][ldc]; full tiles only
template <int BK, bool STAMP>
__global__ __launch_bounds__(512) void proj_deep(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                                 double* __restrict__ C, int ldc, long c_slab, int tiles_m, int tiles_n,
                                                 int ksplit, int kchunk, unsigned long long* stamps) {
    constexpr int T = 64, NT = 512;
    constexpr int RS = BK + 2;                    // A tile in LDS [row][BK + 2]: fragment lanes (i, fk) -> 2 i + fk distinct banks
    constexpr int KS = T + 16;                    // B tile in LDS [k][80]
    constexpr int NA = T * BK / (2 * NT);         // double2 loads per thread and operand and stage
    extern __shared__ double lds[];
    double* As = lds;
    double* Bs = lds + T * RS;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (STAMP) t0 = rt();

    const int t = blockIdx.x;
    const int nx = 8 / ksplit;
    const int xcd = t & 7, j = t >> 3;
    const int ks = xcd / nx;
    const int gi = j / tiles_m;
    const int tm = j - gi * tiles_m;
    const int tn = gi * nx + (xcd - ks * nx);
    const int row0 = tm * T, col0 = tn * T, k_begin = ks * kchunk;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;      // 2 x 4 waves, each 32 rows x 16 columns
    const int fi = lane & 15, fk = lane >> 4;

    // A stage: 64 rows x BK k -> BK / 2 double2 per row; thread -> (row = idx / (BK/2), kp = idx % (BK/2))
    // B stage: BK k x 64 cols -> 32 double2 per k row
    const double* pa[NA];
    const double* pb[NA];
    int la[NA], lb[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int idx = tid + r * NT;
        const int arow = idx / (BK / 2), akp = idx % (BK / 2);
        pa[r] = A + (long)(row0 + arow) * lda + k_begin + 2 * akp;
        la[r] = arow * RS + 2 * akp;
        const int bk = idx / 32, bcp = idx % 32;
        pb[r] = B + (long)(k_begin + bk) * ldb + col0 + 2 * bcp;
        lb[r] = bk * KS + 2 * bcp;
    }
    vg_d2 ra[NA], rb[NA];
    auto ld = [&]() {
#pragma unroll
        for (int r = 0; r < NA; ++r) { ra[r] = *reinterpret_cast<const vg_d2*>(pa[r]); rb[r] = *reinterpret_cast<const vg_d2*>(pb[r]); }
#pragma unroll
        for (int r = 0; r < NA; ++r) { pa[r] += BK; pb[r] += (long)BK * ldb; }
    };
    vg_d4 acc[2] = {(vg_d4){0, 0, 0, 0}, (vg_d4){0, 0, 0, 0}};
    const int nst = kchunk / BK;
    ld();
    for (int s = 0; s < nst; ++s) {
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            *reinterpret_cast<vg_d2*>(As + la[r]) = ra[r];
            *reinterpret_cast<vg_d2*>(Bs + lb[r]) = rb[r];
        }
        __syncthreads();
        if (STAMP && s == 0) t1 = rt();
        if (s + 1 < nst) ld();
        const double* ap0 = As + (wr * 32 + fi) * RS + fk;
        const double* ap1 = ap0 + 16 * RS;
        const double* bp = Bs + fk * KS + wc * 16 + fi;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const double a0 = ap0[kk], a1 = ap1[kk], b0 = bp[kk * KS];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1], 0, 0, 0);
        }
        __syncthreads();
    }
    if (STAMP) t2 = rt();
    double* Cs = C + (long)ks * c_slab;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Cs[(long)(row0 + wr * 32 + mb * 16 + fk + 4 * r) * ldc + col0 + wc * 16 + fi] = acc[mb][r];
    if (STAMP) {
        __builtin_amdgcn_s_waitcnt(0);
        t3 = rt();
        if (tid == 0) { stamps[4 * t] = t0; stamps[4 * t + 1] = t1; stamps[4 * t + 2] = t2; stamps[4 * t + 3] = t3; }
    }
}

__global__ void touch(double* p, long n, double s) {      // stand-in for the producer: rewrites the A operand from all XCDs
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * s;
}

int main() {
    const int M = 256, N = 1024, K = 1024, KSPLIT = 4;
    double *A, *B, *C0, *C1;
    CK(hipMalloc(&A, sizeof(double) * M * K));
    CK(hipMalloc(&B, sizeof(double) * K * N));
    CK(hipMalloc(&C0, sizeof(double) * M * N * KSPLIT));
    CK(hipMalloc(&C1, sizeof(double) * M * N * KSPLIT));
    std::vector<double> hA((size_t)M * K), hB((size_t)K * N);
    srand(1);
    for (auto& v : hA) v = rand() / (double)RAND_MAX - 0.5;
    for (auto& v : hB) v = rand() / (double)RAND_MAX - 0.5;
    CK(hipMemcpy(A, hA.data(), sizeof(double) * M * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), sizeof(double) * K * N, hipMemcpyHostToDevice));
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, 8 * 4 * 256));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(hipFuncSetAttribute((const void*)proj_deep<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)proj_deep<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)proj_deep<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)proj_deep<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    VgGemmBatch gb;
    vg_gemm_init(&gb);
    const int pj = vg_gemm_add(&gb, A, K, 1, B, N, 1, C0, N, M, N, K, KSPLIT, (long)M * N);
    vg_gemm_xcd_group(&gb, pj);
    auto lib = [&]() { CK(vg_gemm_launch(&gb, st, VG_GEMM_TAG_GRAM_PROJECT)); };
    auto lds_bytes = [](int bk) { return (size_t)(64 * (bk + 2) + bk * 80) * 8; };
    auto deep64 = [&](bool stamp) {
        if (stamp) hipLaunchKernelGGL((proj_deep<64, true>), dim3(256), dim3(512), lds_bytes(64), st, A, K, B, N, C1, N, (long)M * N, 4, 16, KSPLIT, K / KSPLIT, stamps);
        else hipLaunchKernelGGL((proj_deep<64, false>), dim3(256), dim3(512), lds_bytes(64), st, A, K, B, N, C1, N, (long)M * N, 4, 16, KSPLIT, K / KSPLIT, stamps);
    };
    auto deep32 = [&](bool stamp) {
        if (stamp) hipLaunchKernelGGL((proj_deep<32, true>), dim3(256), dim3(512), lds_bytes(32), st, A, K, B, N, C1, N, (long)M * N, 4, 16, KSPLIT, K / KSPLIT, stamps);
        else hipLaunchKernelGGL((proj_deep<32, false>), dim3(256), dim3(512), lds_bytes(32), st, A, K, B, N, C1, N, (long)M * N, 4, 16, KSPLIT, K / KSPLIT, stamps);
    };

    // correctness: candidate slabs == library slabs (same k order within a slab is not required: compare the slab sums)
    lib();
    deep64(false);
    CK(hipStreamSynchronize(st));
    std::vector<double> h0((size_t)M * N * KSPLIT), h1((size_t)M * N * KSPLIT);
    CK(hipMemcpy(h0.data(), C0, h0.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), C1, h1.size() * 8, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    for (size_t i = 0; i < h0.size(); ++i) { maxd = std::max(maxd, fabs(h0[i] - h1[i])); maxv = std::max(maxv, fabs(h0[i])); }
    printf("deep64 vs library: max |diff| %.3e (max |value| %.3e)\n", maxd, maxv);
    deep32(false);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h1.data(), C1, h1.size() * 8, hipMemcpyDeviceToHost));
    maxd = 0;
    for (size_t i = 0; i < h0.size(); ++i) maxd = std::max(maxd, fabs(h0[i] - h1[i]));
    printf("deep32 vs library: max |diff| %.3e\n", maxd);

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto fn, bool with_touch) {
        const int REP = 200;
        for (int i = 0; i < 10; ++i) fn();
        CK(hipStreamSynchronize(st));
        float base = 0;
        if (with_touch) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(touch, dim3(M * K / 256), dim3(256), 0, st, A, (long)M * K, 1.0);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&base, e0, e1));
        }
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < REP; ++i) {
            if (with_touch) hipLaunchKernelGGL(touch, dim3(M * K / 256), dim3(256), 0, st, A, (long)M * K, 1.0);
            fn();
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s %s  %7.2f us per launch\n", name, with_touch ? "after a producer rewrote A" : "back to back             ", (ms - base) * 1e3 / REP);
        fflush(stdout);
    };
    for (int wt = 0; wt < 2; ++wt) {
        time_it("library wide kernel (64x64x16)", lib, wt);
        time_it("deep stage BK=64", [&]() { deep64(false); }, wt);
        time_it("deep stage BK=32", [&]() { deep32(false); }, wt);
    }
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 3; ++rep) { if (v == 0) deep64(true); else deep32(true); }
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hs(4 * 256);
        CK(hipMemcpy(hs.data(), stamps, 8 * 4 * 256, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        double s01 = 0, s12 = 0, s23 = 0, late = 0;
        for (int i = 0; i < 256; ++i) tmin = std::min(tmin, hs[4 * i]);
        for (int i = 0; i < 256; ++i) {
            tmax = std::max(tmax, hs[4 * i + 3]);
            late = std::max(late, (double)(hs[4 * i] - tmin));
            s01 += hs[4 * i + 1] - hs[4 * i]; s12 += hs[4 * i + 2] - hs[4 * i + 1]; s23 += hs[4 * i + 3] - hs[4 * i + 2];
        }
        printf("stamps BK=%d: first-WG-start -> last-WG-end %.2f us; last WG starts %.2f us late; mean per WG: first stage in LDS %.2f us, k-loop %.2f us, stores %.2f us\n",
               v == 0 ? 64 : 32, (tmax - tmin) * 0.01, late * 0.01, s01 / 256 * 0.01, s12 / 256 * 0.01, s23 / 256 * 0.01);
    }
    return 0;
}
