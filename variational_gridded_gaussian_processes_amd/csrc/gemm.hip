// Batched strided fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64) plus the
// deterministic split-K slab reduction.
//
// Every contraction on the hot path is "small output, long reduction" (m x m or m x n
// results summed over n grid points), so the kernel is built around split-K: a problem is
// cut into (tiles_m x tiles_n x ksplit) workgroups, each writing its partial 64x64 tile to
// its own slab; slabs are summed in a fixed order (vg_red_kernel, or on the fly when a
// later GEMM consumes them through b_nslab) so results are bitwise reproducible.
//
// Workgroup = 256 threads = 4 waves (2x2), each wave owns a 32x32 sub-tile = 2x2 MFMA
// blocks.  f64 MFMA fragment layout (cdna_hip_programming.md section 3): A[i=lane&15][k=lane>>4],
// B[k=lane>>4][j=lane&15], D[row=(lane>>4)+4*reg][col=lane&15].
// Operands are staged through LDS in the orientation that is contiguous in global memory
// (so both the global read and the LDS write are coalesced / conflict-free):
//   K-contiguous operand -> LDS [row][VG_BK+1]   (row stride 17 doubles: the 16 lanes of a
//                           fragment read 16 different even banks)
//   M/N-contiguous       -> LDS [k][64+16]       (k stride 80 doubles = 128 B mod 256 B, so the
//                           two k rows of a 32-lane LDS group use disjoint bank halves)
#include "gemm_body.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void vg_gemm_kernel(const VgGemmBatch b) {
    __shared__ double lds[2 * VgTile<64, 16>::TILE];
    vg_gemm_body<64, 16>(b, lds, blockIdx.x);
}
// the launch that makes the only pass over Y ([G;H] = [B;V] B^T for both dimensions and S = [B2;V2] Y): same body under
// its own name so that profiler summaries list the step's N-proportional kernel separately from the other GEMM launches
__global__ __launch_bounds__(256) void vg_gemm_gram_project_kernel(const VgGemmBatch b) {
    __shared__ double lds[2 * VgTile<64, 16>::TILE];
    vg_gemm_body<64, 16>(b, lds, blockIdx.x);
}
// same tile with 8 waves (2 per SIMD): used when a launch has about one workgroup per CU, where a single wave per SIMD
// leaves the MFMA pipe idle between its own LDS round trips and barriers
__global__ __launch_bounds__(512) void vg_gemm_gram_project_wide_kernel(const VgGemmBatch b) {
    __shared__ double lds[2 * VgTile<64, 16>::TILE];
    vg_gemm_body<64, 16, 512>(b, lds, blockIdx.x);
}
// Deep-stage variant for the one launch whose shape is known in advance -- S = [B2;V2] Y: A K-contiguous, B N-contiguous, whole
// 64 x 64 tiles, k-chunks that are multiples of 32, XCD-grouped block order -- 64 x 64 x 32 stages: every thread moves one 16-byte
// load per operand and stage (all of a stage's loads in flight together, the next stage prefetched into registers) and the
// workgroup meets at 2 barriers per 32 k instead of per 16.  tools/ubench/project.hip: 14.2 vs 15.8 us per launch back to back,
// bit-identical slabs (same k order).
typedef double vg_d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void vg_gemm_project_deep_kernel(const VgGemmBatch b) {
    constexpr int T = 64, NT = 512, BK = 32;
    constexpr int RS = BK + 2;                    // A tile in LDS [row][BK + 2]: fragment lanes (i, fk) -> 2 i + fk: distinct banks
    constexpr int KS = T + 16;                    // B tile in LDS [k][80]
    constexpr int NA = T * BK / (2 * NT);         // double2 loads per thread, operand and stage
    __shared__ __attribute__((aligned(16))) double lds[T * RS + BK * KS];
    double* As = lds;
    double* Bs = lds + T * RS;
    const VgGemmP& p = b.p[0];
    const int t = blockIdx.x;
    const int nx = 8 / p.ksplit;
    const int xcd = t & 7, j = t >> 3;
    const int ks = xcd / nx;
    const int gi = j / p.tiles_m;
    const int tm = j - gi * p.tiles_m;
    const int tn = gi * nx + (xcd - ks * nx);
    const int row0 = tm * T, col0 = tn * T, k_begin = ks * p.kchunk;
    const int kend = min(p.K, k_begin + p.kchunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;      // 2 x 4 waves, each 32 rows x 16 columns
    const int fi = lane & 15, fk = lane >> 4;
    const long lda = p.sa_m, ldb = p.sb_k;
    const double* pa[NA];
    const double* pb[NA];
    int la[NA], lb[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int idx = tid + r * NT;
        const int arow = idx / (BK / 2), akp = idx % (BK / 2);
        pa[r] = p.A + (long)(row0 + arow) * lda + k_begin + 2 * akp;
        la[r] = arow * RS + 2 * akp;
        const int bk = idx / 32, bcp = idx % 32;
        pb[r] = p.B + (long)(k_begin + bk) * ldb + col0 + 2 * bcp;
        lb[r] = bk * KS + 2 * bcp;
    }
    vg_d2 ra[NA], rb[NA];
    auto ld = [&]() {
#pragma unroll
        for (int r = 0; r < NA; ++r) { ra[r] = *reinterpret_cast<const vg_d2*>(pa[r]); rb[r] = *reinterpret_cast<const vg_d2*>(pb[r]); }
#pragma unroll
        for (int r = 0; r < NA; ++r) { pa[r] += BK; pb[r] += (long)BK * ldb; }
    };
    vg_d4 acc[2] = {(vg_d4){0.0, 0.0, 0.0, 0.0}, (vg_d4){0.0, 0.0, 0.0, 0.0}};
    const int nst = (kend - k_begin) / BK;
    if (nst > 0) ld();
    for (int s = 0; s < nst; ++s) {
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            *reinterpret_cast<vg_d2*>(As + la[r]) = ra[r];
            *reinterpret_cast<vg_d2*>(Bs + lb[r]) = rb[r];
        }
        __syncthreads();
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
    double* Cs = p.C + (long)ks * p.c_slab;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Cs[(long)(row0 + wr * 32 + mb * 16 + fk + 4 * r) * p.ldc + col0 + wc * 16 + fi] = p.alpha * acc[mb][r];
}
static bool vg_project_deep_ok(const VgGemmBatch* b) {
    static const bool off = getenv("VGGP_NO_DEEP_PROJECT") != nullptr;
    if (off || b->nprob != 1) return false;
    const VgGemmP& p = b->p[0];
    return p.xcd_group && p.sa_k == 1 && p.sb_n == 1 && (p.sa_m % 2) == 0 && (p.sb_k % 2) == 0 && (p.M % 64) == 0 && (p.N % 64) == 0 &&
           (p.kchunk % 32) == 0 && (p.K % 32) == 0 && p.b_nslab == 1 && p.a_nslab == 1 && p.tri == VG_TRI_NONE && !p.accum &&
           !p.dotw && !p.dot_out && p.C && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0;
}
// The deep-stage tile for ANY qualifying problem of a batch (whole 64 x 64 tiles, k-chunks that are multiples of 32, each operand
// contiguous along one of its two axes with 16-byte aligned rows, plain operands): the four orientation combinations differ only in
// how a stage is laid out in LDS -- K-contiguous [row][34], M/N-contiguous [k][80] -- exactly as in the generic body.  Written for
// the masked assembly (config 5: two 8.6 GFLOP and three 4.3 GFLOP products per step, which the 64 x 64 x 16 body runs at 24 TF/s).
__device__ __forceinline__ void vg_gemm_deep_body(const VgGemmP& p, double* lds, int t) {
    constexpr int T = 64, NT = 512, BK = 32, RS = BK + 2, KS = T + 16, NA = T * BK / (2 * NT);
    constexpr int TILE = BK * KS > T * RS ? BK * KS : T * RS;
    double* As = lds;
    double* Bs = lds + TILE;
    const int tiles = p.tiles_m * p.tiles_n;
    int ks, tm, tn;
    if (p.xcd_group) {
        const int nx = 8 / p.ksplit;
        const int xcd = t & 7, j = t >> 3;
        ks = xcd / nx;
        const int gi = j / p.tiles_m;
        tm = j - gi * p.tiles_m;
        tn = gi * nx + (xcd - ks * nx);
    } else {
        ks = t / tiles;
        t -= ks * tiles;
        if (p.rev) t = tiles - 1 - t;
        tm = t / p.tiles_n;
        tn = t - tm * p.tiles_n;
    }
    const int row0 = tm * T, col0 = tn * T;
    int k_begin = ks * p.kchunk, k_end = min(p.K, k_begin + p.kchunk);
    if (p.tri) {          // triangular operand: skip the k-range where it vanishes (whole 128-blocks, as in the generic body)
        if (p.tri == VG_TRI_A_LOWER) k_end = min(k_end, ((row0 + T + 127) >> 7) << 7);
        else if (p.tri == VG_TRI_A_UPPER) k_begin = max(k_begin, (row0 >> 7) << 7);
        else if (p.tri == VG_TRI_B_UPPER) k_end = min(k_end, ((col0 + T + 127) >> 7) << 7);
        else if (p.tri == VG_TRI_B_LOWER) k_begin = max(k_begin, (col0 >> 7) << 7);
        else k_begin = max(k_begin, (max(row0, col0) >> 7) << 7);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;      // 2 x 4 waves, each 32 rows x 16 columns
    const int fi = lane & 15, fk = lane >> 4;
    const bool a_kc = p.sa_k == 1, b_nc = p.sb_n == 1;
    const double* pa[NA];
    const double* pb[NA];
    int la[NA], lb[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int idx = tid + r * NT;
        if (a_kc) { const int row = idx / (BK / 2), kp = idx % (BK / 2); pa[r] = p.A + (long)(row0 + row) * p.sa_m + k_begin + 2 * kp; la[r] = row * RS + 2 * kp; }
        else      { const int k = idx / (T / 2), rp = idx % (T / 2);    pa[r] = p.A + (long)(k_begin + k) * p.sa_k + row0 + 2 * rp;  la[r] = k * KS + 2 * rp; }
        if (b_nc) { const int k = idx / (T / 2), cp = idx % (T / 2);    pb[r] = p.B + (long)(k_begin + k) * p.sb_k + col0 + 2 * cp;  lb[r] = k * KS + 2 * cp; }
        else      { const int col = idx / (BK / 2), kp = idx % (BK / 2); pb[r] = p.B + (long)(col0 + col) * p.sb_n + k_begin + 2 * kp; lb[r] = col * RS + 2 * kp; }
    }
    const long da = a_kc ? BK : (long)BK * p.sa_k, db = b_nc ? (long)BK * p.sb_k : BK;
    vg_d2 ra[NA], rb[NA];
    auto ld = [&]() {
#pragma unroll
        for (int r = 0; r < NA; ++r) { ra[r] = *reinterpret_cast<const vg_d2*>(pa[r]); rb[r] = *reinterpret_cast<const vg_d2*>(pb[r]); }
#pragma unroll
        for (int r = 0; r < NA; ++r) { pa[r] += da; pb[r] += db; }
    };
    const int a_si = a_kc ? RS : 1, a_sk = a_kc ? 1 : KS, b_sj = b_nc ? 1 : RS, b_sk = b_nc ? KS : 1;
    vg_d4 acc[2] = {(vg_d4){0.0, 0.0, 0.0, 0.0}, (vg_d4){0.0, 0.0, 0.0, 0.0}};
    const int nst = (k_end - k_begin) / BK;
    if (nst > 0) ld();
    for (int s = 0; s < nst; ++s) {
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            *reinterpret_cast<vg_d2*>(As + la[r]) = ra[r];
            *reinterpret_cast<vg_d2*>(Bs + lb[r]) = rb[r];
        }
        __syncthreads();
        if (s + 1 < nst) ld();
        const double* ap0 = As + (wr * 32 + fi) * a_si + fk * a_sk;
        const double* ap1 = ap0 + 16 * a_si;
        const double* bp = Bs + fk * b_sk + (wc * 16 + fi) * b_sj;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const double a0 = ap0[kk * a_sk], a1 = ap1[kk * a_sk], b0 = bp[kk * b_sk];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1], 0, 0, 0);
        }
        __syncthreads();
    }
    double* Cs = p.C + (long)ks * p.c_slab;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* cp = Cs + (long)(row0 + wr * 32 + mb * 16 + fk + 4 * r) * p.ldc + col0 + wc * 16 + fi;
            const double v = p.alpha * acc[mb][r];
            *cp = p.accum ? *cp + v : v;
        }
}
static bool vg_deep_ok(const VgGemmP& p) {
    const bool a_kc = p.sa_k == 1, a_mc = p.sa_m == 1, b_nc = p.sb_n == 1, b_kc = p.sb_k == 1;
    if (!(a_kc || a_mc) || !(b_nc || b_kc)) return false;
    const long lda = a_kc ? p.sa_m : p.sa_k, ldb = b_nc ? p.sb_k : p.sb_n;
    return (p.M % 64) == 0 && (p.N % 64) == 0 && (p.K % 32) == 0 && (p.kchunk % 32) == 0 && p.a_nslab == 1 && p.b_nslab == 1 &&
           (p.tri == VG_TRI_NONE || p.ksplit == 1) && !p.dotw && !p.dot_out && p.C && (lda % 2) == 0 && (ldb % 2) == 0 &&
           ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0;
}
static bool vg_deep_batch(const VgGemmBatch* b, VgGemmBatch* out) {
    static const bool off = getenv("VGGP_NO_DEEP_GEMM") != nullptr;
    static const char* mt = getenv("VGGP_DEEP_MIN_TILES");
    const int min_tiles = mt ? atoi(mt) : 128;
    static const bool dbg0 = getenv("VGGP_DEEP_DBG") != nullptr;
    if (dbg0 && b->total_tiles < min_tiles && b->p[0].K >= 512)
        fprintf(stderr, "[deep] below threshold: %d tiles, %d problems, p0 M %d N %d K %d ksplit %d\n", b->total_tiles, b->nprob, b->p[0].M, b->p[0].N, b->p[0].K, b->p[0].ksplit);
    if (off || b->total_tiles < min_tiles) return false;
    *out = *b;
    long deep_tiles = 0;
    for (int i = 0; i < out->nprob; ++i) {
        VgGemmP& p = out->p[i];
        p.deep = vg_deep_ok(p) ? 1 : 0;
        if (p.deep) deep_tiles += (long)p.tiles_m * p.tiles_n * p.ksplit;
    }
    static const bool dbg = getenv("VGGP_DEEP_DBG") != nullptr;
    if (dbg && 4 * deep_tiles < 3L * out->total_tiles)
        for (int i = 0; i < out->nprob; ++i) {
            const VgGemmP& p = out->p[i];
            fprintf(stderr, "[deep] NOT deep: prob %d/%d M %d N %d K %d ksplit %d kchunk %d sa (%ld,%ld) sb (%ld,%ld) aslab %d bslab %d tri %d accum %d tiles %d deep %d\n",
                    i, out->nprob, p.M, p.N, p.K, p.ksplit, p.kchunk, p.sa_m, p.sa_k, p.sb_k, p.sb_n, p.a_nslab, p.b_nslab, p.tri, p.accum,
                    p.tiles_m * p.tiles_n * p.ksplit, p.deep);
        }
    return 4 * deep_tiles >= 3L * out->total_tiles;
}
// mixed batches: deep body for the problems flagged by the host (VgGemmP::deep), the generic 8-wave tile for the others
__global__ __launch_bounds__(512) void vg_gemm_deep_kernel(const VgGemmBatch b) {
    constexpr int DT = 32 * 80 > 64 * 34 ? 32 * 80 : 64 * 34;
    __shared__ __attribute__((aligned(16))) double lds[2 * DT];
    const int bid = blockIdx.x;
    int pi = 0;
    for (int i = 1; i < b.nprob; ++i)
        if (bid >= b.p[i].tile_start) pi = i;
    if (b.p[pi].deep) vg_gemm_deep_body(b.p[pi], lds, bid - b.p[pi].tile_start);
    else vg_gemm_body<64, 16, 512>(b, lds, bid);
}
__global__ __launch_bounds__(512) void vg_gemm_wide_kernel(const VgGemmBatch b) {      // the same 8-wave tile for other launches
    __shared__ double lds[2 * VgTile<64, 16>::TILE];
    vg_gemm_body<64, 16, 512>(b, lds, blockIdx.x);
}
__global__ __launch_bounds__(256) void vg_gemm_small_kernel(const VgGemmBatch b) {
    __shared__ double lds[2 * VgTile<32, 32>::TILE];
    vg_gemm_body<32, 32>(b, lds, blockIdx.x);
}

void vg_gemm_init(VgGemmBatch* b) {
    b->nprob = 0;
    b->total_tiles = 0;
}

int vg_gemm_add(VgGemmBatch* b, const double* A, long sa_m, long sa_k, const double* B, long sb_k,
                long sb_n, double* C, int ldc, int M, int N, int K, int ksplit, long c_slab,
                int b_nslab, long b_slab, double alpha, int accum) {
    if (b->nprob >= VG_GEMM_MAXP) return -1;
    VgGemmP& p = b->p[b->nprob];
    p.A = A; p.B = B; p.C = C;
    p.sa_m = sa_m; p.sa_k = sa_k; p.sb_k = sb_k; p.sb_n = sb_n;
    p.M = M; p.N = N; p.K = K; p.ldc = ldc;
    p.alpha = alpha; p.accum = accum;
    if (ksplit < 1) ksplit = 1;
    int ktiles = (K + VG_BK - 1) / VG_BK;
    if (ktiles < 1) ktiles = 1;
    if (ksplit > ktiles) ksplit = ktiles;
    int per = (ktiles + ksplit - 1) / ksplit;
    ksplit = (ktiles + per - 1) / per;          // drop empty splits
    p.ksplit = ksplit;
    p.kchunk = per * VG_BK;
    p.c_slab = c_slab;
    p.b_nslab = b_nslab < 1 ? 1 : b_nslab;
    p.b_slab = b_slab;
    p.a_nslab = 1;
    p.a_slab = 0;
    p.tri = VG_TRI_NONE;
    p.rev = 0;
    p.xcd_group = 0;
    p.dotw = nullptr; p.dot_out = nullptr; p.dotw_ld = 0; p.deep = 0;
    p.tiles_m = (M + VG_BM - 1) / VG_BM;
    p.tiles_n = (N + VG_BN - 1) / VG_BN;
    p.tile_start = b->total_tiles;
    b->total_tiles += p.tiles_m * p.tiles_n * ksplit;
    return b->nprob++;
}

void vg_gemm_xcd_group(VgGemmBatch* b, int prob) {
    static const bool off = getenv("VGGP_NO_XCD_GROUP") != nullptr;
    VgGemmP& p = b->p[prob];
    // needs: first problem of the launch (block index == slot), ksplit | 8, column blocks divisible over the XCDs of a k-chunk
    const bool ok = !off && p.tile_start == 0 && (p.ksplit == 1 || p.ksplit == 2 || p.ksplit == 4 || p.ksplit == 8) &&
                    (p.tiles_n % (8 / p.ksplit)) == 0;
    p.xcd_group = ok ? 1 : 0;
}

// The deep-stage kernel takes a launch of at least 128 tiles of which at least three quarters qualify (same-box sweep of
// VGGP_DEEP_MIN_TILES, tools/ab_masked.sh: masked 2048^2 step 2.18 -> 2.02 ms, 1024^2 step 275.8 -> 271.6 us -- its Gram launch --
// nothing more below 128).  VGGP_NO_DEEP_GEMM=1: off.
static bool vg_deep_batch(const VgGemmBatch* b, VgGemmBatch* out);

static const char* g_last_project_kernel = "";
const char* vg_last_project_kernel() { return g_last_project_kernel; }

hipError_t vg_gemm_launch(const VgGemmBatch* b, hipStream_t st, int tag) {
    if (b->nprob == 0 || b->total_tiles == 0) return hipSuccess;
    // a batch that cannot fill half the chip with 64 x 64 tiles and is not split-K (the m x m x m chain products) runs
    // with 32 x 32 tiles: 4x the workgroups, each with a quarter of the MFMA work per k-tile
    bool small = b->total_tiles < 128;
    for (int i = 0; i < b->nprob; ++i) small = small && b->p[i].ksplit == 1 && b->p[i].K <= 512;
    if (small) {
        VgGemmBatch s = *b;
        s.total_tiles = 0;
        for (int i = 0; i < s.nprob; ++i) {
            VgGemmP& p = s.p[i];
            p.tiles_m = (p.M + 31) / 32;
            p.tiles_n = (p.N + 31) / 32;
            p.kchunk = ((p.K + 31) / 32) * 32;
            p.xcd_group = 0;
            p.tile_start = s.total_tiles;
            s.total_tiles += p.tiles_m * p.tiles_n;
        }
        hipLaunchKernelGGL(vg_gemm_small_kernel, dim3(s.total_tiles), dim3(256), 0, st, s);
        return hipGetLastError();
    }
    static const bool wide = getenv("VGGP_GEMM_NARROW") == nullptr;
    VgGemmBatch deepb;
    // (a 64 x 64 x 32 k-tile variant of the wide kernel -- half the barriers per MFMA -- was measured SLOWER: 28.6 vs 23.5 us)
    if (tag == VG_GEMM_TAG_GRAM_PROJECT && wide && vg_project_deep_ok(b)) {
        g_last_project_kernel = "vg_gemm_project_deep_kernel";
        hipLaunchKernelGGL(vg_gemm_project_deep_kernel, dim3(b->total_tiles), dim3(512), 0, st, *b);
    } else if (tag == VG_GEMM_TAG_GRAM_PROJECT && wide && b->total_tiles <= 320) {
        g_last_project_kernel = "vg_gemm_gram_project_wide_kernel";
        hipLaunchKernelGGL(vg_gemm_gram_project_wide_kernel, dim3(b->total_tiles), dim3(512), 0, st, *b);
    } else if (tag == VG_GEMM_TAG_GRAM_PROJECT) {
        g_last_project_kernel = "vg_gemm_gram_project_kernel";
        hipLaunchKernelGGL(vg_gemm_gram_project_kernel, dim3(b->total_tiles), dim3(256), 0, st, *b);
    }
    else if (vg_deep_batch(b, &deepb)) {
        hipLaunchKernelGGL(vg_gemm_deep_kernel, dim3(deepb.total_tiles), dim3(512), 0, st, deepb);
    }
    else if (tag == VG_GEMM_TAG_WIDE && wide && b->total_tiles <= 320)
        hipLaunchKernelGGL(vg_gemm_wide_kernel, dim3(b->total_tiles), dim3(512), 0, st, *b);
    else
        hipLaunchKernelGGL(vg_gemm_kernel, dim3(b->total_tiles), dim3(256), 0, st, *b);
    return hipGetLastError();
}

// ---- deterministic slab reduction ----------------------------------------------
__global__ __launch_bounds__(256) void vg_red_kernel(const VgRedBatch b) {
    const int bid = blockIdx.x;
    int si = 0;
    for (int i = 1; i < b.nseg; ++i)
        if (bid >= b.s[i].block_start) si = i;
    const VgRedSeg& s = b.s[si];
    const long base = (long)(bid - s.block_start) * 1024;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long i = base + r * 256 + threadIdx.x;
        if (i < s.n) {
            double v = s.in[i];
            for (int k = 1; k < s.nslab; ++k) v += s.in[(long)k * s.slab + i];
            s.out[i] = v;
        }
    }
}

void vg_red_init(VgRedBatch* b) {
    b->nseg = 0;
    b->total_blocks = 0;
}

void vg_red_add(VgRedBatch* b, const double* in, double* out, long n, long slab, int nslab) {
    if (b->nseg >= VG_RED_MAXSEG || n <= 0) return;
    VgRedSeg& s = b->s[b->nseg++];
    s.in = in; s.out = out; s.n = n; s.slab = slab; s.nslab = nslab;
    s.block_start = b->total_blocks;
    b->total_blocks += (int)((n + 1023) / 1024);
}

hipError_t vg_red_launch(const VgRedBatch* b, hipStream_t st) {
    if (b->nseg == 0) return hipSuccess;
    hipLaunchKernelGGL(vg_red_kernel, dim3(b->total_blocks), dim3(256), 0, st, *b);
    return hipGetLastError();
}
