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
#include "common.h"

typedef double vg_d4 __attribute__((ext_vector_type(4)));

#define LDS_KMAJ_STRIDE (VG_BM + 16)   // [k][i]
#define LDS_RMAJ_STRIDE (VG_BK + 1)    // [i][k]
#define LDS_TILE 1280                   // max(16*80, 64*17) doubles

__global__ __launch_bounds__(256) void vg_gemm_kernel(const VgGemmBatch b) {
    __shared__ double lds[2 * LDS_TILE];
    double* As = lds;
    double* Bs = lds + LDS_TILE;

    const int bid = blockIdx.x;
    int pi = 0;
    for (int i = 1; i < b.nprob; ++i)
        if (bid >= b.p[i].tile_start) pi = i;
    const VgGemmP& p = b.p[pi];

    int t = bid - p.tile_start;
    const int tiles = p.tiles_m * p.tiles_n;
    const int ks = t / tiles;
    t -= ks * tiles;
    const int tm = t / p.tiles_n, tn = t - (t / p.tiles_n) * p.tiles_n;
    const int row0 = tm * VG_BM, col0 = tn * VG_BN;
    const int k_begin = ks * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const double* __restrict__ A = p.A;
    const double* __restrict__ B = p.B;
    const long sa_m = p.sa_m, sa_k = p.sa_k, sb_k = p.sb_k, sb_n = p.sb_n;
    const bool a_kc = (sa_k == 1);   // A is K-contiguous
    const bool b_nc = (sb_n == 1);   // B is N-contiguous
    const int M = p.M, N = p.N;
    const int nslab = p.b_nslab;
    const long bslab = p.b_slab;

    // global->register mapping for the 64x16 A tile and 16x64 B tile: 4 elements each
    int a_i[4], a_k[4], b_k[4], b_j[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (a_kc) { a_k[r] = tid & 15; a_i[r] = (tid >> 4) + 16 * r; }
        else      { a_i[r] = tid & 63; a_k[r] = (tid >> 6) + 4 * r; }
        if (b_nc) { b_j[r] = tid & 63; b_k[r] = (tid >> 6) + 4 * r; }
        else      { b_k[r] = tid & 15; b_j[r] = (tid >> 4) + 16 * r; }
    }
    const int a_si = a_kc ? LDS_RMAJ_STRIDE : 1, a_sk = a_kc ? 1 : LDS_KMAJ_STRIDE;
    const int b_sj = b_nc ? 1 : LDS_RMAJ_STRIDE, b_sk = b_nc ? LDS_KMAJ_STRIDE : 1;

    double ra[4], rb[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = row0 + a_i[r], gk = k0 + a_k[r];
            ra[r] = (gi < M && gk < k_end) ? A[gi * sa_m + gk * sa_k] : 0.0;
            const int gj = col0 + b_j[r], gkb = k0 + b_k[r];
            double v = 0.0;
            if (gj < N && gkb < k_end) {
                const double* bp = B + gkb * sb_k + gj * sb_n;
                v = bp[0];
                for (int s = 1; s < nslab; ++s) v += bp[s * bslab];
            }
            rb[r] = v;
        }
    };

    vg_d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (vg_d4){0.0, 0.0, 0.0, 0.0};

    const int fi = lane & 15, fk = lane >> 4;
    if (k_begin < k_end) load_tile(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += VG_BK) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            As[a_i[r] * a_si + a_k[r] * a_sk] = ra[r];
            Bs[b_k[r] * b_sk + b_j[r] * b_sj] = rb[r];
        }
        __syncthreads();
        if (k0 + VG_BK < k_end) load_tile(k0 + VG_BK);   // in flight while the MFMAs run
#pragma unroll
        for (int kk = 0; kk < VG_BK; kk += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                av[mb] = As[(wr * 32 + mb * 16 + fi) * a_si + (kk + fk) * a_sk];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                bv[nb] = Bs[(kk + fk) * b_sk + (wc * 32 + nb * 16 + fi) * b_sj];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
        }
        __syncthreads();
    }

    double* __restrict__ C = p.C + (long)ks * p.c_slab;
    const int ldc = p.ldc;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wr * 32 + mb * 16 + fk + 4 * r;
                const int col = col0 + wc * 32 + nb * 16 + fi;
                if (row < M && col < N) {
                    double* cp = C + (long)row * ldc + col;
                    const double v = p.alpha * acc[mb][nb][r];
                    *cp = p.accum ? *cp + v : v;
                }
            }
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
    p.tiles_m = (M + VG_BM - 1) / VG_BM;
    p.tiles_n = (N + VG_BN - 1) / VG_BN;
    p.tile_start = b->total_tiles;
    b->total_tiles += p.tiles_m * p.tiles_n * ksplit;
    return b->nprob++;
}

hipError_t vg_gemm_launch(const VgGemmBatch* b, hipStream_t st) {
    if (b->nprob == 0 || b->total_tiles == 0) return hipSuccess;
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
