// The tile body of the batched strided fp64 GEMM (see gemm.hip for the design notes): a template shared by the GEMM kernels
// of gemm.hip and by the single-workgroup kernels of eigh.hip, whose launches can carry a GEMM batch as extra workgroups
// ("riders": the projection of Y and the [C;C1;C2] products run beside the eigensolver chain instead of in front of it).
#pragma once
#include "common.h"

typedef double vg_d4 __attribute__((ext_vector_type(4)));

// Tile configurations: <T = 64, BK = 16> for the contractions over the grid (MFMA-bound: 16 MFMAs per wave and k-tile),
// <T = 32, BK = 32> for the m x m x m chain products (latency-bound: 4x the workgroups, half the k-tiles).
// Both move T*BK = 1024 elements per operand and k-tile, i.e. 4 per thread.
template <int T, int BK>
struct VgTile {
    static constexpr int RPAD = BK == 16 ? 1 : 4;            // K-contiguous operand: LDS [row][BK + RPAD]
    static constexpr int RS = BK + RPAD;
    static constexpr int KS = T + 16;                        // M/N-contiguous operand: LDS [k][T + 16]
    static constexpr int TILE = (BK * KS > T * RS) ? BK * KS : T * RS;
    static constexpr int MB = T / 32;                        // 16 x 16 MFMA blocks per wave and dimension
};

template <int T, int BK, int NT = 256>
__device__ __forceinline__ void vg_gemm_body(const VgGemmBatch& b, double* lds, const int bid) {
    using C_ = VgTile<T, BK>;
    constexpr int NR = T * BK / NT;                              // elements per thread, operand and k-tile
    constexpr int WC = NT / 128;                                 // wave grid 2 x WC
    constexpr int MB = T / 32;                                   // 16 x 16 MFMA blocks per wave: rows
    constexpr int NB = T / (16 * WC);                            //                               columns
    double* As = lds;
    double* Bs = lds + C_::TILE;

    int pi = 0;
    for (int i = 1; i < b.nprob; ++i)
        if (bid >= b.p[i].tile_start) pi = i;
    const VgGemmP& p = b.p[pi];

    int t = bid - p.tile_start;
    const int tiles = p.tiles_m * p.tiles_n;
    int ks, tm, tn;
    if (p.xcd_group) {
        // Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share one; observed, used for speed only).  All
        // tile rows of one (column block, k-chunk) stream the SAME block of B (the observations Y in the projection launch):
        // they are given consecutive slots of ONE XCD, so that block crosses the fabric once and is hit in that XCD's L2 by the
        // others; and an XCD only sees ONE k-chunk, i.e. 1 / ksplit of the A operand.  (vg_gemm_xcd_group checks the shape.)
        const int nx = 8 / p.ksplit;                         // XCDs per k-chunk
        const int xcd = t & 7, j = t >> 3;
        ks = xcd / nx;
        const int gi = j / p.tiles_m;
        tm = j - gi * p.tiles_m;
        tn = gi * nx + (xcd - ks * nx);
    } else {
        ks = t / tiles;
        t -= ks * tiles;
        tm = t / p.tiles_n;
        tn = t - tm * p.tiles_n;
    }
    const int row0 = tm * T, col0 = tn * T;
    int k_begin = ks * p.kchunk;
    int k_end = min(p.K, k_begin + p.kchunk);
    if (p.tri) {          // triangular operand: skip the k-range where it vanishes (whole 128-blocks; split-K is not combined with it)
        if (p.tri == VG_TRI_A_LOWER) k_end = min(k_end, ((row0 + T + 127) >> 7) << 7);
        else if (p.tri == VG_TRI_A_UPPER) k_begin = max(k_begin, (row0 >> 7) << 7);
        else if (p.tri == VG_TRI_B_UPPER) k_end = min(k_end, ((col0 + T + 127) >> 7) << 7);
        else if (p.tri == VG_TRI_B_LOWER) k_begin = max(k_begin, (col0 >> 7) << 7);
        else k_begin = max(k_begin, (max(row0, col0) >> 7) << 7);
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;

    const double* __restrict__ A = p.A;
    const double* __restrict__ B = p.B;
    const long sa_m = p.sa_m, sa_k = p.sa_k, sb_k = p.sb_k, sb_n = p.sb_n;
    const bool a_kc = (sa_k == 1);   // A is K-contiguous
    const bool b_nc = (sb_n == 1);   // B is N-contiguous
    const int M = p.M, N = p.N;
    const int nslab = p.b_nslab, anslab = p.a_nslab;
    const long bslab = p.b_slab, aslab = p.a_slab;

    // global->register mapping for the T x BK A tile and BK x T B tile: 4 elements each
    int a_i[NR], a_k[NR], b_k[NR], b_j[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (a_kc) { a_k[r] = tid % BK; a_i[r] = tid / BK + (NT / BK) * r; }
        else      { a_i[r] = tid % T; a_k[r] = tid / T + (NT / T) * r; }
        if (b_nc) { b_j[r] = tid % T; b_k[r] = tid / T + (NT / T) * r; }
        else      { b_k[r] = tid % BK; b_j[r] = tid / BK + (NT / BK) * r; }
    }
    const int a_si = a_kc ? C_::RS : 1, a_sk = a_kc ? 1 : C_::KS;
    const int b_sj = b_nc ? 1 : C_::RS, b_sk = b_nc ? C_::KS : 1;

    // two register stages: while tile i is multiplied, tiles i+1 AND i+2 are in flight (one k-tile of MFMAs is ~0.5 us,
    // an L2 / HBM round trip is 1-2 us: a single stage leaves every k-tile waiting for its operands)
    double ra0[NR], rb0[NR], ra1[NR], rb1[NR];
    auto load_tile = [&](int k0, double (&ra)[NR], double (&rb)[NR]) {
        const double* bp[NR];
        const double* ap[NR];
        bool bok[NR], aok[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int gi = row0 + a_i[r], gk = k0 + a_k[r];
            aok[r] = gi < M && gk < k_end;
            ap[r] = A + (aok[r] ? gi * sa_m + gk * sa_k : 0);
            ra[r] = aok[r] ? ap[r][0] : 0.0;
            const int gj = col0 + b_j[r], gkb = k0 + b_k[r];
            bok[r] = gj < N && gkb < k_end;
            bp[r] = B + (bok[r] ? gkb * sb_k + gj * sb_n : 0);
            rb[r] = bok[r] ? bp[r][0] : 0.0;
        }
        // B given as a sum of slabs (the producer's split-K partials): all 4 elements of up to 3 further slabs are
        // in flight together -- a load-add chain per slab would expose one L2 round trip per slab and element
        for (int s = 1; s < nslab; s += 3) {
            double t[3][NR];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int r = 0; r < NR; ++r) t[u][r] = (bok[r] && s + u < nslab) ? bp[r][(long)(s + u) * bslab] : 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) rb[r] += (t[0][r] + t[1][r]) + t[2][r];
        }
        for (int s = 1; s < anslab; s += 3) {
            double t[3][NR];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int r = 0; r < NR; ++r) t[u][r] = (aok[r] && s + u < anslab) ? ap[r][(long)(s + u) * aslab] : 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) ra[r] += (t[0][r] + t[1][r]) + t[2][r];
        }
    };

    vg_d4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (vg_d4){0.0, 0.0, 0.0, 0.0};

    const int fi = lane & 15, fk = lane >> 4;
    constexpr int WT = T / 2, WTC = T / WC;                    // rows / cols per wave
    auto ktile = [&](double (&ra)[NR], double (&rb)[NR], int knext) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            As[a_i[r] * a_si + a_k[r] * a_sk] = ra[r];
            Bs[b_k[r] * b_sk + b_j[r] * b_sj] = rb[r];
        }
        __syncthreads();
        if (knext < k_end) load_tile(knext, ra, rb);           // refill this stage: two k-tiles ahead
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            double av[MB], bv[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                av[mb] = As[(wr * WT + mb * 16 + fi) * a_si + (kk + fk) * a_sk];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bv[nb] = Bs[(kk + fk) * b_sk + (wc * WTC + nb * 16 + fi) * b_sj];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
        }
        __syncthreads();
    };
    // Interior tiles with whole k-tiles and plain operands (every hot shape of the step) take a branch-free loop: element
    // pointers advance by one add per k-tile and the loads carry no predicate.  (The generic loop's `ok ? *p : 0` loads
    // compile to an exec-mask branch region each -- ~150 branches per k-tile pair -- and bound it by instruction issue.)
    // (slab operands: only for the short m x m x m products -- measured: the long split-K consumer [C;C1] = [B1;V1] S is
    // faster through the generic loop, 22 vs 31 us)
    const bool fast = row0 + T <= M && col0 + T <= N && k_begin < k_end && ((k_end - k_begin) % BK) == 0 &&
                      ((nslab == 1 && anslab == 1) || p.K <= 512);
    if (fast) {
        const double* pa[NR];
        const double* pb[NR];
        int la[NR], lb[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            pa[r] = A + (long)(row0 + a_i[r]) * sa_m + (long)(k_begin + a_k[r]) * sa_k;
            pb[r] = B + (long)(k_begin + b_k[r]) * sb_k + (long)(col0 + b_j[r]) * sb_n;
            la[r] = a_i[r] * a_si + a_k[r] * a_sk;
            lb[r] = b_k[r] * b_sk + b_j[r] * b_sj;
        }
        const long da = (long)BK * sa_k, db = (long)BK * sb_k;
        const int nk = (k_end - k_begin) / BK;
        auto ld = [&](double (&ra)[NR], double (&rb)[NR]) {
#pragma unroll
            for (int r = 0; r < NR; ++r) { ra[r] = *pa[r]; rb[r] = *pb[r]; }
            // operands given as split-K slabs of their producer: three further slabs (12 loads) in flight per round trip
            for (int sl = 1; sl < nslab; sl += 3) {
                const long o0 = (long)sl * bslab, o1 = sl + 1 < nslab ? o0 + bslab : o0, o2 = sl + 2 < nslab ? o0 + 2 * bslab : o0;
                const double w1 = sl + 1 < nslab ? 1.0 : 0.0, w2 = sl + 2 < nslab ? 1.0 : 0.0;      // uniform
#pragma unroll
                for (int r = 0; r < NR; ++r) rb[r] += (pb[r][o0] + w1 * pb[r][o1]) + w2 * pb[r][o2];
            }
            for (int sl = 1; sl < anslab; sl += 3) {
                const long o0 = (long)sl * aslab, o1 = sl + 1 < anslab ? o0 + aslab : o0, o2 = sl + 2 < anslab ? o0 + 2 * aslab : o0;
                const double w1 = sl + 1 < anslab ? 1.0 : 0.0, w2 = sl + 2 < anslab ? 1.0 : 0.0;
#pragma unroll
                for (int r = 0; r < NR; ++r) ra[r] += (pa[r][o0] + w1 * pa[r][o1]) + w2 * pa[r][o2];
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) { pa[r] += da; pb[r] += db; }
        };
        auto kt = [&](double (&ra)[NR], double (&rb)[NR], bool more) {
#pragma unroll
            for (int r = 0; r < NR; ++r) { As[la[r]] = ra[r]; Bs[lb[r]] = rb[r]; }
            __syncthreads();
            if (more) ld(ra, rb);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                double av[MB], bv[NB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[mb] = As[(wr * WT + mb * 16 + fi) * a_si + (kk + fk) * a_sk];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[nb] = Bs[(kk + fk) * b_sk + (wc * WTC + nb * 16 + fi) * b_sj];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
            }
            __syncthreads();
        };
        ld(ra0, rb0);
        if (nk > 1) ld(ra1, rb1);
        for (int it = 0; it < nk; it += 2) {
            kt(ra0, rb0, it + 2 < nk);
            if (it + 1 < nk) kt(ra1, rb1, it + 3 < nk);
        }
    } else {
        if (k_begin < k_end) load_tile(k_begin, ra0, rb0);
        if (k_begin + BK < k_end) load_tile(k_begin + BK, ra1, rb1);
        for (int k0 = k_begin; k0 < k_end; k0 += 2 * BK) {
            ktile(ra0, rb0, k0 + 2 * BK);
            if (k0 + BK < k_end) ktile(ra1, rb1, k0 + 3 * BK);
        }
    }

    double* __restrict__ C = p.C ? p.C + (long)ks * p.c_slab : nullptr;
    const int ldc = p.ldc;
    const double* __restrict__ dotw = p.dotw;
    double dsum = 0.0;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wr * WT + mb * 16 + fk + 4 * r;
                const int col = col0 + wc * WTC + nb * 16 + fi;
                if (row < M && col < N) {
                    const double v = p.alpha * acc[mb][nb][r];
                    if (dotw) dsum += v * dotw[(long)row * p.dotw_ld + col];
                    if (C) {
                        double* cp = C + (long)row * ldc + col;
                        *cp = p.accum ? *cp + v : v;
                    }
                }
            }
    if (p.dot_out) {          // reduction epilogue: lanes -> wave (butterfly) -> workgroup (fixed order) -> one word per tile
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        if (lane == 0) lds[wave] = dsum;            // (the k-loop ended with a barrier: the operand tiles are free)
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < NT / 64; ++w) tot += lds[w];
            p.dot_out[bid - p.tile_start] = tot;
        }
    }
}

