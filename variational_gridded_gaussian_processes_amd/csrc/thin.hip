// "Thin" tail of the ELBO step for numerically rank-deficient Gram matrices (RBF factors: numerical rank ~17 of 128).
//
// The collapsed bound of kronecker_structure.py:249-278 in the eigenbasis of G_d = B_d B_d^T (spec
// oracle/kron.py finish()) only ever sees the NULL eigen-directions of G_d through traces:
//   * D = 1 + lam1 lam2 / v equals 1 whenever one of the two directions is null  ->  sum log D, sum a/D run over range x range;
//   * P = Q1^T (B1 Y^T B2^T) Q2 vanishes on null rows / columns (q^T B = 0 for q in null(B B^T)), and so does beta = P / D;
//   * f_i = (Q^T (H + H^T) Q)_ii vanishes on null directions (H = V B^T), so sum_i f_i r_i = sum_range f_i (r_i - tr) + tr * tr(H + H^T);
//   * sum lam = tr G.
// So the value and the 5-component gradient need the r leading eigenpairs only (r = numerical rank + a margin), and those come
// straight out of the subspace start's Rayleigh-Ritz problem: with V1 (r x m, orthonormal rows spanning range(G)) and the Ritz
// pairs W, theta of V1 G V1^T, the range eigenvectors are E_r = W V1 and every rotated quantity is  W (V1 . V1^T) W^T  of an
// r x r matrix.  The step then needs no full eigendecomposition, no complement basis and no m x m rotation at all
// (tools/studies/thin_rbf_study.py: ELBO 1e-14, gradient 2e-10 against the full evaluation at 1024^2, m = 128, r = 17).
//
// This kernel is the whole m-space stage of such a step in ONE workgroup: the r x r rotations on the matrix cores (operands in
// LDS), the D-stage, the four gradient contractions, the final combination, the subspace-miss check (tr G - sum theta must be
// numerically zero) and the 128-byte burst into the pinned host block.  r1, r2 <= 32.
#include "common.h"

typedef double vt_d4 __attribute__((ext_vector_type(4)));

#define VT_MAXR 32
#define VT_NBUF 15
#define VT_NBIG 11         // sums over the r1 x r2 elements (one element per thread): block-wide reduction
#define VT_NSMALL 14       // sums over <= 32 range directions / <= 128 diagonal elements: one wave
#ifdef VGGP_DIAG
#define VT_STAMP(i) do { if (threadIdx.x == 0 && tt.stamps) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tt.stamps[i] = t_; } } while (0)
#else
#define VT_STAMP(i)
#endif

struct VtTask { int c, a, sa_i, sa_k, b; };      // buffer ids: C = op(A) B, all mp x mp (row stride ld), B row-major

// up to 16 independent products, one 16 x 16 output block per wave and turn (v_mfma_f64_16x16x4: A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], D[row = (lane >> 4) + 4 r][col = lane & 15])
// (kk: the operands are zero beyond the first kk columns / rows -- kk = r rounded up to 8 -- so the reduction stops there)
__device__ __forceinline__ void vt_mm(double* base, int sz, const VtTask* tk, int ntask, int mp, int ld, int kk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, nb = mp >> 4;
    const int fi = lane & 15, fk = lane >> 4;
    for (int t = wave; t < ntask * nb * nb; t += nw) {
        const int task = t / (nb * nb), blk = t - task * nb * nb, bi = blk / nb, bj = blk - bi * nb;
        const VtTask T = tk[task];
        const double* a = base + (long)T.a * sz + (bi * 16 + fi) * T.sa_i + fk * T.sa_k;
        const double* b = base + (long)T.b * sz + fk * ld + bj * 16 + fi;
        double* C = base + (long)T.c * sz;
        vt_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        double av[8], bv[8];                      // all fragments of the block in flight before the first MFMA (kk <= 32)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = 4 * u < kk;
            av[u] = on ? a[(4 * u) * T.sa_k] : 0.0;
            bv[u] = on ? b[(4 * u) * ld] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            if (4 * u < kk) {                     // (wave-uniform)
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bv[u + 1], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(bi * 16 + fk + 4 * r) * ld + bj * 16 + fi] = acc0[r] + acc1[r];
    }
    __syncthreads();
}

// wave-level butterfly of NV values (the NV shuffles of a step issued together)
template <int NV>
__device__ __forceinline__ void vt_wave_sum(double (&v)[NV]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double t[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) t[q] = __shfl_xor(v[q], off);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] += t[q];
    }
}

// buffer ids
enum { B_W1 = 0, B_W1T, B_W2, B_W2T, B_E1, B_F1, B_E2, B_F2, B_P0, B_P1, B_P2, B_T0, B_T1, B_T2, B_T3 };

__global__ __launch_bounds__(1024) void vg_thin_tail_kernel(const VgThinTail tt) {
    extern __shared__ __attribute__((aligned(16))) double vt_dyn[];
    __shared__ double red[16 * VT_NBIG], tot[VT_NBIG + VT_NSMALL];
    __shared__ double dg[4][256];                       // diagonals of G0_1, H0_1, G0_2, H0_2 (traces over ALL directions)
    __shared__ double l1s[VT_MAXR], l2s[VT_MAXR], rs1[VT_MAXR], rl1[VT_MAXR], rs2[VT_MAXR], rl2[VT_MAXR];
    __shared__ VtTask tk[18];
    __shared__ double stage[16];
    __shared__ double jit_s[2], pf_s, seq_s;
    __shared__ int rc_s[2][4], st_s[2][2];
    const int r1 = tt.r1, r2 = tt.r2, tid = threadIdx.x, nthr = blockDim.x;
    const int rmax = r1 > r2 ? r1 : r2, mp = (rmax + 15) & ~15, ld = mp + 2, sz = mp * ld;
    double* base = vt_dyn;
    const double s1 = tt.theta[2], s2 = tt.theta[3], v = tt.theta[4];

    VT_STAMP(0);
    // ---- phase A: one batch of loads, zero padded into LDS; W also transposed -----------------------------------------------
    for (int e = tid; e < sz; e += nthr) {
        const int i = e / ld, j = e - i * ld;
        const bool in1 = i < r1 && j < r1, in2 = i < r2 && j < r2, in12 = i < r1 && j < r2;
        const double w1 = in1 ? tt.W1[i * r1 + j] : 0.0, w2 = in2 ? tt.W2[i * r2 + j] : 0.0;
        const double am1 = in1 ? tt.AM1[i * r1 + j] : 0.0, ah1 = in1 ? tt.AH1[i * r1 + j] : 0.0;
        const double am2 = in2 ? tt.AM2[i * r2 + j] : 0.0, ah2 = in2 ? tt.AH2[i * r2 + j] : 0.0;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if (in12)
            for (int sl = 0; sl < tt.ac_nslab; ++sl) {          // (fixed order: bitwise reproducible)
                const double* ac = tt.AC + sl * tt.ac_slab + i * r2 + j;
                c0 += ac[0]; c1 += ac[(long)r1 * r2]; c2 += ac[2L * r1 * r2];
            }
        base[B_W1 * sz + e] = w1; base[B_W2 * sz + e] = w2;
        if (j < mp) { base[B_W1T * sz + j * ld + i] = w1; base[B_W2T * sz + j * ld + i] = w2; }
        base[B_E1 * sz + e] = am1; base[B_F1 * sz + e] = ah1; base[B_E2 * sz + e] = am2; base[B_F2 * sz + e] = ah2;
        base[B_P0 * sz + e] = c0; base[B_P1 * sz + e] = c1; base[B_P2 * sz + e] = c2;
    }
    // (the two padding columns of the transposed copies are never read: fragments stop at column mp - 1)
    if (tid < VT_MAXR) {
        l1s[tid] = tid < r1 ? s1 * tt.lam1[tid] : 0.0;
        l2s[tid] = tid < r2 ? s2 * tt.lam2[tid] : 0.0;
    }
    // the step's diagnostics: every word by a lane of its own, in the same batch of loads (a lone lane walking them at the end
    // pays one dependent round trip per word)
    if (tid >= 64 && tid < 72) { const int k = (tid - 64) >> 2, q = (tid - 64) & 3; rc_s[k][q] = tt.rcounters[k] ? tt.rcounters[k][q] : 0; }
    if (tid >= 72 && tid < 76) { const int k = (tid - 72) >> 1, q = (tid - 72) & 1; st_s[k][q] = tt.status[k] ? tt.status[k][q] : 0; }
    if (tid >= 76 && tid < 78) jit_s[tid - 76] = tt.jit[tid - 76] ? *tt.jit[tid - 76] : 0.0;
    if (tid == 78) pf_s = tt.peer_fail ? *tt.peer_fail : 0.0;
    if (tid == 79) seq_s = tt.theta[5];
    if (tid >= 128 && tid < 384) {                       // (m <= 256: one diagonal element per lane and matrix)
        const int i = tid - 128;
        dg[0][i] = i < tt.m1 ? tt.G1[(long)i * tt.m1 + i] : 0.0; dg[1][i] = i < tt.m1 ? tt.H1[(long)i * tt.m1 + i] : 0.0;
        dg[2][i] = i < tt.m2 ? tt.G2[(long)i * tt.m2 + i] : 0.0; dg[3][i] = i < tt.m2 ? tt.H2[(long)i * tt.m2 + i] : 0.0;
    }
    // every product of the kernel, tabulated once: C = op(A) B
    if (tid == 0) {
        tk[0] = VtTask{B_T0, B_E1, ld, 1, B_W1T}; tk[1] = VtTask{B_T1, B_F1, ld, 1, B_W1T};       // B1: (AM, AH) W^T
        tk[2] = VtTask{B_T2, B_E2, ld, 1, B_W2T}; tk[3] = VtTask{B_T3, B_F2, ld, 1, B_W2T};
        tk[4] = VtTask{B_E1, B_W1, ld, 1, B_T0}; tk[5] = VtTask{B_F1, B_W1, ld, 1, B_T1};         // B2: W (.)
        tk[6] = VtTask{B_E2, B_W2, ld, 1, B_T2}; tk[7] = VtTask{B_F2, B_W2, ld, 1, B_T3};
        tk[8] = VtTask{B_T0, B_P0, ld, 1, B_W2T}; tk[9] = VtTask{B_T1, B_P1, ld, 1, B_W2T}; tk[10] = VtTask{B_T2, B_P2, ld, 1, B_W2T};   // C1
        tk[11] = VtTask{B_P0, B_W1, ld, 1, B_T0}; tk[12] = VtTask{B_P1, B_W1, ld, 1, B_T1}; tk[13] = VtTask{B_P2, B_W1, ld, 1, B_T2};    // C2
        tk[14] = VtTask{B_W1, B_T0, ld, 1, B_T1}; tk[15] = VtTask{B_W1T, B_T2, ld, 1, B_T1};      // E: beta beta^T, (beta lam2) beta^T,
        tk[16] = VtTask{B_W2, B_T0, 1, ld, B_T0}; tk[17] = VtTask{B_P0, B_T3, 1, ld, B_T0};       //    beta^T beta, (lam1 beta)^T beta
    }
    VT_STAMP(1);
    __syncthreads();
    VT_STAMP(2);

    // ---- phase B: E_d = W_d AM_d W_d^T, F_d = W_d AH_d W_d^T ------------------------------------------------------------------
    const int kk = (rmax + 7) & ~7;
    vt_mm(base, sz, tk, 4, mp, ld, kk);
    vt_mm(base, sz, tk + 4, 4, mp, ld, kk);
    VT_STAMP(3);
    // ---- phase C: P_q = W_1 AC_q W_2^T ------------------------------------------------------------------------------------------
    vt_mm(base, sz, tk + 8, 3, mp, ld, kk);
    vt_mm(base, sz, tk + 11, 3, mp, ld, kk);
    VT_STAMP(4);

    // ---- phase D: D-stage over r1 x r2 (mspace.hip vg_dstage_kernel restricted to the range) ----------------------------------
    // beta -> T0, beta^T -> T1, beta lam2 -> T2, lam1 beta -> T3, (1/D - 1) -> W2T (the W buffers are free now)
    const double rs = sqrt(s1 * s2), iv = 1.0 / v;
    double S[VT_NBIG];
#pragma unroll
    for (int q = 0; q < VT_NBIG; ++q) S[q] = 0.0;
    // (element work by the first four waves only: the block-wide reduction below then costs 4 butterflies instead of 16, and
    //  the butterflies -- ds_bpermute traffic -- are what it costs)
    if (tid < 256)
    for (int e = tid; e < mp * mp; e += 256) {
        const int i = e / mp, j = e - i * mp;
        double b = 0.0, bl2 = 0.0, bl1 = 0.0, w = 0.0;
        if (i < r1 && j < r2) {
            const double l1 = l1s[i], l2 = l2s[j];
            const double a = l1 * l2 * iv, D = 1.0 + a, iD = 1.0 / D;
            const double P = rs * base[B_P0 * sz + i * ld + j], P1 = rs * base[B_P1 * sz + i * ld + j], P2 = rs * base[B_P2 * sz + i * ld + j];
            b = P * iD; bl2 = b * l2; bl1 = b * l1; w = -a * iD;          // w = 1/D - 1
            S[0] += log1p(a); S[1] += P * b; S[2] += b * b; S[3] += a * iD; S[4] += b * b * (2.0 + a); S[5] += b * P1; S[6] += b * P2;
        }
        if (tt.beta_out && i < r1 && j < r2) {                   // the range block of beta and 1/D: all q(v) / posterior(x*) need of the step
            tt.beta_out[i * r2 + j] = b;
            tt.invd_out[i * r2 + j] = 1.0 + w;
        }
        base[B_T0 * sz + i * ld + j] = b; base[B_T1 * sz + j * ld + i] = b;
        base[B_T2 * sz + i * ld + j] = bl2; base[B_T3 * sz + i * ld + j] = bl1;
        base[B_W2T * sz + i * ld + j] = w;
    }
    __syncthreads();
    // row / column sums of (1/D - 1) and of lam_other (1/D - 1)
    if (tid < VT_MAXR) {
        double a0 = 0.0, a1 = 0.0;
        if (tid < r1) for (int j = 0; j < r2; ++j) { const double w = base[B_W2T * sz + tid * ld + j]; a0 += w; a1 += l2s[j] * w; }
        rs1[tid] = a0; rl1[tid] = a1;
    } else if (tid >= 64 && tid < 64 + VT_MAXR) {
        const int j = tid - 64;
        double a0 = 0.0, a1 = 0.0;
        if (j < r2) for (int i = 0; i < r1; ++i) { const double w = base[B_W2T * sz + i * ld + j]; a0 += w; a1 += l1s[i] * w; }
        rs2[j] = a0; rl2[j] = a1;
    }
    VT_STAMP(5);
    // ---- phase E: X1 = beta beta^T -> W1, X1l = (beta lam2) beta^T -> W1T, X2 = beta^T beta -> W2, X2l = (lam1 beta)^T beta -> P0
    vt_mm(base, sz, tk + 14, 4, mp, ld, kk);     // (ends with a barrier: rs*/rl* are visible afterwards too)
    VT_STAMP(6);
    // ---- phase F: contractions --------------------------------------------------------------------------------------------------
    if (tid < 256)
    for (int e = tid; e < mp * mp; e += 256) {
        const int i = e / mp, j = e - i * mp, o = i * ld + j;
        S[7] += base[B_E1 * sz + o] * base[B_W1 * sz + o];            // EX1  = sum E1 o beta beta^T
        S[8] += base[B_F1 * sz + o] * base[B_W1T * sz + o];           // FX1u = sum F1 o (beta lam2) beta^T
        S[9] += base[B_E2 * sz + o] * base[B_W2 * sz + o];            // EX2
        S[10] += base[B_F2 * sz + o] * base[B_P0 * sz + o];           // FX2u = sum F2 o (lam1 beta)^T beta
    }
    VT_STAMP(7);
    if (tid < 256) {
        vt_wave_sum<VT_NBIG>(S);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < VT_NBIG; ++q) red[(tid >> 6) * VT_NBIG + q] = S[q];
        }
    }
    // the sums over range directions and the four traces: wave 1 alone (lane = direction; diagonal elements lane, lane + 64)
    if (tid >= 64 && tid < 128) {
        const int l = tid - 64;
        double T[VT_NSMALL];
#pragma unroll
        for (int q = 0; q < VT_NSMALL; ++q) T[q] = 0.0;
        if (l < r1) {
            const double e1 = base[B_E1 * sz + l * ld + l], f1 = base[B_F1 * sz + l * ld + l];
            T[0] = e1 * rs1[l]; T[1] = f1 * rl1[l]; T[2] = e1 * l1s[l]; T[3] = l1s[l];
            T[8] = l1s[l] > VG_EIG_RANK_CUT * l1s[0] ? 1.0 : 0.0;        // numerical rank (ratios: the outputscale does not matter)
        }
        if (l < r2) {
            const double e2 = base[B_E2 * sz + l * ld + l], f2 = base[B_F2 * sz + l * ld + l];
            T[4] = e2 * rs2[l]; T[5] = f2 * rl2[l]; T[6] = e2 * l2s[l]; T[7] = l2s[l];
            T[9] = l2s[l] > VG_EIG_RANK_CUT * l2s[0] ? 1.0 : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) T[10 + q] = (dg[q][l] + dg[q][l + 64]) + (dg[q][l + 128] + dg[q][l + 192]);
        vt_wave_sum<VT_NSMALL>(T);
        if (l == 0) {
#pragma unroll
            for (int q = 0; q < VT_NSMALL; ++q) tot[VT_NBIG + q] = T[q];
        }
    }
    __syncthreads();
    if (tid < VT_NBIG) {
        double t = 0.0;
        for (int w = 0; w < 4; ++w) t += red[w * VT_NBIG + tid];
        tot[tid] = t;
    }
    __syncthreads();
    VT_STAMP(8);

    // ---- phase G: final combination (mspace.hip vg_final_kernel with the range-only sums) ---------------------------------------
    if (tid == 0) {
        double S[VT_NBIG + 10], tr[4];
#pragma unroll
        for (int q = 0; q < VT_NBIG; ++q) S[q] = tot[q];
        // S[11..13] / [15..17]: sum e r, sum f r_lam, sum e lam of dimension 1 / 2; S[14], S[18]: sums of the scaled Ritz values;
        // S[19], S[20]: numerical ranks
        S[11] = tot[VT_NBIG + 0]; S[12] = tot[VT_NBIG + 1]; S[13] = tot[VT_NBIG + 2]; S[14] = tot[VT_NBIG + 3];
        S[15] = tot[VT_NBIG + 4]; S[16] = tot[VT_NBIG + 5]; S[17] = tot[VT_NBIG + 6]; S[18] = tot[VT_NBIG + 7];
        S[19] = tot[VT_NBIG + 8]; S[20] = tot[VT_NBIG + 9];
#pragma unroll
        for (int q = 0; q < 4; ++q) tr[q] = tot[VT_NBIG + 10 + q];
        const double N = tt.n_total, yy = tt.yy;
        const double is1 = 1.0 / s1, is2 = 1.0 / s2, iv2 = iv * iv, hiv = 0.5 * iv;
        const double sl1 = s1 * tr[0], sl2 = s2 * tr[2];             // sum of ALL eigenvalues = trace
        const double tF1 = 2.0 * s1 * tr[1], tF2 = 2.0 * s2 * tr[3]; // tr(Q^T (H + H^T) Q) = 2 tr H
        const double Nss = N * s1 * s2, sll = sl1 * sl2;
        const double elbo = -0.5 * (N * 1.8378770664093453 + N * log(v) + S[0] + yy * iv - S[1] * iv2) - (Nss - sll) * hiv;
        const double quad1 = 2.0 * S[5] - S[7] - 2.0 * s1 * S[8] * iv;
        const double quad2 = 2.0 * S[6] - S[9] - 2.0 * s2 * S[10] * iv;
        // sum_i ff_i r_lam_i over ALL i = sum_range ff_i (r_lam_i - sl_other) + sl_other * tF, and r_lam_i - sl_other = rl_i
        const double g_l1 = -0.5 * (S[11] + (2.0 * s1 * S[12] + sl2 * tF1) * iv - quad1 * iv2) + sl2 * hiv * (tF1 - S[13]);
        const double g_l2 = -0.5 * (S[15] + (2.0 * s2 * S[16] + sl1 * tF2) * iv - quad2 * iv2) + sl1 * hiv * (tF2 - S[17]);
        const double common = -0.5 * (S[3] - S[2] * iv2);
        const double g_s1 = common * is1 + sll * hiv * is1 - N * s2 * hiv;
        const double g_s2 = common * is2 + sll * hiv * is2 - N * s1 * hiv;
        const double g_v = -0.5 * (N * iv - S[3] * iv - yy * iv2 + S[4] * iv2 * iv) + (Nss - sll) * 0.5 * iv2;
        const double o[6] = {elbo, g_l1, g_l2, g_s1, g_s2, g_v};
#pragma unroll
        for (int q = 0; q < 6; ++q) { if (tt.out) tt.out[q] = o[q]; stage[q] = o[q]; }
        stage[6] = pf_s;
        stage[7] = 0.0;
        // numerical ranks (the host sizes the next step's subspace from them) and the subspace-miss check: tr G - sum theta is
        // the sum of the Rayleigh quotients of ANY orthonormal basis of the complement of span(V1) -- it must be numerically zero
        int* ired = reinterpret_cast<int*>(stage + 10);
        for (int k = 0; k < 2; ++k) {
            const int r = k ? r2 : r1, m = k ? tt.m2 : tt.m1;
            // (S[14], S[18] are sums of the SCALED Ritz values: compare with the scaled trace)
            const double lmax = k ? l2s[0] : l1s[0], trG = k ? sl2 : sl1, sth = k ? S[18] : S[14];
            const int nrank = (int)(k ? S[20] : S[19]);
            // ... and small enough for the BOUND: to first order the directions left out would add (tr G_d - sum theta_d) tr G_other / v
            // to sum log D -- negligible on the headline (1e-16 lam_max eigenvalues), not where the spectrum decays slowly into the
            // cut (m_d = 24 after a jump: 3e-4 of an ELBO of 1.8e4, found by test_rbf_warm_chain_at_small_inducing_counts)
            const double first_order = fabs(trG - sth) * (k ? sl1 : sl2) * iv;
            const bool miss = !(fabs(trG - sth) <= VG_THIN_MISS * lmax * (double)(m - r + 1)) || !(first_order <= VG_THIN_ELBO_TOL * N);
            const int s0 = st_s[k][0], s1w = st_s[k][1], rstat = rc_s[k][2];
            ired[k * 4 + 0] = rc_s[k][0];                        // rotation rounds of the Ritz solve (0 on the Newton path)
            ired[k * 4 + 1] = (rc_s[k][1] & 0xff) | (nrank << 8);
            ired[k * 4 + 2] = rstat;
            ired[k * 4 + 3] = 0;
            ired[8 + k] = s0 ? s0 : (rstat ? rstat : ((s1w & 1) ? VGGP_ENOCONV : ((miss || (s1w & 2)) ? VG_ESUBMISS : 0)));
        }
        stage[8] = jit_s[0];
        stage[9] = jit_s[1];
        stage[15] = seq_s;
    }
    VT_STAMP(9);
    __syncthreads();
    static_assert(sizeof(VgHostOut) == 16 * 8, "VgHostOut layout");
    if (tt.hout && tid < 16) reinterpret_cast<double*>(tt.hout)[tid] = stage[tid];
    VT_STAMP(10);
}

size_t vg_thin_tail_lds(int r1, int r2) {
    const int rmax = r1 > r2 ? r1 : r2, mp = (rmax + 15) & ~15;
    return (size_t)VT_NBUF * mp * (mp + 2) * sizeof(double);
}

__global__ void vg_pivchol_kernel(const VgPivCholArgs a);
hipError_t vg_thin_tail_setup() {          // per device, at vggp_create: the dynamic-LDS ceilings of this file's kernels
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vg_pivchol_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(256 * 65 * sizeof(double)));      // m <= 256 rows x (r <= 64) + 1 columns
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(vg_thin_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)vg_thin_tail_lds(VT_MAXR, VT_MAXR));
}

hipError_t vg_thin_tail_launch(const VgThinTail* tt, hipStream_t st) {
    if (tt->r1 < 1 || tt->r2 < 1 || tt->r1 > VT_MAXR || tt->r2 > VT_MAXR || tt->r1 > tt->m1 || tt->r2 > tt->m2 || tt->m1 > 256 ||
        tt->m2 > 256)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(vg_thin_tail_kernel, dim3(1), dim3(1024), vg_thin_tail_lds(tt->r1, tt->r2), st, *tt);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Cold range finder of the thin chain (api.hip, vg_cold_thin_prepass): r steps of diagonally PIVOTED Cholesky of the Gram matrix
// G (m x m, numerical rank < r): G ~ sum_i l_i l_i^T, pivot = largest residual diagonal.  The columns l_i span range(G) and are
// GRADED -- l_i has zeros at the earlier pivots and norm^2 <= the i-th largest residual -- which is what the row-by-row
// orthonormalisation that follows needs to keep the small range directions apart from the large ones (a random sketch Omega G
// does not have it: every one of its rows is dominated by lam_1, and so is V1 G for an ungraded basis V1).  The residual update
// G - sum l l^T itself cancels, so directions below ~1e-8 lam_1 come out with few digits: the thin chain's own pass "Z = V G"
// (G annihilates what is not range) repairs that, and its miss check decides.  One workgroup of 256 threads per matrix, thread = row;
// a step is one argmax (wave butterflies + one LDS exchange) and one column update with L in LDS: ~1 us.  Steps whose pivot is not
// positive (the residual is rounding noise: rank < r) take the fixed pseudo-random row instead.  Output: the r columns as the ROWS
// of V (r x m).
#define VP_MAXM 256
__global__ __launch_bounds__(256) void vg_pivchol_kernel(const VgPivCholArgs a) {
    const VgPivCholJob j = a.job[blockIdx.x];
    extern __shared__ double vp_dyn[];
    double* L = vp_dyn;                                   // [m][r + 1]
    __shared__ double wmax[4];
    __shared__ int warg[4];
    const int tid = threadIdx.x, m = j.m, r = j.r, ld = r + 1;
    const bool own = tid < m;
    double d = own ? j.G[(long)tid * m + tid] : -1.0;     // residual diagonal of this thread's row
    const double d0_floor = 0.0;
    for (int i = 0; i < r; ++i) {
        // argmax of the residual diagonal
        double v = own ? d : -1.0;
        int arg = tid;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off);
            const int oa = __shfl_xor(arg, off);
            if (ov > v || (ov == v && oa < arg)) { v = ov; arg = oa; }
        }
        if ((tid & 63) == 0) { wmax[tid >> 6] = v; warg[tid >> 6] = arg; }
        __syncthreads();
        double best = wmax[0];
        int p = warg[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (wmax[w] > best || (wmax[w] == best && warg[w] < p)) { best = wmax[w]; p = warg[w]; }
        double li = 0.0;
        if (best > d0_floor) {
            if (own) {
                double cval = j.G[(long)p * m + tid];     // column p = row p (symmetric): coalesced
                for (int q = 0; q < i; ++q) cval -= L[tid * ld + q] * L[p * ld + q];
                li = cval / sqrt(best);
                if (tid == p) li = sqrt(best);
            }
        } else if (own) {
            li = j.Omega[(long)i * m + tid] * 1e-30;      // (rank < r: an arbitrary direction, scaled out of the way of the residuals)
        }
        __syncthreads();                                  // every thread has read row p of L before anyone writes column i
        if (own) {
            L[tid * ld + i] = li;
            d -= li * li;
            if (tid == p) d = -1.0;                       // a pivot is used once
            j.V[(long)i * m + tid] = (best > d0_floor) ? li : j.Omega[(long)i * m + tid];
        }
        __syncthreads();
    }
}

hipError_t vg_pivchol_launch(const VgPivCholJob* jobs, int njobs, hipStream_t st) {
    if (njobs < 1 || njobs > 2) return hipErrorInvalidValue;
    VgPivCholArgs a;
    size_t lds = 0;
    for (int i = 0; i < njobs; ++i) {
        if (jobs[i].m < 1 || jobs[i].m > VP_MAXM || jobs[i].r < 1 || jobs[i].r > jobs[i].m || jobs[i].r > 64) return hipErrorInvalidValue;
        a.job[i] = jobs[i];
        lds = std::max(lds, (size_t)jobs[i].m * (jobs[i].r + 1) * sizeof(double));
    }
    hipLaunchKernelGGL(vg_pivchol_kernel, dim3(njobs), dim3(256), lds, st, a);
    return hipGetLastError();
}
