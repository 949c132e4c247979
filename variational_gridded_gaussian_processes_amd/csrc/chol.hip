// Cholesky factorisation with the psd_safe_cholesky jitter schedule and the explicit inverse of the factor.
//
// Replaces the Cholesky hidden inside lazify(Kuu).inv_matmul(Kuf) and MultivariateNormal.log_prob
// (kronecker_structure.py:269, :273) -- applied to the m_d x m_d per-dimension factors instead of the M x M /
// N x N dense matrices.  The explicit L^{-1} turns every later triangular solve into an MFMA GEMM (validated
// against substitution in DESIGN.md: <= 1e-8 relative on the ill-conditioned RBF factors with the 1e-8 jitter).
//
// Fast path (m <= 128), one workgroup of 1024 threads per (matrix, jitter level):
//   * the four jitter levels (0, 1e-8, 1e-7, 1e-6) are factored CONCURRENTLY by four workgroups; the lowest
//     level that succeeds wins (identical to trying them in order) and only the winner computes the inverse and
//     writes the outputs.  A near-singular RBF factor therefore costs one factorisation latency, not two.
//   * register-resident right-looking Cholesky: thread (ty, tx) of a 32 x 32 grid owns the trailing-matrix
//     elements (ty + 32a, tx + 32b) for the whole factorisation (4 waves per SIMD keep the f64 pipe busy: a lone
//     wave issues one f64 FMA per ~8 cycles); only the pivot column travels through LDS (packed lower
//     triangle), so a column step is one barrier + 2*MT LDS reads + MT^2 FMAs instead of a serial chain of LDS
//     read-modify-writes.
//   * L^{-1} by forward elimination on the identity, same ownership: per step the finished row k is broadcast
//     through a double-buffered LDS row, every thread updates its own elements in registers.
// Generic path (m > 128): the matrix lives in an L2-resident global scratch, one workgroup tries the levels in
// order (slow; only used by the building-block API for large matrices).
#include "gemm_body.h"

#include <cstdlib>

static const int VG_CHOL_FAST_MAX_M = 128;      // fast paths: matrix resident in the LDS of one CU
__constant__ double VG_JITTERS[4] = {0.0, 1e-8, 1e-7, 1e-6};

#define VG_CHOL_MAXJOBS 8
struct VgCholArgs {
    VgCholJob job[VG_CHOL_MAXJOBS];
    int njobs;
    int fast[VG_CHOL_MAXJOBS];
};

__device__ __forceinline__ int vg_ctri(int i) { return (i * (i + 1)) >> 1; }

// ---- fast path -----------------------------------------------------------------------------------------------
template <int MT>
__device__ void vg_chol_fast(const VgCholJob& J, int lvl, double* W, double* rsd, double* rowbuf, int* s_i) {
    const int m = J.m;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;      // 32 x 32 threads, each owns MT x MT elements
    const double jit = VG_JITTERS[lvl];
    const int ldk = J.ldk ? J.ldk : m, ldl = J.ldl ? J.ldl : m;
    int* flags = reinterpret_cast<int*>(J.scratch);          // [4] per matrix, zeroed by the CALLER before the launch

    double a[MT][MT];
    int trow[MT], tcol[MT];                                    // packed-row offsets of my rows / of my columns' rows
#pragma unroll
    for (int ia = 0; ia < MT; ++ia) {
        trow[ia] = vg_ctri(ty + 32 * ia);
        tcol[ia] = vg_ctri(tx + 32 * ia);
#pragma unroll
        for (int ib = 0; ib < MT; ++ib) {
            const int i = ty + 32 * ia, j = tx + 32 * ib;
            a[ia][ib] = (i < m && j <= i) ? J.K[(long)i * ldk + j] + (i == j ? jit : 0.0) : 0.0;
        }
    }

#ifdef VG_CHOL_STAMP
    unsigned long long tA = 0, tB = 0, tC = 0, s0, s1, tstart, tend;
#define CS(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    CS(tstart);
#else
#define CS(var)
#endif
    bool ok = true;
    int pivoff = 0;
    for (int k = 0; k < m; ++k) {
#ifdef VG_CHOL_STAMP
        CS(s0);
#endif
        const int kb = k >> 5, kx = k & 31;
        if (tx == kx) {                                       // owners of column k publish it (rows >= k)
            // kb is wave-uniform: a branch chain over compile-time column indices keeps a[][] statically indexed
            // (a select chain over all MT columns costs hundreds of cycles per step)
#define VG_PUB(KB)                                                                  \
    if (KB < MT && kb == KB) {                                                      \
        _Pragma("unroll") for (int ia = 0; ia < MT; ++ia)                           \
            if (ty + 32 * ia >= k) W[trow[ia] + k] = a[ia][KB < MT ? KB : 0];       \
    }
            VG_PUB(0) else VG_PUB(1) else VG_PUB(2) else VG_PUB(3)
#undef VG_PUB
        }
#ifdef VG_CHOL_STAMP
        CS(s1); tA += s1 - s0;
#endif
        __syncthreads();
#ifdef VG_CHOL_STAMP
        CS(s0); tB += s0 - s1;
#endif
        // all LDS reads of the step are issued together (pivot, my rows' and my columns' entries of column k).
        // No predication: for i <= k or j <= k the values read are stale/garbage, but they only ever touch
        // registers of already-published columns or of the unused upper triangle (L is taken from LDS, never from
        // these registers), so they are don't-cares.  The LDS tile is sized for the padded 16*MT matrix.
        const double piv = W[pivoff];
        pivoff += k + 2;                                      // tri(k+1) + (k+1)
        double li[MT], lj[MT];
#pragma unroll
        for (int ia = 0; ia < MT; ++ia) {
            li[ia] = W[trow[ia] + k];
            lj[ia] = W[tcol[ia] + k];
        }
        if (!(piv > 0.0) || !(piv < 1.0e300)) { ok = false; break; }      // uniform
        double rp = __builtin_amdgcn_rcp(piv);                // 1/piv: hardware seed + 2 Newton steps
        rp = fma(rp, fma(-piv, rp, 1.0), rp);
        rp = fma(rp, fma(-piv, rp, 1.0), rp);
#pragma unroll
        for (int ia = 0; ia < MT; ++ia) {
            const double lr = li[ia] * rp;
#pragma unroll
            for (int ib = 0; ib < MT; ++ib) a[ia][ib] -= lr * lj[ib];
        }
#ifdef VG_CHOL_STAMP
        CS(s1); tC += s1 - s0;
#endif
    }
#ifdef VG_CHOL_STAMP
    CS(tend);
    if ((tid & 63) == 0 && lvl == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(J.scratch) + 16 + (tid >> 6) * 4;
        dbg[0] = tA; dbg[1] = tB; dbg[2] = tC; dbg[3] = tend - tstart;
    }
#endif

    // ---- level selection: the lowest successful level wins (relaxed agent-scope flags, no payload) ----------
    __syncthreads();
    if (J.only_level0) {
        if (tid == 0) { s_i[0] = ok ? 1 : 0; s_i[1] = ok ? 0 : 1; }
    } else if (tid == 0) {
        __hip_atomic_store(&flags[lvl], ok ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int win = ok ? 1 : 0, spins = 0;
        bool all_failed = !ok;
        for (int l = 0; l < lvl; ++l) {
            int f;
            while ((f = __hip_atomic_load(&flags[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                if (++spins > (1 << 22)) { f = 2; break; }   // bounded: a missing sibling counts as failed
                __builtin_amdgcn_s_sleep(4);
            }
            if (f == 1) { win = 0; all_failed = false; }
        }
        s_i[0] = win;
        s_i[1] = (lvl == 3 && all_failed) ? 1 : 0;
    }
    __syncthreads();
    const bool winner = s_i[0] != 0, report_fail = s_i[1] != 0;
    if (report_fail) {
        if (tid == 0) { *J.status = VGGP_ENOTPD; if (J.jitter_out) *J.jitter_out = -1.0; }
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int idx = tid; idx < m * m; idx += blockDim.x) { J.L[(long)(idx / m) * ldl + idx % m] = qnan; J.Linv[idx] = qnan; }
    }
    if (!winner) return;
    if (tid == 0 && J.jitter_out) *J.jitter_out = jit;

    // ---- scale: L[i][k] = W[i][k] / sqrt(W[k][k]) ---------------------------------------------------------------
    for (int k = tid; k < m; k += blockDim.x) rsd[k] = 1.0 / sqrt(W[vg_ctri(k) + k]);      // 1 / L[k][k]
    __syncthreads();
    for (int i = ty; i < m; i += 32)
        for (int k = tx; k <= i; k += 32) {
            const double v = W[vg_ctri(i) + k] * rsd[k];
            W[vg_ctri(i) + k] = v;
            J.L[(long)i * ldl + k] = v;
        }
    for (int i = ty; i < m; i += 32)
        for (int k = tx; k < m; k += 32)
            if (k > i) J.L[(long)i * ldl + k] = 0.0;
    __syncthreads();

    // ---- inverse by forward elimination on the identity (registers hold X, lower triangular) --------------------
#pragma unroll
    for (int ia = 0; ia < MT; ++ia)
#pragma unroll
        for (int ib = 0; ib < MT; ++ib) a[ia][ib] = (ty + 32 * ia == tx + 32 * ib) ? 1.0 : 0.0;
    for (int k = 0; k < m; ++k) {
        const int ka = k >> 5, ky = k & 31;
        double* rb = rowbuf + (k & 1) * 192;
        if (ty == ky) {                                       // owners of row k: finish it and broadcast (zeros beyond k)
            const double rk = rsd[k];                         // 1 / L[k][k]
#define VG_PUBR(KA)                                                                 \
    if (KA < MT && ka == KA) {                                                      \
        _Pragma("unroll") for (int ib = 0; ib < MT; ++ib) {                         \
            a[KA < MT ? KA : 0][ib] *= rk;                                          \
            rb[tx + 32 * ib] = a[KA < MT ? KA : 0][ib];                             \
        }                                                                           \
    }
            VG_PUBR(0) else VG_PUBR(1) else VG_PUBR(2) else VG_PUBR(3)
#undef VG_PUBR
        }
        __syncthreads();
        double li[MT], xr[MT];
#pragma unroll
        for (int ia = 0; ia < MT; ++ia) {
            const int i = ty + 32 * ia;
            li[ia] = (i > k) ? W[trow[ia] + k] : 0.0;         // finished rows (i <= k) must stay untouched
            xr[ia] = rb[tx + 32 * ia];
        }
#pragma unroll
        for (int ia = 0; ia < MT; ++ia)
#pragma unroll
            for (int ib = 0; ib < MT; ++ib) a[ia][ib] -= li[ia] * xr[ib];
    }
#pragma unroll
    for (int ia = 0; ia < MT; ++ia)
#pragma unroll
        for (int ib = 0; ib < MT; ++ib) {
            const int i = ty + 32 * ia, j = tx + 32 * ib;
            if (i < m && j < m) J.Linv[(long)i * m + j] = (j <= i) ? a[ia][ib] : 0.0;
        }
}

// ---- generic path (any m <= 1024, matrix in global scratch, levels in order) --------------------------------------
__device__ void vg_chol_generic(const VgCholJob& J, double* sd) {
    const int m = J.m, ld = m + 1;
    double* W = J.scratch;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tx = tid & 31, ty = tid >> 5, nty = nthr >> 5;
    int used = -1;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const double jit = VG_JITTERS[attempt];
        for (int idx = tid; idx < m * m; idx += nthr) {
            const int i = idx / m, j = idx - i * m;
            W[i * ld + j] = J.K[idx] + (i == j ? jit : 0.0);
        }
        __threadfence_block();
        __syncthreads();
        bool ok = true;
        for (int k = 0; k < m; ++k) {
            const double piv = W[k * ld + k];
            if (!(piv > 0.0) || !(piv < 1.0e300)) { ok = false; break; }
            const double rp = 1.0 / piv;
            for (int i = k + 1 + ty; i < m; i += nty) {
                const double lik = W[i * ld + k] * rp;
                for (int j = k + 1 + tx; j <= i; j += 32) W[i * ld + j] -= lik * W[j * ld + k];
            }
            __threadfence_block();
            __syncthreads();
        }
        if (ok) { used = attempt; break; }
        __syncthreads();
    }
    if (used < 0) {
        if (tid == 0) { *J.status = VGGP_ENOTPD; if (J.jitter_out) *J.jitter_out = -1.0; }
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int idx = tid; idx < m * m; idx += nthr) { J.L[idx] = qnan; J.Linv[idx] = qnan; }
        return;
    }
    if (tid == 0 && J.jitter_out) *J.jitter_out = VG_JITTERS[used];
    for (int k = tid; k < m; k += nthr) sd[k] = sqrt(W[k * ld + k]);
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += nthr) {
        const int i = idx / m, k = idx - i * m;
        if (i > k) W[i * ld + k] /= sd[k];
        else if (i == k) W[i * ld + k] = sd[k];
    }
    __threadfence_block();
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += nthr) {
        const int i = idx / m, j = idx - i * m;
        J.L[idx] = (j <= i) ? W[i * ld + j] : 0.0;
    }
    // column j of L^{-1} kept in the (unused) upper triangle: 8 lanes split each forward-substitution dot product
    const int sub = tid & 7, grp = tid >> 3, ngrp = nthr >> 3;
    for (int jb = 0; jb < m; jb += ngrp) {
        const int j = jb + grp;
        if (j < m) {
            const double xj = 1.0 / W[j * ld + j];
            for (int i = j + 1; i < m; ++i) {
                double part = 0.0;
                for (int k = j + sub; k < i; k += 8) {
                    const double xk = (k == j) ? xj : W[j * ld + k];
                    part += W[i * ld + k] * xk;
                }
                part += __shfl_xor(part, 1);
                part += __shfl_xor(part, 2);
                part += __shfl_xor(part, 4);
                const double xi = -part / W[i * ld + i];
                if (sub == 0) W[j * ld + i] = xi;
                __threadfence_block();
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += nthr) {
        const int i = idx / m, j = idx - i * m;
        double v = 0.0;
        if (i == j) v = 1.0 / W[i * ld + i];
        else if (i > j) v = W[j * ld + i];
        J.Linv[idx] = v;
    }
}


// ==== MFMA-blocked path (m <= 128): left-looking Cholesky on 16-wide panels ==========================================
// One workgroup of 8 waves per (matrix, jitter level); the matrix (lower triangle, padded to 16 nb) lives in LDS.
// Per panel p:   (1) wave w forms the transposed Schur block U^T of block row bi = p + w with f64 MFMAs
//                    (U^T = K[bi][p]^T - L[p][:p] L[bi][:p]^T, accumulators only);
//                (2) wave 0 factors the 16 x 16 diagonal block: 16 pivot steps entirely inside the wave (no workgroup
//                    barrier; the pivot column and the pivot row of the running inverse travel through LDS), giving
//                    L11 and X11 = L11^{-1} (Gaussian elimination on [A | I], rows scaled by 1/sqrt(pivot) at the end);
//                (3) the other waves finish their block: L[bi][p]^T = X11 U^T -- U^T is still in the accumulator
//                    registers, whose layout is exactly the MFMA B operand, so this is 4 MFMAs and no LDS round trip.
// Two workgroup barriers per panel instead of one per column: the factorisation's critical path is the 16 in-wave pivot
// steps per panel.  The full inverse follows blockwise, X[bi][bj] = -X[bi][bi] sum_{bj<=bk<bi} L[bi][bk] X[bk][bj]; block
// column bj is owned by wave bj, so this phase needs no barrier at all.  X blocks live in the unused upper block triangle.
#define VG_CB 16
#define VG_CLD 132                         // LDS row stride (doubles): 16 rows x 4 k of an MFMA operand hit 32 distinct banks per half-wave
typedef double vg_cd4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double vg_crcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double vg_crsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = fma(y * 0.5, fma(-x * y, y, 1.0), y);
    y = fma(y * 0.5, fma(-x * y, y, 1.0), y);
    return y;
}
#define VG_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ void vg_chol_mfma(const VgCholJob& J, int lvl, double* Lm, double* Dinv, double* Db, double* colbuf, double* xrow,
                             double* sdv, int* s_i) {
    const int m = J.m, nb = (m + VG_CB - 1) / VG_CB, mp = nb * VG_CB;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fk = lane >> 4;
    const double jit = VG_JITTERS[lvl];
    const int ldk = J.ldk ? J.ldk : m, ldl = J.ldl ? J.ldl : m;
    int* flags = reinterpret_cast<int*>(J.scratch);          // [4] per matrix, zeroed by the CALLER before the launch
#ifdef VG_CHOL_STAMP
    unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tF = 0, tU = 0, tS = 0, q0, q1;
#define CM(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    CM(T[0]);
#else
#define CM(var)
#endif

    // lower triangle of K (+ jitter), identity on the padding; everything above the diagonal blocks starts at zero.
    // A wave takes whole rows (coalesced, no integer division).  ALL loads of the thread are issued before the first LDS
    // store: K was written by the previous kernel (cold in this XCD's L2), and four dependent batches of four rows cost
    // 5 us at m = 128 where one batch costs 2.
    {
        const int nw = nthr >> 6;                      // 8 waves: rows wave, wave + 8, ... -> at most 16 rows per wave (mp <= 128)
        double v[16][2];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = wave + u * nw;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = lane + 64 * h;
                v[u][h] = (i < m && j <= i) ? J.K[(long)i * ldk + j] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = wave + u * nw;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = lane + 64 * h;
                if (i < mp && j < mp) Lm[i * VG_CLD + j] = v[u][h] + (i == j ? (i < m ? jit : 1.0) : 0.0);
            }
        }
        for (int i = wave + 16 * nw; i < mp; i += nw)      // (fewer than 8 waves: never taken with the 512-thread launch)
            for (int h = 0; h < 2; ++h) {
                const int j = lane + 64 * h;
                if (j < mp) Lm[i * VG_CLD + j] = ((i < m && j <= i) ? J.K[(long)i * ldk + j] : 0.0) + (i == j ? (i < m ? jit : 1.0) : 0.0);
            }
    }
    for (int idx = tid; idx < nb * VG_CB * VG_CB; idx += nthr) Dinv[idx] = 0.0;
    if (tid == 0) s_i[2] = 0;
    __syncthreads();
    CM(T[1]);

    bool ok = true;
    for (int p = 0; p < nb; ++p) {
        CM(q0);
        const int bi = p + wave;
        vg_cd4 ut = {0.0, 0.0, 0.0, 0.0};
        const bool mine = bi < nb;
        if (mine) {
            // acc[c][i] = sum_k L[16p + c][k] L[16bi + i][k],  k < 16p
            vg_cd4 acc = {0.0, 0.0, 0.0, 0.0};
            const double* ap = Lm + (p * VG_CB + fi) * VG_CLD + fk;
            const double* bp = Lm + (bi * VG_CB + fi) * VG_CLD + fk;
            // (the diagonal block, bi == p, arrives as a running Schur complement: every earlier panel subtracted its
            //  contribution in step (3), spread over the waves -- wave 0's chain of up to 28 dependent MFMAs per panel, 8.4 us
            //  of the factorisation's critical path at m = 128, is gone)
            if (wave > 0)
                for (int k0 = 0; k0 < p * VG_CB; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k0], bp[k0], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = fk + 4 * r, i = fi;                    // D layout: row c (panel column), col i (row inside block bi)
                int gr = bi * VG_CB + i, gc = p * VG_CB + c;
                if (gc > gr) { const int t = gr; gr = gc; gc = t; }  // diagonal block: symmetric, stored lower
                ut[r] = Lm[gr * VG_CLD + gc] - acc[r];
            }
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Db[(fk + 4 * r) * 17 + fi] = ut[r];
            }
        }
        CM(q1);
#ifdef VG_CHOL_STAMP
        tU += q1 - q0;
#endif
        if (wave == 0) {
            // ---- 16 x 16 diagonal block: in-wave elimination on [A | I]; lane (i, q) holds columns 4q .. 4q+3 of row i ----
            VG_WAVE_SYNC();
            const int i = fi, q = fk;
            double a[4], x[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { a[c] = Db[i * 17 + 4 * q + c]; x[c] = (i == 4 * q + c) ? 1.0 : 0.0; }
            bool good = true;
            // FOUR pivot columns per LDS round trip: columns k .. k+3 (and rows k .. k+3 of the running inverse) are published
            // together; every lane then runs the four eliminations of the 4 x 4 pivot block itself, on the handful of entries of
            // those columns it needs -- its own row, the pivot rows, the rows that mirror its own columns (the trailing matrix
            // stays symmetric, so row k+r at column j is column k+r at row j) -- and applies one rank-4 update.  The chain is
            // 4 round trips per diagonal block instead of 8 (two pivots each: 910 cycles per pair, 24 of the launch's 42 us).
#pragma unroll
            for (int k = 0; k < VG_CB; k += 4) {
                if (q == (k >> 2)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) colbuf[(k + c) * 16 + i] = a[c];
                }
                if ((i >> 2) == (k >> 2)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xrow[(i & 3) * 16 + 4 * q + c] = x[c];
                }
                VG_WAVE_SYNC();
                double Cp[4][4], Ci[4], Cj[4][4], Yr[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Ci[r] = colbuf[(k + r) * 16 + i];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        Cp[r][c] = colbuf[(k + r) * 16 + k + c];
                        Cj[r][c] = colbuf[(k + r) * 16 + 4 * q + c];
                        Yr[r][c] = xrow[r * 16 + 4 * q + c];
                    }
                }
                double l[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double pr = Cp[r][r];
                    if (!(pr > 0.0) || !(pr < 1.0e300)) { good = false; break; }           // wave-uniform
                    const double inv = vg_crcp(pr);
                    l[r] = (i > k + r) ? Ci[r] * inv : 0.0;
#pragma unroll
                    for (int t = r + 1; t < 4; ++t) {
                        const double f = Cp[r][t] * inv;                                   // multiplier of pivot row k+t at pivot k+r
#pragma unroll
                        for (int sidx = t; sidx < 4; ++sidx) Cp[t][sidx] -= Cp[r][sidx] * f;
                        Ci[t] -= Ci[r] * f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) { Cj[t][c] -= Cj[r][c] * f; Yr[t][c] -= Yr[r][c] * f; }
                    }
                }
                if (!good) break;
                VG_WAVE_SYNC();                                                    // everybody has read columns k+1 .. k+3 ...
                if (q == 0 && i > k) {                                             // ... before they are replaced by their final values
#pragma unroll
                    for (int t = 1; t < 4; ++t) colbuf[(k + t) * 16 + i] = Ci[t];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a[c] -= l[0] * Cj[0][c] + l[1] * Cj[1][c] + l[2] * Cj[2][c] + l[3] * Cj[3][c];
                    x[c] -= l[0] * Yr[0][c] + l[1] * Yr[1][c] + l[2] * Yr[2][c] + l[3] * Yr[3][c];
                }
            }
            if (good) {
                if (lane < 16) sdv[lane] = vg_crsq(colbuf[lane * 16 + lane]);      // 1 / L[k][k]
                VG_WAVE_SYNC();
                const double si = sdv[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = 4 * q + c;
                    Lm[(p * VG_CB + i) * VG_CLD + p * VG_CB + k] = (k <= i) ? colbuf[k * 16 + i] * sdv[k] : 0.0;
                    Dinv[(p * VG_CB + i) * VG_CB + k] = (k <= i) ? x[c] * si : 0.0;
                }
            } else if (lane == 0) {
                s_i[2] = 1;
            }
        }
        CM(q0);
#ifdef VG_CHOL_STAMP
        tF += q0 - q1;
#endif
        __syncthreads();
        if (s_i[2]) { ok = false; break; }
        if (mine && wave > 0) {
            // L[bi][p]^T = X11 U^T : A operand X11[c'][k] from LDS, B operand for k-step r is accumulator register r of U^T
            vg_cd4 lt = {0.0, 0.0, 0.0, 0.0};
            const double* xp = Dinv + (p * VG_CB + fi) * VG_CB + fk;
#pragma unroll
            for (int r = 0; r < 4; ++r) lt = __builtin_amdgcn_mfma_f64_16x16x4f64(xp[4 * r], ut[r], lt, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Lm[(bi * VG_CB + fi) * VG_CLD + p * VG_CB + fk + 4 * r] = lt[r];
            // running Schur complement of this block row's own diagonal block: D[bi] -= L[bi][p] L[bi][p]^T.  lt[s] at lane
            // (fi, fk) is L[16 bi + fi][16 p + fk + 4 s]: exactly the A operand AND the B operand of k-step s.
            vg_cd4 dd = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) dd = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[r], lt[r], dd, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = fk + 4 * r;
                if (fi <= row) Lm[(bi * VG_CB + row) * VG_CLD + bi * VG_CB + fi] -= dd[r];       // stored lower
            }
        }
        __syncthreads();
        CM(q1);
#ifdef VG_CHOL_STAMP
        tS += q1 - q0;
#endif
    }
    CM(T[2]);

    // ---- level selection: the lowest successful level wins (relaxed agent-scope flags, no payload) ----------
    if (J.only_level0) {
        if (tid == 0) { s_i[0] = ok ? 1 : 0; s_i[1] = ok ? 0 : 1; }
    } else if (tid == 0) {
        __hip_atomic_store(&flags[lvl], ok ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int win = ok ? 1 : 0, spins = 0;
        bool all_failed = !ok;
        for (int l = 0; l < lvl; ++l) {
            int f;
            while ((f = __hip_atomic_load(&flags[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                if (++spins > (1 << 22)) { f = 2; break; }   // bounded: a missing sibling counts as failed
                __builtin_amdgcn_s_sleep(4);
            }
            if (f == 1) { win = 0; all_failed = false; }
        }
        s_i[0] = win;
        s_i[1] = (lvl == 3 && all_failed) ? 1 : 0;
    }
    __syncthreads();
    const bool winner = s_i[0] != 0, report_fail = s_i[1] != 0;
    if (report_fail) {
        if (tid == 0) { *J.status = VGGP_ENOTPD; if (J.jitter_out) *J.jitter_out = -1.0; }
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int idx = tid; idx < m * m; idx += nthr) { J.L[(long)(idx / m) * ldl + idx % m] = qnan; if (J.Linv) J.Linv[idx] = qnan; }
        if (J.Dinv_out) for (int idx = tid; idx < nb * VG_CB * VG_CB; idx += nthr) J.Dinv_out[idx] = qnan;
    }
    if (!winner) return;
    if (tid == 0 && J.jitter_out) *J.jitter_out = jit;
    CM(T[3]);
    // the inverses of the 16 x 16 diagonal blocks, [nb][16][16]: all a substitution (trsm.hip) needs of the inverse
    if (J.Dinv_out) for (int idx = tid; idx < nb * VG_CB * VG_CB; idx += nthr) J.Dinv_out[idx] = Dinv[idx];

    // ---- L out (the upper block triangle of Lm is about to receive X) ----
    for (int i = wave; i < m; i += (nthr >> 6))
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = lane + 64 * h;
            if (j < m) J.L[(long)i * ldl + j] = (j <= i) ? Lm[i * VG_CLD + j] : 0.0;
        }
    CM(T[4]);
    // (no full inverse wanted: the step takes L^{-1} from the substitution kernel -- an identity right-hand side in the same
    //  launch as B = L^{-1} A -- which removes this phase and the store below, 12 us of 58 at m = 128, from the critical path)
    if (!J.Linv) return;
    // ---- X = L^{-1}: block column bj by wave bj, rows top-down; X[bi][bj] is kept at block position (bj, bi) ----
    for (int bj = wave; bj < nb; bj += (nthr >> 6)) {
        for (int bi = bj + 1; bi < nb; ++bi) {
            vg_cd4 t = {0.0, 0.0, 0.0, 0.0};
            const double* ap = Lm + (bi * VG_CB + fi) * VG_CLD + fk;                 // L[16bi + i][k]
            // bk = bj: X[bj][bj] = Dinv block
            {
                const double* bp = Dinv + (bj * VG_CB + fk) * VG_CB + fi;            // X11[k][j]
#pragma unroll
                for (int k0 = 0; k0 < VG_CB; k0 += 4)
                    t = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[bj * VG_CB + k0], bp[k0 * VG_CB], t, 0, 0, 0);
            }
            for (int bk = bj + 1; bk < bi; ++bk) {
                const double* bp = Lm + (bj * VG_CB + fk) * VG_CLD + bk * VG_CB + fi;    // X[bk][bj][k][j] at (16bj + k, 16bk + j)
#pragma unroll
                for (int k0 = 0; k0 < VG_CB; k0 += 4)
                    t = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[bk * VG_CB + k0], bp[k0 * VG_CLD], t, 0, 0, 0);
            }
            vg_cd4 xb = {0.0, 0.0, 0.0, 0.0};
            const double* xp = Dinv + (bi * VG_CB + fi) * VG_CB + fk;
#pragma unroll
            for (int r = 0; r < 4; ++r) xb = __builtin_amdgcn_mfma_f64_16x16x4f64(xp[4 * r], t[r], xb, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Lm[(bj * VG_CB + fk + 4 * r) * VG_CLD + bi * VG_CB + fi] = -xb[r];
            VG_WAVE_SYNC();                                                        // own stores before own next-row loads
        }
    }
    __syncthreads();
    CM(T[5]);
    for (int i = wave; i < m; i += (nthr >> 6))
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = lane + 64 * h;
            const int bi = i >> 4, bj = j >> 4;
            double v = 0.0;
            if (bi == bj) v = Dinv[i * VG_CB + (j & 15)];
            else if (bi > bj) v = Lm[(bj * VG_CB + (i & 15)) * VG_CLD + bi * VG_CB + (j & 15)];
            if (j < m) J.Linv[(long)i * m + j] = v;
        }
#ifdef VG_CHOL_STAMP
    CM(T[6]);
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(J.scratch) + 16 + wave * 12;
        for (int i = 0; i < 7; ++i) dbg[i] = T[i] - T[0];
        dbg[7] = tU; dbg[8] = tF; dbg[9] = tS;
    }
#endif
}

// `rider`: optional GEMM batch executed by extra workgroups of the launch (linear block ids >= 4 * njobs), see eigh.hip
__global__ __launch_bounds__(512) void vg_chol_mfma_kernel(const VgCholArgs a, const VgGemmBatch rider) {
    extern __shared__ double vg_cm_dyn[];
    if ((int)blockIdx.x >= 4 * a.njobs) {
        vg_gemm_body<64, 16, 512>(rider, vg_cm_dyn, blockIdx.x - 4 * a.njobs);
        return;
    }
    __shared__ double Db[16 * 17];
    __shared__ double colbuf[16 * 16];
    __shared__ double xrow[4 * 16];
    __shared__ double sdv[16];
    __shared__ int s_i[4];
    const VgCholJob& J = a.job[blockIdx.x >> 2];
    const int lvl = blockIdx.x & 3;
    if (J.only_level0 && lvl > 0) return;
    const int nb = (J.m + VG_CB - 1) / VG_CB, mp = nb * VG_CB;
    vg_chol_mfma(J, lvl, vg_cm_dyn, vg_cm_dyn + mp * VG_CLD, Db, colbuf, xrow, sdv, s_i);
}

__global__ __launch_bounds__(1024) void vg_chol_kernel(const VgCholArgs a) {
    extern __shared__ double vg_chol_dyn[];
    __shared__ double sd[1024];
    __shared__ double rowbuf[2 * 192];
    __shared__ int s_i[2];
    const VgCholJob& J = a.job[blockIdx.y];
    const int lvl = blockIdx.x;
    if (J.only_level0 && lvl > 0) return;
    if (a.fast[blockIdx.y]) {
        if (J.m <= 64) vg_chol_fast<2>(J, lvl, vg_chol_dyn, sd, rowbuf, s_i);
        else vg_chol_fast<4>(J, lvl, vg_chol_dyn, sd, rowbuf, s_i);
    } else if (lvl == 0) {
        vg_chol_generic(J, sd);
    }
}


// ---- 128 < m <= 256: two 128-blocks, all four jitter levels side by side (api.hip vg_chol_big_enqueue) ---------------------------
// Kc[lvl] = K + jitter_lvl I  (four full copies: the levels are then plain only_level0 jobs whose status words say which survived)
__global__ void vg_jitcopy_kernel(const double* K, int m, double* Kc) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x, mm = (long)m * m;
    if (idx >= 4 * mm) return;
    const int lvl = (int)(idx / mm);
    const long e = idx - lvl * mm;
    Kc[idx] = K[e] + ((e / m == e % m) ? VG_JITTERS[lvl] : 0.0);
}
hipError_t vg_jitcopy_launch(const double* K, int m, double* Kc, hipStream_t st) {
    hipLaunchKernelGGL(vg_jitcopy_kernel, dim3((unsigned)((4L * m * m + 255) / 256)), dim3(256), 0, st, K, m, Kc);
    return hipGetLastError();
}
// The lowest level whose two block factorisations both succeeded wins: its factor (upper triangle zeroed), the inverses of its
// 16 x 16 diagonal blocks, the jitter value; VGGP_ENOTPD when none did.  st8: [lvl][stage] status words of the 8 block jobs.
__global__ void vg_cholsel_kernel(const double* Lc, const double* Dc, const int* st8, int m, int nb16, double* L0, double* Dinv0,
                                  double* jitter_out, int* status) {
    __shared__ int s_lvl;
    if (threadIdx.x == 0 && blockIdx.x == 0) {}
    if (threadIdx.x == 0) {
        int lvl = -1;
        for (int l = 3; l >= 0; --l) if (st8[2 * l] == 0 && st8[2 * l + 1] == 0) lvl = l;
        s_lvl = lvl;
    }
    __syncthreads();
    const int lvl = s_lvl;
    const long mm = (long)m * m;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (lvl < 0) { *status = VGGP_ENOTPD; *jitter_out = -1.0; }
        else *jitter_out = VG_JITTERS[lvl];
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < mm; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / m), j = (int)(e - (long)i * m);
        L0[e] = lvl < 0 ? qnan : (j <= i ? Lc[lvl * mm + e] : 0.0);
    }
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)nb16 * 256; e += (long)gridDim.x * blockDim.x)
        Dinv0[e] = lvl < 0 ? qnan : Dc[(long)lvl * nb16 * 256 + e];
}
hipError_t vg_cholsel_launch(const double* Lc, const double* Dc, const int* st8, int m, double* L0, double* Dinv0, double* jitter_out,
                             int* status, hipStream_t st) {
    hipLaunchKernelGGL(vg_cholsel_kernel, dim3(64), dim3(256), 0, st, Lc, Dc, st8, m, (m + 15) / 16, L0, Dinv0, jitter_out, status);
    return hipGetLastError();
}

hipError_t vg_chol_setup() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vg_chol_mfma_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(vg_chol_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
}

hipError_t vg_chol_launch(const VgCholJob* jobs, int njobs, hipStream_t st, const VgGemmBatch* rider) {
    if (njobs < 1 || njobs > VG_CHOL_MAXJOBS) return hipErrorInvalidValue;
    VgCholArgs a;
    a.njobs = njobs;
    size_t lds = 0;
    static const bool legacy = getenv("VGGP_CHOL_LEGACY") != nullptr;      // A/B switch: register-resident column kernel
    bool all_fast = !legacy;
    for (int j = 0; j < njobs; ++j) all_fast = all_fast && jobs[j].m >= 1 && jobs[j].m <= VG_CHOL_FAST_MAX_M;
    for (int j = 0; j < njobs; ++j)
        if (!jobs[j].Linv && !all_fast) return hipErrorInvalidValue;      // only the MFMA path can leave the inverse out
    if (all_fast) {
        for (int j = 0; j < njobs; ++j) {
            a.job[j] = jobs[j];
            a.fast[j] = 1;
            const size_t mp = (size_t)((jobs[j].m + VG_CB - 1) / VG_CB) * VG_CB;
            const size_t need = (mp * VG_CLD + mp * VG_CB) * sizeof(double);
            if (need > lds) lds = need;
        }
        VgGemmBatch rb;
        rb.nprob = 0; rb.total_tiles = 0;
        if (rider && rider->nprob > 0 && rider->total_tiles > 0) {
            rb = *rider;
            const size_t rl = 2 * VgTile<64, 16>::TILE * sizeof(double);
            if (rl > lds) lds = rl;
        }
        hipLaunchKernelGGL(vg_chol_mfma_kernel, dim3(4 * njobs + rb.total_tiles), dim3(512), lds, st, a, rb);
        return hipGetLastError();
    }
    if (rider && rider->nprob > 0) return hipErrorInvalidValue;      // (callers check vg_chol_can_ride first)
    for (int j = 0; j < njobs; ++j) {
        a.job[j] = jobs[j];
        if (jobs[j].m > 1024 || jobs[j].m < 1) return hipErrorInvalidValue;
        a.fast[j] = jobs[j].m <= VG_CHOL_FAST_MAX_M;
        if (a.fast[j]) {
            const size_t mp = jobs[j].m <= 64 ? 64 : 128;
            const size_t need = mp * (mp + 1) / 2 * sizeof(double);
            if (need > lds) lds = need;
        }
    }
    hipLaunchKernelGGL(vg_chol_kernel, dim3(4, njobs), dim3(1024), lds, st, a);
    return hipGetLastError();
}
