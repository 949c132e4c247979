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
#include "common.h"

__constant__ double VG_JITTERS[4] = {0.0, 1e-8, 1e-7, 1e-6};

struct VgCholArgs {
    VgCholJob job[4];
    int njobs;
    int fast[4];
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

static const int VG_CHOL_FAST_MAX_M = 128;      // 4 x 4 register tile per thread; LDS holds the padded packed triangle

hipError_t vg_chol_setup() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(vg_chol_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
}

hipError_t vg_chol_launch(const VgCholJob* jobs, int njobs, hipStream_t st) {
    if (njobs < 1 || njobs > 4) return hipErrorInvalidValue;
    VgCholArgs a;
    a.njobs = njobs;
    size_t lds = 0;
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
