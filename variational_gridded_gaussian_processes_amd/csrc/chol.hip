// Cholesky factorisation with the psd_safe_cholesky jitter schedule and the explicit inverse of
// the factor, one workgroup (1024 threads) per matrix, matrix resident in LDS (m <= 136 in fp64;
// larger factors work out of an L2-resident global scratch with the same code).
//
// Replaces the Cholesky hidden inside lazify(Kuu).inv_matmul(Kuf) and MultivariateNormal.log_prob
// (kronecker_structure.py:269, :273) -- applied to the m_d x m_d per-dimension factors instead of
// the M x M / N x N dense matrices.  The explicit L^{-1} turns every later triangular solve into
// an MFMA GEMM (numerically validated against substitution in DESIGN.md: <= 1e-8 relative on the
// ill-conditioned RBF factors with the 1e-8 jitter).
//
// Phase 1  right-looking Cholesky on the lower triangle, one barrier per column; the column
//          scaling is deferred (L[i][k] = W[i][k] / sqrt(W[k][k]) once at the end).
// Phase 2  L^{-1} column by column: 8 lanes per column split the dot product of the forward
//          substitution and combine with 3 xor-shuffles; the solution is kept in the (unused)
//          upper triangle of the same LDS tile.
#include "common.h"

__constant__ double VG_JITTERS[4] = {0.0, 1e-8, 1e-7, 1e-6};

struct VgCholArgs {
    VgCholJob job[4];
    int njobs;
    int use_lds[4];
};

template <bool INLDS>
__device__ void vg_chol_body(const VgCholJob& J, double* W, double* sd, int* s_flag) {
    const int m = J.m, ld = m + 1;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tx = tid & 31, ty = tid >> 5;     // 32 x 32 thread tile for the trailing update
    const int nty = nthr >> 5;
    int used = -1;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const double jit = VG_JITTERS[attempt];
        for (int idx = tid; idx < m * m; idx += nthr) {
            const int i = idx / m, j = idx - i * m;
            W[i * ld + j] = J.K[idx] + (i == j ? jit : 0.0);
        }
        if (!INLDS) __threadfence_block();
        __syncthreads();
        bool ok = true;
        for (int k = 0; k < m; ++k) {
            const double piv = W[k * ld + k];
            if (!(piv > 0.0) || !(piv < 1.0e300)) { ok = false; break; }   // uniform: same value everywhere
            const double rp = 1.0 / piv;
            for (int i = k + 1 + ty; i < m; i += nty) {
                const double lik = W[i * ld + k] * rp;
                for (int j = k + 1 + tx; j <= i; j += 32) W[i * ld + j] -= lik * W[j * ld + k];
            }
            if (!INLDS) __threadfence_block();
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

    // deferred column scaling
    for (int k = tid; k < m; k += nthr) sd[k] = sqrt(W[k * ld + k]);
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += nthr) {
        const int i = idx / m, k = idx - i * m;
        if (i > k) W[i * ld + k] /= sd[k];
        else if (i == k) W[i * ld + k] = sd[k];
    }
    if (!INLDS) __threadfence_block();
    __syncthreads();

    // write L (before the upper triangle is reused for the inverse)
    for (int idx = tid; idx < m * m; idx += nthr) {
        const int i = idx / m, j = idx - i * m;
        J.L[idx] = (j <= i) ? W[i * ld + j] : 0.0;
    }

    // explicit inverse: column j of L^{-1} stored at W[j][i], i > j
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
                if (!INLDS) __threadfence_block();
            }
        }
    }
    if (!INLDS) __threadfence_block();
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
    __shared__ int s_flag;
    const VgCholJob& J = a.job[blockIdx.x];
    if (a.use_lds[blockIdx.x]) vg_chol_body<true>(J, vg_chol_dyn, sd, &s_flag);
    else vg_chol_body<false>(J, J.scratch, sd, &s_flag);
}

static const int VG_CHOL_LDS_MAX_M = 136;

hipError_t vg_chol_setup() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(vg_chol_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
}

hipError_t vg_chol_launch(const VgCholJob* jobs, int njobs, hipStream_t st) {
    if (njobs < 1 || njobs > 4) return hipErrorInvalidValue;
    VgCholArgs a;
    a.njobs = njobs;
    size_t lds = 0;
    for (int j = 0; j < njobs; ++j) {
        a.job[j] = jobs[j];
        if (jobs[j].m > 1024) return hipErrorInvalidValue;
        a.use_lds[j] = jobs[j].m <= VG_CHOL_LDS_MAX_M;
        if (a.use_lds[j]) {
            size_t need = (size_t)jobs[j].m * (jobs[j].m + 1) * sizeof(double);
            if (need > lds) lds = need;
        }
    }
    hipLaunchKernelGGL(vg_chol_kernel, dim3(njobs), dim3(1024), lds, st, a);
    return hipGetLastError();
}
