// m-space stage of the ELBO step: everything after the eigendecompositions is O(m1 m2) element
// work plus wavefront-shuffle reductions.  Implements the collapsed bound of
// kronecker_structure.py:249-278 in the eigenbasis (spec: oracle/kron.py finish()):
//
//   lam_d = s_d lam0_d,  a = lam1 lam2^T / v,  D = 1 + a,  P = sqrt(s1 s2) P0,  beta = P / D
//   ELBO  = -1/2 [N log 2pi + N log v + sum log D + yy/v - sum(P beta)/v^2]
//           - (N s1 s2 - sum(lam1) sum(lam2)) / (2 v)
// and its analytic gradient with respect to theta = (ell1, ell2, s1, s2, v).
// All factor quantities arrive at unit outputscale; s_d enters only here.
#include "common.h"

__device__ __forceinline__ double vg_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-wide sum, result valid in every thread; `red` has >= 16 doubles
__device__ __forceinline__ double vg_block_sum(double v, double* red) {
    v = vg_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}

// ---- D-stage: one workgroup per row i1 ---------------------------------------------
__global__ __launch_bounds__(256) void vg_dstage_kernel(const VgMspace ms) {
    __shared__ double red[16];
    const int i1 = blockIdx.x, m2 = ms.m2;
    const long msz = (long)ms.m1 * m2;
    const double s1 = ms.theta[2], s2 = ms.theta[3], v = ms.theta[4];
    const double rs = sqrt(s1 * s2);
    const double l1 = s1 * ms.lam1[i1];
    double acc[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) acc[q] = 0.0;
    for (int i2 = threadIdx.x; i2 < m2; i2 += blockDim.x) {
        const long e = (long)i1 * m2 + i2;
        const double l2 = s2 * ms.lam2[i2];
        const double a = l1 * l2 / v;
        const double D = 1.0 + a, iD = 1.0 / D;
        const double P = rs * ms.P3[e], P1 = rs * ms.P3[msz + e], P2 = rs * ms.P3[2 * msz + e];
        const double b = P * iD;
        ms.beta[e] = b;
        ms.bl2[e] = b * l2;
        ms.bl1[e] = b * l1;
        ms.invD[e] = iD;
        acc[0] += log1p(a);
        acc[1] += P * b;
        acc[2] += b * b;
        acc[3] += a * iD;
        acc[4] += b * b * (2.0 + a);
        acc[5] += b * P1;
        acc[6] += b * P2;
        acc[8] += iD;
        acc[9] += l2 * iD;
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) acc[q] = vg_block_sum(acc[q], red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ms.rowpart[i1 * 8 + q] = acc[q];
        ms.r1[i1] = acc[8];
        ms.r1l[i1] = acc[9];
    }
}

hipError_t vg_dstage_launch(const VgMspace* ms, hipStream_t st) {
    hipLaunchKernelGGL(vg_dstage_kernel, dim3(ms->m1), dim3(256), 0, st, *ms);
    return hipGetLastError();
}

// ---- final reduction: ONE launch ------------------------------------------------------------------------------------
// VG_NPART workgroups compute the column sums of 1/D and lam1/D and slices of the four m x m dot products sum(E o X),
// sum(F o Xl); the workgroup that draws the last ticket then does the O(m) combinations (all partial sums of all
// workgroups are visible to it: release fence before the ticket, acquire after) and writes the 6 results plus the step's
// diagnostics (jitter levels, status words, Jacobi counters) straight into the pinned host block.
#define VG_NPART 64
#define VG_NFIN 23          // scalars the final combination needs

// sums NV values per thread over the workgroup in one pass (wave shuffles, then one LDS stage); result in every thread
template <int NV>
__device__ __forceinline__ void vg_block_sum_n(double (&v)[NV], double* red /* >= 4 * NV */) {
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = vg_wave_sum(v[q]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[(threadIdx.x >> 6) * NV + q] = v[q];
    }
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += red[w * NV + q];
        v[q] = t;
    }
}

__device__ void vg_final_body(const VgMspace& ms, double* red) {
    const int m1 = ms.m1, m2 = ms.m2;
    const double s1 = ms.theta[2], s2 = ms.theta[3], v = ms.theta[4];
    const double N = ms.n_total, yy = ms.yy;
    // S[0..6]: row partials; [7..10]: dot products; [11],[12]: sum lam; [13..17]: dimension 1 sums; [18..22]: dimension 2
    double S[VG_NFIN];
#pragma unroll
    for (int q = 0; q < VG_NFIN; ++q) S[q] = 0.0;
    for (int i = threadIdx.x; i < m1; i += blockDim.x) {
#pragma unroll
        for (int q = 0; q < 7; ++q) S[q] += ms.rowpart[i * 8 + q];
        S[11] += ms.lam1[i];
        const double e = ms.E1[(long)i * m1 + i], f = 2.0 * s1 * ms.F1[(long)i * m1 + i];
        S[13] += e; S[14] += e * ms.r1[i]; S[15] += f * ms.r1l[i]; S[16] += f; S[17] += e * s1 * ms.lam1[i];
    }
    for (int i = threadIdx.x; i < m2; i += blockDim.x) {
        S[12] += ms.lam2[i];
        const double e = ms.E2[(long)i * m2 + i], f = 2.0 * s2 * ms.F2[(long)i * m2 + i];
        S[18] += e; S[19] += e * ms.r2[i]; S[20] += f * ms.r2l[i]; S[21] += f; S[22] += e * s2 * ms.lam2[i];
    }
    for (int i = threadIdx.x; i < VG_NPART; i += blockDim.x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) S[7 + q] += ms.dotpart[i * 4 + q];
    }
    vg_block_sum_n<VG_NFIN>(S, red);
    if (threadIdx.x == 0) {
        const double sl1 = s1 * S[11], sl2 = s2 * S[12];
        const double EX1 = S[7], FX1 = 2.0 * s1 * S[8], EX2 = S[9], FX2 = 2.0 * s2 * S[10];
        const double v2 = v * v;
        const double elbo = -0.5 * (N * 1.8378770664093453 + N * log(v) + S[0] + yy / v - S[1] / v2)
                            - (N * s1 * s2 - sl1 * sl2) / (2.0 * v);
        const double quad1 = 2.0 * S[5] - EX1 - FX1 / v;
        const double quad2 = 2.0 * S[6] - EX2 - FX2 / v;
        const double g_l1 = -0.5 * (S[14] + S[15] / v - (double)m2 * S[13] - quad1 / v2) + sl2 / (2.0 * v) * (S[16] - S[17]);
        const double g_l2 = -0.5 * (S[19] + S[20] / v - (double)m1 * S[18] - quad2 / v2) + sl1 / (2.0 * v) * (S[21] - S[22]);
        const double common = -0.5 * (S[3] - S[2] / v2);
        const double g_s1 = common / s1 + sl1 * sl2 / (2.0 * v * s1) - N * s2 / (2.0 * v);
        const double g_s2 = common / s2 + sl1 * sl2 / (2.0 * v * s2) - N * s1 / (2.0 * v);
        const double g_v = -0.5 * (N / v - S[3] / v - yy / v2 + S[4] / (v2 * v)) + (N * s1 * s2 - sl1 * sl2) / (2.0 * v2);
        const double o[6] = {elbo, g_l1, g_l2, g_s1, g_s2, g_v};
#pragma unroll
        for (int q = 0; q < 6; ++q) ms.out[q] = o[q];
        if (ms.hout) {
#pragma unroll
            for (int q = 0; q < 6; ++q) ms.hout->out[q] = o[q];
            for (int k = 0; k < 2; ++k) {
                ms.hout->jitter[k] = ms.jit[k] ? *ms.jit[k] : 0.0;
                ms.hout->status[k] = ms.status[k] ? *ms.status[k] : 0;
                for (int q = 0; q < 4; ++q) ms.hout->counters[k][q] = ms.counters[k] ? ms.counters[k][q] : 0;
            }
        }
    }
}

__global__ __launch_bounds__(256) void vg_partial_kernel(const VgMspace ms) {
    __shared__ double red[4 * VG_NFIN];
    __shared__ int s_last;
    const int m1 = ms.m1, m2 = ms.m2, b = blockIdx.x;
    const double s1 = ms.theta[2];
    // column sums: block b owns columns b, b + VG_NPART, ...
    for (int i2 = b; i2 < m2; i2 += VG_NPART) {
        double ac[2] = {0.0, 0.0};
        for (int i1 = threadIdx.x; i1 < m1; i1 += blockDim.x) {
            const double iD = ms.invD[(long)i1 * m2 + i2];
            ac[0] += iD;
            ac[1] += s1 * ms.lam1[i1] * iD;
        }
        vg_block_sum_n<2>(ac, red);
        if (threadIdx.x == 0) { ms.r2[i2] = ac[0]; ms.r2l[i2] = ac[1]; }
    }
    // dot-product slices
    double dp[4] = {0.0, 0.0, 0.0, 0.0};
    for (long idx = (long)b * blockDim.x + threadIdx.x; idx < (long)m1 * m1; idx += (long)VG_NPART * blockDim.x) {
        dp[0] += ms.E1[idx] * ms.X1[idx];
        dp[1] += ms.F1[idx] * ms.X1l[idx];
    }
    for (long idx = (long)b * blockDim.x + threadIdx.x; idx < (long)m2 * m2; idx += (long)VG_NPART * blockDim.x) {
        dp[2] += ms.E2[idx] * ms.X2[idx];
        dp[3] += ms.F2[idx] * ms.X2l[idx];
    }
    vg_block_sum_n<4>(dp, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ms.dotpart[b * 4 + q] = dp[q];
        __threadfence();                                         // release this workgroup's partial sums ...
        const int t = atomicAdd(ms.ticket, 1);                   // ... before drawing the ticket
        s_last = (t == VG_NPART - 1);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                             // acquire: every other workgroup's partial sums
    if (threadIdx.x == 0) *ms.ticket = 0;                        // self-cleaning for the next launch
    vg_final_body(ms, red);
}

hipError_t vg_final_launch(const VgMspace* ms, hipStream_t st) {
    if (!ms->ticket) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vg_partial_kernel, dim3(VG_NPART), dim3(256), 0, st, *ms);
    return hipGetLastError();
}

// ---- small element-wise helpers for q(v) / posterior --------------------------------
__global__ void vg_sq_kernel(const double* in, double* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * in[i];
}
hipError_t vg_scale_sq_launch(const double* in, double* out_sq, long n, hipStream_t st) {
    hipLaunchKernelGGL(vg_sq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out_sq, n);
    return hipGetLastError();
}

// w_mean = beta * sqrt(s1 s2) / v ;   w_var (second half of the buffer) = invD - 1
__global__ void vg_qv_weights_kernel(const double* theta, const double* beta, const double* invD, double* w, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double rs = sqrt(theta[2] * theta[3]);
        w[i] = beta[i] * rs / theta[4];
        w[n + i] = invD[i] - 1.0;
    }
}
hipError_t vg_qv_weights_launch(const double* theta, const double* beta, const double* invD, double* w_mean,
                                long n, hipStream_t st) {
    hipLaunchKernelGGL(vg_qv_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, theta, beta,
                       invD, w_mean, n);
    return hipGetLastError();
}

// mode 0: x *= s1*s2 ; mode 1: x = s1*s2*(1 + x)
__global__ void vg_scale_kernel(double* x, long n, const double* theta, int mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double s12 = theta[2] * theta[3];
        x[i] = mode == 0 ? x[i] * s12 : s12 * (1.0 + x[i]);
    }
}
hipError_t vg_scale_launch(double* x, long n, const double* theta, int mode, hipStream_t st) {
    hipLaunchKernelGGL(vg_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, theta, mode);
    return hipGetLastError();
}

// posterior combine: mean[p] = sum_i T1[i][p] U[i][p] (U already carries sqrt(s1 s2)/v);  var[p] = s1 s2 (1 + sum_i T1[i][p]^2 Uv[i][p])
__global__ void vg_post_combine_kernel(const double* theta, const double* T1, const double* U, const double* Uv,
                                       int m1, long ns, double* mean, double* var) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ns) return;
    double am = 0.0, av = 0.0;
    for (int i = 0; i < m1; ++i) {
        const double t = T1[(long)i * ns + p];
        am += t * U[(long)i * ns + p];
        av += t * t * Uv[(long)i * ns + p];
    }
    const double s12 = theta[2] * theta[3];
    mean[p] = am;
    var[p] = s12 * (1.0 + av);
}
hipError_t vg_post_combine_launch(const double* theta, const double* T1, const double* U, const double* Uv,
                                  const double* unused, int m1, int m2, long ns, double* mean, double* var,
                                  hipStream_t st) {
    (void)unused; (void)m2;
    hipLaunchKernelGGL(vg_post_combine_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, theta, T1, U,
                       Uv, m1, ns, mean, var);
    return hipGetLastError();
}

// sum of squares: two-pass deterministic
__global__ __launch_bounds__(256) void vg_sumsq_kernel(const double* y, long n, double* partial) {
    __shared__ double red[16];
    double t = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        t += y[i] * y[i];
    t = vg_block_sum(t, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void vg_sumsq_final_kernel(const double* partial, int nb, double* out) {
    __shared__ double red[16];
    double t = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) t += partial[i];
    t = vg_block_sum(t, red);
    if (threadIdx.x == 0) out[0] = t;
}
hipError_t vg_sumsq_launch(const double* y, long n, double* partial, double* out, hipStream_t st) {
    int nb = (int)((n + 256L * 8 - 1) / (256L * 8));
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(vg_sumsq_kernel, dim3(nb), dim3(256), 0, st, y, n, partial);
    hipLaunchKernelGGL(vg_sumsq_final_kernel, dim3(1), dim3(256), 0, st, partial, nb, out);
    return hipGetLastError();
}

__global__ void vg_clear_kernel(const VgClearArgs a) {
    const int b = blockIdx.x;
    if (b < a.n)
        for (int i = threadIdx.x; i < a.nwords[b]; i += blockDim.x) a.ptr[b][i] = 0;
}
hipError_t vg_clear_launch(const VgClearArgs* a, hipStream_t st) {
    if (a->n <= 0) return hipSuccess;
    hipLaunchKernelGGL(vg_clear_kernel, dim3(a->n), dim3(64), 0, st, *a);
    return hipGetLastError();
}
