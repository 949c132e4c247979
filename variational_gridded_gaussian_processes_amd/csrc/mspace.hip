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

// ---- final reduction in two launches -----------------------------------------------------------------------------
// (1) vg_partial_kernel, VG_NPART workgroups: column sums of 1/D and lam1/D, and slices of the four m x m dot
//     products sum(E o X), sum(F o Xl) per dimension;  (2) vg_final_kernel, one workgroup: O(m) combinations only.
#define VG_NPART 64

__global__ __launch_bounds__(256) void vg_partial_kernel(const VgMspace ms) {
    __shared__ double red[16];
    const int m1 = ms.m1, m2 = ms.m2, b = blockIdx.x;
    const double s1 = ms.theta[2];
    // column sums: block b owns columns b, b + VG_NPART, ...
    for (int i2 = b; i2 < m2; i2 += VG_NPART) {
        double a = 0.0, c = 0.0;
        for (int i1 = threadIdx.x; i1 < m1; i1 += blockDim.x) {
            const double iD = ms.invD[(long)i1 * m2 + i2];
            a += iD;
            c += s1 * ms.lam1[i1] * iD;
        }
        a = vg_block_sum(a, red);
        c = vg_block_sum(c, red);
        if (threadIdx.x == 0) { ms.r2[i2] = a; ms.r2l[i2] = c; }
    }
    // dot-product slices
    double ex1 = 0, fx1 = 0, ex2 = 0, fx2 = 0;
    for (long idx = (long)b * blockDim.x + threadIdx.x; idx < (long)m1 * m1; idx += (long)VG_NPART * blockDim.x) {
        ex1 += ms.E1[idx] * ms.X1[idx];
        fx1 += ms.F1[idx] * ms.X1l[idx];
    }
    for (long idx = (long)b * blockDim.x + threadIdx.x; idx < (long)m2 * m2; idx += (long)VG_NPART * blockDim.x) {
        ex2 += ms.E2[idx] * ms.X2[idx];
        fx2 += ms.F2[idx] * ms.X2l[idx];
    }
    ex1 = vg_block_sum(ex1, red); fx1 = vg_block_sum(fx1, red);
    ex2 = vg_block_sum(ex2, red); fx2 = vg_block_sum(fx2, red);
    if (threadIdx.x == 0) {
        ms.dotpart[b * 4 + 0] = ex1; ms.dotpart[b * 4 + 1] = fx1;
        ms.dotpart[b * 4 + 2] = ex2; ms.dotpart[b * 4 + 3] = fx2;
    }
}

struct VgDimSums { double se, ser, sfr, sf, sel; };

__device__ VgDimSums vg_dim_sums(const double* E, const double* F, const double* lam0, double s, const double* r,
                                 const double* rl, int m, double* red) {
    VgDimSums o;
    double se = 0, ser = 0, sfr = 0, sf = 0, sel = 0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double e = E[(long)i * m + i], f = 2.0 * s * F[(long)i * m + i];
        se += e;
        ser += e * r[i];
        sfr += f * rl[i];
        sf += f;
        sel += e * s * lam0[i];
    }
    o.se = vg_block_sum(se, red);
    o.ser = vg_block_sum(ser, red);
    o.sfr = vg_block_sum(sfr, red);
    o.sf = vg_block_sum(sf, red);
    o.sel = vg_block_sum(sel, red);
    return o;
}

__global__ __launch_bounds__(256) void vg_final_kernel(const VgMspace ms) {
    __shared__ double red[16];
    const int m1 = ms.m1, m2 = ms.m2;
    const double s1 = ms.theta[2], s2 = ms.theta[3], v = ms.theta[4];
    const double N = ms.n_total, yy = ms.yy;

    double S[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        double t = 0.0;
        for (int i = threadIdx.x; i < m1; i += blockDim.x) t += ms.rowpart[i * 8 + q];
        S[q] = vg_block_sum(t, red);
    }
    double dp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double t = 0.0;
        for (int i = threadIdx.x; i < VG_NPART; i += blockDim.x) t += ms.dotpart[i * 4 + q];
        dp[q] = vg_block_sum(t, red);
    }
    double t1 = 0.0, t2 = 0.0;
    for (int i = threadIdx.x; i < m1; i += blockDim.x) t1 += ms.lam1[i];
    for (int i = threadIdx.x; i < m2; i += blockDim.x) t2 += ms.lam2[i];
    const double sl1 = s1 * vg_block_sum(t1, red), sl2 = s2 * vg_block_sum(t2, red);

    const VgDimSums d1 = vg_dim_sums(ms.E1, ms.F1, ms.lam1, s1, ms.r1, ms.r1l, m1, red);
    const VgDimSums d2 = vg_dim_sums(ms.E2, ms.F2, ms.lam2, s2, ms.r2, ms.r2l, m2, red);
    const double EX1 = dp[0], FX1 = 2.0 * s1 * dp[1], EX2 = dp[2], FX2 = 2.0 * s2 * dp[3];

    if (threadIdx.x == 0) {
        const double v2 = v * v;
        const double elbo = -0.5 * (N * 1.8378770664093453 + N * log(v) + S[0] + yy / v - S[1] / v2)
                            - (N * s1 * s2 - sl1 * sl2) / (2.0 * v);
        const double quad1 = 2.0 * S[5] - EX1 - FX1 / v;
        const double quad2 = 2.0 * S[6] - EX2 - FX2 / v;
        const double g_l1 = -0.5 * (d1.ser + d1.sfr / v - (double)m2 * d1.se - quad1 / v2)
                            + sl2 / (2.0 * v) * (d1.sf - d1.sel);
        const double g_l2 = -0.5 * (d2.ser + d2.sfr / v - (double)m1 * d2.se - quad2 / v2)
                            + sl1 / (2.0 * v) * (d2.sf - d2.sel);
        const double common = -0.5 * (S[3] - S[2] / v2);
        const double g_s1 = common / s1 + sl1 * sl2 / (2.0 * v * s1) - N * s2 / (2.0 * v);
        const double g_s2 = common / s2 + sl1 * sl2 / (2.0 * v * s2) - N * s1 / (2.0 * v);
        const double g_v = -0.5 * (N / v - S[3] / v - yy / v2 + S[4] / (v2 * v))
                           + (N * s1 * s2 - sl1 * sl2) / (2.0 * v2);
        ms.out[0] = elbo;
        ms.out[1] = g_l1;
        ms.out[2] = g_l2;
        ms.out[3] = g_s1;
        ms.out[4] = g_s2;
        ms.out[5] = g_v;
    }
}

hipError_t vg_final_launch(const VgMspace* ms, hipStream_t st) {
    hipLaunchKernelGGL(vg_partial_kernel, dim3(VG_NPART), dim3(256), 0, st, *ms);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(vg_final_kernel, dim3(1), dim3(256), 0, st, *ms);
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
