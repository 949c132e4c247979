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

// agent-scope (all XCDs) write-through store / cache-bypassing load: the per-XCD L2s are not coherent with each other
// for ordinary accesses, and a full __threadfence() costs a whole-L2 write-back per call
__device__ __forceinline__ void vg_st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double vg_ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NV>
__device__ __forceinline__ void vg_block_sum_n(double (&v)[NV], double* red /* >= 4 * NV */) {
    // butterfly with the NV shuffles of a step issued together: one cross-lane latency per step instead of one per value
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double t[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) t[q] = __shfl_xor(v[q], off);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] += t[q];
    }
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

// ---- D-stage: one workgroup per row i1 ---------------------------------------------
__global__ __launch_bounds__(256) void vg_dstage_kernel(const VgMspace ms) {
    __shared__ double red[4 * 10];
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
    vg_block_sum_n<10>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ms.rowpart[i1 * 8 + q] = acc[q];
        ms.r1[i1] = acc[8];
        ms.r1l[i1] = acc[9];
        if (ms.ol) { ms.ol[i1] = 1.0; ms.ol[ms.m1 + i1] = l1; }      // left operand of the column-sum product [r2; r2l] = ol invD
    }
}

hipError_t vg_dstage_launch(const VgMspace* ms, hipStream_t st) {
    hipLaunchKernelGGL(vg_dstage_kernel, dim3(ms->m1), dim3(256), 0, st, *ms);
    return hipGetLastError();
}

// ---- final reduction: ONE workgroup -----------------------------------------------------------------------------------
// Everything the combination needs arrives reduced: the D-stage left the row sums, the beta launch (api.hip finish_enqueue,
// step 9) the column sums [r2; r2l] = [1; s1 lam1] (1/D) and -- through the GEMM reduction epilogue -- the per-tile partial sums
// of the four dot products sum(E o beta beta^T) = sum((E^T beta) o beta) etc.  One workgroup loads ~25 words per thread in a single
// batch, sums them and writes the 6 results plus the step's diagnostics (jitter levels, status words, Jacobi counters) straight
// into the pinned host block.  (Round 1 / early round 2: 64 workgroups formed the four m x m dot products and the column sums
// here and met through a ticket; 17.6 us -- the ticket round trip and 1 MB of operands -- against ~7 now.)
#define VG_NFIN 23          // scalars the final combination needs
#define VG_NDOT 64          // slots per dot product in VgMspace::dotp (unused slots stay zero)

// S[0..6]: row partials; [7..10]: dot products; [11],[12]: sum lam; [13..17]: dimension 1 sums; [18..22]: dimension 2
__global__ __launch_bounds__(256) void vg_final_kernel(const VgMspace ms) {
    __shared__ double red[4 * VG_NFIN];
    __shared__ double stage[16];
    const int m1 = ms.m1, m2 = ms.m2, i = threadIdx.x;
    const double s1 = ms.theta[2], s2 = ms.theta[3], v = ms.theta[4];
    const double N = ms.n_total, yy = ms.yy;
    const bool o1 = i < m1, o2 = i < m2, od = i < VG_NDOT;
    // one batch of loads (m <= 256 = blockDim.x: one element per thread and array)
    double rp[7], dpv[4];
#pragma unroll
    for (int q = 0; q < 7; ++q) rp[q] = o1 ? ms.rowpart[i * 8 + q] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) dpv[q] = od ? ms.dotp[q * VG_NDOT + i] : 0.0;
    const double la1 = o1 ? ms.lam1[i] : 0.0, e1 = o1 ? ms.E1[(long)i * m1 + i] : 0.0, f1 = o1 ? ms.F1[(long)i * m1 + i] : 0.0;
    const double r1 = o1 ? ms.r1[i] : 0.0, r1l = o1 ? ms.r1l[i] : 0.0;
    const double la2 = o2 ? ms.lam2[i] : 0.0, e2 = o2 ? ms.E2[(long)i * m2 + i] : 0.0, f2 = o2 ? ms.F2[(long)i * m2 + i] : 0.0;
    const double r2 = o2 ? ms.r2[i] : 0.0, r2l = o2 ? ms.r2l[i] : 0.0;
    double S[VG_NFIN];
#pragma unroll
    for (int q = 0; q < 7; ++q) S[q] = rp[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) S[7 + q] = dpv[q];
    S[11] = la1; S[12] = la2;
    const double ff1 = 2.0 * s1 * f1, ff2 = 2.0 * s2 * f2;
    S[13] = e1; S[14] = e1 * r1; S[15] = ff1 * r1l; S[16] = ff1; S[17] = e1 * s1 * la1;
    S[18] = e2; S[19] = e2 * r2; S[20] = ff2 * r2l; S[21] = ff2; S[22] = e2 * s2 * la2;
    vg_block_sum_n<VG_NFIN>(S, red);
    if (threadIdx.x == 0) {
        // one thread, dependent f64 chain: reciprocals once, multiplications after (a software f64 division is ~30
        // dependent instructions; the original 17 of them cost ~6 us here)
        const double iv = 1.0 / v, is1 = 1.0 / s1, is2 = 1.0 / s2;
        const double iv2 = iv * iv, hiv = 0.5 * iv;
        const double sl1 = s1 * S[11], sl2 = s2 * S[12];
        const double EX1 = S[7], FX1 = 2.0 * s1 * S[8], EX2 = S[9], FX2 = 2.0 * s2 * S[10];
        const double Nss = N * s1 * s2, sll = sl1 * sl2;
        const double elbo = -0.5 * (N * 1.8378770664093453 + N * log(v) + S[0] + yy * iv - S[1] * iv2) - (Nss - sll) * hiv;
        const double quad1 = 2.0 * S[5] - EX1 - FX1 * iv;
        const double quad2 = 2.0 * S[6] - EX2 - FX2 * iv;
        const double g_l1 = -0.5 * (S[14] + S[15] * iv - (double)m2 * S[13] - quad1 * iv2) + sl2 * hiv * (S[16] - S[17]);
        const double g_l2 = -0.5 * (S[19] + S[20] * iv - (double)m1 * S[18] - quad2 * iv2) + sl1 * hiv * (S[21] - S[22]);
        const double common = -0.5 * (S[3] - S[2] * iv2);
        const double g_s1 = common * is1 + sll * hiv * is1 - N * s2 * hiv;
        const double g_s2 = common * is2 + sll * hiv * is2 - N * s1 * hiv;
        const double g_v = -0.5 * (N * iv - S[3] * iv - yy * iv2 + S[4] * iv2 * iv) + (Nss - sll) * 0.5 * iv2;
        const double o[6] = {elbo, g_l1, g_l2, g_s1, g_s2, g_v};
#pragma unroll
        for (int q = 0; q < 6; ++q) { ms.out[q] = o[q]; stage[q] = o[q]; }
        stage[6] = ms.peer_fail ? *ms.peer_fail : 0.0;          // ranks that failed their partials (multi-rank step), else 0
        stage[7] = 0.0;
    }
    // results + diagnostics go to the pinned host block as ONE 128-byte burst: the block is staged in LDS (layout of
    // VgHostOut: out[8], jitter[2], counters[2][4], status[2]) and 16 lanes store one 8-byte word each
    if (ms.hout) {
        int* ired = reinterpret_cast<int*>(stage + 10);
        if (threadIdx.x < 2) stage[8 + threadIdx.x] = ms.jit[threadIdx.x] ? *ms.jit[threadIdx.x] : 0.0;
        if (threadIdx.x >= 32 && threadIdx.x < 40) {
            const int k = (threadIdx.x - 32) >> 2, q = (threadIdx.x - 32) & 3;
            ired[k * 4 + q] = ms.counters[k] ? ms.counters[k][q] : 0;
        }
        if (threadIdx.x >= 64 && threadIdx.x < 66) {
            // status word of the dimension: Cholesky status, else the eigensolver's replay-timeout flag (word 1)
            const int* sp = ms.status[threadIdx.x - 64];
            const int s0 = sp ? sp[0] : 0, s1w = sp ? sp[1] : 0;
            // (word 1: bit 0 = replay timeout, bit 1 = the subspace start missed part of the range -> the host repeats the step)
            ired[8 + threadIdx.x - 64] = s0 ? s0 : ((s1w & 1) ? VGGP_ENOCONV : ((s1w & 2) ? VG_ESUBMISS : 0));
        }
        if (threadIdx.x == 66) stage[15] = ms.theta[5];          // the step's sequence number travels back with the results
        __syncthreads();
        static_assert(sizeof(VgHostOut) == 16 * 8, "VgHostOut layout");
        if (threadIdx.x < 16) reinterpret_cast<double*>(ms.hout)[threadIdx.x] = stage[threadIdx.x];
    }
}

hipError_t vg_final_launch(const VgMspace* ms, hipStream_t st) {
    if (!ms->dotp || !ms->r2 || !ms->r2l) return hipErrorInvalidValue;
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
// e_d = +1: Kuu_d = s_d K0 (R_d = sqrt(s_d) L0 Q);  e_d = -1: Kuu_d = K0 / s_d (inter-domain VFF / B1 features)
__global__ void vg_qv_weights_kernel(const double* theta, const double* beta, const double* invD, double* w, long n, int e1, int e2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double rs = sqrt((e1 > 0 ? theta[2] : 1.0 / theta[2]) * (e2 > 0 ? theta[3] : 1.0 / theta[3]));
        w[i] = beta[i] * rs / theta[4];
        w[n + i] = invD[i] - 1.0;
    }
}
hipError_t vg_qv_weights_launch(const double* theta, const double* beta, const double* invD, double* w_mean,
                                long n, hipStream_t st, int e1, int e2) {
    hipLaunchKernelGGL(vg_qv_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, theta, beta,
                       invD, w_mean, n, e1, e2);
    return hipGetLastError();
}

// read-out weights: w[0..n) = beta sqrt(s1 s2) / v (mean), w[n..2n) = D - 1 (literal reference) or 1/D - 1
__global__ void vg_readout_weights_kernel(const double* theta, const double* beta, const double* invD, double* w, long n, int literal) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        w[i] = beta[i] * sqrt(theta[2] * theta[3]) / theta[4];
        w[n + i] = literal ? 1.0 / invD[i] - 1.0 : invD[i] - 1.0;
    }
}
hipError_t vg_readout_weights_launch(const double* theta, const double* beta, const double* invD, double* w, long n, int literal,
                                     hipStream_t st) {
    hipLaunchKernelGGL(vg_readout_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, theta, beta, invD, w, n, literal);
    return hipGetLastError();
}
// var[a][b] = s1 s2 (kd1[a] kd2[b] + var[a][b])
__global__ void vg_readout_var_kernel(const double* theta, const double* kd1, const double* kd2, long mv1, long mv2, double* var) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < mv1 * mv2) var[i] = theta[2] * theta[3] * (kd1[i / mv2] * kd2[i % mv2] + var[i]);
}
hipError_t vg_readout_var_launch(const double* theta, const double* kd1, const double* kd2, long mv1, long mv2, double* var, hipStream_t st) {
    hipLaunchKernelGGL(vg_readout_var_kernel, dim3((unsigned)((mv1 * mv2 + 255) / 256)), dim3(256), 0, st, theta, kd1, kd2, mv1, mv2, var);
    return hipGetLastError();
}

// mode 0: x *= s1*s2 ; mode 1: x = s1*s2*(1 + x)
__global__ void vg_scale_kernel(double* x, long n, const double* theta, int mode, int e1, int e2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double s12 = (e1 > 0 ? theta[2] : 1.0 / theta[2]) * (e2 > 0 ? theta[3] : 1.0 / theta[3]);
        x[i] = mode == 0 ? x[i] * s12 : s12 * (1.0 + x[i]);
    }
}
hipError_t vg_scale_launch(double* x, long n, const double* theta, int mode, hipStream_t st, int e1, int e2) {
    hipLaunchKernelGGL(vg_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, theta, mode, e1, e2);
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


// ---- inducing-point gradient (vggp_zgrad, api.hip): the two element-wise stages ---------------------------------------------
// Weights of the lengthscale-gradient formula seen as a linear functional of (Mk, V): dELBO = <W_E, Q^T Mk Q> + <W_F, Q^T (H + H^T) Q>
// + <beta / v^2, P_d>  (oracle/kron.py z_grad).  One dimension per launch:
//   W_E  = -X / (2 v^2) + diag(-r / 2 + m_o / 2 - So lam / (2 v)),   W_F = -Xl / (2 v^3) + diag(-rl / (2 v) + So / (2 v)),
// lam = s_self lam0 (scaled eigenvalues), So = s_o sum(lam0_o); outputs W_E and W_F + W_F^T.
__global__ __launch_bounds__(256) void vg_zw_kernel(const double* theta, int self, const double* lam_self, const double* lam_other,
                                                   int m, int m_other, const double* X, const double* Xl, const double* r,
                                                   const double* rl, double* WE, double* WFs) {
    __shared__ double red[16];
    const double s_self = theta[2 + self], s_o = theta[3 - self], v = theta[4];
    double t = 0.0;
    for (int i = threadIdx.x; i < m_other; i += blockDim.x) t += lam_other[i];
    t = vg_block_sum(t, red);
    const double So = s_o * t;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)m * m; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / m), j = (int)(e - (long)i * m);
        double we = -X[e] / (2.0 * v * v), wf = -(Xl[e] + Xl[(long)j * m + i]) / (2.0 * v * v * v);
        if (i == j) {
            we += -0.5 * r[i] + 0.5 * (double)m_other - So * s_self * lam_self[i] / (2.0 * v);
            wf += 2.0 * (-0.5 * rl[i] / v + So / (2.0 * v));
        }
        WE[e] = we;
        WFs[e] = wf;
    }
}
hipError_t vg_zw_launch(const double* theta, int self, const double* lam_self, const double* lam_other, int m, int m_other,
                        const double* X, const double* Xl, const double* r, const double* rl, double* WE, double* WFs, hipStream_t st) {
    hipLaunchKernelGGL(vg_zw_kernel, dim3((unsigned)((m * m + 255) / 256)), dim3(256), 0, st, theta, self, lam_self, lam_other, m, m_other,
                       X, Xl, r, rl, WE, WFs);
    return hipGetLastError();
}

// g_i = -ell [ sum_k Abar_ik dA0_ik / (z_i - x_k) + sum_{j != i} (Kbar_ij + Kbar_ji) dK0_ij / (z_i - z_j) ]: for a stationary kernel
// d kappa(z_i, x) / d z_i = -(d kappa / d ell) ell / (z_i - x) (0 where z_i = x).  One workgroup per inducing point.
__global__ __launch_bounds__(256) void vg_zdot_kernel(const double* theta, int self, const double* z, const double* x, int m, long n,
                                                     const double* Abar, const double* dA0, const double* Kbar, const double* dK0,
                                                     double* out) {
    __shared__ double red[16];
    const int i = blockIdx.x;
    const double zi = z[i];
    double acc = 0.0;
    for (long k = threadIdx.x; k < n; k += blockDim.x) {
        const double d = zi - x[k];
        if (d != 0.0) acc += Abar[(long)i * n + k] * dA0[(long)i * n + k] / d;
    }
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        const double d = zi - z[j];
        if (j != i && d != 0.0) acc += (Kbar[(long)i * m + j] + Kbar[(long)j * m + i]) * dK0[(long)i * m + j] / d;
    }
    acc = vg_block_sum(acc, red);
    if (threadIdx.x == 0) out[i] = -theta[self] * acc;
}
hipError_t vg_zdot_launch(const double* theta, int self, const double* z, const double* x, int m, long n, const double* Abar,
                          const double* dA0, const double* Kbar, const double* dK0, double* out, hipStream_t st) {
    hipLaunchKernelGGL(vg_zdot_kernel, dim3(m), dim3(256), 0, st, theta, self, z, x, m, n, Abar, dA0, Kbar, dK0, out);
    return hipGetLastError();
}
