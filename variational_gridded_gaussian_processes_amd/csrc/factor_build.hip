// Per-dimension factor build at unit outputscale: A0 = Kuf_d / s_d  (m x n), K0 = Kuu_d / s_d
// (m x m) and their derivatives with respect to the lengthscale, fused in one pass.
//
// Replaces (reference paths relative to the reference checkout):
//   pairwise  kernel_d(Z), kernel(Z, x)            kronecker_structure.py:318-319, :336-337
//   B0-spline _Kuu_along_dim / _Kuf_along_dim       kronecker_structure.py:723-739, :768-790
//                                                  (== gridded_kronecker_structure.py:1307-1374)
// The kernel is HBM-write-bound: per output element one exp(), two 8-byte stores
// (value + d/d ell), coordinates re-read from L1/L2.  Lanes run along the observation
// index p (the contiguous axis of the row-major [m][n] outputs) so every wave store is a
// full 512-B line; the m+1 mesh / inducing coordinates are wave-uniform broadcasts.
#include "common.h"

#define VG_FB_MAXJOBS 4
struct VgFactorArgs {
    VgFactorJob job[VG_FB_MAXJOBS];
    int block_start[2 * VG_FB_MAXJOBS + 1];   // per job: A part then K part
    int njobs;
};

__device__ __forceinline__ void vg_kappa(int kind, double dist, double ell, double& v, double& dv) {
    const double r = dist / ell;
    if (kind == VGGP_KIND_MATERN12) {
        const double e = exp(-r);
        v = e;
        dv = e * r / ell;
    } else if (kind == VGGP_KIND_MATERN32) {
        const double a = 1.7320508075688772 * r;
        const double e = exp(-a);
        v = (1.0 + a) * e;
        dv = a * a * e / ell;
    } else if (kind == VGGP_KIND_MATERN52) {
        const double a = 2.23606797749979 * r;
        const double e = exp(-a);
        v = (1.0 + a + a * a / 3.0) * e;
        dv = (a * a / 3.0) * (1.0 + a) * e / ell;
    } else {   // RBF
        const double e = exp(-0.5 * r * r);
        v = e;
        dv = e * r * r / ell;
    }
}

// B0 cell-integral cross-covariance, cell k = (a, b], point x (kronecker_structure.py:768-790)
__device__ __forceinline__ void vg_b0_A(double a, double b, double x, double ell, double& v, double& dv) {
    const double ua = fabs(x - a), ub = fabs(x - b);
    const double ea = exp(-ua / ell), eb = exp(-ub / ell);
    const double E1 = ell * ea, E2 = ell * eb;
    const double dE1 = ea * (1.0 + ua / ell), dE2 = eb * (1.0 + ub / ell);
    if (x > a && x <= b) {
        v = 2.0 * ell - (E1 + E2);
        dv = 2.0 - (dE1 + dE2);
    } else {
        const double sg = (x <= a) ? 1.0 : -1.0;
        v = sg * (E1 - E2);
        dv = sg * (dE1 - dE2);
    }
}

// B0 cell-cell covariance, Toeplitz in kd = |i-j| (kronecker_structure.py:723-739), written in the
// cancellation-free form r_k = e^{-k t} 4 sinh^2(t/2), r_0 = 2 (expm1(-t) + t), t = delta/ell.
__device__ __forceinline__ void vg_b0_K(int kd, double delta, double ell, double& v, double& dv) {
    const double t = delta / ell;
    double r, dr;
    if (kd == 0) {
        const double em1 = expm1(-t);
        r = 2.0 * (em1 + t);
        dr = (2.0 * t / ell) * em1;
    } else {
        const double sh = sinh(0.5 * t);
        const double s4 = 4.0 * sh * sh;
        const double e = exp(-(double)kd * t);
        r = e * s4;
        dr = (t / ell) * e * ((double)kd * s4 - 2.0 * sinh(t));
    }
    v = ell * ell * r;
    dv = 2.0 * ell * r + ell * ell * dr;
}

// Reference-literal variant (VGGP_FLAG_B0_F32_KDELTA): the products c*delta are rounded to float32 as in
// the reference (float32 mesh attributes), then exp(-(c delta)_f32 / ell) in float64, three-term form.
__device__ __forceinline__ void vg_b0_K_f32(int kd, double delta, double ell, double& v, double& dv) {
    const float df = (float)delta;
    double r, dr;
    if (kd == 0) {
        const double t = (double)df / ell;
        const double e = exp(-t);
        r = 2.0 * (e + t - 1.0);
        dr = 2.0 * (e * t / ell - t / ell);
    } else {
        const double a0 = (double)((float)(kd - 1) * df), a1 = (double)((float)(kd + 1) * df),
                     a2 = (double)((float)kd * df);
        const double e0 = exp(-a0 / ell), e1 = exp(-a1 / ell), e2 = exp(-a2 / ell);
        r = e0 + e1 - 2.0 * e2;
        dr = (e0 * a0 + e1 * a1 - 2.0 * e2 * a2) / (ell * ell);
    }
    v = ell * ell * r;
    dv = 2.0 * ell * r + ell * ell * dr;
}

__global__ __launch_bounds__(256) void vg_factor_kernel(const VgFactorArgs args, const double* __restrict__ theta,
                                                       double* theta_copy, const VgClearArgs clr) {
    const int bid = blockIdx.x;
    if (bid == 0) {                                  // step prologue duties (see vg_factor_build_launch)
        if (theta_copy && threadIdx.x < 5) theta_copy[threadIdx.x] = theta[threadIdx.x];
        for (int k = 0; k < clr.n; ++k)
            for (int i = threadIdx.x; i < clr.nwords[k]; i += 256) clr.ptr[k][i] = 0;
    }
    int part = 0;
    for (int i = 1; i < 2 * args.njobs; ++i)
        if (bid >= args.block_start[i]) part = i;
    const VgFactorJob& J = args.job[part >> 1];
    const bool kpart = part & 1;
    const double ell = (J.theta_idx >= 0) ? theta[J.theta_idx] : J.ell_imm;
    const int m = J.m;
    const long ncols = kpart ? m : J.n;
    const long total = (long)m * ncols;
    const long idx = (long)(bid - args.block_start[part]) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx / ncols);
    const int p = (int)(idx - (long)k * ncols);
    double v, dv;
    if (J.basis == VGGP_BASIS_ONE) {
        v = 1.0;
        dv = 0.0;
    } else if (J.basis == VGGP_BASIS_B0) {
        if (kpart) {
            const int kd = k > p ? k - p : p - k;
            if (J.flags & VGGP_FLAG_B0_F32_KDELTA) vg_b0_K_f32(kd, J.grid[1] - J.grid[0], ell, v, dv);
            else vg_b0_K(kd, J.grid[1] - J.grid[0], ell, v, dv);
        } else {
            vg_b0_A(J.grid[k], J.grid[k + 1], J.x[p], ell, v, dv);
        }
    } else if (J.basis == VGGP_BASIS_VFF) {
        // grid = [a, b, omega_0 .. omega_M], m = 2M + 1: rows 0..M cosine features, M+1..2M sine features
        const double a = J.grid[0], b = J.grid[1];
        const int M = (m - 1) >> 1;
        if (kpart) {
            // unit-scale Kuu factor: diag(alpha0) + beta0 beta0^T, alpha0 = (b-a)/4 c (1/ell + w^2 ell), c = 2 at w = 0
            const double w = J.grid[2 + (k <= M ? k : k - M)];
            const double cc = (b - a) * 0.25 * (k == 0 ? 2.0 : 1.0);
            v = (k <= M && p <= M) ? 1.0 : 0.0;
            dv = 0.0;
            if (k == p) { v += cc * (1.0 / ell + w * w * ell); dv = cc * (-1.0 / (ell * ell) + w * w); }
        } else {
            const double x = J.x[p];
            const bool inside = x >= a && x < b;
            if (inside) {
                const double w = J.grid[2 + (k <= M ? k : k - M)];
                v = k <= M ? cos(w * (x - a)) : sin(w * (x - a));
                dv = 0.0;
            } else {
                const double r = fmin(fabs(x - a), fabs(x - b)), e = exp(-r / ell);
                v = k <= M ? e : 0.0;
                dv = k <= M ? r / (ell * ell) * e : 0.0;
            }
        }
    } else if (J.basis == VGGP_BASIS_B1) {
        // grid = knot mesh v_0 .. v_{m-1}
        const double d = J.grid[1] - J.grid[0];
        if (kpart) {
            // (A ell + B / ell + BC) / 2: A tridiagonal (2d/3, d/6; d/3 at the ends), B (2/d, -1/d; 1/d at the ends), BC = ends
            const int kd = k > p ? k - p : p - k;
            const bool end = (k == 0 || k == m - 1);
            double Aij = 0.0, Bij = 0.0, BCij = 0.0;
            if (kd == 0) { Aij = end ? d / 3.0 : 2.0 * d / 3.0; Bij = end ? 1.0 / d : 2.0 / d; BCij = end ? 1.0 : 0.0; }
            else if (kd == 1) { Aij = d / 6.0; Bij = -1.0 / d; }
            v = 0.5 * (Aij * ell + Bij / ell + BCij);
            dv = 0.5 * (Aij - Bij / (ell * ell));
        } else {
            const double x = J.x[p];
            const bool in = x >= J.grid[0] && x <= J.grid[m - 1];
            v = in ? fmax(0.0, 1.0 - fabs(x - J.grid[k]) / d) : 0.0;
            dv = 0.0;
        }
    } else {
        const double other = kpart ? J.grid[p] : J.x[p];
        vg_kappa(J.kind, fabs(J.grid[k] - other), ell, v, dv);
    }
    if (kpart) {
        if (J.K0) J.K0[idx] = v;
        if (J.dK0) J.dK0[idx] = dv;
    } else {
        if (J.A0) J.A0[idx] = v;
        if (J.dA0) J.dA0[idx] = dv;
    }
}

hipError_t vg_factor_build_launch(const VgFactorJob* jobs, int njobs, const double* theta_dev, hipStream_t st,
                                  double* theta_copy, const VgClearArgs* clr) {
    if (njobs < 1 || njobs > VG_FB_MAXJOBS) return hipErrorInvalidValue;
    VgFactorArgs a;
    a.njobs = njobs;
    int blocks = 0;
    for (int j = 0; j < njobs; ++j) {
        a.job[j] = jobs[j];
        const long na = (jobs[j].A0 || jobs[j].dA0) ? (long)jobs[j].m * jobs[j].n : 0;
        const long nk = (jobs[j].K0 || jobs[j].dK0) ? (long)jobs[j].m * jobs[j].m : 0;
        a.block_start[2 * j] = blocks;
        blocks += (int)((na + 255) / 256);
        a.block_start[2 * j + 1] = blocks;
        blocks += (int)((nk + 255) / 256);
    }
    a.block_start[2 * njobs] = blocks;
    if (blocks == 0) return hipSuccess;
    VgClearArgs c0;
    c0.n = 0;
    hipLaunchKernelGGL(vg_factor_kernel, dim3(blocks), dim3(256), 0, st, a, theta_dev, theta_copy, clr ? *clr : c0);
    return hipGetLastError();
}
