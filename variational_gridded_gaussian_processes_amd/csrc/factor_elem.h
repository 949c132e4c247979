// Element formulas of the per-dimension factors (unit outputscale): value and d/d ell of Kuf_d / Kuu_d for every basis.
// Shared by the factor kernel (factor_build.hip) and the kernels that evaluate prior covariances on the fly (api.hip).
// Reference: kronecker_structure.py:318-319, :336-337 (pairwise), :723-739, :768-790 (B0), :400-480 (VFF), :560-628 (B1).
#pragma once
#include "common.h"

// inv_ell = 1 / ell (a launch constant): two f64 divisions per element become multiplications (the divisions were a third of
// the pairwise kernel's instructions: 3.8 -> TB/s at 8192^2, profiles/r2_factor_build_8192.txt)
__device__ __forceinline__ void vg_kappa(int kind, double dist, double inv_ell, double& v, double& dv) {
    const double r = dist * inv_ell;
    if (kind == VGGP_KIND_MATERN12) {
        const double e = exp(-r);
        v = e;
        dv = e * r * inv_ell;
    } else if (kind == VGGP_KIND_MATERN32) {
        const double a = 1.7320508075688772 * r;
        const double e = exp(-a);
        v = (1.0 + a) * e;
        dv = a * a * e * inv_ell;
    } else if (kind == VGGP_KIND_MATERN52) {
        const double a = 2.23606797749979 * r;
        const double e = exp(-a);
        v = (1.0 + a + a * a * (1.0 / 3.0)) * e;
        dv = (a * a * (1.0 / 3.0)) * (1.0 + a) * e * inv_ell;
    } else {   // RBF
        const double e = exp(-0.5 * r * r);
        v = e;
        dv = e * r * r * inv_ell;
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

// one element of any basis; x = coordinate of column p (an observation for the A part, an inducing coordinate for the K part)
__device__ __forceinline__ void vg_factor_elem(const VgFactorJob& J, bool kpart, int k, int p, double x, double gk, double gk1,
                                               double ell, double& v, double& dv) {
    const int m = J.m;
    if (J.basis == VGGP_BASIS_ONE) {
        v = 1.0;
        dv = 0.0;
    } else if (J.basis == VGGP_BASIS_B0) {
        if (kpart) {
            const int kd = k > p ? k - p : p - k;
            if (J.flags & VGGP_FLAG_B0_F32_KDELTA) vg_b0_K_f32(kd, J.grid[1] - J.grid[0], ell, v, dv);
            else vg_b0_K(kd, J.grid[1] - J.grid[0], ell, v, dv);
        } else {
            vg_b0_A(gk, gk1, x, ell, v, dv);
        }
    } else if (J.basis == VGGP_BASIS_VFF) {
        // grid = [a, b, omega_0 .. omega_M], m = 2M + 1: rows 0..M cosine features, M+1..2M sine features
        const double a = J.grid[0], b = J.grid[1];
        const int M = (m - 1) >> 1;
        const double w = J.grid[2 + (k <= M ? k : k - M)];
        if (kpart) {
            // unit-scale Kuu factor: diag(alpha0) + beta0 beta0^T, alpha0 = (b-a)/4 c (1/ell + w^2 ell), c = 2 at w = 0
            const double cc = (b - a) * 0.25 * (k == 0 ? 2.0 : 1.0);
            v = (k <= M && p <= M) ? 1.0 : 0.0;
            dv = 0.0;
            if (k == p) { v += cc * (1.0 / ell + w * w * ell); dv = cc * (-1.0 / (ell * ell) + w * w); }
        } else {
            const bool inside = x >= a && x < b;
            if (inside) {
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
            const bool in = x >= J.grid[0] && x <= J.grid[m - 1];
            v = in ? fmax(0.0, 1.0 - fabs(x - gk) / d) : 0.0;
            dv = 0.0;
        }
    } else {
        vg_kappa(J.kind, fabs(gk - x), 1.0 / ell, v, dv);
    }
}

