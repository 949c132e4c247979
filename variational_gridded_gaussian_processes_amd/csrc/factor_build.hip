// Per-dimension factor build at unit outputscale: A0 = Kuf_d / s_d  (m x n), K0 = Kuu_d / s_d
// (m x m) and their derivatives with respect to the lengthscale, fused in one pass.
//
// Replaces (reference paths relative to the reference checkout):
//   pairwise  kernel_d(Z), kernel(Z, x)            kronecker_structure.py:318-319, :336-337
//   B0-spline _Kuu_along_dim / _Kuf_along_dim       kronecker_structure.py:723-739, :768-790
//                                                  (== gridded_kronecker_structure.py:1307-1374)
// The kernel is HBM-write-bound (16 B per output element: value + d/d ell, one exp() each).  2-D tiling: a 256-thread
// workgroup owns an (inducing-feature rows) x (observation columns) tile, WIDE (tmw rows x 512 columns, the four waves
// side by side) when the matrix has >= 512 columns, TALL (4 tmw rows x 128 columns, one wave per row group) otherwise.
// Lanes run along the observation index p (the contiguous axis of the row-major [m][n] outputs) and own TWO adjacent
// columns: their coordinates are read once (one coalesced 16-B load) and stay in registers for all rows of the tile, and
// every store is a 16-B store (a wave writes 1 KiB of one output row).  The mesh / inducing coordinates of the tile's
// rows are staged once in LDS and read as wave-uniform broadcasts.  No per-thread division anywhere (one 32-bit division
// per workgroup maps the block index to its tile).  B0 cell integrals: cell k's right edge is cell k+1's left edge, so a
// thread walking down the rows evaluates ONE exponential per element, not two.
#include "common.h"
#include "factor_elem.h"

#define VG_FB_MAXJOBS 4
struct VgFactorArgs {
    VgFactorJob job[VG_FB_MAXJOBS];
    int block_start[2 * VG_FB_MAXJOBS + 1];   // per job: A part then K part
    int njobs;
};

#define VG_FB_TN_WIDE 512
#define VG_FB_TN_TALL 128
#define VG_FB_MAXROWS 64          // rows per tile (tall tiles: 4 * tmw), bounds the LDS staging of the row coordinates

struct VgFbPart {                 // one output matrix pair (value, d/d ell) of one job
    int job, kpart;               // kpart: the m x m matrix K0 (columns = inducing features) instead of the m x n matrix A0
    int ncols, tiles_c, tmw, wide, block_start;
};
struct VgFactorArgs2 {
    VgFactorJob job[VG_FB_MAXJOBS];
    VgFbPart part[2 * VG_FB_MAXJOBS];
    int nparts;
};

__global__ __launch_bounds__(256) void vg_factor_kernel(const VgFactorArgs2 args, const double* __restrict__ theta,
                                                       double* theta_copy, const VgClearArgs clr) {
    __shared__ double s_grid[VG_FB_MAXROWS + 2];
    const int bid = blockIdx.x, tid = threadIdx.x;
    if (bid == 0) {                                  // step prologue duties (see vg_factor_build_launch)
        if (theta_copy && tid < 6) theta_copy[tid] = theta[tid];     // 5 hyper-parameters + the step's sequence number
        for (int k = 0; k < clr.n; ++k)
            for (int i = tid; i < clr.nwords[k]; i += 256) clr.ptr[k][i] = 0;
    }
    int pi = 0;
    for (int i = 1; i < args.nparts; ++i)
        if (bid >= args.part[i].block_start) pi = i;
    const VgFbPart& P = args.part[pi];
    const VgFactorJob& J = args.job[P.job];
    const bool kpart = P.kpart != 0;
    const double ell = (J.theta_idx >= 0) ? theta[J.theta_idx] : J.ell_imm;
    const int m = J.m, ncols = P.ncols, tmw = P.tmw;
    const int t = bid - P.block_start;
    const int tr = t / P.tiles_c, tc = t - tr * P.tiles_c;            // the only division: once per workgroup, 32-bit
    const int lane = tid & 63, wave = tid >> 6;
    const int rows_tile = P.wide ? tmw : 4 * tmw;
    const int row0 = tr * rows_tile;
    const int c = tc * (P.wide ? VG_FB_TN_WIDE : VG_FB_TN_TALL) + (P.wide ? wave * 128 : 0) + 2 * lane;
    const int rw0 = P.wide ? 0 : wave * tmw;                         // this wave's first row inside the tile

    // stage the row coordinates of the tile (mesh knots k .. k+rows for B0 cells, inducing coordinates otherwise)
    const int glen = (J.basis == VGGP_BASIS_B0) ? m + 1 : m;
    const bool rowgrid = J.basis == VGGP_BASIS_B0 || J.basis == VGGP_BASIS_POINTS || J.basis == VGGP_BASIS_B1;
    if (tid <= rows_tile) {
        const int gi = row0 + tid;
        s_grid[tid] = (rowgrid && gi < glen) ? J.grid[gi] : 0.0;
    }
    // this lane's two column coordinates: one 16-B load when aligned
    const double* __restrict__ xsrc = kpart ? J.grid : J.x;
    const bool colcoord = !(kpart && (J.basis == VGGP_BASIS_VFF)) && J.basis != VGGP_BASIS_ONE && xsrc != nullptr;
    double x0 = 0.0, x1 = 0.0;
    if (colcoord) {
        if (c + 1 < ncols && ((reinterpret_cast<uintptr_t>(xsrc + c) & 15) == 0)) {
            const double2 xx = *reinterpret_cast<const double2*>(xsrc + c);
            x0 = xx.x; x1 = xx.y;
        } else {
            if (c < ncols) x0 = xsrc[c];
            if (c + 1 < ncols) x1 = xsrc[c + 1];
        }
    }
    __syncthreads();
    if (c >= ncols) return;
    double* __restrict__ O = kpart ? J.K0 : J.A0;
    double* __restrict__ dO = kpart ? J.dK0 : J.dA0;
    const bool vec = c + 1 < ncols && (ncols & 1) == 0 && ((reinterpret_cast<uintptr_t>(O) | reinterpret_cast<uintptr_t>(dO)) & 15) == 0;
    const bool two = c + 1 < ncols;

    if (J.basis == VGGP_BASIS_B0 && !kpart) {
        // cell k = (g_k, g_k+1]: E(g) = ell e^{-|x-g|/ell}, dE(g) = e^{-|x-g|/ell} (1 + |x-g|/ell); the right edge of one
        // row is the left edge of the next: one exponential per element
        auto edge = [&](double g, double x, double& E, double& dE) {
            const double u = fabs(x - g), e = exp(-u / ell);
            E = ell * e;
            dE = e * (1.0 + u / ell);
        };
        double Ea0, dEa0, Ea1, dEa1;
        edge(s_grid[rw0], x0, Ea0, dEa0);
        edge(s_grid[rw0], x1, Ea1, dEa1);
        for (int r = 0; r < tmw; ++r) {
            const int k = row0 + rw0 + r;
            if (k >= m) break;
            const double a = s_grid[rw0 + r], b = s_grid[rw0 + r + 1];
            double Eb0, dEb0, Eb1, dEb1;
            edge(b, x0, Eb0, dEb0);
            edge(b, x1, Eb1, dEb1);
            double v0, d0, v1, d1;
            if (x0 > a && x0 <= b) { v0 = 2.0 * ell - (Ea0 + Eb0); d0 = 2.0 - (dEa0 + dEb0); }
            else { const double sg = (x0 <= a) ? 1.0 : -1.0; v0 = sg * (Ea0 - Eb0); d0 = sg * (dEa0 - dEb0); }
            if (x1 > a && x1 <= b) { v1 = 2.0 * ell - (Ea1 + Eb1); d1 = 2.0 - (dEa1 + dEb1); }
            else { const double sg = (x1 <= a) ? 1.0 : -1.0; v1 = sg * (Ea1 - Eb1); d1 = sg * (dEa1 - dEb1); }
            const long o = (long)k * ncols + c;
            if (vec) {
                if (O) *reinterpret_cast<double2*>(O + o) = make_double2(v0, v1);
                if (dO) *reinterpret_cast<double2*>(dO + o) = make_double2(d0, d1);
            } else {
                if (O) { O[o] = v0; if (two) O[o + 1] = v1; }
                if (dO) { dO[o] = d0; if (two) dO[o + 1] = d1; }
            }
            Ea0 = Eb0; dEa0 = dEb0; Ea1 = Eb1; dEa1 = dEb1;
        }
        return;
    }
    if (J.basis == VGGP_BASIS_B0 && kpart && !(J.flags & VGGP_FLAG_B0_F32_KDELTA)) {
        // Toeplitz in kd = |k - p| (vg_b0_K): everything but e^{-kd t} is a constant of the launch -- one exponential per element
        const double t = (J.grid[1] - J.grid[0]) / ell, sh = sinh(0.5 * t), s4 = 4.0 * sh * sh, sh2 = 2.0 * sinh(t);
        const double em1 = expm1(-t), l2 = ell * ell, toe = t / ell;
        const double v_diag = l2 * 2.0 * (em1 + t), d_diag = 2.0 * ell * 2.0 * (em1 + t) + l2 * (2.0 * toe) * em1;
        auto el = [&](int kd, double& v, double& dv) {
            const double e = exp(-(double)kd * t), r = e * s4, dr = toe * e * ((double)kd * s4 - sh2);
            v = kd ? l2 * r : v_diag;
            dv = kd ? 2.0 * ell * r + l2 * dr : d_diag;
        };
        for (int r = 0; r < tmw; ++r) {
            const int k = row0 + rw0 + r;
            if (k >= m) break;
            double v0, d0, v1, d1;
            el(k > c ? k - c : c - k, v0, d0);
            el(k > c + 1 ? k - c - 1 : c + 1 - k, v1, d1);
            const long o = (long)k * ncols + c;
            if (vec) {
                if (O) *reinterpret_cast<double2*>(O + o) = make_double2(v0, v1);
                if (dO) *reinterpret_cast<double2*>(dO + o) = make_double2(d0, d1);
            } else {
                if (O) { O[o] = v0; if (two) O[o + 1] = v1; }
                if (dO) { dO[o] = d0; if (two) dO[o + 1] = d1; }
            }
        }
        return;
    }
    for (int r = 0; r < tmw; ++r) {
        const int k = row0 + rw0 + r;
        if (k >= m) break;
        const double gk = s_grid[rw0 + r], gk1 = s_grid[rw0 + r + 1];
        double v0, d0, v1 = 0.0, d1 = 0.0;
        vg_factor_elem(J, kpart, k, c, x0, gk, gk1, ell, v0, d0);
        if (two) vg_factor_elem(J, kpart, k, c + 1, x1, gk, gk1, ell, v1, d1);
        const long o = (long)k * ncols + c;
        if (vec) {
            if (O) *reinterpret_cast<double2*>(O + o) = make_double2(v0, v1);
            if (dO) *reinterpret_cast<double2*>(dO + o) = make_double2(d0, d1);
        } else {
            if (O) { O[o] = v0; if (two) O[o + 1] = v1; }
            if (dO) { dO[o] = d0; if (two) dO[o + 1] = d1; }
        }
    }
}

hipError_t vg_factor_build_launch(const VgFactorJob* jobs, int njobs, const double* theta_dev, hipStream_t st,
                                  double* theta_copy, const VgClearArgs* clr) {
    if (njobs < 1 || njobs > VG_FB_MAXJOBS) return hipErrorInvalidValue;
    VgFactorArgs2 a;
    a.nparts = 0;
    int blocks = 0;
    long total = 0;
    for (int j = 0; j < njobs; ++j) {
        total += ((jobs[j].A0 || jobs[j].dA0) ? (long)jobs[j].m * jobs[j].n : 0) + ((jobs[j].K0 || jobs[j].dK0) ? (long)jobs[j].m * jobs[j].m : 0);
    }
    // rows per wave: small launches (the step's 2 MB of factors) are latency-bound and want many workgroups; large ones
    // amortise the per-thread set-up (coordinate loads, LDS staging) over more rows
    const int tmw = total >= (1L << 24) ? 16 : (total >= (1L << 21) ? 8 : (total >= (1L << 18) ? 4 : 2));
    for (int j = 0; j < njobs; ++j) {
        a.job[j] = jobs[j];
        for (int kp = 0; kp < 2; ++kp) {
            const bool want = kp ? (jobs[j].K0 || jobs[j].dK0) : (jobs[j].A0 || jobs[j].dA0);
            const int ncols = kp ? jobs[j].m : jobs[j].n;
            if (!want || ncols <= 0 || jobs[j].m <= 0) continue;
            VgFbPart& P = a.part[a.nparts++];
            P.job = j; P.kpart = kp; P.ncols = ncols; P.tmw = tmw;
            P.wide = ncols >= VG_FB_TN_WIDE ? 1 : 0;
            const int tn = P.wide ? VG_FB_TN_WIDE : VG_FB_TN_TALL, rows = P.wide ? tmw : 4 * tmw;
            P.tiles_c = (ncols + tn - 1) / tn;
            P.block_start = blocks;
            blocks += P.tiles_c * ((jobs[j].m + rows - 1) / rows);
        }
    }
    if (blocks == 0) return hipSuccess;
    VgClearArgs c0;
    c0.n = 0;
    hipLaunchKernelGGL(vg_factor_kernel, dim3(blocks), dim3(256), 0, st, a, theta_dev, theta_copy, clr ? *clr : c0);
    return hipGetLastError();
}
