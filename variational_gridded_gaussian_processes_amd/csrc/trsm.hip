// Batched triangular solves  L X = R  /  L^T X = R  on the CDNA4 matrix cores, by substitution.
//
// Replaces the triangular solves hidden inside lazify(Kuu).inv_matmul(Kuf) (kronecker_structure.py:269, :170-172,
// :217-227, :844-848): B = L^{-1} Kuf_d, V = L^{-1} dKuf_d/d ell, X = L^{-1} dKuu_d/d ell for the per-dimension Cholesky
// factors (m <= 256), and the diagonal-block solves of the Kronecker solve (vggp_kron_solve).
//
// Why substitution and not "explicit inverse times right-hand side": on the ill-conditioned RBF factors (cond(L) ~ 1e5 with
// the 1e-8 jitter) the product L^{-1} A carries the full conditioning of L into every element of B, and the posterior
// variance of a large grid (1e-6 of the prior variance at 4096 x 4096) then shows it at 2e-4 relative.  Blocked
// substitution with 16 x 16 diagonal blocks keeps the error at the level of LAPACK's dtrsm (measured in numpy on the same
// factors: 6e-8 against 9e-7 for the explicit inverse, tools/studies/trsm_accuracy.py).
//
// Mapping.  A wave owns a strip of 16 right-hand-side columns for ALL m rows: the strips are independent, so the
// substitution needs no workgroup barrier (one barrier after the factor has been staged in LDS).  Block row ib (16 rows) of the strip:
//     T     = R[ib] - sum_{jb < ib} L[ib][jb] X[jb]        4 x v_mfma_f64_16x16x4 per (ib, jb), accumulator = T
//     X[ib] = Dinv[ib] T                                   4 more; Dinv[ib] = inverse of the 16 x 16 diagonal block
// The accumulator layout D[row = (lane >> 4) + 4 r][col = lane & 15] IS the B-operand layout of the next MFMA
// (B[k = (lane >> 4) + 4 kk][j = lane & 15]), so T feeds the Dinv product and every finished X[ib] feeds the later block rows
// straight from registers: m / 16 blocks x 4 doubles per lane stay resident (64 doubles at m = 256).  L blocks are A
// operands read from LDS (m <= 128, staged once per workgroup) or from global memory (L2-resident; 16 rows x 32 B per load).
// L^T X = R runs the same code backwards (block rows from last to first) with the strides of L swapped.
// The 16 x 16 diagonal-block inverses come from the caller: the diagonal blocks of the explicit inverse the Cholesky
// kernel produces anyway (chol.hip) -- the diagonal blocks of L^{-1} are the inverses of the diagonal blocks of L -- or
// vg_tri_diaginv_launch below.
#include "common.h"

#include <cstdlib>

typedef double vg_td4 __attribute__((ext_vector_type(4)));

#define VG_TRSM_MAXJOBS 16
struct VgTrsmArgs {
    VgTrsmJob job[VG_TRSM_MAXJOBS];
    int block_start[VG_TRSM_MAXJOBS + 1];
    int njobs;
    int dbg;      // diagnostic builds of the timing study only (VGGP_TRSM_DBG): 1 = stop after staging, 2 = no MFMA chain
};

// NB = number of 16-row blocks the strip is unrolled for (m <= 16 NB); rows beyond m behave like an identity extension.
// LDSL: the lower triangle of L is staged once per workgroup in LDS (m <= 128: 128 x 130 doubles; row stride = 2 mod 32
// doubles, so the 16 rows x 2 k of a 32-lane ds_read_b64 group hit 32 distinct bank pairs) and every A operand is an LDS
// read.  All right-hand-side blocks of the strip are loaded before the first MFMA, so the substitution itself never waits
// for HBM.  (m > 128 is blocked by the caller: api.hip trsm_batch.)
template <int NB, bool LDSL>
__device__ __forceinline__ void vg_trsm_strip(const VgTrsmJob& J, long c0, bool active, double* sL, int dbg) {
    const int lane = threadIdx.x & 63;
    const int fi = lane & 15, fk = lane >> 4;
    const int m = J.m;
    const long ncols = J.ncols;
    const long col = c0 + fi;
    const bool cok = active && col < ncols;
    const double* __restrict__ Lg = J.L;
    const double* __restrict__ Rg = J.R;
    const double* __restrict__ Dg = J.Dinv;
    double* __restrict__ Xg = J.X;
    constexpr int S = 16 * NB + 2;
    // strides of the staged factor as seen by the substitution: element (i, k) of the lower-triangular operator
    const int si = J.trans ? 1 : S, sk = J.trans ? S : 1;
    const long di = J.trans ? 1 : J.dinv_ld, dk = J.trans ? J.dinv_ld : 1;
    // ---- every global load of the workgroup is requested up front, in ONE batch: the factor rows to stage (whole rows, L is
    // zero above the diagonal; a wave takes a row per load instruction, 16 B per lane), all right-hand-side blocks and all
    // diagonal-block inverses of the strip.  The operands were written by the previous kernels on other XCDs, so every
    // DEPENDENT batch of loads costs a full 2-4 us round trip (staging in four batches of eight rows: 26 us per launch).
    const int wave = threadIdx.x >> 6;
    const bool vec = (J.ldl & 1) == 0 && (reinterpret_cast<uintptr_t>(Lg) & 15) == 0;
    const int c2 = 2 * lane;
    constexpr int RW = 4 * NB;                  // rows of the factor per wave
    double2 v[RW];
    if (c2 < 16 * NB) {
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const int row = wave + 4 * u;             // (whole rows: predicating the lanes above the diagonal block off turns the
            const int rr = row < m ? row : 0, cc = c2 + 1 < m ? c2 : 0;               //  batch of loads into exec-mask regions: 18.0 vs 15.8 us)
            if (vec) v[u] = *reinterpret_cast<const double2*>(Lg + (long)rr * J.ldl + cc);
            else { v[u].x = Lg[(long)rr * J.ldl + cc]; v[u].y = Lg[(long)rr * J.ldl + cc + (cc + 1 < m ? 1 : 0)]; }
            if (c2 + 1 == m && row < m) v[u].x = Lg[(long)row * J.ldl + c2];            // odd m: last column of the row
        }
    }
    vg_td4 rb[NB], dvb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = b * 16 + fk + 4 * r;
            if (J.rhs_ident) rb[b][r] = (cok && row < m && row == col) ? 1.0 : 0.0;
            else rb[b][r] = (cok && row < m) ? Rg[(long)row * J.r_sk + col * J.r_sc] : 0.0;
            const int i = fi, k = fk + 4 * r;
            const bool in = b * 16 + i < m && b * 16 + k < m;
            dvb[b][r] = in ? Dg[(long)b * J.dinv_blk + (long)i * di + (long)k * dk] : (i == k ? 1.0 : 0.0);
        }
    // stage: zero outside the m x m matrix (a stray NaN in the padding would survive the multiplication by a zero operand)
    if (c2 < 16 * NB) {
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const int row = wave + 4 * u;
            double x = v[u].x, y = v[u].y;
            if (row >= m || c2 >= m) x = 0.0;
            if (row >= m || c2 + 1 >= m) y = 0.0;
            sL[row * S + c2] = x;
            sL[row * S + c2 + 1] = y;
        }
    }
    __syncthreads();
    if (!active) return;
    if (dbg == 1) { if (cok) Xg[col * J.x_sc] = rb[0][0] + dvb[0][0] + sL[lane]; return; }
    // A operands of one block row = 4 doubles per solved block.  They are read from LDS one block row AHEAD of their use
    // (the reads of row s+1 are issued before the MFMA chain of row s and land while it runs): a read placed next to its
    // MFMA exposes the LDS latency once per MFMA -- 176 times per strip at m = 128 (measured: 27 us per launch).
    constexpr int NA = 4 * (NB > 1 ? NB - 1 : 1);
    auto a_load = [&](int s, double (&av)[NA]) {
        const int ib = J.trans ? (NB - 1 - s) : s;
#pragma unroll
        for (int sp = 0; sp < NB - 1; ++sp) {
            if (sp >= s) break;
            const int jb = J.trans ? (NB - 1 - sp) : sp;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int i = ib * 16 + fi, k = jb * 16 + fk + 4 * kk;
                av[4 * sp + kk] = -sL[i * si + k * sk];
            }
        }
    };
    double a0[NA], a1[NA];
    vg_td4 xb[NB];
    auto row = [&](int s, double (&acur)[NA], double (&anext)[NA]) {
        // trans: L^T is upper triangular -> block rows from last to first; `ib` is the block row of L^T X = R
        const int ib = J.trans ? (NB - 1 - s) : s;
        if (s + 1 < NB) a_load(s + 1, anext);
        __builtin_amdgcn_sched_barrier(0);
        vg_td4 acc = rb[ib];
#pragma unroll
        for (int sp = 0; sp < NB - 1; ++sp) {
            if (sp >= s || dbg == 2) break;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[4 * sp + kk], xb[sp][kk], acc, 0, 0, 0);
        }
        // X[ib] = Dinv[ib] T  (rows of the block beyond m: identity)
        vg_td4 x = (vg_td4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(dvb[ib][kk], acc[kk], x, 0, 0, 0);
        xb[s] = x;
        __builtin_amdgcn_sched_barrier(0);
    };
    // (blocks of the identity extension -- rows >= m -- are staged as zeros with unit Dinv: they cost MFMAs, not correctness)
#pragma unroll
    for (int s = 0; s < NB; s += 2) {
        row(s, a0, a1);
        if (s + 1 < NB) row(s + 1, a1, a0);
    }
    // the solution blocks stayed in registers (they are the B operands of the later block rows): one burst of stores at the
    // end -- stores between the block rows put a wait in front of every MFMA chain (0.6 us per block row)
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        const int ib = J.trans ? (NB - 1 - s) : s;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rowi = ib * 16 + fk + 4 * r;
            if (cok && rowi < m) Xg[(long)rowi * J.x_sk + col * J.x_sc] = xb[s][r];
        }
    }
}

__global__ __launch_bounds__(256) void vg_trsm_kernel(const VgTrsmArgs a) {
    extern __shared__ double vg_trsm_lds[];
    const int bid = blockIdx.x;
    int ji = 0;
    for (int i = 1; i < a.njobs; ++i)
        if (bid >= a.block_start[i]) ji = i;
    const VgTrsmJob& J = a.job[ji];
    const long c0 = ((long)(bid - a.block_start[ji]) * 4 + (threadIdx.x >> 6)) * 16;
    const bool active = c0 < J.ncols;
    const int nb = (J.m + 15) >> 4;
    // LDS rows of a transposed solve are read as columns: zero the staging area's unwritten part is not needed (reads
    // stay inside the lower triangle's blocks: block (ib, jb) with jb < ib, or its mirror for trans)
    if (nb <= 1) vg_trsm_strip<1, true>(J, c0, active, vg_trsm_lds, a.dbg);
    else if (nb <= 2) vg_trsm_strip<2, true>(J, c0, active, vg_trsm_lds, a.dbg);
    else if (nb <= 4) vg_trsm_strip<4, true>(J, c0, active, vg_trsm_lds, a.dbg);
    else vg_trsm_strip<8, true>(J, c0, active, vg_trsm_lds, a.dbg);
}

hipError_t vg_trsm_setup() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(vg_trsm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               128 * 130 * 8);
}

hipError_t vg_trsm_launch(const VgTrsmJob* jobs, int njobs, hipStream_t st) {
    if (njobs < 1 || njobs > VG_TRSM_MAXJOBS) return hipErrorInvalidValue;
    VgTrsmArgs a;
    a.njobs = njobs;
    int blocks = 0;
    for (int j = 0; j < njobs; ++j) {
        if (jobs[j].m < 1 || jobs[j].m > 128) return hipErrorInvalidValue;      // larger factors: blocked by the caller (api.hip)
        a.job[j] = jobs[j];
        a.block_start[j] = blocks;
        blocks += (int)((jobs[j].ncols + 63) / 64);
    }
    a.block_start[njobs] = blocks;
    static const char* dbg = getenv("VGGP_TRSM_DBG");
    a.dbg = dbg ? atoi(dbg) : 0;
    if (blocks == 0) return hipSuccess;
    size_t lds = 0;
    for (int j = 0; j < njobs; ++j) {
        const int nb = (jobs[j].m + 15) / 16;
        const int nbt = nb <= 1 ? 1 : nb <= 2 ? 2 : nb <= 4 ? 4 : 8;                     // template instance
        const size_t need = (size_t)(16 * nbt) * (16 * nbt + 2) * sizeof(double);
        if (need > lds) lds = need;
    }
    hipLaunchKernelGGL(vg_trsm_kernel, dim3(blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

// ---- inverses of the 16 x 16 diagonal blocks of a lower-triangular matrix ------------------------------------------------
// One wave per block: lane (i = lane & 15, c4 = lane >> 4) owns row i of columns 4 c4 .. 4 c4 + 3 of X = D^{-1}; forward
// substitution row by row (x_i = (e_i - sum_{k<i} d_ik x_k) / d_ii), rows exchanged with wave shuffles.  Out: [nblk][16][16].
__global__ __launch_bounds__(256) void vg_tri_diaginv_kernel(const double* __restrict__ L, long ldl, int m, double* __restrict__ out,
                                                             const double* __restrict__ Lb, long ldlb, int mb, double* __restrict__ outb,
                                                             int nblk_a) {
    int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= nblk_a) {                        // second matrix of the launch
        blk -= nblk_a; L = Lb; ldl = ldlb; m = mb; out = outb;
        if (!L) return;
    }
    if (blk * 16 >= m) return;
    const int lane = threadIdx.x & 63, i = lane & 15, c4 = lane >> 4;
    const int r0 = blk * 16;
    double d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = (r0 + i < m && r0 + k < m && k <= i) ? L[(long)(r0 + i) * ldl + r0 + k] : (i == k ? 1.0 : 0.0);
    double x[4] = {0.0, 0.0, 0.0, 0.0};
    // row r of X becomes final at step r; every later row subtracts d[i][r] * X[r][:]
    double accv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) accv[q] = (i == 4 * c4 + q) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double piv = __shfl(d[r], (lane & 48) | r, 64);             // d[r][r] lives in lane row r (same column group)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double xr = __shfl(accv[q], (lane & 48) | r, 64) / piv;    // finished X[r][4 c4 + q]
            if (i == r) x[q] = xr;
            else if (i > r) accv[q] -= d[r] * xr;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) out[(long)blk * 256 + i * 16 + 4 * c4 + q] = x[q];
}

hipError_t vg_tri_diaginv_launch(const double* L, long ldl, int m, double* out, hipStream_t st, const double* L2, long ldl2, int m2,
                                 double* out2) {
    const int nblk = (m + 15) / 16, nblk_a = ((nblk + 3) / 4) * 4, nblk2 = L2 ? (m2 + 15) / 16 : 0;
    hipLaunchKernelGGL(vg_tri_diaginv_kernel, dim3((nblk_a + nblk2 + 3) / 4), dim3(256), 0, st, L, ldl, m, out, L2, ldl2, m2, out2, nblk_a);
    return hipGetLastError();
}
