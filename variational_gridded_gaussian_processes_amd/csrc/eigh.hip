// Symmetric eigendecomposition G = Q diag(lam) Q^T of the m_d x m_d Gram factors by parallel
// cyclic (round-robin) two-sided Jacobi.  This is what makes the full-grid ELBO closed-form:
// Sigma = Kuu + Kuf Kuf^T / sigma^2 (kronecker_structure.py:134-150) is diagonalised by
// (L1 Q1) (x) (L2 Q2) -- SURVEY.md section 7.0 -- so no M x M matrix is ever formed.
//
// Kernel 1 (vg_jacobi_kernel, one workgroup of 1024 threads per matrix, G resident in LDS):
//   each round applies m/2 disjoint plane rotations.  The rotation angles come from the 2x2
//   diagonal blocks; every 2x2 block G[{p_a,q_a},{p_b,q_b}] is then updated by ONE thread with
//   both its row and its column rotation (J_a^T . J_b), so each element is read and written
//   once per round and a round needs two barriers.  Rounds in which no pair exceeds the
//   threshold are skipped after the angle phase.  Rotations are appended to a log instead of
//   being applied to the eigenvector matrix here (that would not fit LDS next to G).
// Kernel 2 (vg_replay_kernel, m/16 workgroups per matrix): replays the rotation log on a block
//   of columns of Q^T; every wave owns its columns outright, so there is no barrier per round.
#include "gemm_body.h"

#include <algorithm>
#include <cstdlib>

#define VG_EIG_TOL 1e-13         // default relative off-diagonal threshold (VgEigJob::tol)
#define VG_SUB_MISS 1e-12        // subspace start: largest admissible Rayleigh quotient of a complement row, relative to lam_max
#define VG_EIG_TOL_SUB 1e-12     // ... of a sparse_first job (see vg_jacobi_body)
#ifndef VG_BJ_MAX_M
#define VG_BJ_MAX_M 128            // block Jacobi up to this size, scalar cyclic Jacobi beyond
#endif
#define VG_EIG_DONE (1 << 30)   // progress word: rounds published | DONE
#define VG_EIG_POLISH (1 << 28) // ... | POLISH: after the logged rotations, Q^T <- (I + E + E^2/2) Q^T with E in gwork
#define VG_EIG_DIRECT (1 << 27) // ... | DIRECT: the producer wrote Q^T itself (Newton start, see vg_newton_diag): nothing to replay or store
#define VG_EIG_NEWTON_MAX_M 48   // largest problem the Newton start takes (five m x m LDS buffers behind the two packed copies of G)
#define VG_POLISH_EMAX 1e-3     // largest first-order rotation the polish accepts
typedef double vg_bd4 __attribute__((ext_vector_type(4)));
#define VG_EIG_LAG 9             // rounds whose log stores may still be in flight: vmcnt(16) with >= 2 VMEM ops per storing wave per round, +1
#ifndef VG_SPARSE_OK
#define VG_SPARSE_OK 1
#endif
#define VG_UB 3            // blocks of one thread whose loads are batched
#define VG_MAXMINE 6       // >= ceil(half*(half+1)/2 / 1024) for every m that fits LDS (half <= 92)

struct VgEigArgs {
    VgEigJob job[2];
    int njobs;
    int use_lds[2];
    int fast[2];       // dense sweeps with fixed addresses (two copies of G fit LDS)
    int rp_cols;       // replay columns per workgroup (4 per working wave)
    int nx;            // workgroups per job: 1 producer + nx - 1 replay
    int neig;          // njobs * nx; linear block ids >= neig are tiles of the rider
};

// circle-method pairing of m2 (even) players in round r: pair index k -> (p, q)
__device__ __forceinline__ void vg_pair(int m2, int r, int k, int& p, int& q) {
    const int n1 = m2 - 1;
    if (k == 0) { p = r; q = n1; return; }
    p = r + k; if (p >= n1) p -= n1;
    q = r - k; if (q < 0) q += n1;
}

__device__ __forceinline__ int vg_tri(int i) { return (i * (i + 1)) >> 1; }
// canonical (lower-triangular, packed) address of the symmetric element (i, j)
__device__ __forceinline__ int vg_sym(int i, int j) { return i >= j ? vg_tri(i) + j : vg_tri(j) + i; }

// per-pair record kept in LDS for the round: indices and their packed-row offsets (no integer multiplies in
// the update loop: v_mul_lo_u32 is quarter rate), rotation (c, s)
struct __attribute__((aligned(16))) VgPairRec { int p, q, tp, tq; };
// a rotating pair of the current round, compacted: indices, packed-row offsets, rotation, pair index k
struct __attribute__((aligned(16))) VgActRec { int p, q, tp, tq; double c, s; int k, pad; };
__device__ __forceinline__ int vg_symo(int i, int ti, int j, int tj) { return i >= j ? ti + j : tj + i; }

// one 2x2 block G[{pa,qa},{pb,qb}] <- Ja^T . Jb, read once / written once at its canonical addresses
__device__ __forceinline__ void vg_block(double* W, const VgPairRec& A, double ca, double sa, const VgPairRec& B,
                                         double cb, double sb) {
    const int a00 = vg_symo(A.p, A.tp, B.p, B.tp), a01 = vg_symo(A.p, A.tp, B.q, B.tq);
    const int a10 = vg_symo(A.q, A.tq, B.p, B.tp), a11 = vg_symo(A.q, A.tq, B.q, B.tq);
    const double g00 = W[a00], g01 = W[a01], g10 = W[a10], g11 = W[a11];
    const double h00 = cb * g00 - sb * g01, h01 = sb * g00 + cb * g01;
    const double h10 = cb * g10 - sb * g11, h11 = sb * g10 + cb * g11;
    W[a00] = ca * h00 - sa * h10;
    W[a10] = sa * h00 + ca * h10;
    W[a01] = ca * h01 - sa * h11;
    W[a11] = sa * h01 + ca * h11;
}
// diagonal block of pair (p, q): three distinct elements
__device__ __forceinline__ void vg_block_diag(double* W, const VgPairRec& A, double c, double s) {
    const int app = A.tp + A.p, aqq = A.tq + A.q, apq = vg_symo(A.p, A.tp, A.q, A.tq);
    const double gpp = W[app], gqq = W[aqq], gpq = W[apq];
    const double cc = c * c, ss = s * s, cs2 = 2.0 * c * s * gpq;
    W[app] = cc * gpp - cs2 + ss * gqq;
    W[aqq] = ss * gpp + cs2 + cc * gqq;
    W[apq] = (cc - ss) * gpq + c * s * (gpp - gqq);
}

// fast reciprocal / reciprocal square root: hardware seed + two Newton steps (full double accuracy for
// normal inputs; the rotation only needs c^2 + s^2 = 1 to rounding, not a correctly rounded angle)
__device__ __forceinline__ double vg_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double vg_rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = fma(y * 0.5, fma(-x * y, y, 1.0), y);
    y = fma(y * 0.5, fma(-x * y, y, 1.0), y);
    return y;
}

// Workgroup barrier of the round loop.  With G in LDS only LDS traffic has to be ordered, so the barrier
// waits on lgkmcnt alone: a plain __syncthreads() also emits s_waitcnt vmcnt(0) and would stall every round
// (~1.5 us) on the acknowledgement of the rotation-log global stores, which nobody in this kernel reads.
template <bool INLDS>
__device__ __forceinline__ void vg_round_barrier() {
    if (INLDS) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        __threadfence_block();
        __syncthreads();
    }
}


// =====================================================================================================================
// Dense sweeps with FIXED addresses (m2 <= 128, two packed copies of G in LDS).
// vg_pair() is the circle method: player x sits at relative position u = (x - r) mod n1 in round r (n1 = m2 - 1, the
// last player never moves) and pair k is always (position k, position n1 - k).  If the matrix is stored BY POSITION
// and physically shifted by one position per round, every 2x2 block (al, be) = rows {al, n1-al} x cols {be, n1-be}
// lives at the same addresses in every round, and so do its destinations (rows/cols shifted by -1, wrapped).  So:
//   * a thread owns the same <= 2 off-diagonal blocks for the whole solve; their 4 read and 4 write addresses are
//     computed once (the moving-index variant below spends more VALU issue on addresses than on the rotations);
//   * reads come from one copy, writes go to the other (ping-pong): ONE barrier per round;
//   * the diagonal is carried in closed form (d_p -= t g_pq, d_q += t g_pq, g_pq <- 0), so there is no angle phase:
//     the off-diagonal element of a pair of the NEXT round is always an output of one fixed block of THIS round; those
//     `half` blocks sit in wave 0, whose lanes compute the next rotations right after their block update while the other
//     15 waves are still moving data.
// The rotation sequence is the one of the moving-index variant (same pairs, same order), so the replay workgroups and
// the log format are unchanged.  A round costs a full pass even when few pairs rotate; when a sweep gets sparse the
// solver continues with the moving-index variant (inactive rounds ~10x cheaper) -- at a sweep boundary the position
// layout is the identity again.
#ifdef VG_EIG_STAMP
#define VG_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define VG_STAMP(var)
#endif
#define VG_FAST_LOGWAVE 15
__device__ __forceinline__ int vg_fshift(int u, int n1) { return u == n1 ? n1 : (u == 0 ? n1 - 1 : u - 1); }

struct VgAngle { double c, s, t; bool rot; };
__device__ __forceinline__ VgAngle vg_angle3(double gpp, double gqq, double gpq, double thr) {
    VgAngle a{1.0, 0.0, 0.0, fabs(gpq) > thr};
    if (a.rot) {
        const double d = gqq - gpp, o = 2.0 * gpq;
        const double ad = fabs(d), ao = fabs(o);
        const double ib = vg_rcp(fmax(ad, ao));
        const double dn = ad * ib, on = ao * ib;
        const double h2 = dn * dn + on * on;
        double t = on * vg_rcp(dn + h2 * vg_rsq(h2));
        if ((d >= 0.0) != (o >= 0.0)) t = -t;
        a.t = t;
        a.c = vg_rsq(1.0 + t * t);
        a.s = t * a.c;
    }
    return a;
}

// One scan of the strict lower triangle (identity layout, diagonal in Dc): E_ij = g_ij / (g_ii - g_jj) for the elements
// above thr, stored speculatively (strict lower triangle, row-major, write-through) in J.gwork; returns whether the polish
// R = I + E + E^2/2 may replace the remaining sweeps (see vg_jacobi_fast).  rs: 48 doubles of LDS scratch.
__device__ __forceinline__ bool vg_polish_scan(const VgEigJob& J, const double* Wc, const double* Dc, double thr, double* rs) {
    const int m = J.m, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    double sE = 0.0, sG = 0.0, mE = 0.0;
    double* E = J.gwork;
    for (int i = 1 + wave; i < m; i += (nthr >> 6)) {
        const int ti = vg_tri(i);
        const double di = Dc[i];
        for (int j = lane; j < i; j += 64) {
            const double g = Wc[ti + j];
            double e = 0.0;
            sG += g * g;
            if (fabs(g) > thr) {
                e = g / (di - Dc[j]);
                sE += e * e;
                mE = fmax(mE, fabs(e));
            }
            __hip_atomic_store(&E[i * m + j], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sE += __shfl_xor(sE, off); sG += __shfl_xor(sG, off); mE = fmax(mE, __shfl_xor(mE, off));
    }
    if (lane == 0) { rs[wave] = sE; rs[16 + wave] = sG; rs[32 + wave] = mE; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    sE = sG = mE = 0.0;
    for (int w = 0; w < (nthr >> 6); ++w) { sE += rs[w]; sG += rs[16 + w]; mE = fmax(mE, rs[32 + w]); }
    const double budget = (double)m * thr;
    return mE <= VG_POLISH_EMAX && 4.0 * sE * sG <= budget * budget;      // NaN / inf (equal diagonals) fail both
}

// returns the buffer that holds G (packed, identity layout, diagonal included) when the phase ends; converged is set
// when a whole sweep rotated nothing.  Wa must already hold the packed lower triangle.
__device__ __forceinline__ double* vg_jacobi_fast(const VgEigJob& J, double* Wa, double* Wb, double2* cs, double* Dd, int* nact_s, double thr,
                                  int& nlog, int& sweeps, int& status, bool& converged, bool& polished) {
    const int m = J.m, m2 = m + (m & 1), half = m2 >> 1, n1 = m2 - 1;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    // the start basis came out of vg_refine_launch: the matrix may already be within the polish bound
    if (J.polish && J.polish0) {
        double* Dt = Dd + 576;
        for (int i = tid; i < m; i += nthr) Dt[i] = Wa[vg_tri(i) + i];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (vg_polish_scan(J, Wa, Dt, thr, Dd + 520)) {
            polished = true;
            converged = true;
            if (tid == 0) { nact_s[0] = 0; nact_s[1] = 0; }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            return Wa;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // ---- block ownership (constant for the whole phase) ----
    int rd[2][4], wr[2][4], sal[2], sbe[2];
    bool ok[2] = {false, false};
    int sel = 0, dpos = 0, dqos = 0, dpw = 0, dqw = 0, wE = 0;
    const bool desig = wave == 0 && lane < half;
    {
        const int ngen = (half * (half - 1)) / 2 - half;            // blocks that are not designated
        const int per = nthr - 64;
        int g[2];
        // waves 1..15: one block from the front of the enumeration (short rows) and its mirror from the back (long rows),
        // so every wave carries the same mix; wave 0's second half-slot takes the blocks in the middle that remain
        const int hfront = (ngen + 1) / 2, t = tid - 64;
        if (wave == 0) { g[0] = -1; g[1] = (per + lane < ngen - per) ? per + lane : -1; }
        else { g[0] = (t < hfront) ? t : -1; g[1] = (ngen - 1 - t >= hfront) ? ngen - 1 - t : -1; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int al = 0, be = 0;
            if (u == 0 && wave == 0) {
                if (desig) {
                    const int k = lane;
                    if (k == 0) { al = 1; be = 0; sel = 1; }
                    else if (k == 1) { al = 2; be = 0; sel = 0; }
                    else if (k <= half - 2) { al = k + 1; be = k - 1; sel = 1; }
                    else { al = half - 1; be = half - 2; sel = 3; }
                    ok[u] = true;
                }
            } else if (g[u] >= 0 && g[u] < ngen) {
                int rest = g[u];
                al = 2;
                for (;;) {
                    const int cnt = al == half - 1 ? al - 2 : al - 1;
                    if (rest < cnt) break;
                    rest -= cnt;
                    ++al;
                }
                be = (al < half - 1 && rest >= al - 2) ? rest + 1 : rest;
                ok[u] = true;
            }
            sal[u] = al; sbe[u] = be;
            const int pa = al, qa = n1 - al, pb = be, qb = n1 - be;
            const int pa2 = vg_fshift(pa, n1), qa2 = vg_fshift(qa, n1), pb2 = vg_fshift(pb, n1), qb2 = vg_fshift(qb, n1);
            rd[u][0] = vg_sym(pa, pb); rd[u][1] = vg_sym(pa, qb); rd[u][2] = vg_sym(qa, pb); rd[u][3] = vg_sym(qa, qb);
            wr[u][0] = vg_sym(pa2, pb2); wr[u][1] = vg_sym(pa2, qb2); wr[u][2] = vg_sym(qa2, pb2); wr[u][3] = vg_sym(qa2, qb2);
        }
        if (desig) {
            dpos = lane; dqos = n1 - lane;                                  // this lane's pair, in any round's layout
            dpw = vg_fshift(dpos, n1); dqw = vg_fshift(dqos, n1);           // ... and one round later
            wE = vg_sym(dpw, dqw);
        }
    }
    double2* cs0 = cs;            // rotations of even rounds
    double2* cs1 = cs + 256;      // ... of odd rounds
    double* D0 = Dd;              // diag(G_R) for even R (position layout of round R)
    double* D1 = Dd + 256;
    // ---- prologue: rotations of round 0 from the diagonal blocks of the initial matrix ----
    double eprime = 0.0, tmax = 0.0;
    if (desig) {
        const double e = Wa[vg_sym(dpos, dqos)], dp = Wa[vg_tri(dpos) + dpos], dq = Wa[vg_tri(dqos) + dqos];
        const VgAngle a = vg_angle3(dp, dq, e, thr);
        cs0[lane] = make_double2(a.c, a.s);
        D1[dpw] = dp - a.t * e;
        D1[dqw] = dq + a.t * e;
        eprime = a.rot ? 0.0 : e;
        const unsigned long long bal = __ballot(a.rot);
        if (lane == 0) nact_s[0] = __popcll(bal);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    converged = false;
    int R = 0;
#ifdef VG_EIG_STAMP
    unsigned long long fU = 0, fA = 0, fB = 0, fN = 0, fs0, fs1;
#endif
    const int switch_below = (half * n1 * J.fast_switch) >> 8;     // continue here while >= this many PAIRS of a sweep rotate
    for (; sweeps < VG_EIG_MAXSWEEP && !status;) {
        int active_rounds = 0, rotations = 0;
        for (int r = 0; r < n1; ++r, ++R) {
            const int cur = R & 1;
#ifdef VG_EIG_STAMP
            VG_STAMP(fs0);
#endif
            const int na = nact_s[cur];
            const double* src = cur ? Wb : Wa;
            double* dst = cur ? Wa : Wb;
            const double2* csc = cur ? cs1 : cs0;
            double2* csn = cur ? cs0 : cs1;
            const double* Dr = cur ? D0 : D1;          // diag(G_{R+1}), written one round ago
            double* Dw = cur ? D1 : D0;                // receives diag(G_{R+2})
            if (na > 0) {
                if (nlog >= J.max_rounds) { status = VGGP_ENOCONV; break; }
                if (wave == VG_FAST_LOGWAVE) {
                    // same hand-off as below: publish the count of rounds that are certainly complete, then store
                    if (lane == 0 && nlog >= VG_EIG_LAG)
                        __hip_atomic_store(&J.counters[3], nlog - VG_EIG_LAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane < half) {
                        const double2 v = csc[lane];
                        double* lp = reinterpret_cast<double*>(&J.rotlog[(long)nlog * half + lane]);
                        __hip_atomic_store(lp, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(lp + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane == 0) __hip_atomic_store(&J.roundlog[nlog], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ++nlog;
                ++active_rounds;
                rotations += na;
            }
            if (desig) dst[wE] = eprime;               // in-pair element after this round's rotation (0, or untouched)
            double out0 = 0.0, out1 = 0.0, out2 = 0.0, out3 = 0.0;
#pragma unroll
            for (int u = 1; u >= 0; --u) {             // wave 0 does its ordinary half-slot first, the designated blocks last
                if (!ok[u]) continue;
                const double g0 = src[rd[u][0]], g1 = src[rd[u][1]], g2 = src[rd[u][2]], g3 = src[rd[u][3]];
                const double2 A = csc[sal[u]], B = csc[sbe[u]];
                const double h00 = B.x * g0 - B.y * g1, h01 = B.y * g0 + B.x * g1;
                const double h10 = B.x * g2 - B.y * g3, h11 = B.y * g2 + B.x * g3;
                out0 = A.x * h00 - A.y * h10;
                out2 = A.y * h00 + A.x * h10;
                out1 = A.x * h01 - A.y * h11;
                out3 = A.y * h01 + A.x * h11;
                dst[wr[u][0]] = out0; dst[wr[u][1]] = out1; dst[wr[u][2]] = out2; dst[wr[u][3]] = out3;
            }
#ifdef VG_EIG_STAMP
            VG_STAMP(fs1); fU += fs1 - fs0;
#endif
            if (desig) {
                // rotation of round R+1 for this lane's pair: its off-diagonal element is one of the outputs above
                const double e = sel == 0 ? out0 : (sel == 1 ? out1 : out3);
                const double dp = Dr[dpos], dq = Dr[dqos];
                const VgAngle a = vg_angle3(dp, dq, e, thr);
                csn[lane] = make_double2(a.c, a.s);
                Dw[dpw] = dp - a.t * e;
                Dw[dqw] = dq + a.t * e;
                eprime = a.rot ? 0.0 : e;
                const unsigned long long bal = __ballot(a.rot);
                if (lane == 0) nact_s[cur ^ 1] = __popcll(bal);
                tmax = fmax(tmax, fabs(a.t));
                if (r == n1 - 1) {                     // free screen for the polish: no large angle in the whole sweep
                    const unsigned long long big = __ballot(tmax > VG_POLISH_EMAX);
                    if (lane == 0) Dd[512] = big ? 1.0 : 0.0;
                    tmax = 0.0;
                }
            }
#ifdef VG_EIG_STAMP
            VG_STAMP(fs0); fA += fs0 - fs1;
#endif
            if (wave == VG_FAST_LOGWAVE && na > 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef VG_EIG_STAMP
            VG_STAMP(fs1); fB += fs1 - fs0; ++fN;
#endif
        }
        if (status) break;
        ++sweeps;
        if (active_rounds == 0) { converged = true; break; }
        // ---- first-order polish instead of the remaining sweeps.  With every off-diagonal element small against the gap of
        // its diagonal pair, the rotation that finishes the job is R = I + E + E^2/2, E_ij = g_ij / (g_ii - g_jj) (skew;
        // R R^T = I + E^4/4), what is left behind is E Goff + (E Goff)^T to leading order, and the eigenvalues move by
        // sum_j E_ij g_ij -- all bounded by ||E||_F ||Goff||_F.  When that bound is below the total the element-wise
        // threshold admits anyway (m thr = tol ||G||_F), E goes to gwork and the replay workgroups apply R: the second
        // dense sweep of a warm start on a well-separated spectrum (Matern kernels; ~127 rounds) becomes one scan.
        // Clustered spectra (RBF: dozens of numerically null eigenvalues, |E| ~ 1 among them) never pass the test.
        // Tried only when the next sweep would be another dense one (a sparse tail is cheaper than the polish itself).
        if (J.polish && rotations >= switch_below && Dd[512] == 0.0) {
            const double* Wc = (R & 1) ? Wb : Wa;           // off-diagonals of the current matrix, identity layout
            const double* Dc = (R & 1) ? D1 : D0;           // its diagonal
            if (vg_polish_scan(J, Wc, Dc, thr, Dd + 520)) {
                polished = true;
                converged = true;
                break;
            }
        }
        if (rotations < switch_below) break;
        if (sweeps >= VG_EIG_MAXSWEEP) status = VGGP_ENOCONV;
    }
    // ---- epilogue: identity layout again (R is a multiple of n1); restore the diagonal, drain the log wave ----
#ifdef VG_EIG_STAMP
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(J.gwork) + 64 + wave * 4;
        dbg[0] = fU; dbg[1] = fA; dbg[2] = fB; dbg[3] = fN;
    }
#endif
    double* Wfin = (R & 1) ? Wb : Wa;
    const double* Df = (R & 1) ? D1 : D0;
    if (R == 0) {
        for (int i = tid; i < m2; i += nthr) (void)i;       // nothing ran: Wa is untouched
    } else {
        for (int i = tid; i < m2; i += nthr) Wfin[vg_tri(i) + i] = Df[i];
    }
    if (wave == VG_FAST_LOGWAVE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) { nact_s[0] = 0; nact_s[1] = 0; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return Wfin;
}


// ---- Newton start for small, nearly diagonal problems (the Ritz matrix of the subspace start) -------------------------------
// H0 = V1 G V1^T of a warm step is diagonal up to first order in the hyper-parameter change (tools/studies/
// ritz_sweeps_study.py: max |h_ij / (h_jj - h_ii)| = 7e-3 at a 1 % change of the lengthscale, r = 24), so the diagonalising
// rotation is found by a Newton iteration instead of Jacobi sweeps:  E_ij = h_ij / (h_jj - h_ii) for the elements above the
// threshold (skew), R = I + E + E^2 / 2, W <- W R re-orthonormalised by one Newton-Schulz step, H <- W^T H0 W from the ORIGINAL
// matrix (so neither the O(E^3) departure of R from orthogonality nor rounding accumulates).  Quadratic: 7e-3 -> 1e-5 -> 2e-10
// -> below the threshold, i.e. three iterations of six r x r x r products on the matrix cores against 46 Jacobi rounds
// of 0.63 us plus the hand-off to the replay workgroups.  A start that is not nearly diagonal (some |E_ij| >= 0.3, a zero gap)
// or does not converge in VG_NEWTON_MAXIT iterations returns false and the Jacobi sweeps run as usual from the untouched G.
// Buffers (mp x mp zero padded, mp = m rounded up to 16, row stride mp + 2): NB + {0: H0, 1: W, 2: H, 3: E / R / Gram, 4: products}.
#define VG_NEWTON_MAXIT 6
typedef double vg_nd4 __attribute__((ext_vector_type(4)));
// C = epilogue(op(A) B) on the matrix cores, all matrices mp x mp (mp = m rounded up to 16, zero padded) with row stride
// ld = mp + 2: wave w owns the 16 x 16 block w of C.  (A scalar product loop reads 16 bytes of LDS per multiply-add: 221 KB per
// 24^3 product, 0.7 us at the LDS rate -- measured 1.4 us; the MFMA fragments need 1 byte per multiply-add.)
// EPI 0: C = A B;  1: C = I + X + 0.5 A B (X read at C's own positions);  2: C = 1.5 I - 0.5 A B.   C must not alias A, B or X.
template <int EPI>
__device__ __forceinline__ void vg_nt_mm(double* C, const double* A, int sa_i, int sa_k, const double* B, int mp, int ld, const double* X = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nb = mp >> 4;
    const int fi = lane & 15, fk = lane >> 4;
    for (int blk = wave; blk < nb * nb; blk += blockDim.x >> 6) {
        const int bi = blk / nb, bj = blk - bi * nb;
        const double* a = A + (bi * 16 + fi) * sa_i + fk * sa_k;
        const double* b = B + fk * ld + bj * 16 + fi;
        vg_nd4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
        for (int k0 = 0; k0 < mp; k0 += 16) {
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = a[(k0 + 4 * u) * sa_k]; bv[u] = b[(k0 + 4 * u) * ld]; }
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = bi * 16 + fk + 4 * r, col = bj * 16 + fi;
            double v = acc0[r] + acc1[r];
            if (EPI == 1) v = ((row == col) ? 1.0 : 0.0) + X[row * ld + col] + 0.5 * v;
            if (EPI == 2) v = ((row == col) ? 1.5 : 0.0) - 0.5 * v;
            C[row * ld + col] = v;
        }
    }
    __syncthreads();
}
// (buffers: five mp x ld areas behind NB; on success *Hout / *Wout point at the diagonalised matrix and at W, whose COLUMNS are
//  the eigenvectors)
__device__ __forceinline__ bool vg_newton_diag(int m, const double* Wpk, double* NB, double thr, int* flags /* 4 ints of LDS */, int& iters,
                                               const double** Hout, const double** Wout) {
    const int mp = (m + 15) & ~15, ld = mp + 2, sz = mp * ld, tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    double *H0 = NB, *Wn = NB + sz, *H = NB + 2 * sz, *E = NB + 3 * sz, *T = NB + 4 * sz;
    for (int e = tid; e < sz; e += nthr) {
        const int i = e / ld, j = e - i * ld;
        const double v = (i < m && j < m) ? Wpk[vg_sym(i, j)] : 0.0;
        H0[e] = v; H[e] = v;
        Wn[e] = (i == j && i < m) ? 1.0 : 0.0;
        E[e] = 0.0; T[e] = 0.0;
    }
    if (tid < 4) flags[tid] = 0;
    __syncthreads();
    for (int it = 0; it < VG_NEWTON_MAXIT; ++it) {
        // E from the strict lower triangle of H (skew).  flags: [0] some element above the threshold, [1] a rotation too large,
        // [2] some |E_ij| > 1e-5 (R = I + E + E^2/2 is then not orthogonal to rounding: a Newton-Schulz step follows),
        // [3] some |E_ij| > 1e-8 (the E^2 term matters)
        int fl = 0;
        for (int i = wave; i < m; i += nw)
            for (int j = lane; j < m; j += 64) {
                double ev = 0.0;
                if (i != j) {
                    const int lo = i > j ? i : j, hi = i > j ? j : i;          // element (lo, hi) of the lower triangle
                    const double h = H[lo * ld + hi];
                    if (fabs(h) > thr) {
                        const double gap = H[hi * ld + hi] - H[lo * ld + lo];  // e_(lo,hi) = h / (h_hihi - h_lolo)
                        const double q = h / gap, aq = fabs(q);
                        fl |= 1;
                        if (!(aq < 0.3)) fl |= 2;                              // also catches NaN / inf
                        if (aq > 1e-5) fl |= 4;
                        if (aq > 1e-8) fl |= 8;
                        ev = i > j ? q : -q;
                    }
                }
                E[i * ld + j] = ev;
            }
        if (fl & 1) atomicOr(&flags[0], 1);
        if (fl & 2) atomicOr(&flags[1], 1);
        if (fl & 4) atomicOr(&flags[2], 1);
        if (fl & 8) atomicOr(&flags[3], 1);
        __syncthreads();
        const int f0 = flags[0], f1 = flags[1], f2 = flags[2], f3 = flags[3];
        __syncthreads();
        if (tid < 4) flags[tid] = 0;
        if (f1) return false;
        if (!f0) { iters = it; *Hout = H; *Wout = Wn; return true; }
        double* R = E;
        if (f3) { vg_nt_mm<1>(T, E, ld, 1, E, mp, ld, E); R = T; }              // R = I + E + E^2 / 2   (else R = I + E: below)
        else {
            for (int i = tid; i < mp; i += nthr) E[i * ld + i] = 1.0;
            __syncthreads();
        }
        double* WR = (R == T) ? E : T;                                         // the free buffer
        if (it == 0) WR = R;                                                   // W = I: W R is R itself
        else vg_nt_mm<0>(WR, Wn, ld, 1, R, mp, ld);                            // W R
        if (!f3) {
            // every |E_ij| <= 1e-8: what W <- W (I + E) leaves off the diagonal is O(|E|^2 ||H||) <= 1e-16 ||H||, far below any
            // threshold, and the diagonal moves at second order too -- no need to form W^T H0 W again or to look at it
            iters = it + 1; *Hout = H; *Wout = WR; return true;
        }
        double* Wnew = WR;
        if (f2) {
            double* S = (WR == E) ? T : E;
            vg_nt_mm<2>(S, WR, 1, ld, WR, mp, ld);                             // 1.5 I - 0.5 (W R)^T (W R)
            vg_nt_mm<0>(Wn, WR, ld, 1, S, mp, ld);                             // W <- (W R) (1.5 I - 0.5 Gram)
            Wnew = Wn;
        }
        // (W moved into another buffer when the Newton-Schulz step was skipped: rotate the roles instead of copying)
        double* spare = (Wnew == Wn) ? nullptr : Wn;
        double* P = (Wnew == E) ? T : ((Wnew == T) ? E : T);                   // a buffer that is neither W nor H0 / H
        vg_nt_mm<0>(P, H0, ld, 1, Wnew, mp, ld);                               // H0 W
        vg_nt_mm<0>(H, Wnew, 1, ld, P, mp, ld);                                // H <- W^T H0 W
        if (spare) {                                                           // Wn <-> the buffer that now holds W
            if (Wnew == E) E = spare; else T = spare;
            Wn = Wnew;
        }
    }
    return false;
}

#ifdef VG_EIG_RT
#define RT(i) do { if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); reinterpret_cast<unsigned long long*>(J.gwork)[(long)J.m * J.m + 8 + (i)] = t_;   /* past the polish matrix E */ } } while (0)
#else
#define RT(i)
#endif

template <bool INLDS>
__device__ __forceinline__ void vg_jacobi_body(const VgEigJob& J, double* W, double2* cs, VgPairRec* pq, VgActRec* actrec, unsigned char* isact,
                               int* nact_s, double* red, bool fast) {
    const int m = J.m;
    const int m2 = m + (m & 1), half = m2 >> 1;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tx = tid & 31, ty = tid >> 5, nty = nthr >> 5;

    RT(0);
    // load the lower triangle (zero padded) and the Frobenius norm
    // (a wave takes whole rows -- coalesced, no integer division -- and four rows' loads are in flight together)
    double ss = 0.0, dmax = 0.0, nmax = 0.0;       // (dmax, nmax: largest diagonal element / largest one from row null_from on)
    if (INLDS && J.Hl && J.Hr) {
        // the matrix is given as a product: G = Hl Hr^T (m x hk factors, row-major; the Ritz matrix H = (V1 G) V1^T of the
        // subspace start) -- one 16 x 16 block of the lower triangle per wave on the matrix cores, operands straight from global
        // memory, 8 k-steps of loads in flight; saves the GEMM launch that used to form it
        const int lane_ = tid & 63, wave_ = tid >> 6, nw_ = nthr >> 6, fi = lane_ & 15, fk = lane_ >> 4;
        const int nb = (m2 + 15) >> 4, hk = J.hk;
        // small problems: both factors are staged in LDS first (behind the packed copies, where the Newton start's buffers go
        // later) -- one coalesced batch of loads for the whole workgroup instead of four dependent batches of 32-byte segments
        // per wave (measured: 10 us -> 3 us for the 24 x 128 factors)
        const bool staged = m <= VG_EIG_NEWTON_MAX_M && hk <= 128;
        const int ldh = hk + 2, mp = nb * 16;
        double* Tl = W + 2 * ((m2 * (m2 + 1)) >> 1);
        double* Vl = Tl + mp * ldh;
        if (staged) {
            double tv[8], vv[8];                                   // mp * hk <= 48 * 128 = 6 * 1024 elements per factor
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + u * nthr;
                const bool okl = idx < m * hk;
                tv[u] = okl ? J.Hl[idx] : 0.0;
                vv[u] = okl ? J.Hr[idx] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + u * nthr;
                if (idx < mp * hk) {
                    const int i = idx / hk, k = idx - i * hk;
                    Tl[i * ldh + k] = tv[u];
                    Vl[i * ldh + k] = vv[u];
                }
            }
        }
        for (int i = tid; i < ((m2 * (m2 + 1)) >> 1); i += nthr) W[i] = 0.0;
        __syncthreads();
        for (int blk = wave_; blk < ((nb * (nb + 1)) >> 1); blk += nw_) {
            int bi = 0, rest = blk;
            while (rest > bi) { rest -= bi + 1; ++bi; }
            const int bj = rest;
            const int ra = bi * 16 + fi, rb = bj * 16 + fi;
            const double* pa = staged ? Tl + ra * ldh + fk : J.Hl + (long)(ra < m ? ra : 0) * hk + fk;
            const double* pb = staged ? Vl + rb * ldh + fk : J.Hr + (long)(rb < m ? rb : 0) * hk + fk;
            const double ma = (staged || ra < m) ? 1.0 : 0.0, mb = (staged || rb < m) ? 1.0 : 0.0;
            vg_nd4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = 0; k0 < hk; k0 += 32) {
                double av[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 4 * u;
                    av[u] = (k + fk < hk) ? pa[k] * ma : 0.0;
                    bv[u] = (k + fk < hk) ? pb[k] * mb : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bv[u + 1], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int row = bi * 16 + fk + 4 * r4, col = bj * 16 + fi;
                if (row < m && col <= row) {
                    const double v = acc0[r4] + acc1[r4];
                    W[vg_tri(row) + col] = v;
                    ss += (row == col) ? v * v : 2.0 * v * v;
                }
            }
        }
    } else if (INLDS && m2 <= 128 && nthr == 1024) {
        // m <= 128 with the full workgroup: 16 waves x 8 rows = the whole matrix in ONE batch of loads (G comes straight from
        // the previous kernel, cold in this XCD's L2: two dependent batches of four rows cost 4.4 us, one costs ~2.5)
        const int lane_ = tid & 63, wave_ = tid >> 6;
        double v[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = wave_ + 16 * u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = lane_ + 64 * h;
                v[u][h] = (i < m && j <= i) ? J.G[i * m + j] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = wave_ + 16 * u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = lane_ + 64 * h;
                if (i < m2 && j <= i) {
                    W[vg_tri(i) + j] = v[u][h];
                    ss += (i == j) ? v[u][h] * v[u][h] : 2.0 * v[u][h] * v[u][h];
                    if (i == j) { dmax = fmax(dmax, v[u][h]); if (i >= J.null_from) nmax = fmax(nmax, fabs(v[u][h])); }
                }
            }
        }
    } else {
        const int lane_ = tid & 63, wave_ = tid >> 6, nw_ = nthr >> 6;
        for (int i0 = wave_; i0 < m2; i0 += 4 * nw_) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nw_;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int j = lane_ + 64 * h;
                    v[u][h] = (i < m && j <= i && j < m) ? J.G[i * m + j] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nw_;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int j = lane_ + 64 * h;
                    if (i < m2 && j <= i) {
                        W[vg_tri(i) + j] = v[u][h];
                        ss += (i == j) ? v[u][h] * v[u][h] : 2.0 * v[u][h] * v[u][h];
                    }
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off);
        dmax = fmax(dmax, __shfl_xor(dmax, off));
        nmax = fmax(nmax, __shfl_xor(nmax, off));
    }
    if ((tid & 63) == 0) { red[tid >> 6] = ss; red[16 + (tid >> 6)] = dmax; red[32 + (tid >> 6)] = nmax; }
    if (tid == 0) { nact_s[0] = 0; nact_s[1] = 0; }
    if (!INLDS) __threadfence_block();
    __syncthreads();
    double fro = 0.0;
    for (int w = 0; w < (nthr >> 6); ++w) fro += red[w];
    // Subspace start: the rows from null_from on are the previous complement projected off the new range; they must be
    // numerically null for the new matrix too.  After a jump of the hyper-parameters (measured: > 5 % in a lengthscale) they are
    // not -- the range has moved out of the predicted subspace -- and since those rows were never re-orthonormalised the result
    // would be silently wrong.  The complement's Rayleigh quotients tell: 1e-16 lam_max on a valid start, 3e-10 at a 6 % jump.
    // The flag makes the host repeat the step cold (api.hip: VG_ESUBMISS).
    if (J.null_from > 0 && J.null_from < m && tid == 0) {
        double dm = 0.0, nm = 0.0;
        for (int w = 0; w < (nthr >> 6); ++w) { dm = fmax(dm, red[16 + w]); nm = fmax(nm, red[32 + w]); }
        if (!(nm <= VG_SUB_MISS * dm) && J.err) atomicOr(J.err, 2);
    }
    // sparse_first (subspace start): G arrives as E G E^T through two GEMMs, i.e. with rounding noise of ~sqrt(m) eps ||G||
    // in every element (measured at 1024^2 RBF, m = 128: 20 elements of the range block at 1-5x the default threshold, nothing
    // else above it; tools/dbg_gw.py) -- the default threshold sits below that floor and the solver then spends 15-20 rounds
    // (40 us) rotating noise.  VG_EIG_TOL_SUB = 64 eps m puts the threshold above it.
    const double thr = (J.tol > 0.0 ? J.tol : (J.sparse_first ? VG_EIG_TOL_SUB : VG_EIG_TOL)) * sqrt(fro) / (double)m;
    int nlog = 0, sweeps = 0, status = 0;
    bool converged = false, polished = false;
    RT(1);
    bool direct = false;
    double* NB = nullptr;
    const double *ntH = nullptr, *ntW = nullptr;
    if (INLDS && J.newton && !J.Qt0 && m <= VG_EIG_NEWTON_MAX_M) {
        NB = W + 2 * ((m2 * (m2 + 1)) >> 1);
        int iters = 0;
        __shared__ int ntflags[4];
        direct = vg_newton_diag(m, W, NB, thr, ntflags, iters, &ntH, &ntW);
        if (direct) {
            const int ldn = ((m + 15) & ~15) + 2;
            for (int i = tid; i < m; i += nthr) W[vg_tri(i) + i] = ntH[i * ldn + i];
            converged = true;
            sweeps = iters;
        }
        if (tid == 0) { nact_s[0] = 0; nact_s[1] = 0; }
        __syncthreads();
    }
    if (INLDS && fast && !J.sparse_first && !direct)   // dense sweeps with fixed addresses; the second copy of G follows the first in LDS
        W = vg_jacobi_fast(J, W, W + ((m2 * (m2 + 1)) >> 1), cs, reinterpret_cast<double*>(pq), nact_s, thr, nlog, sweeps,
                           status, converged, polished);

    RT(2);
    // round-independent block -> thread map: canonical blocks (al >= be) enumerated row by row, dealt round-robin
    int my_al[VG_MAXMINE], my_be[VG_MAXMINE];
    int nmine = 0;
    {
        const int nblk = (half * (half + 1)) >> 1;
        int al = 0, rowstart = 0;                       // rowstart = al*(al+1)/2
#pragma unroll
        for (int u = 0; u < VG_MAXMINE; ++u) {
            const int idx = tid + u * nthr;
            my_al[u] = my_be[u] = 0;
            if (INLDS && !converged && idx < nblk) {       // (not needed when the dense phase already finished the job)
                while (rowstart + al + 1 <= idx) { rowstart += al + 1; ++al; }
                my_al[u] = al;
                my_be[u] = idx - rowstart;
                nmine = u + 1;
            }
        }
    }

#ifdef VG_EIG_STAMP
    unsigned long long tP = 0, tB1 = 0, tU = 0, tB2 = 0, t0s, t1s;
#endif
    // ---- moving-index rounds: the matrix stays in place, pair k of round r is vg_pair(m2, r, k).  Cost follows the number
    // of pairs that actually rotate: the angle lanes also finish their own diagonal block and compact the rotating pairs
    // into a list; the update then visits only the blocks (a, b) of listed pairs a -- one wave per listed pair, one lane
    // per partner pair b -- so a round with a handful of rotations (the late sweeps of a warm start: hundreds of rounds
    // with 1-4 rotations each) costs a few hundred cycles, and a round with none costs one barrier.
    const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    int rr = 0;                                    // running round count: the two counters alternate across sweep
    // Each sweep starts with ONE scan of the strict lower triangle that flags the rounds holding a super-threshold element
    // (pair (p, q) belongs to round (p + q) / 2 mod n1, or p when q is the resting player); only flagged rounds are
    // visited.  What a rotation lifts above the threshold in an already passed round is caught by the next scan, and
    // an empty scan is the convergence test -- so the tail of a warm start (a nearly empty sweep plus the verification
    // sweep, ~230 rounds at one barrier each) costs two scans.
    __shared__ unsigned int rmask[8];              // m2 - 1 <= 255 rounds
    const int n1r = m2 - 1, inv2 = (n1r + 1) >> 1; // 2 * inv2 = n1r + 1 = 1 (mod n1r)
    for (int sweep = sweeps; sweep < VG_EIG_MAXSWEEP && !converged && !status; ++sweep) {      // boundaries too (m2 - 1 is odd)
        bool any = false;
        if (tid < 8) rmask[tid] = 0u;
        vg_round_barrier<INLDS>();
        for (int i = 1 + wave; i < m2; i += nwave) {
            const int ti = vg_tri(i);
            for (int j = lane; j < i; j += 64) {
                if (fabs(W[ti + j]) > thr) {
                    int rnd = (i == n1r) ? j : (int)(((long)(i + j) * inv2) % n1r);
                    atomicOr(&rmask[rnd >> 5], 1u << (rnd & 31));
                }
            }
        }
        vg_round_barrier<INLDS>();
        unsigned int mk[8];
        bool none = true;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) { mk[w8] = rmask[w8]; none = none && mk[w8] == 0u; }
        if (none) { ++sweeps; break; }             // uniform: every off-diagonal element is at or below the threshold
        for (int r = 0; r < m2 - 1; ++r) {
            if (!((mk[r >> 5] >> (r & 31)) & 1u)) continue;
            const int par = (rr++) & 1;
#ifdef VG_EIG_STAMP
            VG_STAMP(t0s);
#endif
            if (tid < half) {
                int p, q;
                vg_pair(m2, r, tid, p, q);
                const int tp = vg_tri(p), tq = vg_tri(q);
                const int apq = vg_symo(p, tp, q, tq);
                const double gpq = W[apq], gpp = W[tp + p], gqq = W[tq + q];       // one LDS round trip for all three
                const bool rot = fabs(gpq) > thr;
                double c = 1.0, s = 0.0;
                // compaction of the rotating pairs: ballot + prefix count when the angle lanes are one wave (no LDS atomic
                // on the critical path), LDS counter otherwise.  List order is irrelevant: the listed blocks are disjoint.
                int slot = 0;
                if (half <= 64) {
                    const unsigned long long bal = __ballot(rot);
                    slot = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                    if (tid == 0) nact_s[par] = __popcll(bal);
                } else if (rot) {
                    slot = atomicAdd(&nact_s[par], 1);
                }
                if (rot) {
                    const VgAngle a = vg_angle3(gpp, gqq, gpq, thr);
                    c = a.c; s = a.s;
                    W[tp + p] = gpp - a.t * gpq;               // the pair's own diagonal block, in closed form
                    W[tq + q] = gqq + a.t * gpq;
                    W[apq] = 0.0;
                    actrec[slot] = VgActRec{p, q, tp, tq, c, s, tid, 0};
                }
                isact[tid] = rot;
                cs[tid] = make_double2(c, s);
                pq[tid] = VgPairRec{p, q, tp, tq};
            }
            if (tid == nthr - 1 && half > 64) nact_s[par ^ 1] = 0;
#ifdef VG_EIG_STAMP
            VG_STAMP(t1s); tP += t1s - t0s;
#endif
            vg_round_barrier<INLDS>();
#ifdef VG_EIG_STAMP
            VG_STAMP(t0s); tB1 += t0s - t1s;
#endif
            const int na = nact_s[par];
            if (na == 0) continue;                 // uniform: nothing to rotate in this round
            if (nlog >= J.max_rounds) { status = VGGP_ENOCONV; break; }
            // hand-off to the replay workgroups (cdna guide G16, sc1 form): log entries are stored write-through;
            // at the end of every logged round the storing waves wait until all but that round's own stores are
            // done (counted vmcnt, so the wait never exposes store latency).  Hence, after the previous round's
            // barrier, every round logged before the previous one is complete and one lane may publish that count.
            if (tid == 0 && nlog >= VG_EIG_LAG)
                __hip_atomic_store(&J.counters[3], nlog - VG_EIG_LAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < half) {
                double* lp = reinterpret_cast<double*>(&J.rotlog[(long)nlog * half + tid]);
                __hip_atomic_store(lp, cs[tid].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(lp + 1, cs[tid].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) __hip_atomic_store(&J.roundlog[nlog], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++nlog;
            any = true;
            for (int sidx = wave; sidx < na; sidx += nwave) {
                const VgActRec R = actrec[sidx];                          // wave-uniform: pair, offsets, rotation, pair index
                const int a = R.k;
                const VgPairRec A{R.p, R.q, R.tp, R.tq};
                for (int b = lane; b < half; b += 64) {
                    const bool both = isact[b];                           // these three loads do not depend on R:
                    const double2 rbv = cs[b];                            // everything the block needs arrives in one
                    const VgPairRec B = pq[b];                            // LDS round trip
                    if (b == a) continue;                                 // diagonal block: done by the angle lane
                    if (both && a < b) continue;                          // the larger listed index does the shared block
                    // vg_block wants (row pair, column pair) in canonical order al > be; rbv is the identity if b rests
                    if (a > b) vg_block(W, A, R.c, R.s, B, rbv.x, rbv.y);
                    else vg_block(W, B, rbv.x, rbv.y, A, R.c, R.s);
                }
            }
#ifdef VG_EIG_STAMP
            VG_STAMP(t1s); tU += t1s - t0s;
#endif
            // <= 4 VMEM ops per wave per round; write-through (sc1) stores take a few microseconds to be acknowledged,
            // so VG_EIG_LAG rounds are left in flight: everything older is done when this returns
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            vg_round_barrier<INLDS>();
#ifdef VG_EIG_STAMP
            VG_STAMP(t0s); tB2 += t0s - t1s;
#endif
        }
        ++sweeps;
        if (status || !any) break;
        if (sweep == VG_EIG_MAXSWEEP - 1) status = VGGP_ENOCONV;
    }
    RT(3);
    // every storing wave drains its log stores BEFORE the barrier that precedes the DONE publication
    // (__syncthreads() alone does not wait for vmcnt on gfx950)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Eigenpairs leave SORTED by decreasing eigenvalue (rank by counting; ties by index).  The next step warm-starts from
    // this basis, and cyclic Jacobi on the strongly graded Gram matrices converges in fewer sweeps when the diagonal is
    // ordered (RBF, m = 128, 1% hyper-parameter change: 5 rotating sweeps instead of 6-7; unordered bases are what a
    // Jacobi solver leaves behind).  The replay workgroups permute their rows of Q^T with the same ranks.
    double* dg = reinterpret_cast<double*>(cs);         // the diagonal, staged contiguously (the rotation array is free now)
    for (int i = tid; i < m; i += nthr) dg[i] = W[vg_tri(i) + i];
    __syncthreads();
    // rank of eigenvalue i = number of eigenvalues ahead of it: 8 lanes per eigenvalue share the count (m <= 128 with 1024
    // threads; a lone lane per eigenvalue walks all m entries: 3 us at m = 128), then the lane group's leader writes
    const int rgrp = (m * 8 <= nthr) ? 8 : 1;
    for (int i = tid / rgrp; i < m; i += nthr / rgrp) {
        const int part = tid % rgrp;
        const double li = dg[i];
        int rank = i;
        if (J.perm) {
            rank = 0;
#pragma unroll 8
            for (int j = part; j < m; j += rgrp) {
                const double lj = dg[j];
                rank += (lj > li || (lj == li && j < i)) ? 1 : 0;
            }
            if (rgrp == 8) { rank += __shfl_xor(rank, 1); rank += __shfl_xor(rank, 2); rank += __shfl_xor(rank, 4); }
            if (part == 0) __hip_atomic_store(&J.perm[i], rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (part != 0) continue;
        J.lam[rank] = li;
        if (direct) reinterpret_cast<int*>(pq)[i] = rank;     // (the pair records are free now)
    }
    if (direct) {                                             // eigenvector i = column i of the Newton iterate
        __syncthreads();
        const int* rk = reinterpret_cast<const int*>(pq);
        const int ldn = ((m + 15) & ~15) + 2;
        const double* Wn = ntW;
        for (int idx = tid; idx < m * m; idx += nthr) {
            const int i = idx / m, j = idx - i * m;
            const double v = Wn[j * ldn + i];
            J.Qt[rk[i] * m + j] = v;
            if (J.Qt2) J.Qt2[rk[i] * m + j] = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ranks are at the L2 before DONE is published
    __syncthreads();
    int nrank = 0;
    if (tid < 64) {                                          // wave 0: largest eigenvalue and numerical rank by wave reductions
        double lmax = 0.0;
        for (int j = tid; j < m; j += 64) lmax = fmax(lmax, dg[j]);
        for (int off = 32; off > 0; off >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, off));
        for (int j = tid; j < m; j += 64) nrank += dg[j] > VG_EIG_RANK_CUT * lmax ? 1 : 0;
        for (int off = 32; off > 0; off >>= 1) nrank += __shfl_xor(nrank, off);
    }
    if (tid == 0) {
        J.counters[0] = nlog;
        J.counters[1] = sweeps | (nrank << 8);               // numerical rank rides above the sweep count
        J.counters[2] = status;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&J.counters[3], nlog | VG_EIG_DONE | (polished ? VG_EIG_POLISH : 0) | (direct ? VG_EIG_DIRECT : 0), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    RT(4);
#ifdef VG_EIG_STAMP
    if ((tid & 63) == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(J.gwork) + (tid >> 6) * 4;
        dbg[0] = tP; dbg[1] = tB1; dbg[2] = tU; dbg[3] = tB2;
    }
#endif
}

// ---- replay role: a block of VG_RP_COLS columns of Q^T per workgroup, fed by the producer's log ----------
#define VG_RP_CHUNK_BYTES 16384

__device__ __forceinline__ void vg_replay_body(const VgEigJob& J, int cblock, int VG_RP_COLS, double* dyn, int* s_sync) {
    const int m = J.m, m2 = m + (m & 1), half = m2 >> 1;
    const int VG_RP_LD = VG_RP_COLS + 1, csh = VG_RP_COLS == 64 ? 6 : (VG_RP_COLS == 32 ? 5 : 4);
    const int j0 = cblock * VG_RP_COLS;
    if (j0 >= m2) return;
    double* T = dyn;                                          // [m2][65]
    double2* chunk = reinterpret_cast<double2*>(T + (long)m2 * VG_RP_LD);
    int* rchunk = reinterpret_cast<int*>(chunk + VG_RP_CHUNK_BYTES / sizeof(double2));
    int rounds_per_chunk = VG_RP_CHUNK_BYTES / (half * (int)sizeof(double2));
    if (rounds_per_chunk < 1) rounds_per_chunk = 1;
    if (rounds_per_chunk > 1024) rounds_per_chunk = 1024;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;

    for (int idx = tid; idx < m2 * VG_RP_COLS; idx += nthr) {
        const int i = idx >> csh, jj = idx & (VG_RP_COLS - 1), j = j0 + jj;
        double v = (i == j) ? 1.0 : 0.0;
        if (J.Qt0 && i < m && j < m) v = J.Qt0[i * m + j];
        T[i * VG_RP_LD + jj] = v;
    }
    if (J.cp_src && J.cp_dst) {
        for (int idx = tid; idx < m * VG_RP_COLS; idx += nthr) {
            const int i = idx >> csh, j = j0 + (idx & (VG_RP_COLS - 1));
            if (j < m) J.cp_dst[i * m + j] = J.cp_src[i * m + j];
        }
    }
    const int jj = wave * 4 + (lane & 3);                   // this lane's column inside the block
    const bool colwave = wave * 4 < VG_RP_COLS;              // waves beyond the tile only help with the loads
    int consumed = 0;
    bool polish = false;
    for (;;) {
        // one lane polls the producer's progress word (relaxed, agent scope), then the workgroup rendezvous
        if (tid == 0) {
            int pw, spins = 0;
            for (;;) {
                pw = __hip_atomic_load(&J.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int avail = (pw & (VG_EIG_DONE - 1)) - consumed;
                if ((pw & VG_EIG_DONE) || avail >= rounds_per_chunk || avail >= 8) break;      // small batches: short tail after DONE
                if (++spins > (1 << 24)) { pw = VG_EIG_DONE | 0x20000000; if (J.err) atomicOr(J.err, 1); break; }   // bounded spin: never hang the GPU
                __builtin_amdgcn_s_sleep(8);
            }
            s_sync[0] = pw;
        }
        __syncthreads();
        const int pw = s_sync[0];
        const bool done = (pw & VG_EIG_DONE) != 0;
        if (pw & 0x20000000) break;                          // producer never showed up (timeout)
        polish = done && (pw & VG_EIG_POLISH);
        if (done && (pw & VG_EIG_DIRECT)) return;             // uniform: the producer wrote Q^T itself
        const int published = pw & 0x07ffffff;
        const int nr = min(rounds_per_chunk, published - consumed);
        if (nr > 0) {
            for (int idx = tid; idx < nr * half; idx += nthr) {
                const double* lp = reinterpret_cast<const double*>(&J.rotlog[(long)consumed * half + idx]);
                const double cx = __hip_atomic_load(lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double cy = __hip_atomic_load(lp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                chunk[idx] = make_double2(cx, cy);
            }
            for (int idx = tid; idx < nr; idx += nthr)
                rchunk[idx] = __hip_atomic_load(&J.roundlog[consumed + idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            for (int kk = 0; colwave && kk < nr; ++kk) {
                const int r = rchunk[kk];
                // the pairs of one round touch disjoint rows: issue all loads of a batch of 4 before the math
                for (int al0 = lane >> 2; al0 < half; al0 += 64) {
                    int ap[4], aq[4];
                    double2 c[4];
                    double tp[4], tq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int al = al0 + 16 * u;
                        c[u] = make_double2(1.0, 0.0);
                        ap[u] = aq[u] = jj;
                        if (al < half) {
                            c[u] = chunk[kk * half + al];
                            int p, q;
                            vg_pair(m2, r, al, p, q);
                            ap[u] = p * VG_RP_LD + jj;
                            aq[u] = q * VG_RP_LD + jj;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { tp[u] = T[ap[u]]; tq[u] = T[aq[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (c[u].y != 0.0) {                 // identity rotations (pair below threshold) are skipped
                            T[ap[u]] = c[u].x * tp[u] - c[u].y * tq[u];
                            T[aq[u]] = c[u].y * tp[u] + c[u].x * tq[u];
                        }
                    }
                }
            }
            consumed += nr;
        }
        __syncthreads();                                     // s_sync / chunk reuse
        if (done && consumed >= published) break;
    }
    __syncthreads();
    if (polish && VG_RP_COLS == 16) {
        // T <- T + E (T + E T / 2) on the matrix cores: wave w owns rows 16w .. 16w+15 of the slice; its 16 x 128 block of E
        // (the producer left E in gwork) is loaded ONCE, straight into MFMA A-operand registers, and serves both passes;
        // the B operands are single LDS reads of T (pass 1) and T1 (pass 2).  v_mfma_f64_16x16x4: lane (fi, fk) supplies
        // A[fi][4 kk + fk], B[4 kk + fk][fi] and receives D[fk + 4 r][fi].
        double* T1 = reinterpret_cast<double*>(rchunk + 1024);
        const int fi = lane & 15, fk = lane >> 4, row0 = wave * 16;
        const bool act = row0 < m;
        double ea[32];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int k = kk * 4 + fk;
            const int i = row0 + fi;
            ea[kk] = 0.0;                                   // E is skew: the producer stored the strict lower triangle
            if (act && i < m && k < m && k != i) {
                const double v = __hip_atomic_load(&J.gwork[k < i ? (long)i * m + k : (long)k * m + i], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                ea[kk] = k < i ? v : -v;
            }
        }
        vg_bd4 keep = {0.0, 0.0, 0.0, 0.0};
        for (int pass = 0; pass < 2; ++pass) {
            const double* Tin = pass ? T1 : T;
            if (act) {
                vg_bd4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    const int k = kk * 4 + fk;
                    if (kk * 4 < m2) {                                   // wave-uniform
                        const double bv = k < m2 ? Tin[k * VG_RP_LD + fi] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ea[kk], bv, acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = row0 + fk + 4 * r;
                    if (i < m2) {
                        const double t0 = T[i * VG_RP_LD + fi];
                        if (pass == 0) T1[i * VG_RP_LD + fi] = t0 + 0.5 * acc[r];
                        else keep[r] = t0 + acc[r];
                    }
                }
            }
            __syncthreads();
        }
        if (act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = row0 + fk + 4 * r;
                if (i < m2) T[i * VG_RP_LD + fi] = keep[r];
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < m * VG_RP_COLS; idx += nthr) {
        const int i = idx >> csh, j = j0 + (idx & (VG_RP_COLS - 1));
        if (j < m) {
            const double v = T[i * VG_RP_LD + (idx & (VG_RP_COLS - 1))];
            int di = i;
            if (J.perm) {
                di = __hip_atomic_load(&J.perm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                di = di < 0 ? 0 : (di >= m ? m - 1 : di);           // never out of range, even after a producer timeout
            }
            J.Qt[di * m + j] = v;
            if (J.Qt2) J.Qt2[di * m + j] = v;
        }
    }
#ifdef VG_EIG_RT
    __syncthreads();
    if (cblock == 0) RT(5);
#endif
}

// =====================================================================================================================
// Block Jacobi (m <= 128): the same m-1 rotation rounds per sweep, but organised so that every round works on 32 x 32
// data instead of the whole matrix, and the long-range part of the update is dense f64 MFMA work.
//   * indices are grouped in nb = 2 ceil(m/32) blocks of 16; an OUTER round pairs the blocks round-robin (nb/2 pairs);
//   * each pair (I, J) is a 32 x 32 sub-problem S = G[IJ, IJ], gathered to LDS and rotated by a 4-wave group:
//     outer round 0 does a full 31-round sweep of S (this is where within-block pairs are annihilated, every block is
//     in exactly one pair), outer rounds >= 1 only the 16 rounds of cross pairs (i in I, j in J).  Rotations are
//     accumulated into U (32 x 32, U <- U J);
//   * afterwards the off-diagonal super-blocks are updated with MFMA, G_ab <- U_a^T G_ab U_b (the first product's
//     accumulator tile is, register for register, the B operand of the second), the diagonal ones are S itself;
//   * the four U of an outer round are logged (write-through) for the replay workgroups, which apply
//     Qt[IJ, cols] <- U^T Qt[IJ, cols] with MFMA on their 16-column tile.
// Per outer sweep: 31 + 16 (nb - 2) = m-ish inner rounds, each ~4x cheaper than a full-matrix round, plus nb - 1 applies.
#define VG_BJ_SLOT (2 * 32 * 32)                // doubles per pair slot: S then U
// 32 x 32 tiles are stored with row stride 32 and the column XOR-swizzled by 16 on odd rows: a half-wave that touches
// rows {r, r+1} x 16 columns (every S / U access pattern here) then hits 32 distinct 8-byte banks (stride 33 gave
// bank = 2 (r + c) mod 64, i.e. up to 4-way conflicts).
__device__ __forceinline__ int vg_sw(int r, int c) { return (r << 5) + (c ^ ((r & 1) << 4)); }
#define VG_BJ_MAXSWEEP 30

__device__ __forceinline__ int vg_bgidx(int I, int Jb, int l) { return ((l < 16) ? I : Jb) * 16 + (l & 15); }

// local pairings of the 32 indices of a sub-problem.  full == true: circle method, round t in [0, 31);
// full == false: cross pairs only (i in the first block, j in the second), round t in [0, 16)
__device__ __forceinline__ void vg_bpair(bool full, int t, int k, int& p, int& q) {
    if (full) vg_pair(32, t, k, p, q);
    else { p = k; q = 16 + ((k + t) & 15); }
}
// inverse: which pair of round t holds local index x, and is x its p (pos 0) or its q (pos 1)
__device__ __forceinline__ void vg_bpair_inv(bool full, int t, int x, int& k, int& pos) {
    if (full) {
        if (x == 31) { k = 0; pos = 1; return; }
        int d = x - t; if (d < 0) d += 31;
        if (d == 0) { k = 0; pos = 0; }
        else if (d <= 15) { k = d; pos = 0; }
        else { k = 31 - d; pos = 1; }
    } else {
        if (x < 16) { k = x; pos = 0; }
        else { k = ((x - 16) - t) & 15; pos = 1; }
    }
}
// rotation (c, s) annihilating spq, or identity below the threshold
__device__ __forceinline__ double2 vg_angle(double spp, double sqq, double spq, double thr, bool& rot) {
    rot = fabs(spq) > thr;
    if (!rot) return make_double2(1.0, 0.0);
    const double d = sqq - spp, o = 2.0 * spq;
    const double h2 = d * d + o * o;
    double tt = fabs(o) * vg_rcp(fabs(d) + h2 * vg_rsq(h2));
    if ((d >= 0.0) != (o >= 0.0)) tt = -tt;
    const double c = vg_rsq(1.0 + tt * tt);
    return make_double2(c, tt * c);
}

__device__ __forceinline__ void vg_bjacobi_body(const VgEigJob& J, double* dyn, double2* cs, VgPairRec* pq, int* flags, double* red) {
    const int m = J.m;
    const int nb = 2 * ((m + 31) / 32), Mp = 16 * nb, npair = nb >> 1;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int grp = tid >> 8, lt = tid & 255;
    double* Gp = dyn;                                            // packed lower triangle of the padded matrix
    double* slots = dyn + ((Mp * (Mp + 1) / 2 + 1) & ~1);
    double* S = slots + grp * VG_BJ_SLOT;
    double* U = S + 32 * 32;

    double ss = 0.0;
    for (int idx = tid; idx < Mp * Mp; idx += nthr) {
        const int i = idx / Mp, j = idx - i * Mp;
        if (j > i) continue;
        const double v = (i < m && j < m) ? J.G[i * m + j] : 0.0;
        Gp[vg_tri(i) + j] = v;
        ss += (i == j) ? v * v : 2.0 * v * v;
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    double fro = 0.0;
    for (int w = 0; w < (nthr >> 6); ++w) fro += red[w];
    const double thr = (J.tol > 0.0 ? J.tol : VG_EIG_TOL) * sqrt(fro) / (double)m;

    // capacity of the log buffer in outer rounds (it was sized for the scalar kernel's (c, s) log)
    const int cap_rounds = (int)(J.log_bytes / ((long)npair * 1024 * sizeof(double)));
    double* ulog = reinterpret_cast<double*>(J.rotlog);

#ifdef VG_EIG_STAMP
    unsigned long long tP = 0, tB1 = 0, tU = 0, tB2 = 0, tG = 0, tA = 0, t0s, t1s, nin_tot = 0, nout = 0;
#define BST(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define BST(var)
#endif
    int nlog = 0, sweeps = 0, status = 0;
    for (int sweep = 0; sweep < VG_BJ_MAXSWEEP; ++sweep) {
        bool any_sweep = false;
        for (int R = 0; R < nb - 1; ++R) {
            int I = 0, Jb = 1;
#ifdef VG_EIG_STAMP
            BST(t0s);
#endif
            if (grp < npair) vg_pair(nb, R, grp, I, Jb);
            // ---- gather S, U = I -----------------------------------------------------------------------------
            if (grp < npair)
                for (int e = lt; e < 1024; e += 256) {
                    const int r = e >> 5, c = e & 31;
                    const int gi = vg_bgidx(I, Jb, r), gj = vg_bgidx(I, Jb, c);
                    S[vg_sw(r, c)] = Gp[vg_sym(gi, gj)];
                    U[vg_sw(r, c)] = (r == c) ? 1.0 : 0.0;
                }
            if (tid == 0) { flags[0] = 0; flags[1] = 0; }
            vg_round_barrier<true>();
#ifdef VG_EIG_STAMP
            BST(t1s); tG += t1s - t0s; ++nout;
#endif
            // ---- inner rounds ----------------------------------------------------------------------------------
            // Software pipeline: the angles of round t+1 are computed by 16 lanes per group WHILE round t is applied.
            // Those lanes read the (still old) S together with everybody else before barrier 1, rotate the three
            // elements they need in registers, and publish cs[t+1]; S itself is only written between barrier 1 and 2.
            const bool full = (R == 0);
            const int nin = full ? 31 : 16;
            // angle lanes: 16 lanes of wave `grp` of group `grp` (waves 0, 5, 10, 15 sit on four different SIMDs)
            const bool pl = grp < npair && (lt >> 6) == grp && (lt & 63) < 16;
            const int kl = lt & 63;
            if (pl) {                                                // prologue: angles of round 0 straight from S
                int p, q;
                vg_bpair(full, 0, kl, p, q);
                bool rot;
                cs[grp * 16 + kl] = vg_angle(S[vg_sw(p, p)], S[vg_sw(q, q)], S[vg_sw(p, q)], thr, rot);
                if (rot) flags[0] = 1;
            }
            vg_round_barrier<true>();
            bool any_outer = false;
            if (!full) {
                // ---- cross rounds (R >= 1): p_k = k, q_k = 16 + ((k + t) & 15): most addresses are loop invariant ----
                const int a = lt >> 4, b = lt & 15;
                const int swa = (a & 1) << 4;
                const int rowa = a << 5;
                const int a00 = rowa + (b ^ swa);                        // S[pa][pb] and U[a][pb]
                // angle lane k: pair A = k (holds p' = k as its p), pair B = (k+1)&15 (holds q' as its q)
                const int kB = (kl + 1) & 15;
                const int swk = (kl & 1) << 4, swB = (kB & 1) << 4;
                for (int t = 0; t < 16; ++t) {
                    const int cur = t & 1, nxt = cur ^ 1;
                    const bool active = flags[cur] != 0;                 // uniform
                    const double2* csc = cs + cur * 64 + grp * 16;
                    if (tid == nthr - 1) flags[nxt] = 0;
                    int a01 = 0, a10 = 0, a11 = 0;
                    double g00 = 0, g01 = 0, g10 = 0, g11 = 0, v0p = 0, v0q = 0, v1p = 0, v1q = 0;
                    double2 ca = make_double2(1.0, 0.0), cb = ca;
                    if (active && grp < npair) {
                        const int qa = 16 + ((a + t) & 15), qb = 16 + ((b + t) & 15);
                        const int swq = (qa & 1) << 4;
                        a01 = rowa + (qb ^ swa);
                        a10 = (qa << 5) + (b ^ swq);
                        a11 = (qa << 5) + (qb ^ swq);
                        ca = csc[a];
                        cb = csc[b];
                        g00 = S[a00]; g01 = S[a01]; g10 = S[a10]; g11 = S[a11];
                        v0p = U[a00]; v0q = U[a01]; v1p = U[a00 + 512]; v1q = U[a01 + 512];
                    }
                    double b00 = 0, b01 = 0, b10 = 0, b11 = 0, ea = 0, eb = 0, ec = 0, fa = 0, fb = 0, fc = 0;
                    double2 cA = make_double2(1.0, 0.0), cB = cA;
                    const bool pnext = pl && (t + 1 < 16);
                    if (pnext) {
                        const int qA = 16 + ((kl + t) & 15), qB = 16 + ((kl + t + 1) & 15);       // qB = q' of the next pair
                        const int swqA = (qA & 1) << 4, swqB = (qB & 1) << 4;
                        if (active) { cA = csc[kl]; cB = csc[kB]; }
                        b00 = S[(kl << 5) + (kB ^ swk)]; b01 = S[(kl << 5) + (qB ^ swk)];
                        b10 = S[(qA << 5) + (kB ^ swqA)]; b11 = S[(qA << 5) + (qB ^ swqA)];
                        ea = S[(kl << 5) + (kl ^ swk)]; eb = S[(kl << 5) + (qA ^ swk)]; ec = S[(qA << 5) + (qA ^ swqA)];
                        fa = S[(kB << 5) + (kB ^ swB)]; fb = S[(kB << 5) + (qB ^ swB)]; fc = S[(qB << 5) + (qB ^ swqB)];
                    }
                    if (active) vg_round_barrier<true>();                // all loads done: S / U may now be overwritten
                    if (active && grp < npair) {
                        const double h00 = cb.x * g00 - cb.y * g01, h01 = cb.y * g00 + cb.x * g01;
                        const double h10 = cb.x * g10 - cb.y * g11, h11 = cb.y * g10 + cb.x * g11;
                        S[a00] = ca.x * h00 - ca.y * h10;
                        S[a10] = ca.y * h00 + ca.x * h10;
                        S[a01] = ca.x * h01 - ca.y * h11;
                        S[a11] = ca.y * h01 + ca.x * h11;
                        U[a00] = cb.x * v0p - cb.y * v0q;
                        U[a01] = cb.y * v0p + cb.x * v0q;
                        U[a00 + 512] = cb.x * v1p - cb.y * v1q;
                        U[a01 + 512] = cb.y * v1p + cb.x * v1q;
                    }
                    if (pnext) {
                        // p' is the p of pair A (column 0 of J_A = (c, -s)); q' is the q of pair B (column 1 = (s, c))
                        const double uA0 = cA.x, uA1 = -cA.y, uB0 = cB.y, uB1 = cB.x;
                        const double n_pq = uA0 * (b00 * uB0 + b01 * uB1) + uA1 * (b10 * uB0 + b11 * uB1);
                        const double n_pp = uA0 * (ea * uA0 + eb * uA1) + uA1 * (eb * uA0 + ec * uA1);
                        const double n_qq = uB0 * (fa * uB0 + fb * uB1) + uB1 * (fb * uB0 + fc * uB1);
                        bool rot;
                        cs[nxt * 64 + grp * 16 + kl] = vg_angle(n_pp, n_qq, n_pq, thr, rot);
                        if (rot) flags[nxt] = 1;
                    }
                    vg_round_barrier<true>();
                    any_outer |= active;
                }
            } else
            for (int t = 0; t < nin; ++t) {
                const int cur = t & 1, nxt = cur ^ 1;
#ifdef VG_EIG_STAMP
                BST(t0s); ++nin_tot;
#endif
                const bool active = flags[cur] != 0;                 // uniform
                const double2* csc = cs + cur * 64 + grp * 16;
                if (tid == nthr - 1) flags[nxt] = 0;
                // ---- phase L: every load of the round ----------------------------------------------------------
                int a00 = 0, a01 = 0, a10 = 0, a11 = 0, u0p = 0, u0q = 0, u1p = 0, u1q = 0;
                double g00 = 0, g01 = 0, g10 = 0, g11 = 0, v0p = 0, v0q = 0, v1p = 0, v1q = 0;
                double2 ca = make_double2(1.0, 0.0), cb = ca;
                if (active && grp < npair) {
                    const int a = lt >> 4, b = lt & 15;
                    int pa, qa, pb, qb;
                    vg_bpair(full, t, a, pa, qa);
                    vg_bpair(full, t, b, pb, qb);
                    ca = csc[a];
                    cb = csc[b];
                    a00 = vg_sw(pa, pb); a01 = vg_sw(pa, qb); a10 = vg_sw(qa, pb); a11 = vg_sw(qa, qb);
                    u0p = vg_sw(a, pb); u0q = vg_sw(a, qb);
                    u1p = vg_sw(a + 16, pb); u1q = vg_sw(a + 16, qb);
                    g00 = S[a00]; g01 = S[a01]; g10 = S[a10]; g11 = S[a11];
                    v0p = U[u0p]; v0q = U[u0q]; v1p = U[u1p]; v1q = U[u1q];
                }
                double n_pp = 0, n_qq = 0, n_pq = 0;                 // next pair's three elements (angle lanes)
                double b00 = 0, b01 = 0, b10 = 0, b11 = 0, ea = 0, eb = 0, ec = 0, fa = 0, fb = 0, fc = 0;
                double2 cA = make_double2(1.0, 0.0), cB = cA;
                int ip = 0, iq = 0;
                const bool pnext = pl && (t + 1 < nin);
                if (pnext) {
                    int pn, qn, kA, kB, pA, qA, pB, qB;
                    vg_bpair(full, t + 1, kl, pn, qn);
                    vg_bpair_inv(full, t, pn, kA, ip);
                    vg_bpair_inv(full, t, qn, kB, iq);
                    vg_bpair(full, t, kA, pA, qA);
                    vg_bpair(full, t, kB, pB, qB);
                    if (active) { cA = csc[kA]; cB = csc[kB]; }
                    b00 = S[vg_sw(pA, pB)]; b01 = S[vg_sw(pA, qB)];
                    b10 = S[vg_sw(qA, pB)]; b11 = S[vg_sw(qA, qB)];
                    ea = S[vg_sw(pA, pA)]; eb = S[vg_sw(pA, qA)]; ec = S[vg_sw(qA, qA)];
                    fa = S[vg_sw(pB, pB)]; fb = S[vg_sw(pB, qB)]; fc = S[vg_sw(qB, qB)];
                }
#ifdef VG_EIG_STAMP
                BST(t1s); tP += t1s - t0s;
#endif
                if (active) vg_round_barrier<true>();                // all loads done: S / U may now be overwritten
#ifdef VG_EIG_STAMP
                BST(t0s); tB1 += t0s - t1s;
#endif
                // ---- phase C: stores of round t, angles of round t+1 ------------------------------------------------
                if (active && grp < npair) {
                    const double h00 = cb.x * g00 - cb.y * g01, h01 = cb.y * g00 + cb.x * g01;
                    const double h10 = cb.x * g10 - cb.y * g11, h11 = cb.y * g10 + cb.x * g11;
                    S[a00] = ca.x * h00 - ca.y * h10;
                    S[a10] = ca.y * h00 + ca.x * h10;
                    S[a01] = ca.x * h01 - ca.y * h11;
                    S[a11] = ca.y * h01 + ca.x * h11;
                    U[u0p] = cb.x * v0p - cb.y * v0q;
                    U[u0q] = cb.y * v0p + cb.x * v0q;
                    U[u1p] = cb.x * v1p - cb.y * v1q;
                    U[u1q] = cb.y * v1p + cb.x * v1q;
                }
                if (pnext) {
                    // column i of J = (c, -s) for i = 0, (s, c) for i = 1;  new[i][j] = u_i^T B v_j
                    const double uA0 = ip ? cA.y : cA.x, uA1 = ip ? cA.x : -cA.y;
                    const double uB0 = iq ? cB.y : cB.x, uB1 = iq ? cB.x : -cB.y;
                    n_pq = uA0 * (b00 * uB0 + b01 * uB1) + uA1 * (b10 * uB0 + b11 * uB1);
                    n_pp = uA0 * (ea * uA0 + eb * uA1) + uA1 * (eb * uA0 + ec * uA1);
                    n_qq = uB0 * (fa * uB0 + fb * uB1) + uB1 * (fb * uB0 + fc * uB1);
                    bool rot;
                    cs[nxt * 64 + grp * 16 + kl] = vg_angle(n_pp, n_qq, n_pq, thr, rot);
                    if (rot) flags[nxt] = 1;
                }
#ifdef VG_EIG_STAMP
                BST(t1s); tU += t1s - t0s;
#endif
                vg_round_barrier<true>();
#ifdef VG_EIG_STAMP
                BST(t0s); tB2 += t0s - t1s;
#endif
                any_outer |= active;
            }
#ifdef VG_EIG_STAMP
            BST(t0s);
#endif
            if (!any_outer) continue;                           // uniform: this block pairing was already diagonal
            any_sweep = true;
            if (nlog >= cap_rounds) { status = VGGP_ENOCONV; break; }
            // ---- hand-off: previous outer rounds' log stores are long done; drain, rendezvous, publish, then log ----
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            vg_round_barrier<true>();
            if (tid == 0) {
                __hip_atomic_store(&J.counters[3], nlog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&J.roundlog[nlog], R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (grp < npair) {
                double* dst = ulog + ((long)nlog * npair + grp) * 1024;
                for (int e = lt; e < 1024; e += 256)
                    __hip_atomic_store(dst + e, U[vg_sw(e >> 5, e & 31)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ++nlog;
            // ---- off-diagonal super-blocks: X = U_a^T (G_ab U_b), one 32 x 16 strip per wave ---------------------
            const int nunits = npair * (npair - 1);
            vg_bd4 X0 = {0.0, 0.0, 0.0, 0.0}, X1 = {0.0, 0.0, 0.0, 0.0};
            int Ia = 0, Ja = 0, Ib = 0, Jbb = 0, jb = 0;
            if (wave < nunits) {
                const int sbi = wave >> 1;
                jb = wave & 1;
                int al = 1;
                while ((al * (al + 1)) / 2 <= sbi) ++al;
                const int be = sbi - (al * (al - 1)) / 2;
                vg_pair(nb, R, al, Ia, Ja);
                vg_pair(nb, R, be, Ib, Jbb);
                const double* Ua = slots + al * VG_BJ_SLOT + 32 * 32;
                const double* Ub = slots + be * VG_BJ_SLOT + 32 * 32;
                const int fi = lane & 15, fk = lane >> 4;
                const int gi0 = vg_bgidx(Ia, Ja, fi), gi1 = vg_bgidx(Ia, Ja, 16 + fi);
                vg_bd4 T0 = {0.0, 0.0, 0.0, 0.0}, T1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int k = kk * 4 + fk;
                    const int gk = vg_bgidx(Ib, Jbb, k);
                    const double bval = Ub[vg_sw(k, jb * 16 + fi)];
                    T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Gp[vg_sym(gi0, gk)], bval, T0, 0, 0, 0);
                    T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Gp[vg_sym(gi1, gk)], bval, T1, 0, 0, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k0 = kk * 4 + fk, k1 = 16 + kk * 4 + fk;
                    X0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ua[vg_sw(k0, fi)], T0[kk], X0, 0, 0, 0);
                    X1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ua[vg_sw(k0, 16 + fi)], T0[kk], X1, 0, 0, 0);
                    X0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ua[vg_sw(k1, fi)], T1[kk], X0, 0, 0, 0);
                    X1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ua[vg_sw(k1, 16 + fi)], T1[kk], X1, 0, 0, 0);
                }
            }
            vg_round_barrier<true>();                            // every strip has read its G_ab before anyone writes
            if (wave < nunits) {
                const int fi = lane & 15, fk = lane >> 4;
                const int gj = vg_bgidx(Ib, Jbb, jb * 16 + fi);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Gp[vg_sym(vg_bgidx(Ia, Ja, fk + 4 * r), gj)] = X0[r];
                    Gp[vg_sym(vg_bgidx(Ia, Ja, 16 + fk + 4 * r), gj)] = X1[r];
                }
            }
            if (grp < npair)                                     // diagonal super-block = the rotated S itself
                for (int e = lt; e < 1024; e += 256) {
                    const int r = e >> 5, c = e & 31;
                    const int gi = vg_bgidx(I, Jb, r), gj = vg_bgidx(I, Jb, c);
                    if (gi >= gj) Gp[vg_tri(gi) + gj] = S[vg_sw(r, c)];
                }
            vg_round_barrier<true>();
#ifdef VG_EIG_STAMP
            BST(t1s); tA += t1s - t0s;
#endif
        }
        ++sweeps;
        if (status || !any_sweep) break;
        if (sweep == VG_BJ_MAXSWEEP - 1) status = VGGP_ENOCONV;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < m; i += nthr) J.lam[i] = Gp[vg_tri(i) + i];
    if (tid == 0) {
        J.counters[0] = nlog;
        J.counters[1] = sweeps;
        J.counters[2] = status;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&J.counters[3], nlog | VG_EIG_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef VG_EIG_STAMP
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(J.gwork) + wave * 8;
        dbg[0] = tP; dbg[1] = tB1; dbg[2] = tU; dbg[3] = tB2; dbg[4] = tG; dbg[5] = tA; dbg[6] = nin_tot; dbg[7] = nout;
    }
#endif
}

// replay role for the block log: a 16-column tile of Q^T per workgroup; wave g applies pair g's U with MFMA
__device__ __forceinline__ void vg_breplay_body(const VgEigJob& J, int cblock, double* dyn, int* s_sync) {
    const int m = J.m;
    const int nb = 2 * ((m + 31) / 32), Mp = 16 * nb, npair = nb >> 1;
    const int j0 = cblock * 16;
    if (j0 >= Mp) return;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    double* T = dyn;                                            // [Mp][17]
    double* Ubuf = T + Mp * 17 + (Mp & 1);                      // [npair][32][32]
    const double* ulog = reinterpret_cast<const double*>(J.rotlog);
    for (int idx = tid; idx < Mp * 16; idx += nthr) {
        const int i = idx >> 4, jj = idx & 15, j = j0 + jj;
        double v = (i == j) ? 1.0 : 0.0;
        if (J.Qt0 && i < m && j < m) v = J.Qt0[i * m + j];
        T[i * 17 + jj] = v;
    }
    int consumed = 0;
    for (;;) {
        if (tid == 0) {
            int pw, spins = 0;
            for (;;) {
                pw = __hip_atomic_load(&J.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((pw & VG_EIG_DONE) || (pw & (VG_EIG_DONE - 1)) > consumed) break;
                if (++spins > (1 << 24)) { pw = VG_EIG_DONE | 0x20000000; if (J.err) atomicOr(J.err, 1); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            s_sync[0] = pw;
        }
        __syncthreads();
        const int pw = s_sync[0];
        if (pw & 0x20000000) break;
        const bool done = (pw & VG_EIG_DONE) != 0;
        const int published = pw & 0x0fffffff;
        if (published > consumed) {
            const double* src = ulog + (long)consumed * npair * 1024;
            for (int idx = tid; idx < npair * 1024; idx += nthr)
                Ubuf[idx] = __hip_atomic_load(src + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) s_sync[1] = __hip_atomic_load(&J.roundlog[consumed], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int R = s_sync[1];
            if (wave < npair) {
                int I, Jb;
                vg_pair(nb, R, wave, I, Jb);
                const int fi = lane & 15, fk = lane >> 4;
                const double* Ug = Ubuf + wave * 1024;
                double b[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) b[kk] = T[vg_bgidx(I, Jb, kk * 4 + fk) * 17 + fi];
                vg_bd4 A0 = {0.0, 0.0, 0.0, 0.0}, A1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int k = kk * 4 + fk;
                    A0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ug[k * 32 + fi], b[kk], A0, 0, 0, 0);
                    A1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ug[k * 32 + 16 + fi], b[kk], A1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T[vg_bgidx(I, Jb, fk + 4 * r) * 17 + fi] = A0[r];
                    T[vg_bgidx(I, Jb, 16 + fk + 4 * r) * 17 + fi] = A1[r];
                }
            }
            ++consumed;
        }
        __syncthreads();
        if (done && consumed >= published) break;
    }
    __syncthreads();
    for (int idx = tid; idx < m * 16; idx += nthr) {
        const int i = idx >> 4, j = j0 + (idx & 15);
        if (j < m) {
            const double v = T[i * 17 + (idx & 15)];
            J.Qt[i * m + j] = v;
            if (J.Qt2) J.Qt2[i * m + j] = v;
        }
    }
}

// One launch, up to three roles.  Linear block id = job * nx + bx: bx == 0 is the Jacobi producer of the job's matrix, bx >= 1 replay its
// rotation log on column block bx-1 of Q^T while the producer is still running (the replay is ~3x faster
// per round, so it finishes a few microseconds after the producer).  At most 2*(1+16) workgroups of up to 1024 threads / 136 KB LDS:
// co-resident on an otherwise idle MI355X, NOT guaranteed on a shared GPU -- hence the bounded spin of the replay
// workgroups and the error word (VgEigJob::err) they raise on a timeout (surfaces as VGGP_ENOCONV).
// `rider`: optional GEMM batch executed by extra workgroups of this launch (a parameter of its own: inside VgEigArgs the
// compiler copies the whole argument block to scratch memory at the start of every workgroup)
__global__ __launch_bounds__(1024) void vg_eigh_kernel(const VgEigArgs a, const VgGemmBatch rider) {
    extern __shared__ __attribute__((aligned(16))) double vg_eig_dyn[];   // double2 views of it are read with ds_read_b128
    __shared__ double2 cs[512];
    __shared__ VgPairRec pq[512];
    __shared__ VgActRec actrec[128];          // rotating pairs of the current round (half <= 128)
    __shared__ unsigned char isact[512];
    __shared__ int nact_s[2];
    __shared__ double red[48];
    const int bid = blockIdx.x;
    if (bid >= a.neig) {          // rider role: one 64 x 64 tile of the attached GEMM batch, 8 waves (the other 8 leave)
        if (threadIdx.x >= 512) return;
        vg_gemm_body<64, 16, 512>(rider, vg_eig_dyn, bid - a.neig);
        return;
    }
    const int by = bid >= a.nx ? 1 : 0, bx = bid - by * a.nx;      // at most two jobs (no division: keeps the index scalar)
    const VgEigJob& J = a.job[by];
    const bool block_mode = J.block && J.m <= VG_BJ_MAX_M;
    if (bx == 0) {
        if (block_mode) vg_bjacobi_body(J, vg_eig_dyn, cs, pq, nact_s, red);
        else if (a.use_lds[by]) vg_jacobi_body<true>(J, vg_eig_dyn, cs, pq, actrec, isact, nact_s, red, a.fast[by] != 0);
        else vg_jacobi_body<false>(J, J.gwork, cs, pq, actrec, isact, nact_s, red, false);
    } else {
        if (block_mode) vg_breplay_body(J, bx - 1, vg_eig_dyn, nact_s);
        else vg_replay_body(J, bx - 1, a.rp_cols, vg_eig_dyn, nact_s);
    }
}

static const int VG_EIG_LDS_MAX_M = 184;      // packed lower triangle: 184*185/2*8 B = 136 KB (+ ~19 KB static)

size_t vg_eigh_log_bytes(int m) {
    const size_t m2 = m + (m & 1);
    size_t scalar = (size_t)VG_EIG_MAXSWEEP * (m2 - 1 > 0 ? m2 - 1 : 1) * (m2 / 2) * sizeof(double2);
    if (m > VG_BJ_MAX_M) return scalar;
    const size_t nb = 2 * ((m + 31) / 32), np = nb / 2;
    const size_t blk = (size_t)VG_BJ_MAXSWEEP * (nb - 1) * np * 1024 * sizeof(double);
    return blk > scalar ? blk : scalar;
}

// ---- row orthonormalisation (subspace start) --------------------------------------------------------------------------
// The rows of Z = V G are dominated by what leaks onto the leading eigenvectors (row k is lambda_k v_k + sum_j phi_kj lambda_j u_j
// with phi ~ 1e-3), so their Gram matrix is numerically singular and Cholesky-QR is out; the deflation has to run top-down.
// Classical Gram-Schmidt with re-orthogonalisation, row by row, everything in LDS: the k dot products of a row are taken
// by the 16 waves in parallel, then 128 lanes subtract; a third pass when a pass removed most of the row.
struct VgRowQrArgs { VgRowQrJob job[2]; int njobs; };
// NE: elements of a row per lane in the in-block pass of wave 0 (2: m <= 128, 4: m <= 256); CP: with the pass-through copy (its 16
// registers per lane live across the whole row loop: the m > 128 instance, which only the thin chain uses, goes without -- with it
// the kernel spilled 66 VGPRs and the m <= 128 instance paid 6 us for it)
template <int NE, bool CP>
__device__ __forceinline__ void vg_rowqr_body(const VgRowQrJob& J, double* V) {
    const int r = J.r, m = J.m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the pass-through copy: loads now (registers), stores after the last row -- a global store inside the row loop would be
    // waited for at every barrier (__syncthreads drains vmcnt)
    double cpv[CP ? 16 : 1];
    if (CP) {
#pragma unroll
        for (int u = 0; u < (CP ? 16 : 1); ++u) { const long i = tid + u * 1024L; cpv[u] = i < J.cp_n ? J.cp_src[i] : 0.0; }
    }
    for (int i = tid; i < r * m; i += 1024) V[i] = J.Z[i];
    __syncthreads();
    // Block Gram-Schmidt, 4 rows at a time (r is a multiple of 4): the block is projected off all finished rows twice
    // (4 k0 dot products per pass, 16 lanes each, one barrier; 4 x EW threads subtract), then wave 0 alone orthonormalises the
    // four rows among themselves in registers (NE elements per lane, wave reductions, no workgroup barrier).
    // "Twice is enough", but only where it is needed (Kahan / Parlett): a projection is repeated when it removed more than
    // half of a row's squared norm -- the trailing rows of Z, which are dominated by what leaks onto the leading eigenvectors;
    // the leading rows of a warm start lose almost nothing and take one pass.  The norms come for free: |x|^2 rides in the same
    // reduction as the dot products c_j, and |x - sum c_j q_j|^2 = |x|^2 - sum c_j^2 for orthonormal q_j.
    __shared__ double cb[4 * 64];
    __shared__ double nb2[4];
    __shared__ int need2[4];
    constexpr int EW = 64 * NE;                              // threads per row in the subtraction
    const int grp = tid >> 4, l16 = lane & 15;               // 64 groups of 16 lanes
    for (int k0 = 0; k0 < r; k0 += 4) {
        double* vb = V + k0 * m;
        for (int pass = 0; pass < 2 && k0 > 0; ++pass) {
            for (int p = grp; p < 4 * k0 + (pass == 0 ? 4 : 0); p += 64) {
                const int bb = p & 3, j = p >> 2;
                const double* vj = j < k0 ? V + j * m : vb + bb * m;        // j == k0: the row's own squared norm
                double s = 0.0;
                for (int e = l16; e < m; e += 16) s += vj[e] * vb[bb * m + e];
                s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
                if (l16 == 0) { if (j < k0) cb[bb * 64 + j] = s; else nb2[bb] = s; }
            }
            __syncthreads();
            if (tid < 4 * EW) {
                const int e = tid & (EW - 1), bb = tid / EW;
                if (e < m) {
                    const double* cc = cb + bb * 64;
                    double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
                    int j = 0;
                    for (; j + 3 < k0; j += 4) {
                        x0 += cc[j] * V[j * m + e];
                        x1 += cc[j + 1] * V[(j + 1) * m + e];
                        x2 += cc[j + 2] * V[(j + 2) * m + e];
                        x3 += cc[j + 3] * V[(j + 3) * m + e];
                    }
                    for (; j < k0; ++j) x0 += cc[j] * V[j * m + e];
                    vb[bb * m + e] -= (x0 + x1) + (x2 + x3);
                }
            }
            if (pass == 0 && tid >= 1020) {                                  // one lane per row decides about the second pass
                const int bb = tid - 1020;
                const double* cc = cb + bb * 64;
                double sc2 = 0.0;
                for (int j = 0; j < k0; ++j) sc2 += cc[j] * cc[j];
                need2[bb] = !(nb2[bb] - sc2 >= 0.5 * nb2[bb]);              // (also for NaN)
            }
            __syncthreads();
            if (pass == 0 && !(need2[0] | need2[1] | need2[2] | need2[3])) break;      // uniform
        }
        if (wave == 0) {
            double x[4][NE];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int u = 0; u < NE; ++u) x[bb][u] = lane + 64 * u < m ? vb[bb * m + lane + 64 * u] : 0.0;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                double n2 = 0.0;
                for (int pass = 0; pass < 2; ++pass) {
                    // the (up to three) dot products with the block's finished rows and the row's own squared norm: four
                    // independent reductions through one butterfly
                    double c[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int jb = 0; jb < bb; ++jb)
#pragma unroll
                        for (int u = 0; u < NE; ++u) c[jb] += x[bb][u] * x[jb][u];
#pragma unroll
                    for (int u = 0; u < NE; ++u) c[3] += x[bb][u] * x[bb][u];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
                        for (int jb = 0; jb < 4; ++jb)
                            if (jb < bb || jb == 3) c[jb] += __shfl_xor(c[jb], off);
                    }
                    double sc2 = 0.0;
#pragma unroll
                    for (int jb = 0; jb < bb; ++jb) {
#pragma unroll
                        for (int u = 0; u < NE; ++u) x[bb][u] -= c[jb] * x[jb][u];
                        sc2 += c[jb] * c[jb];
                    }
                    n2 = c[3] - sc2;                                         // |x|^2 after the projection
                    if (n2 >= 0.5 * c[3]) break;                             // wave-uniform: little was removed, once is enough
                }
                const double sc = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
#pragma unroll
                for (int u = 0; u < NE; ++u) x[bb][u] *= sc;
            }
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int u = 0; u < NE; ++u)
                    if (lane + 64 * u < m) vb[bb * m + lane + 64 * u] = x[bb][u];
        }
        __syncthreads();
    }
    for (int i = tid; i < r * m; i += 1024) J.V1[i] = V[i];
    if (CP) {
#pragma unroll
        for (int u = 0; u < (CP ? 16 : 1); ++u) { const long i = tid + u * 1024L; if (i < J.cp_n) J.cp_dst[i] = cpv[u]; }
        for (long i = tid + 16 * 1024L; i < J.cp_n; i += 1024) J.cp_dst[i] = J.cp_src[i];      // (m <= 128: never taken)
    } else {
        for (long i = tid; i < J.cp_n; i += 1024) J.cp_dst[i] = J.cp_src[i];
    }
}
// (two kernels, not one with a branch: the register allocation of a kernel is the maximum over its paths)
template <int NE, bool CP>
__global__ __launch_bounds__(1024) void vg_rowqr_kernel_t(const VgRowQrArgs a, const VgGemmBatch rider) {
    extern __shared__ __attribute__((aligned(16))) double vq_dyn[];
    if ((int)blockIdx.x >= a.njobs) {          // rider role (see vg_eigh_kernel)
        // (the upper 8 waves leave before the body's barriers: on gfx9 -- this library is built for gfx950 only, vggp_create
        //  refuses anything else -- s_barrier counts the waves of the workgroup that have not terminated, so the remaining 8
        //  synchronise among themselves; HIP's portable model does not promise that)
        if (threadIdx.x >= 512) return;
        vg_gemm_body<64, 16, 512>(rider, vq_dyn, blockIdx.x - a.njobs);
        return;
    }
    vg_rowqr_body<NE, CP>(a.job[blockIdx.x], vq_dyn);
}


static const size_t VG_RIDER_LDS = 2 * VgTile<64, 16>::TILE * sizeof(double);

hipError_t vg_rowqr_launch(const VgRowQrJob* jobs, int njobs, hipStream_t st, const VgGemmBatch* rider) {
    if (njobs < 1 || njobs > 2) return hipErrorInvalidValue;
    VgRowQrArgs a;
    a.njobs = njobs;
    VgGemmBatch rb;
    rb.nprob = 0; rb.total_tiles = 0;
    size_t lds = 0;
    if (rider && rider->nprob > 0 && rider->total_tiles > 0) { rb = *rider; lds = VG_RIDER_LDS; }
    for (int j = 0; j < njobs; ++j) {
        a.job[j] = jobs[j];
        if (jobs[j].r < 1 || jobs[j].r > 64 || jobs[j].m < jobs[j].r || jobs[j].m > 256) return hipErrorInvalidValue;
        const size_t need = (size_t)jobs[j].r * jobs[j].m * sizeof(double);
        if (need > lds) lds = need;
    }
    // (separate kernels, not branches: a kernel's register allocation is the maximum over its paths -- the pass-through copy holds
    //  32 registers per lane across the whole row loop, which only the full subspace chain needs)
    bool wide = false, copy = false;
    for (int j = 0; j < njobs; ++j) { wide = wide || jobs[j].m > 128; copy = copy || jobs[j].cp_n > 0; }
    const dim3 grid(njobs + rb.total_tiles), blk(1024);
    if (wide) hipLaunchKernelGGL((vg_rowqr_kernel_t<4, false>), grid, blk, lds, st, a, rb);
    else if (copy) hipLaunchKernelGGL((vg_rowqr_kernel_t<2, true>), grid, blk, lds, st, a, rb);
    else hipLaunchKernelGGL((vg_rowqr_kernel_t<2, false>), grid, blk, lds, st, a, rb);
    return hipGetLastError();
}

__global__ void vg_identity_kernel(double* A, int m) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < m * m) A[idx] = (idx / m == idx % m) ? 1.0 : 0.0;
}
hipError_t vg_identity_launch(double* A, int m, hipStream_t st) {
    hipLaunchKernelGGL(vg_identity_kernel, dim3((m * m + 255) / 256), dim3(256), 0, st, A, m);
    return hipGetLastError();
}

struct VgRefineArgs { VgRefineJob job[2]; int njobs; };
__global__ __launch_bounds__(1024) void vg_refine_kernel(const VgRefineArgs a, const VgGemmBatch rider) {
    extern __shared__ double vr_dyn[];         // m <= 128: the packed lower triangle of Gw
    __shared__ double red[16];
    __shared__ double dg[256];
    __shared__ unsigned char nl[256];          // rows that were numerically null in the previous step (VgRefineJob::lam_prev)
    if ((int)blockIdx.x >= a.njobs) {          // rider role (see vg_eigh_kernel)
        if (threadIdx.x >= 512) return;
        vg_gemm_body<64, 16, 512>(rider, vr_dyn, blockIdx.x - a.njobs);
        return;
    }
    const VgRefineJob& J = a.job[blockIdx.x];
    const int m = J.m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double emax = J.emax > 0.0 ? J.emax : VG_POLISH_EMAX;
    if (J.lam_prev) {                          // (the Newton chain keeps the start order: the largest is not necessarily the first)
        if (wave == 0) {
            double lm = 0.0;
            for (int i = lane; i < m; i += 64) lm = fmax(lm, fabs(J.lam_prev[i]));
            for (int off = 32; off > 0; off >>= 1) lm = fmax(lm, __shfl_xor(lm, off));
            if (lane == 0) red[0] = lm;
        }
        __syncthreads();
        if (tid < 256) nl[tid] = (tid < m && fabs(J.lam_prev[tid]) <= J.null_cut * red[0]) ? 1 : 0;
        __syncthreads();
    } else {
        if (tid < 256) nl[tid] = 0;
        __syncthreads();
    }
    if (m <= 128) {
        // One batch of loads for the whole matrix (16 elements per thread, coalesced); the three passes below then run on
        // registers and LDS.  Gw comes straight from the previous kernel: the three dependent passes over global memory of the
        // general path below cost 27 us at m = 128, this one ~7.
        const int mm = m * m;
        double g[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int idx = tid + u * 1024; g[u] = idx < mm ? J.Gw[idx] : 0.0; }
        double ss = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = tid + u * 1024;
            if (idx < mm) {
                const int i = idx / m, j = idx - i * m;
                ss += g[u] * g[u];
                if (j <= i) vr_dyn[vg_tri(i) + j] = g[u];
                if (i == j) dg[i] = g[u];
            }
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        double fro = 0.0;
        for (int w = 0; w < 16; ++w) fro += red[w];
        const double thr = (J.tol > 0.0 ? J.tol : VG_EIG_TOL) * sqrt(fro) / (double)m;
        double nfloor = -1.0;                                   // diagonal entries at or below this are "at the rounding floor"
        if (J.noise > 0.0) {
            double dm = fmax(lane < m ? fabs(dg[lane]) : 0.0, lane + 64 < m ? fabs(dg[lane + 64]) : 0.0);
            for (int off = 32; off > 0; off >>= 1) dm = fmax(dm, __shfl_xor(dm, off));
            nfloor = J.noise * dm;
        }
        __syncthreads();
        // every thread forms the quotient of ITS elements from the lower-triangle value (E is exactly skew), coalesced stores
        double e[16], mE = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = tid + u * 1024;
            e[u] = 0.0;
            if (idx < mm) {
                const int i = idx / m, j = idx - i * m;
                if (i != j) {
                    const int lo = i > j ? i : j, hi = i > j ? j : i;
                    const double gl = vr_dyn[vg_tri(lo) + hi];
                    if (fabs(gl) > thr && !(fabs(dg[lo]) <= nfloor && fabs(dg[hi]) <= nfloor) && !(nl[lo] && nl[hi])) {
                        const double q = gl / (dg[lo] - dg[hi]);
                        mE = fmax(mE, fabs(q));
                        if (!(fabs(q) <= emax)) mE = 1e300;                    // NaN / inf
                        e[u] = i > j ? q : -q;
                    }
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) mE = fmax(mE, __shfl_xor(mE, off));
        if (lane == 0) red[wave] = mE;
        __syncthreads();
        mE = 0.0;
        for (int w = 0; w < 16; ++w) mE = fmax(mE, red[w]);
        const bool ok = mE <= emax;
        if (!ok && J.flag && tid == 0) atomicOr(J.flag, 2);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = tid + u * 1024;
            if (idx < mm) {
                const int i = idx / m, j = idx - i * m;
                const double ev = ok ? e[u] : 0.0;
                J.E[idx] = ev;
                J.R1[idx] = (i == j) ? 1.0 : ev;
            }
        }
        return;
    }
    if (J.flag) {
        // Newton chain, m > 128: ONE pass over the lower triangle instead of three dependent ones (norm, largest quotient, output) --
        // 93 us per launch at m = 256, three launches per step.  The element threshold comes from the diagonal alone (its norm is a
        // lower bound of the Frobenius norm: a few more negligible elements get a quotient), and a quotient above emax raises the
        // reject bit but no longer blanks E: the chain's last kernel refuses the result anyway and the stored bases stay untouched.
        for (int i = tid; i < m; i += 1024) dg[i] = J.Gw[i * m + i];
        __syncthreads();
        double sd = 0.0, dm = 0.0;
        for (int i = lane; i < m; i += 64) { sd += dg[i] * dg[i]; dm = fmax(dm, fabs(dg[i])); }
        for (int off = 32; off > 0; off >>= 1) { sd += __shfl_xor(sd, off); dm = fmax(dm, __shfl_xor(dm, off)); }
        const double thr1 = (J.tol > 0.0 ? J.tol : VG_EIG_TOL) * sqrt(sd) / (double)m;
        const double nfl1 = J.noise > 0.0 ? J.noise * dm : -1.0;
        bool big = false;
        // 64 x 64 tiles of the lower block triangle through LDS, so that BOTH the tile and its mirror image are written with
        // coalesced rows (the mirror image straight from the registers was 64 cache lines per store instruction)
        double* tl = vr_dyn;                                   // [64][65]
        const int nbt = (m + 63) >> 6, tr = tid >> 6, tc = tid & 63;
        for (int bi = 0; bi < nbt; ++bi)
            for (int bj = 0; bj <= bi; ++bj) {
                double ev[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = tr + 16 * u, ii = bi * 64 + r, jj = bj * 64 + tc;
                    double e = 0.0;
                    if (ii < m && jj < m && jj < ii) {
                        const double g = J.Gw[ii * m + jj];
                        if (fabs(g) > thr1 && !(fabs(dg[ii]) <= nfl1 && fabs(dg[jj]) <= nfl1) && !(nl[ii] && nl[jj])) {
                            e = g / (dg[ii] - dg[jj]);
                            if (!(fabs(e) <= emax)) { big = true; e = 0.0; }
                        }
                    }
                    ev[u] = e;
                    tl[r * 65 + tc] = e;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = tr + 16 * u, ii = bi * 64 + r, jj = bj * 64 + tc;
                    if (bi != bj) {
                        if (ii < m && jj < m) { J.E[ii * m + jj] = ev[u]; J.R1[ii * m + jj] = ev[u]; }
                        const int i2 = bj * 64 + r, j2 = bi * 64 + tc;            // mirror tile, element (i2, j2) = -e(j2, i2)
                        if (i2 < m && j2 < m) { const double t = -tl[tc * 65 + r]; J.E[i2 * m + j2] = t; J.R1[i2 * m + j2] = t; }
                    } else if (ii < m && jj < m) {                                // diagonal tile: lower part as computed, upper part mirrored
                        const double t = jj < ii ? ev[u] : (jj > ii ? -tl[tc * 65 + r] : 0.0);
                        J.E[ii * m + jj] = t; J.R1[ii * m + jj] = jj == ii ? 1.0 : t;
                    }
                }
                __syncthreads();
            }
        if (__any(big) && lane == 0) atomicOr(J.flag, 2);
        return;
    }
    // a wave takes whole rows (coalesced, no integer division); m <= 256: at most 4 column chunks and 16 rows per wave
    double ss = 0.0;
    for (int i = wave; i < m; i += 16)
        for (int j = lane; j < m; j += 64) { const double g = J.Gw[i * m + j]; ss += g * g; }
    for (int i = tid; i < m; i += 1024) dg[i] = J.Gw[i * m + i];
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    double fro = 0.0;
    for (int w = 0; w < 16; ++w) fro += red[w];
    const double thr = (J.tol > 0.0 ? J.tol : VG_EIG_TOL) * sqrt(fro) / (double)m;
    double nfloor = -1.0;
    if (J.noise > 0.0) {
        double dm = 0.0;
        for (int i = lane; i < m; i += 64) dm = fmax(dm, fabs(dg[i]));
        for (int off = 32; off > 0; off >>= 1) dm = fmax(dm, __shfl_xor(dm, off));
        nfloor = J.noise * dm;
    }
    __syncthreads();
    // E from the lower triangle (the value at (i, j), i > j, decides for (j, i) too: E is exactly skew); the transposed
    // element is written by the same lane, so one pass computes every quotient once
    double mE = 0.0;
    for (int i = wave; i < m; i += 16)
        for (int j = lane; j < i; j += 64) {
            const double g = J.Gw[i * m + j];
            if (fabs(g) > thr && !(fabs(dg[i]) <= nfloor && fabs(dg[j]) <= nfloor) && !(nl[i] && nl[j])) mE = fmax(mE, fabs(g / (dg[i] - dg[j])));
        }
    for (int off = 32; off > 0; off >>= 1) mE = fmax(mE, __shfl_xor(mE, off));
    if (lane == 0) red[wave] = mE;
    __syncthreads();
    mE = 0.0;
    for (int w = 0; w < 16; ++w) mE = fmax(mE, red[w]);
    const bool ok = mE <= emax;                             // false also for NaN / inf
    if (!ok && J.flag && tid == 0) atomicOr(J.flag, 2);
    for (int i = wave; i < m; i += 16)
        for (int j = lane; j <= i; j += 64) {
            if (j == i) { J.E[i * m + i] = 0.0; J.R1[i * m + i] = 1.0; continue; }
            const double g = J.Gw[i * m + j];
            const double e = (ok && fabs(g) > thr && !(fabs(dg[i]) <= nfloor && fabs(dg[j]) <= nfloor) && !(nl[i] && nl[j])) ? g / (dg[i] - dg[j]) : 0.0;
            J.E[i * m + j] = e;  J.R1[i * m + j] = e;
            J.E[j * m + i] = -e; J.R1[j * m + i] = -e;
        }
}

hipError_t vg_refine_launch(const VgRefineJob* jobs, int njobs, hipStream_t st, const VgGemmBatch* rider) {
    if (njobs < 1 || njobs > 2) return hipErrorInvalidValue;
    VgRefineArgs a;
    a.njobs = njobs;
    VgGemmBatch rb;
    rb.nprob = 0; rb.total_tiles = 0;
    if (rider && rider->nprob > 0 && rider->total_tiles > 0) rb = *rider;
    for (int j = 0; j < njobs; ++j) {
        a.job[j] = jobs[j];
        if (jobs[j].m < 1 || jobs[j].m > 256) return hipErrorInvalidValue;
    }
    size_t lds = 0;
    for (int j = 0; j < njobs; ++j)
        if (jobs[j].m <= 128) lds = std::max(lds, (size_t)jobs[j].m * (jobs[j].m + 1) / 2 * sizeof(double));
        else if (jobs[j].flag) lds = std::max(lds, (size_t)64 * 65 * sizeof(double));      // the tile buffer of the one-pass path
    if (rb.total_tiles > 0) lds = std::max(lds, (size_t)(2 * VgTile<64, 16>::TILE * sizeof(double)));
    hipLaunchKernelGGL(vg_refine_kernel, dim3(njobs + rb.total_tiles), dim3(1024), lds, st, a, rb);
    return hipGetLastError();
}

// ---- Newton chain: the last look at Gw = S G S^T ------------------------------------------------------------------------------
struct VgNewtonCheckArgs { VgNewtonCheckJob job[2]; int njobs; };
__global__ __launch_bounds__(1024) void vg_newton_check_kernel(const VgNewtonCheckArgs a) {
    __shared__ double red[3 * 16];
    const VgNewtonCheckJob& J = a.job[blockIdx.x];
    const int m = J.m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double dgc[256];
    __shared__ unsigned char nlc[256];       // rows that were numerically null in the previous step (J.lam still holds its eigenvalues here)
    double ss = 0.0, mo = 0.0, lmax = 0.0;
    if (J.null_cut > 0.0) {
        if (wave == 0) {
            double lm = 0.0;
            for (int i = lane; i < m; i += 64) lm = fmax(lm, fabs(J.lam[i]));
            for (int off = 32; off > 0; off >>= 1) lm = fmax(lm, __shfl_xor(lm, off));
            if (lane == 0) red[0] = lm;
        }
        __syncthreads();
        if (tid < 256) nlc[tid] = (tid < m && fabs(J.lam[tid]) <= J.null_cut * red[0]) ? 1 : 0;
        __syncthreads();
    } else {
        if (tid < 256) nlc[tid] = 0;
        __syncthreads();
    }
    for (int i = tid; i < m; i += 1024) { const double g = J.Gw[(long)i * m + i]; dgc[i] = g; J.lam[i] = g; }
    __syncthreads();
    for (int i = lane; i < m; i += 64) lmax = fmax(lmax, dgc[i]);
    for (int off = 32; off > 0; off >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, off));
    const double nfloor = J.noise > 0.0 ? J.noise * lmax : -1.0;
    for (int i = wave; i < m; i += 16)
        for (int j = lane; j < m; j += 64) {
            const double g = J.Gw[(long)i * m + j];
            ss += g * g;
            if (i != j && !(fabs(dgc[i]) <= nfloor && fabs(dgc[j]) <= nfloor) && !(nlc[i] && nlc[j])) mo = fmax(mo, fabs(g));
        }
    for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off); mo = fmax(mo, __shfl_xor(mo, off)); }
    if (lane == 0) { red[wave] = ss; red[16 + wave] = mo; }
    __syncthreads();
    double fro = 0.0; mo = 0.0;
    for (int w = 0; w < 16; ++w) { fro += red[w]; mo = fmax(mo, red[16 + w]); }
    __syncthreads();
    int nr = 0;
    for (int i = tid; i < m; i += 1024) nr += J.Gw[(long)i * m + i] > VG_EIG_RANK_CUT * lmax ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) nr += __shfl_xor(nr, off);
    if (lane == 0) red[wave] = (double)nr;
    __syncthreads();
    if (tid == 0) {
        int nrank = 0;
        for (int w = 0; w < 16; ++w) nrank += (int)red[w];
        const double thr = J.tol * sqrt(fro) / (double)m;
        const bool bad = !(mo <= thr) || (J.flag && (*J.flag & 2));
        // counters[3]: diagnostics of a miss -- bit 1: a rotation was rejected, bit 2: not converged; bits 8..: -log10 of the largest
        // off-diagonal element relative to the threshold (how far from convergence)
        const int rej = (J.flag && (*J.flag & 2)) ? 2 : 0, ncv = !(mo <= thr) ? 4 : 0;
        const int lg = mo > 0.0 && thr > 0.0 ? (int)fmin(99.0, fmax(-99.0, 10.0 * log10(mo / thr))) : -99;
        J.counters[0] = 0; J.counters[1] = (J.iters & 0xff) | (nrank << 8); J.counters[2] = 0; J.counters[3] = rej | ncv | ((lg + 100) << 8);
        if (bad && J.err) atomicOr(J.err, 2);
    }
}
// dst_a <- src_a (and dst_b <- src_b) unless bit 1 of *err is set: the Newton chain only replaces the stored bases when it converged,
// so that the host can repeat a missed step on the regular chain from the SAME warm start
struct VgCopyIfArgs { const int* err[2]; const double* src_a[2]; double* dst_a[2]; const double* src_b[2]; double* dst_b[2]; long n[2]; int njobs; };
__global__ void vg_copy_if_kernel(const VgCopyIfArgs a) {
    const int k = blockIdx.y;
    if (k >= a.njobs || (*a.err[k] & 2)) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n[k]; i += (long)gridDim.x * blockDim.x) {
        if (a.dst_b[k]) a.dst_b[k][i] = a.src_b[k][i];          // (b first: a may overwrite b's source)
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n[k]; i += (long)gridDim.x * blockDim.x) a.dst_a[k][i] = a.src_a[k][i];
}
hipError_t vg_copy_if_launch(const int* const* err, const double* const* src_a, double* const* dst_a, const double* const* src_b,
                             double* const* dst_b, const long* n, int njobs, hipStream_t st) {
    VgCopyIfArgs a;
    a.njobs = njobs;
    for (int k = 0; k < njobs; ++k) { a.err[k] = err[k]; a.src_a[k] = src_a[k]; a.dst_a[k] = dst_a[k]; a.src_b[k] = src_b[k]; a.dst_b[k] = dst_b[k]; a.n[k] = n[k]; }
    hipLaunchKernelGGL(vg_copy_if_kernel, dim3(64, njobs), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t vg_newton_check_launch(const VgNewtonCheckJob* jobs, int njobs, hipStream_t st) {
    if (njobs < 1 || njobs > 2) return hipErrorInvalidValue;
    VgNewtonCheckArgs a;
    a.njobs = njobs;
    for (int j = 0; j < njobs; ++j) { a.job[j] = jobs[j]; if (jobs[j].m < 1 || jobs[j].m > 256) return hipErrorInvalidValue; }
    hipLaunchKernelGGL(vg_newton_check_kernel, dim3(njobs), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t vg_eigh_setup() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vg_refine_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vg_rowqr_kernel_t<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vg_rowqr_kernel_t<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vg_rowqr_kernel_t<4, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);      // r x m <= 64 x 256 doubles
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(vg_eigh_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
}

hipError_t vg_eigh_launch(const VgEigJob* jobs, int njobs, hipStream_t st, const VgGemmBatch* rider) {
    if (njobs < 1 || njobs > 2) return hipErrorInvalidValue;
    VgEigArgs a;
    a.njobs = njobs;
    VgGemmBatch rb;
    rb.nprob = 0; rb.total_tiles = 0;
    if (rider && rider->nprob > 0 && rider->total_tiles > 0) rb = *rider;
    static const char* fs_env = getenv("VGGP_EIG_FAST_SWITCH");      // tuning / A-B switch (0 disables the dense phase)
    static const char* tol_env = getenv("VGGP_EIG_TOL");
    static const bool no_polish = getenv("VGGP_EIG_NO_POLISH") != nullptr;
    static const bool no_newton = getenv("VGGP_EIG_NO_NEWTON") != nullptr;
    size_t lds = 0;
    int maxm2 = 0;
    for (int j = 0; j < njobs; ++j) maxm2 = jobs[j].m + 1 > maxm2 ? jobs[j].m + 1 : maxm2;
    // 16 columns (4 working waves) per replay workgroup: with more, the consumer's own CU becomes the bottleneck
    const int VG_RP_COLS = 16, VG_RP_LD = VG_RP_COLS + 1;
    a.rp_cols = VG_RP_COLS;
    maxm2 = 0;
    for (int j = 0; j < njobs; ++j) {
        a.job[j] = jobs[j];
        if (fs_env) a.job[j].fast_switch = atoi(fs_env);
        if (tol_env) a.job[j].tol = atof(tol_env);
        if (no_polish) a.job[j].polish = 0;
        if (no_newton) a.job[j].newton = 0;
        const int m = jobs[j].m;
        if (m < 1 || m > 256) return hipErrorInvalidValue;
        const int m2 = m + (m & 1);
        a.use_lds[j] = m <= VG_EIG_LDS_MAX_M;
        size_t need = a.use_lds[j] ? (size_t)m2 * (m2 + 1) / 2 * sizeof(double) : 0;
        a.fast[j] = a.use_lds[j] && !jobs[j].block && a.job[j].fast_switch > 0 && m2 >= 16 && m2 <= 128;
        if (a.fast[j]) need *= 2;
        size_t rp = (size_t)m2 * VG_RP_LD * sizeof(double) + VG_RP_CHUNK_BYTES + 4096;
        if (a.fast[j] && a.job[j].polish)      // + T1
            rp += (size_t)m2 * VG_RP_LD * sizeof(double);
        if (jobs[j].block && m <= VG_BJ_MAX_M) {
            const size_t nb = 2 * ((m + 31) / 32), Mp = 16 * nb, np = nb / 2;
            need = (((Mp * (Mp + 1) / 2 + 1) & ~size_t(1)) + np * VG_BJ_SLOT) * sizeof(double);
            rp = (Mp * 17 + 2 + np * 1024) * sizeof(double);
        }
        if (a.use_lds[j] && a.job[j].newton && !jobs[j].Qt0 && m <= VG_EIG_NEWTON_MAX_M && !(jobs[j].block && m <= VG_BJ_MAX_M))
            need = (size_t)m2 * (m2 + 1) * sizeof(double) + 5 * (size_t)((m + 15) & ~15) * (((m + 15) & ~15) + 2) * sizeof(double);      // two packed copies + five padded buffers
        if (a.use_lds[j] && jobs[j].Hl && m <= VG_EIG_NEWTON_MAX_M && jobs[j].hk <= 128) {      // staging area of the two factors
            const size_t stg = (size_t)m2 * (m2 + 1) * sizeof(double) + 2 * (size_t)((m2 + 15) & ~15) * (jobs[j].hk + 2) * sizeof(double);
            if (stg > need) need = stg;
        }
        if (rp > need) need = rp;
        if (need > lds) lds = need;
        if (m2 > maxm2) maxm2 = m2;
    }
    if (lds > 136 * 1024) return hipErrorInvalidValue;
    const int ncb = (maxm2 + VG_RP_COLS - 1) / VG_RP_COLS;
    a.nx = 1 + ncb;
    a.neig = njobs * a.nx;
    if (rb.total_tiles > 0 && lds < VG_RIDER_LDS) lds = VG_RIDER_LDS;
    hipLaunchKernelGGL(vg_eigh_kernel, dim3(a.neig + rb.total_tiles), dim3(1024), lds, st, a, rb);
    return hipGetLastError();
}
