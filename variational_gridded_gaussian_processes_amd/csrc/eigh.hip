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
#include "common.h"

#define VG_EIG_TOL 1e-13

struct VgEigArgs {
    VgEigJob job[2];
    int njobs;
    int use_lds[2];
};

// circle-method pairing of m2 (even) players in round r: pair index k -> (p, q)
__device__ __forceinline__ void vg_pair(int m2, int r, int k, int& p, int& q) {
    const int n1 = m2 - 1;
    if (k == 0) { p = r; q = n1; return; }
    p = r + k; if (p >= n1) p -= n1;
    q = r - k; if (q < 0) q += n1;
}

template <bool INLDS>
__device__ void vg_jacobi_body(const VgEigJob& J, double* W, double2* cs, int* pq, volatile int* flag, double* red) {
    const int m = J.m;
    const int m2 = m + (m & 1), half = m2 >> 1, ld = m2 + 1;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tx = tid & 31, ty = tid >> 5, nty = nthr >> 5;

    // load (zero padded) and Frobenius norm
    double ss = 0.0;
    for (int idx = tid; idx < m2 * m2; idx += nthr) {
        const int i = idx / m2, j = idx - i * m2;
        const double v = (i < m && j < m) ? J.G[i * m + j] : 0.0;
        W[i * ld + j] = v;
        ss += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    if (!INLDS) __threadfence_block();
    __syncthreads();
    double fro = 0.0;
    for (int w = 0; w < (nthr >> 6); ++w) fro += red[w];
    const double thr = VG_EIG_TOL * sqrt(fro) / (double)m;

    int nlog = 0, sweeps = 0, status = 0;
    for (int sweep = 0; sweep < VG_EIG_MAXSWEEP; ++sweep) {
        bool any = false;
        for (int r = 0; r < m2 - 1; ++r) {
            const int par = r & 1;
            if (tid < half) {
                int p, q;
                vg_pair(m2, r, tid, p, q);
                const double gpp = W[p * ld + p], gqq = W[q * ld + q], gpq = W[p * ld + q];
                double c = 1.0, s = 0.0;
                if (fabs(gpq) > thr) {
                    const double tau = (gqq - gpp) / (2.0 * gpq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + hypot(1.0, tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                    flag[par] = 1;
                }
                cs[tid] = make_double2(c, s);
                pq[tid] = p | (q << 16);
            }
            if (tid == nthr - 1) flag[par ^ 1] = 0;
            __syncthreads();
            if (!flag[par]) continue;          // uniform: nothing to rotate in this round
            if (nlog >= J.max_rounds) { status = VGGP_ENOCONV; break; }
            if (tid < half) J.rotlog[(long)nlog * half + tid] = cs[tid];
            if (tid == 0) J.roundlog[nlog] = r;
            ++nlog;
            any = true;
            for (int al = ty; al < half; al += nty) {
                const int pa = pq[al] & 0xffff, qa = pq[al] >> 16;
                const double ca = cs[al].x, sa = cs[al].y;
                for (int be = tx; be < half; be += 32) {
                    const int pb = pq[be] & 0xffff, qb = pq[be] >> 16;
                    const double cb = cs[be].x, sb = cs[be].y;
                    const double g00 = W[pa * ld + pb], g01 = W[pa * ld + qb];
                    const double g10 = W[qa * ld + pb], g11 = W[qa * ld + qb];
                    // right rotation (columns pb, qb), then left rotation (rows pa, qa)
                    const double h00 = cb * g00 - sb * g01, h01 = sb * g00 + cb * g01;
                    const double h10 = cb * g10 - sb * g11, h11 = sb * g10 + cb * g11;
                    W[pa * ld + pb] = ca * h00 - sa * h10;
                    W[qa * ld + pb] = sa * h00 + ca * h10;
                    W[pa * ld + qb] = ca * h01 - sa * h11;
                    W[qa * ld + qb] = sa * h01 + ca * h11;
                }
            }
            if (!INLDS) __threadfence_block();
            __syncthreads();
        }
        ++sweeps;
        if (status || !any) break;
        if (sweep == VG_EIG_MAXSWEEP - 1) status = VGGP_ENOCONV;
    }
    __syncthreads();
    for (int i = tid; i < m; i += nthr) J.lam[i] = W[i * ld + i];
    if (tid == 0) {
        J.counters[0] = nlog;
        J.counters[1] = sweeps;
        J.counters[2] = status;
    }
}

__global__ __launch_bounds__(1024) void vg_jacobi_kernel(const VgEigArgs a) {
    extern __shared__ double vg_eig_dyn[];
    __shared__ double2 cs[512];
    __shared__ int pq[512];
    __shared__ int flag[2];
    __shared__ double red[16];
    const VgEigJob& J = a.job[blockIdx.x];
    if (a.use_lds[blockIdx.x]) vg_jacobi_body<true>(J, vg_eig_dyn, cs, pq, flag, red);
    else vg_jacobi_body<false>(J, J.gwork, cs, pq, flag, red);
}

// ---- rotation-log replay on column blocks of Q^T ---------------------------------
#define VG_RP_COLS 16          // columns per workgroup (4 per wave)
#define VG_RP_LD 17
#define VG_RP_CHUNK_BYTES 16384

__global__ __launch_bounds__(256) void vg_replay_kernel(const VgEigArgs a) {
    extern __shared__ double vg_rp_dyn[];
    const VgEigJob& J = a.job[blockIdx.y];
    const int m = J.m, m2 = m + (m & 1), half = m2 >> 1;
    const int j0 = blockIdx.x * VG_RP_COLS;
    if (j0 >= m2) return;
    double* T = vg_rp_dyn;                                   // [m2][17]
    double2* chunk = reinterpret_cast<double2*>(T + (long)m2 * VG_RP_LD);
    int rounds_per_chunk = VG_RP_CHUNK_BYTES / (half * (int)sizeof(double2));
    if (rounds_per_chunk < 1) rounds_per_chunk = 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int idx = tid; idx < m2 * VG_RP_COLS; idx += 256) {
        const int i = idx >> 4, jj = idx & 15, j = j0 + jj;
        double v = (i == j) ? 1.0 : 0.0;
        if (J.Qt0 && i < m && j < m) v = J.Qt0[i * m + j];
        T[i * VG_RP_LD + jj] = v;
    }
    const int nlog = J.counters[0];
    const int jj = wave * 4 + (lane & 3);                   // this lane's column inside the block
    for (int k0 = 0; k0 < nlog; k0 += rounds_per_chunk) {
        const int nr = min(rounds_per_chunk, nlog - k0);
        __syncthreads();                                     // previous chunk fully consumed / T initialised
        for (int idx = tid; idx < nr * half; idx += 256) chunk[idx] = J.rotlog[(long)k0 * half + idx];
        __syncthreads();
        for (int kk = 0; kk < nr; ++kk) {
            const int r = J.roundlog[k0 + kk];
            for (int al = lane >> 2; al < half; al += 16) {
                int p, q;
                vg_pair(m2, r, al, p, q);
                const double2 c = chunk[kk * half + al];
                const double tp = T[p * VG_RP_LD + jj], tq = T[q * VG_RP_LD + jj];
                T[p * VG_RP_LD + jj] = c.x * tp - c.y * tq;
                T[q * VG_RP_LD + jj] = c.y * tp + c.x * tq;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < m * VG_RP_COLS; idx += 256) {
        const int i = idx >> 4, j = j0 + (idx & 15);
        if (j < m) J.Qt[i * m + j] = T[i * VG_RP_LD + (idx & 15)];
    }
}

static const int VG_EIG_LDS_MAX_M = 136;

hipError_t vg_eigh_setup() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vg_jacobi_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(vg_replay_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}

hipError_t vg_eigh_launch(const VgEigJob* jobs, int njobs, hipStream_t st, hipEvent_t mid) {
    if (njobs < 1 || njobs > 2) return hipErrorInvalidValue;
    VgEigArgs a;
    a.njobs = njobs;
    size_t lds = 0, lds_rp = 0;
    int maxm2 = 0;
    for (int j = 0; j < njobs; ++j) {
        a.job[j] = jobs[j];
        const int m = jobs[j].m;
        if (m < 1 || m > 1024) return hipErrorInvalidValue;
        const int m2 = m + (m & 1);
        a.use_lds[j] = m <= VG_EIG_LDS_MAX_M;
        if (a.use_lds[j]) {
            size_t need = (size_t)m2 * (m2 + 1) * sizeof(double);
            if (need > lds) lds = need;
        }
        size_t rp = (size_t)m2 * VG_RP_LD * sizeof(double) + VG_RP_CHUNK_BYTES + 8192;
        if (rp > lds_rp) lds_rp = rp;
        if (m2 > maxm2) maxm2 = m2;
    }
    hipLaunchKernelGGL(vg_jacobi_kernel, dim3(njobs), dim3(1024), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (mid) { e = hipEventRecord(mid, st); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(vg_replay_kernel, dim3((maxm2 + VG_RP_COLS - 1) / VG_RP_COLS, njobs), dim3(256), lds_rp,
                       st, a);
    return hipGetLastError();
}
