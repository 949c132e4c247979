// Masked / partially observed grids (BASELINE config 5): only the grid points with W[j][i] = 1 are observed, which
// is how the reference sees scattered data (it simply receives the subset as X, y and runs the dense algebra of
// kronecker_structure.py:249-278).  Phi = Kuf W Kuf^T is then no Kronecker product, so
//     Sigma~ = I + rho Phi~0,   rho = s1 s2 / sigma^2,   Phi~0 = sum_obs (b1_i (x) b2_j)(b1_i (x) b2_j)^T
// is ASSEMBLED in M-space (M = m1 m2) from the per-dimension factors and factored densely.  The N x N / M x N
// matrices of the reference are still never formed.  Specification: oracle/kron.py elbo_step_masked().
//
//   assembly   T[i,(a,b)] = sum_j W[j,i] B2[a,j] B2[b,j]                (one GEMM over the mask)
//              R[(i1,k1),(a,b)] = sum_i B1[i1,i] B1[k1,i] T[i,(a,b)]    (one GEMM)  -> permuted into Sigma~
//              (the same two GEMMs with V in place of one B give the two derivative matrices Phi~'_1, Phi~'_2)
//   factor     blocked right-looking Cholesky with 128-wide panels: diagonal blocks by the register-resident
//              single-workgroup kernel (chol.hip), panels and trailing updates by the MFMA GEMM; then the blocked
//              inverse of the factor and Sigma~^{-1} = L^{-T} L^{-1} by GEMM
//   gradient   analytic, from Sigma~^{-1}, its partial traces, <Sigma~^{-1}, Phi~'_d> and three n2 x n1 products.
#include "ctx.h"

#include <algorithm>

#define VG_MB 128                 // panel width of the blocked Cholesky
#define VG_MD_NPART 256           // partial sums per reduction job
#define VG_MD_MAXJOBS 24

// ---- small kernels -------------------------------------------------------------------------------------------------
__global__ void vgm_pairprod_kernel(const double* Xa, const double* Xb, int ma, int mb, long n, double* out) {
    // out[(a*mb + b)][j] = Xa[a][j] * Xb[b][j]
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)ma * mb * n) return;
    const long ab = idx / n, j = idx - ab * n;
    const int a = (int)(ab / mb), b = (int)(ab - (long)a * mb);
    out[idx] = Xa[a * n + j] * Xb[b * n + j];
}

// R[(i1*m1+k1)][(i2*m2+k2)] -> out[(i1*m2+i2)][(k1*m2+k2)] ; mode 1: out = I + rho R (rho = s1 s2 / v from theta)
__global__ void vgm_permute_kernel(const double* R, int m1, int m2, const double* theta, int mode, double* out) {
    const long M = (long)m1 * m2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const long row = idx / M, col = idx - row * M;
    const int i1 = (int)(row / m2), i2 = (int)(row - (long)i1 * m2);
    const int k1 = (int)(col / m2), k2 = (int)(col - (long)k1 * m2);
    const double r = R[((long)i1 * m1 + k1) * ((long)m2 * m2) + ((long)i2 * m2 + k2)];
    if (mode == 1) {
        const double rho = theta[2] * theta[3] / theta[4];
        out[idx] = (row == col ? 1.0 : 0.0) + rho * r;
    } else {
        out[idx] = r;
    }
}

// out[j] = sum_a Xa[a][j] * Xb[a][j]   (column dot products, X is [m][n])
__global__ void vgm_coldot_kernel(const double* Xa, const double* Xb, int m, long n, double* out) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int a = 0; a < m; ++a) s += Xa[a * n + j] * Xb[a * n + j];
    out[j] = s;
}

// wcol[i] = sum_j W[j][i] v[j]  (n1 outputs) ; wrow[j] = sum_i W[j][i] u[i]  (n2 outputs)
// two stages (row slabs, then a deterministic sum over the slabs): a single pass with one thread per column is 8 workgroups
// walking 2048 rows each -- 0.5 ms of pure latency at n = 2048
__global__ void vgm_wcol_part_kernel(const double* W, const double* v, long n1, long n2, int nslab, double* part) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const long per = (n2 + nslab - 1) / nslab, j0 = blockIdx.y * per, j1 = j0 + per < n2 ? j0 + per : n2;
    double s = 0.0;
    for (long j = j0; j < j1; ++j) s += W[j * n1 + i] * v[j];
    part[(long)blockIdx.y * n1 + i] = s;
}
__global__ void vgm_wcol_sum_kernel(const double* part, long n1, int nslab, double* wcol) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    double s = 0.0;
    for (int k = 0; k < nslab; ++k) s += part[(long)k * n1 + i];
    wcol[i] = s;
}
__global__ __launch_bounds__(256) void vgm_wrow_kernel(const double* W, const double* u, long n1, long n2, double* wrow) {
    __shared__ double red[4];
    const long j = blockIdx.x;
    double s = 0.0;
    for (long i = threadIdx.x; i < n1; i += 256) s += W[j * n1 + i] * u[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) wrow[j] = red[0] + red[1] + red[2] + red[3];
}

// Xs[a][j] = X[a][j] * w[j]
__global__ void vgm_scalecols_kernel(const double* X, const double* w, int m, long n, double* Xs) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)m * n) return;
    Xs[idx] = X[idx] * w[idx % n];
}

// partial traces of the M x M matrix S (M = m1 m2): which = 1: out[a][b] = sum_j S[(a,j)][(b,j)]  (m1 x m1)
//                                                   which = 2: out[a][b] = sum_i S[(i,a)][(i,b)]  (m2 x m2)
__global__ void vgm_ptrace_kernel(const double* S, int m1, int m2, int which, double* out) {
    const long M = (long)m1 * m2;
    const int md = which == 1 ? m1 : m2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)md * md) return;
    const int a = (int)(idx / md), b = (int)(idx - (long)a * md);
    double s = 0.0;
    if (which == 1) for (int j = 0; j < m2; ++j) s += S[((long)a * m2 + j) * M + ((long)b * m2 + j)];
    else            for (int i = 0; i < m1; ++i) s += S[((long)i * m2 + a) * M + ((long)i * m2 + b)];
    out[idx] = s;
}

// generic reductions, VG_MD_NPART partial sums per job:
//   op 0: sum a[i*sa] * b[i*sb]     op 1: sum log(a[i*sa])     op 2: sum a[i*sa]     op 3: sum a[i] * b[i] * c[i]
struct VgmRedJob { const double* a; const double* b; const double* c; long n, sa, sb; int op; };
struct VgmRedArgs { VgmRedJob job[VG_MD_MAXJOBS]; int njobs; double* partial; };
__global__ __launch_bounds__(256) void vgm_red_kernel(const VgmRedArgs A) {
    __shared__ double red[4];
    const VgmRedJob& J = A.job[blockIdx.y];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < J.n; i += (long)VG_MD_NPART * 256) {
        if (J.op == 0) s += J.a[i * J.sa] * J.b[i * J.sb];
        else if (J.op == 1) s += log(J.a[i * J.sa]);
        else if (J.op == 2) s += J.a[i * J.sa];
        else s += J.a[i] * J.b[i] * J.c[i];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) A.partial[blockIdx.y * VG_MD_NPART + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// job indices of the masked step (order of the partial-sum buffer)
enum { RJ_LOGDET = 0, RJ_Q, RJ_AA, RJ_TRS, RJ_TRPHI, RJ_MK1PTS, RJ_TRMK1, RJ_SPHI1, RJ_AC1, RJ_MKA1, RJ_Z1, RJ_HV1, RJ_MK1PT,
       RJ_MK2PTS, RJ_TRMK2, RJ_SPHI2, RJ_AC2, RJ_MKA2, RJ_Z2, RJ_HV2, RJ_MK2PT, RJ_COUNT };

// partial[job][VG_MD_NPART] -> scal[job].  Row-sharded step: the sums over this rank's grid rows (local_mask bit set) add up
// over the ranks; every other scalar is computed identically on every rank from all-reduced inputs, so only rank 0
// contributes it to the second (tiny) all-reduce -- exact, and all ranks end up with identical scalars.
__global__ void vgm_sum_kernel(const double* partial, double* scal, unsigned local_mask, int keep_replicated) {
    const int j = threadIdx.x;
    if (j >= RJ_COUNT) return;
    double s = 0.0;
    for (int k = 0; k < VG_MD_NPART; ++k) s += partial[j * VG_MD_NPART + k];
    scal[j] = (((local_mask >> j) & 1u) || keep_replicated) ? s : 0.0;
}

struct VgmFinalArgs { const double* theta; const double* scal; double* out; double N, yy; int m1, m2; };
__global__ void vgm_final_kernel(const VgmFinalArgs A) {
    __shared__ double S[RJ_COUNT];
    if (threadIdx.x < RJ_COUNT) S[threadIdx.x] = A.scal[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double s1 = A.theta[2], s2 = A.theta[3], v = A.theta[4], N = A.N, yy = A.yy;
    const double M = (double)A.m1 * A.m2, rho = s1 * s2 / v, ss = s1 * s2;
    const double logdet = 2.0 * S[RJ_LOGDET], q = S[RJ_Q], aa = S[RJ_AA], trS = S[RJ_TRS], trPhi = S[RJ_TRPHI];
    const double elbo = -0.5 * (N * 1.8378770664093453 + N * log(v) + logdet + yy / v - (ss / (v * v)) * q)
                        - (N * ss - ss * trPhi) / (2.0 * v);
    const double trSP = (M - trS) / rho, aPa = (q - aa) / rho;
    const double common = -0.5 * (rho * trSP - (ss / (v * v)) * q + (ss / (v * v)) * rho * aPa);
    const double g_s1 = common / s1 - (N * s2 - s2 * trPhi) / (2.0 * v);
    const double g_s2 = common / s2 - (N * s1 - s1 * trPhi) / (2.0 * v);
    const double g_v = -0.5 * (N / v - (rho / v) * trSP - yy / (v * v) + 2.0 * ss * q / (v * v * v) - (ss * rho / (v * v * v)) * aPa)
                       + (N * ss - ss * trPhi) / (2.0 * v * v);
    const double ld1 = S[RJ_MK1PTS] - (double)A.m2 * S[RJ_TRMK1] + 2.0 * rho * S[RJ_SPHI1];
    const double quad1 = 2.0 * S[RJ_AC1] - S[RJ_MKA1] - 2.0 * rho * S[RJ_Z1];
    const double g_l1 = -0.5 * (ld1 - (ss / (v * v)) * quad1) + (ss / (2.0 * v)) * (2.0 * S[RJ_HV1] - S[RJ_MK1PT]);
    const double ld2 = S[RJ_MK2PTS] - (double)A.m1 * S[RJ_TRMK2] + 2.0 * rho * S[RJ_SPHI2];
    const double quad2 = 2.0 * S[RJ_AC2] - S[RJ_MKA2] - 2.0 * rho * S[RJ_Z2];
    const double g_l2 = -0.5 * (ld2 - (ss / (v * v)) * quad2) + (ss / (2.0 * v)) * (2.0 * S[RJ_HV2] - S[RJ_MK2PT]);
    A.out[0] = elbo; A.out[1] = g_l1; A.out[2] = g_l2; A.out[3] = g_s1; A.out[4] = g_s2; A.out[5] = g_v;
}

// q(v) diagonal: var[a] = s1 s2 sum_{b,c} Lk[a][b] Sinv[b][c] Lk[a][c] with Lk = L1 (x) L2, computed as rowdot(Lk Sinv, Lk)
__global__ void vgm_kron_kernel(const double* L1, const double* L2, int m1, int m2, double* Lk) {
    const long M = (long)m1 * m2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const long row = idx / M, col = idx - row * M;
    Lk[idx] = L1[(row / m2) * m1 + col / m2] * L2[(row % m2) * m2 + col % m2];
}
// e_d = +1: Kuu_d = s_d K0;  e_d = -1: Kuu_d = K0 / s_d (inter-domain VFF / B1 features): q(v) = L (...) with L = s^(e/2) L0
__global__ void vgm_rowdot_scale_kernel(const double* A, const double* B, long M, const double* theta, double* out, int e1, int e2) {
    const long a = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= M) return;
    double s = 0.0;
    for (long b = 0; b < M; ++b) s += A[a * M + b] * B[a * M + b];
    out[a] = (e1 > 0 ? theta[2] : 1.0 / theta[2]) * (e2 > 0 ? theta[3] : 1.0 / theta[3]) * s;
}
__global__ void vgm_scale_rho_kernel(double* x, long n, const double* theta, int e1, int e2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= (e1 > 0 ? theta[2] : 1.0) * (e2 > 0 ? theta[3] : 1.0) / theta[4];      // s_d^((1 + e_d) / 2) / sigma^2
}

#define VGM_MAX_M 16384      // dense M x M solver: ~10 M^2 doubles of workspace (21 GB at the limit), O(M^3) per step
#define VGM_LAUNCH1D(kern, n, st, ...) \
    hipLaunchKernelGGL(kern, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, st, __VA_ARGS__)

// ---- masked workspace ------------------------------------------------------------------------------------------------
struct VgMasked {
    long M = 0, n1 = 0, n2 = 0;
    int m1 = 0, m2 = 0, nblk = 0;
    bool scattered = false;        // n1 == n2 == number of points: the [n2][n1] grid buffers shrink to per-point vectors
    double *Sg, *Lg, *Xg, *Sinv, *R, *Phip, *DI, *Tmp;          // M x M (Phip: two of them; Tmp: 128 x M)
    double *PP1, *PP1v, *PP2, *PP2v, *T, *Tv;
    double *UB, *UV, *Zb, *Zv1, *Zv2, *B1s, *B2s;
    double *nb1, *nb2, *hv1, *hv2, *wn1, *wn2, *PT1, *PT2, *PTS1, *PTS2, *MkA1, *MkA2, *a0, *partial, *out;
    double *cholscratch, *choljit;
    int* cholstatus;
    // the all-reduce buffer of the row-sharded masked step: [slack 2 m2^2 | C, C1, C2 (3 M) | wn2 (n1) | R3 (3 M^2)];
    // the collective covers everything after the slack (vg_partials_enqueue leaves G2, H2 in it, which this path ignores)
    double *mpay, *R3, *scal;
    void* mem = nullptr;
    size_t bytes = 0;
};

static void vgm_layout(VgMasked& w, char* base, size_t& off) {
    auto take = [&](size_t count) {
        off = (off + 255) & ~size_t(255);
        double* p = base ? reinterpret_cast<double*>(base + off) : nullptr;
        off += count * sizeof(double);
        return p;
    };
    const size_t M = w.M, MM = M * M, n1 = w.n1, n2 = w.n2, m1 = w.m1, m2 = w.m2;
    w.Sg = take(MM); w.Lg = take(MM); w.Xg = take(MM); w.Sinv = take(MM); w.R = take(MM); w.Phip = take(2 * MM);
    w.DI = take((size_t)w.nblk * VG_MB * VG_MB); w.Tmp = take((size_t)VG_MB * M);
    w.PP1 = take(m1 * m1 * n1); w.PP1v = take(m1 * m1 * n1); w.PP2 = take(m2 * m2 * n2); w.PP2v = take(m2 * m2 * n2);
    const size_t grid = w.scattered ? n1 : n1 * n2;            // scattered: zb, zv1, zv2 are per-point vectors
    w.T = take(w.scattered ? 256 * (m1 > m2 ? m1 * m1 : m2 * m2) : n1 * m2 * m2);      // scattered: split-K slabs of the small long-K products
    w.Tv = take(w.scattered ? 256 : n1 * m2 * m2);
    w.UB = take(m1 * n2); w.UV = take(m1 * n2); w.Zb = take(grid); w.Zv1 = take(grid); w.Zv2 = take(grid);
    w.B1s = take(m1 * n1); w.B2s = take(m2 * n2);
    w.nb1 = take(n1); w.nb2 = take(n2); w.hv1 = take(n1); w.hv2 = take(n2); w.wn1 = take(n2);
    const size_t wn2len = w.scattered ? 0 : n1;                 // (scattered: the all-reduce payload is C, C1, C2 | R3, the same
    w.mpay = take(2 * m2 * m2 + 3 * M + wn2len + 3 * MM);       //  length on every rank whatever its number of points)
    w.wn2 = base ? w.mpay + 2 * m2 * m2 + 3 * M : nullptr;
    w.R3 = base ? w.wn2 + wn2len : nullptr;
    w.scal = take(32);
    w.PT1 = take(m1 * m1); w.PT2 = take(m2 * m2); w.PTS1 = take(m1 * m1); w.PTS2 = take(m2 * m2);
    w.MkA1 = take(M); w.MkA2 = take(M); w.a0 = take(M);
    w.partial = take(VG_MD_MAXJOBS * VG_MD_NPART); w.out = take(8);
    w.cholscratch = take(VG_MB * (VG_MB + 1)); w.choljit = take(8);
    w.cholstatus = reinterpret_cast<int*>(take(8));
}

static int vgm_prepare(vggp_ctx* c) {
    VgMasked* w = reinterpret_cast<VgMasked*>(c->masked);
    if (!w) { w = new VgMasked(); c->masked = w; }
    const long m1 = c->desc.m1, m2 = c->desc.m2;
    const bool sc = (c->desc.flags & VGGP_FLAG_SCATTERED) != 0;
    if (w->mem && w->M == m1 * m2 && w->n1 == c->desc.n1 && w->n2 == c->desc.n2 && w->m1 == m1 && w->scattered == sc) return VGGP_OK;
    if (w->mem) { VG_HIP(hipFree(w->mem)); w->mem = nullptr; }
    w->M = m1 * m2; w->m1 = (int)m1; w->m2 = (int)m2; w->n1 = c->desc.n1; w->n2 = c->desc.n2; w->scattered = sc;
    w->nblk = (int)((w->M + VG_MB - 1) / VG_MB);
    size_t off = 0;
    vgm_layout(*w, nullptr, off);
    w->bytes = off + 4096;
    VG_HIP(hipMalloc(&w->mem, w->bytes));
    VG_HIP(hipMemset(w->mem, 0, w->bytes));
    off = 0;
    vgm_layout(*w, reinterpret_cast<char*>(w->mem), off);
    return VGGP_OK;
}

void vg_masked_free(vggp_ctx* c) {
    VgMasked* w = reinterpret_cast<VgMasked*>(c->masked);
    if (!w) return;
    if (w->mem) (void)hipFree(w->mem);
    delete w;
    c->masked = nullptr;
}

static int gemm1(const double* A, long sa_m, long sa_k, const double* B, long sb_k, long sb_n, double* C, int ldc, int M,
                 int N, int K, hipStream_t st, double alpha = 1.0, int accum = 0) {
    VgGemmBatch g;
    vg_gemm_init(&g);
    vg_gemm_add(&g, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, M, N, K, 1, 0, 1, 0, alpha, accum);
    VG_HIP(vg_gemm_launch(&g, st));
    return VGGP_OK;
}

// C (M x N, contiguous) = A B with a SHORT output and a LONG reduction (m x m results summed over 1e5 points): one workgroup per
// tile would walk the whole K alone (5.8 ms at K = 100 000, measured), so the reduction is split over up to 256 workgroups whose
// slabs are summed in fixed order.  scratch: >= 256 * M * N doubles.
static int gemm_longk(const double* A, long sa_m, long sa_k, const double* B, long sb_k, long sb_n, double* C, int M, int N, int K,
                      double* scratch, hipStream_t st) {
    int ks = K / 512;
    if (ks > 256) ks = 256;
    if (ks < 2) return gemm1(A, sa_m, sa_k, B, sb_k, sb_n, C, N, M, N, K, st);
    VgGemmBatch g;
    vg_gemm_init(&g);
    const int ip = vg_gemm_add(&g, A, sa_m, sa_k, B, sb_k, sb_n, scratch, N, M, N, K, ks, (long)M * N);
    const int slabs = g.p[ip].ksplit;
    VG_HIP(vg_gemm_launch(&g, st));
    VgRedBatch r;
    vg_red_init(&r);
    vg_red_add(&r, scratch, C, (long)M * N, (long)M * N, slabs);
    VG_HIP(vg_red_launch(&r, st));
    return VGGP_OK;
}

// S (M x M, SPD, destroyed) -> L (lower), X = L^{-1}, optionally Sinv = S^{-1}.  Blocked right-looking Cholesky with
// VG_MB-wide panels: diagonal blocks by the single-workgroup kernel (chol.hip), panels and trailing updates by the MFMA
// GEMM; then the blocked inverse of the factor.  Shared with vggp_cholesky_inverse for m > 128 (ctx.h).
int vg_blocked_chol_inverse(const VgDenseChol& w, hipStream_t st) {
    const int M = (int)w.M;
    const int nblk = (M + VG_MB - 1) / VG_MB;
    int rc;
    VG_HIP(hipMemsetAsync(w.L, 0, sizeof(double) * w.M * w.M, st));
    VG_HIP(hipMemsetAsync(w.X, 0, sizeof(double) * w.M * w.M, st));
    for (int kb = 0; kb < nblk; ++kb) {
        const int k0 = kb * VG_MB, nbk = std::min(VG_MB, M - k0), rest = M - k0 - nbk;
        VgClearArgs clr;
        clr.n = 1; clr.ptr[0] = reinterpret_cast<int*>(w.scratch); clr.nwords[0] = 16;
        VG_HIP(vg_clear_launch(&clr, st));
        VgCholJob j{w.S + (long)k0 * M + k0, w.L + (long)k0 * M + k0, w.DI + (long)kb * VG_MB * VG_MB, w.scratch, w.jit,
                    w.status, nbk};
        j.ldk = M; j.ldl = M; j.only_level0 = 1;
        VG_HIP(vg_chol_launch(&j, 1, st));
        if (rest > 0) {
            // panel: L[i, kb] = A[i, kb] Linv_kk^T ;  trailing: A[i, j] -= L[i, kb] L[j, kb]^T
            const double* DIk = w.DI + (long)kb * VG_MB * VG_MB;
            if ((rc = gemm1(w.S + (long)(k0 + nbk) * M + k0, M, 1, DIk, 1, nbk, w.L + (long)(k0 + nbk) * M + k0, M, rest, nbk, nbk, st))) return rc;
            // (only the lower triangle of the trailing matrix is read later: column strips that start on the diagonal, one launch)
            const double* Lp = w.L + (long)(k0 + nbk) * M + k0;
            double* St = w.S + (long)(k0 + nbk) * M + (k0 + nbk);
            const int nstrip = std::min(VG_GEMM_MAXP, (rest + 1023) / 1024);
            const int wd = (((rest + nstrip - 1) / nstrip) + VG_MB - 1) / VG_MB * VG_MB;
            VgGemmBatch g;
            vg_gemm_init(&g);
            for (int c0 = 0; c0 < rest; c0 += wd)
                vg_gemm_add(&g, Lp + (long)c0 * M, M, 1, Lp + (long)c0 * M, 1, M, St + (long)c0 * M + c0, M, rest - c0, std::min(wd, rest - c0),
                            nbk, 1, 0, 1, 0, -1.0, 1);
            VG_HIP(vg_gemm_launch(&g, st));
        }
    }
    // blocked inverse of the lower factor: X[k,k] = inv(L_kk); X[i, :i] = -inv(L_ii) (L[i, :i] X[:i, :i])
    for (int kb = 0; kb < nblk; ++kb) {
        const int k0 = kb * VG_MB, nbk = std::min(VG_MB, M - k0);
        VG_HIP(hipMemcpy2DAsync(w.X + (long)k0 * M + k0, sizeof(double) * M, w.DI + (long)kb * VG_MB * VG_MB, sizeof(double) * nbk,
                                sizeof(double) * nbk, nbk, hipMemcpyDeviceToDevice, st));
        if (kb == 0) continue;
        {   // (X[:k0, :k0] is lower triangular: the tiles skip the k-range above their columns)
            VgGemmBatch g;
            vg_gemm_init(&g);
            const int i = vg_gemm_add(&g, w.L + (long)k0 * M, M, 1, w.X, M, 1, w.Tmp, k0, nbk, k0, k0);
            g.p[i].tri = VG_TRI_B_LOWER;
            VG_HIP(vg_gemm_launch(&g, st));
        }
        if ((rc = gemm1(w.DI + (long)kb * VG_MB * VG_MB, nbk, 1, w.Tmp, k0, 1, w.X + (long)k0 * M, M, nbk, k0, nbk, st, -1.0, 0))) return rc;
    }
    if (!w.Sinv) return VGGP_OK;
    VgGemmBatch g;                                                     // Sinv = X^T X, X lower triangular: the sum runs over k >= max(i, j)
    vg_gemm_init(&g);
    const int i = vg_gemm_add(&g, w.X, 1, M, w.X, M, 1, w.Sinv, M, M, M, M);
    g.p[i].tri = VG_TRI_A_UPPER_B_LOWER;
    VG_HIP(vg_gemm_launch(&g, st));
    return VGGP_OK;
}

static int dense_chol_inverse(vggp_ctx* c, VgMasked& w, hipStream_t st) {
    (void)c;
    VgDenseChol d{w.Sg, w.Lg, w.Xg, w.DI, w.Tmp, w.cholscratch, w.choljit, w.cholstatus, w.M, w.Sinv};
    return vg_blocked_chol_inverse(d, st);
}

extern "C" int vggp_elbo_step_masked(vggp_ctx* c, const double* Ym, const double* W, double n_obs, double yy_obs,
                                     const double theta[5], double* elbo_out, double grad_out[5], vggp_info* info,
                                     void* stream) {
    if (!c || !c->planned) { vg_set_error("vggp_elbo_step_masked: context not planned"); return VGGP_ESTATE; }
    VG_REQUIRE(Ym && W && theta && elbo_out && grad_out, "vggp_elbo_step_masked: null argument");
    c->have_masked = false;          // a failed step must not leave an earlier step's state readable (qv_masked / posterior_masked)
    VG_REQUIRE(!(c->desc.flags & VGGP_FLAG_SCATTERED), "vggp_elbo_step_masked: the context was planned for scattered points");
    const long m1 = c->desc.m1, m2 = c->desc.m2, n1 = c->desc.n1, n2 = c->desc.n2, M = m1 * m2;
    VG_REQUIRE(M <= VGM_MAX_M, "vggp_elbo_step_masked: M = m1*m2 = %ld too large for the dense masked solver (<= %d)", M, VGM_MAX_M);
    VG_REQUIRE(m1 * m1 * n1 < (1L << 31) && m2 * m2 * n2 < (1L << 31) && M * M < (1L << 31) * 4, "masked problem too large");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    for (int i = 0; i < 5; ++i) {
        VG_REQUIRE(theta[i] > 0.0 && std::isfinite(theta[i]), "theta[%d]=%g must be positive and finite", i, theta[i]);
        c->h_theta[i] = theta[i];
    }
    int rc = vgm_prepare(c);
    if (rc) return rc;
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    // factor build, Cholesky, B|V, Mk and the projections C, C1, C2 of the masked observations (unit outputscale).
    // Row-sharded job (n_ranks > 1): Ym, W are this rank's row slab; every sum over grid rows below is a PARTIAL sum that
    // lands in the all-reduce buffer w.mpay (C, C1, C2 | wn2 | the three M x M assembly matrices) -- ONE collective --
    // after which Sigma~, its factorisation and a0 are replicated on every rank.
    if ((rc = vg_partials_enqueue(c, Ym, w.mpay, st))) return rc;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const double *B1 = d1.BV, *V1 = d1.BV + m1 * n1, *B2 = d2.BV, *V2 = d2.BV + m2 * n2;
    const double *C0 = w.mpay + 2 * m2 * m2, *C1 = C0 + M, *C2 = C1 + M;

    // column statistics
    VGM_LAUNCH1D(vgm_coldot_kernel, n1, st, B1, B1, (int)m1, n1, w.nb1);
    VGM_LAUNCH1D(vgm_coldot_kernel, n2, st, B2, B2, (int)m2, n2, w.nb2);
    VGM_LAUNCH1D(vgm_coldot_kernel, n1, st, V1, B1, (int)m1, n1, w.hv1);
    VGM_LAUNCH1D(vgm_coldot_kernel, n2, st, V2, B2, (int)m2, n2, w.hv2);
    {
        const long mm = m2 * m2;                                // w.T ([m2 m2][n1], written by the assembly below) is the scratch
        const int nslab = (int)(mm < 64 ? mm : 64);
        hipLaunchKernelGGL(vgm_wcol_part_kernel, dim3((unsigned)((n1 + 255) / 256), (unsigned)nslab), dim3(256), 0, st, W, w.nb2,
                           n1, n2, nslab, w.T);
        VGM_LAUNCH1D(vgm_wcol_sum_kernel, n1, st, w.T, n1, nslab, w.wn2);           // partial over this rank's rows
    }
    hipLaunchKernelGGL(vgm_wrow_kernel, dim3((unsigned)n2), dim3(256), 0, st, W, w.nb1, n1, n2, w.wn1);
    // assembly (partial over this rank's rows: T sums over j, R = PP1 T is linear in T)
    VGM_LAUNCH1D(vgm_pairprod_kernel, m2 * m2 * n2, st, B2, B2, (int)m2, (int)m2, n2, w.PP2);
    VGM_LAUNCH1D(vgm_pairprod_kernel, m2 * m2 * n2, st, B2, V2, (int)m2, (int)m2, n2, w.PP2v);
    VGM_LAUNCH1D(vgm_pairprod_kernel, m1 * m1 * n1, st, B1, B1, (int)m1, (int)m1, n1, w.PP1);
    VGM_LAUNCH1D(vgm_pairprod_kernel, m1 * m1 * n1, st, B1, V1, (int)m1, (int)m1, n1, w.PP1v);
    VG_HIP(hipGetLastError());
    if ((rc = gemm1(W, 1, n1, w.PP2, 1, n2, w.T, (int)(m2 * m2), (int)n1, (int)(m2 * m2), (int)n2, st))) return rc;
    if ((rc = gemm1(W, 1, n1, w.PP2v, 1, n2, w.Tv, (int)(m2 * m2), (int)n1, (int)(m2 * m2), (int)n2, st))) return rc;
    if ((rc = gemm1(w.PP1, n1, 1, w.T, m2 * m2, 1, w.R3, (int)(m2 * m2), (int)(m1 * m1), (int)(m2 * m2), (int)n1, st))) return rc;
    if ((rc = gemm1(w.PP1v, n1, 1, w.T, m2 * m2, 1, w.R3 + M * M, (int)(m2 * m2), (int)(m1 * m1), (int)(m2 * m2), (int)n1, st))) return rc;
    if ((rc = gemm1(w.PP1, n1, 1, w.Tv, m2 * m2, 1, w.R3 + 2 * M * M, (int)(m2 * m2), (int)(m1 * m1), (int)(m2 * m2), (int)n1, st))) return rc;
    // the collective of the masked step (no-op on a single-rank context)
    if ((rc = vg_allreduce(c, w.mpay + 2 * m2 * m2, 3 * M + n1 + 3 * M * M, st))) return rc;
    VGM_LAUNCH1D(vgm_permute_kernel, M * M, st, w.R3, (int)m1, (int)m2, c->theta, 1, w.Sg);
    VGM_LAUNCH1D(vgm_permute_kernel, M * M, st, w.R3 + M * M, (int)m1, (int)m2, c->theta, 0, w.Phip);
    VGM_LAUNCH1D(vgm_permute_kernel, M * M, st, w.R3 + 2 * M * M, (int)m1, (int)m2, c->theta, 0, w.Phip + M * M);
    // dense factorisation and inverse of Sigma~
    if ((rc = dense_chol_inverse(c, w, st))) return rc;
    // a0 = Sinv c0 ; A0 = mat(a0) (m1 x m2)
    if ((rc = gemm1(w.Sinv, M, 1, C0, 1, 1, w.a0, 1, (int)M, 1, (int)M, st))) return rc;
    // Mk1 A0, A0 Mk2 ; UB = A0 B2, UV = A0 V2 ; Zb^T = UB^T B1, Zv1^T = UB^T V1, Zv2^T = UV^T B1   ([n2][n1] like W)
    if ((rc = gemm1(d1.Mk, m1, 1, w.a0, m2, 1, w.MkA1, (int)m2, (int)m1, (int)m2, (int)m1, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, d2.Mk, m2, 1, w.MkA2, (int)m2, (int)m1, (int)m2, (int)m2, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, B2, n2, 1, w.UB, (int)n2, (int)m1, (int)n2, (int)m2, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, V2, n2, 1, w.UV, (int)n2, (int)m1, (int)n2, (int)m2, st))) return rc;
    if ((rc = gemm1(w.UB, 1, n2, B1, n1, 1, w.Zb, (int)n1, (int)n2, (int)n1, (int)m1, st))) return rc;
    if ((rc = gemm1(w.UB, 1, n2, V1, n1, 1, w.Zv1, (int)n1, (int)n2, (int)n1, (int)m1, st))) return rc;
    if ((rc = gemm1(w.UV, 1, n2, B1, n1, 1, w.Zv2, (int)n1, (int)n2, (int)n1, (int)m1, st))) return rc;
    // PT_d = B_d diag(w) B_d^T, partial traces of Sinv
    VGM_LAUNCH1D(vgm_scalecols_kernel, m1 * n1, st, B1, w.wn2, (int)m1, n1, w.B1s);
    VGM_LAUNCH1D(vgm_scalecols_kernel, m2 * n2, st, B2, w.wn1, (int)m2, n2, w.B2s);
    // (short outputs, long reductions: split-K with the assembly's T buffer -- free by now -- as slab scratch)
    if ((rc = gemm_longk(w.B1s, n1, 1, B1, 1, n1, w.PT1, (int)m1, (int)m1, (int)n1, w.T, st))) return rc;
    if ((rc = gemm_longk(w.B2s, n2, 1, B2, 1, n2, w.PT2, (int)m2, (int)m2, (int)n2, w.T, st))) return rc;
    VGM_LAUNCH1D(vgm_ptrace_kernel, m1 * m1, st, w.Sinv, (int)m1, (int)m2, 1, w.PTS1);
    VGM_LAUNCH1D(vgm_ptrace_kernel, m2 * m2, st, w.Sinv, (int)m1, (int)m2, 2, w.PTS2);
    // reductions
    VgmRedArgs ra;
    ra.njobs = RJ_COUNT;
    ra.partial = w.partial;
    auto job = [&](int k, const double* a, const double* b, long n, long sa, long sb, int op, const double* cc = nullptr) {
        ra.job[k] = VgmRedJob{a, b, cc, n, sa, sb, op};
    };
    job(RJ_LOGDET, w.Lg, nullptr, M, M + 1, 0, 1);
    job(RJ_Q, C0, w.a0, M, 1, 1, 0);
    job(RJ_AA, w.a0, w.a0, M, 1, 1, 0);
    job(RJ_TRS, w.Sinv, nullptr, M, M + 1, 0, 2);
    job(RJ_TRPHI, w.nb1, w.wn2, n1, 1, 1, 0);
    job(RJ_MK1PTS, d1.Mk, w.PTS1, m1 * m1, 1, 1, 0);
    job(RJ_TRMK1, d1.Mk, nullptr, m1, m1 + 1, 0, 2);
    job(RJ_SPHI1, w.Sinv, w.Phip, M * M, 1, 1, 0);
    job(RJ_AC1, w.a0, C1, M, 1, 1, 0);
    job(RJ_MKA1, w.MkA1, w.a0, M, 1, 1, 0);
    job(RJ_Z1, W, w.Zb, n1 * n2, 1, 1, 3, w.Zv1);
    job(RJ_HV1, w.hv1, w.wn2, n1, 1, 1, 0);
    job(RJ_MK1PT, d1.Mk, w.PT1, m1 * m1, 1, 1, 0);
    job(RJ_MK2PTS, d2.Mk, w.PTS2, m2 * m2, 1, 1, 0);
    job(RJ_TRMK2, d2.Mk, nullptr, m2, m2 + 1, 0, 2);
    job(RJ_SPHI2, w.Sinv, w.Phip + M * M, M * M, 1, 1, 0);
    job(RJ_AC2, w.a0, C2, M, 1, 1, 0);
    job(RJ_MKA2, w.MkA2, w.a0, M, 1, 1, 0);
    job(RJ_Z2, W, w.Zb, n1 * n2, 1, 1, 3, w.Zv2);
    job(RJ_HV2, w.hv2, w.wn1, n2, 1, 1, 0);
    job(RJ_MK2PT, d2.Mk, w.PT2, m2 * m2, 1, 1, 0);
    hipLaunchKernelGGL(vgm_red_kernel, dim3(VG_MD_NPART, RJ_COUNT), dim3(256), 0, st, ra);
    // scalars that are sums over this rank's grid rows (everything else is replicated): second, tiny collective
    const unsigned local_mask = (1u << RJ_Z1) | (1u << RJ_Z2) | (1u << RJ_HV2) | (1u << RJ_MK2PT);
    const bool multi = c->n_ranks > 1 || c->comm || c->cb;
    hipLaunchKernelGGL(vgm_sum_kernel, dim3(1), dim3(64), 0, st, w.partial, w.scal, local_mask, (!multi || c->rank == 0) ? 1 : 0);
    if ((rc = vg_allreduce(c, w.scal, RJ_COUNT, st))) return rc;
    VgmFinalArgs fa{c->theta, w.scal, w.out, n_obs, yy_obs, (int)m1, (int)m2};
    hipLaunchKernelGGL(vgm_final_kernel, dim3(1), dim3(64), 0, st, fa);
    VG_HIP(hipGetLastError());
    // readback
    VG_HIP(hipMemcpyAsync(c->h_out->out, w.out, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    for (int k = 0; k < 2; ++k) {
        VG_HIP(hipMemcpyAsync(&c->h_out->jitter[k], c->d[k].jitter, sizeof(double), hipMemcpyDeviceToHost, st));
        VG_HIP(hipMemcpyAsync(&c->h_out->status[k], c->d[k].status, sizeof(int), hipMemcpyDeviceToHost, st));
    }
    VG_HIP(hipMemcpyAsync(&c->h_out->counters[1][3], w.cholstatus, sizeof(int), hipMemcpyDeviceToHost, st));
    { const int wrc_ = vg_comm_wait(c, st); if (wrc_) return wrc_; }      // (polls the communicator: a dead peer is an error, not a hang)
    const int hs = c->h_out->counters[1][3];
    *elbo_out = c->h_out->out[0];
    for (int i = 0; i < 5; ++i) grad_out[i] = c->h_out->out[1 + i];
    int status = c->h_out->status[0] ? c->h_out->status[0] : (c->h_out->status[1] ? c->h_out->status[1] : hs);
    if (info) {
        info->jitter1 = c->h_out->jitter[0]; info->jitter2 = c->h_out->jitter[1];
        info->sweeps1 = info->sweeps2 = info->rounds1 = info->rounds2 = 0;
        info->status = status; info->polished = 0;
    }
    if (status) { vg_set_error("masked step: a factor is not positive definite"); return VGGP_ENOTPD; }
    c->have_masked = true;
    return VGGP_OK;
}

// ---- scattered observations ----------------------------------------------------------------------------------------------
// N points (x1_k, x2_k, y_k) that form no grid (satellite tracks: the reference's notebooks 6 / 61 / 7 feed them to the same
// _elbo(), kronecker_structure.py:249-278, whose _Kuf(x) :808-823 takes any x).  Kuf[:, k] = a1(x1_k) (x) a2(x2_k): with the
// per-dimension factors evaluated AT THE POINTS (B_d is m_d x N), Phi~0 = sum_k (b1_k (x) b2_k)(b1_k (x) b2_k)^T is a single
// GEMM over the points between the row-pair products, R[(i1,k1),(i2,k2)] = sum_k (B1[i1,k] B1[k1,k]) (B2[i2,k] B2[k2,k]);
// everything after the assembly -- dense factorisation of Sigma~ = I + rho Phi~0, a0, the gradient's scalars, the final
// combination, the read-outs -- is the masked step's, with sums over observed grid points replaced by sums over the points.
// Specification: oracle/kron.py elbo_step_scattered (== the literal dense restatement to 1e-14).  Several ranks: each plans ITS
// points (desc.n_total = the number of points over all ranks, yy = the global sum of squares) -- two all-reduces, see below.
extern "C" int vggp_elbo_step_scattered(vggp_ctx* c, const double* y, double yy, const double theta[5], double* elbo_out,
                                        double grad_out[5], vggp_info* info, void* stream) {
    if (!c || !c->planned) { vg_set_error("vggp_elbo_step_scattered: context not planned"); return VGGP_ESTATE; }
    VG_REQUIRE(y && theta && elbo_out && grad_out, "vggp_elbo_step_scattered: null argument");
    VG_REQUIRE(c->desc.flags & VGGP_FLAG_SCATTERED, "vggp_elbo_step_scattered: plan the context with VGGP_FLAG_SCATTERED");
    c->have_masked = false;
    const long m1 = c->desc.m1, m2 = c->desc.m2, N = c->desc.n1, M = m1 * m2;
    VG_REQUIRE(M <= VGM_MAX_M, "vggp_elbo_step_scattered: M = m1*m2 = %ld too large for the dense solver (<= %d)", M, VGM_MAX_M);
    VG_REQUIRE(m1 * m1 * N < (1L << 31) && m2 * m2 * N < (1L << 31) && M * M < (1L << 31) * 4, "scattered problem too large");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    for (int i = 0; i < 5; ++i) {
        VG_REQUIRE(theta[i] > 0.0 && std::isfinite(theta[i]), "theta[%d]=%g must be positive and finite", i, theta[i]);
        c->h_theta[i] = theta[i];
    }
    int rc = vgm_prepare(c);
    if (rc) return rc;
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    if ((rc = vg_partials_enqueue(c, nullptr, nullptr, st))) return rc;          // factors at the points: B|V, Mk (unit outputscale)
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const double *B1 = d1.BV, *V1 = d1.BV + m1 * N, *B2 = d2.BV, *V2 = d2.BV + m2 * N;
    double *C0 = w.mpay + 2 * m2 * m2, *C1 = C0 + M, *C2 = C1 + M;
    // projections: C0 = B1 diag(y) B2^T, C1 = V1 diag(y) B2^T, C2 = B1 diag(y) V2^T   (m1 x m2, K = N)
    VGM_LAUNCH1D(vgm_scalecols_kernel, m1 * N, st, B1, y, (int)m1, N, w.B1s);
    VGM_LAUNCH1D(vgm_scalecols_kernel, m1 * N, st, V1, y, (int)m1, N, w.UV);           // (UV is free until the a0 stage)
    if ((rc = gemm_longk(w.B1s, N, 1, B2, 1, N, C0, (int)m1, (int)m2, (int)N, w.T, st))) return rc;
    if ((rc = gemm_longk(w.UV, N, 1, B2, 1, N, C1, (int)m1, (int)m2, (int)N, w.T, st))) return rc;
    if ((rc = gemm_longk(w.B1s, N, 1, V2, 1, N, C2, (int)m1, (int)m2, (int)N, w.T, st))) return rc;
    // per-point statistics
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, B1, B1, (int)m1, N, w.nb1);
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, B2, B2, (int)m2, N, w.nb2);
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, V1, B1, (int)m1, N, w.hv1);
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, V2, B2, (int)m2, N, w.hv2);
    // assembly: one GEMM over the points per matrix
    VGM_LAUNCH1D(vgm_pairprod_kernel, m2 * m2 * N, st, B2, B2, (int)m2, (int)m2, N, w.PP2);
    VGM_LAUNCH1D(vgm_pairprod_kernel, m2 * m2 * N, st, B2, V2, (int)m2, (int)m2, N, w.PP2v);
    VGM_LAUNCH1D(vgm_pairprod_kernel, m1 * m1 * N, st, B1, B1, (int)m1, (int)m1, N, w.PP1);
    VGM_LAUNCH1D(vgm_pairprod_kernel, m1 * m1 * N, st, B1, V1, (int)m1, (int)m1, N, w.PP1v);
    VG_HIP(hipGetLastError());
    if ((rc = gemm1(w.PP1, N, 1, w.PP2, 1, N, w.R3, (int)(m2 * m2), (int)(m1 * m1), (int)(m2 * m2), (int)N, st))) return rc;
    if ((rc = gemm1(w.PP1v, N, 1, w.PP2, 1, N, w.R3 + M * M, (int)(m2 * m2), (int)(m1 * m1), (int)(m2 * m2), (int)N, st))) return rc;
    if ((rc = gemm1(w.PP1, N, 1, w.PP2v, 1, N, w.R3 + 2 * M * M, (int)(m2 * m2), (int)(m1 * m1), (int)(m2 * m2), (int)N, st))) return rc;
    // point-sharded job (n_ranks > 1: every rank holds its own points): C0, C1, C2 and the three assembly matrices are partial
    // sums over this rank's points -- ONE all-reduce of 3 M + 3 M^2 doubles, then Sigma~, its factorisation and a0 are replicated
    if ((rc = vg_allreduce(c, C0, 3 * M + 3 * M * M, st))) return rc;
    VGM_LAUNCH1D(vgm_permute_kernel, M * M, st, w.R3, (int)m1, (int)m2, c->theta, 1, w.Sg);
    VGM_LAUNCH1D(vgm_permute_kernel, M * M, st, w.R3 + M * M, (int)m1, (int)m2, c->theta, 0, w.Phip);
    VGM_LAUNCH1D(vgm_permute_kernel, M * M, st, w.R3 + 2 * M * M, (int)m1, (int)m2, c->theta, 0, w.Phip + M * M);
    if ((rc = dense_chol_inverse(c, w, st))) return rc;
    if ((rc = gemm1(w.Sinv, M, 1, C0, 1, 1, w.a0, 1, (int)M, 1, (int)M, st))) return rc;
    if ((rc = gemm1(d1.Mk, m1, 1, w.a0, m2, 1, w.MkA1, (int)m2, (int)m1, (int)m2, (int)m1, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, d2.Mk, m2, 1, w.MkA2, (int)m2, (int)m1, (int)m2, (int)m2, st))) return rc;
    // UB = A0 B2, UV = A0 V2 (m1 x N); zb_k = b1_k^T A0 b2_k, zv1_k = v1_k^T A0 b2_k, zv2_k = b1_k^T A0 v2_k
    if ((rc = gemm1(w.a0, m2, 1, B2, N, 1, w.UB, (int)N, (int)m1, (int)N, (int)m2, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, V2, N, 1, w.UV, (int)N, (int)m1, (int)N, (int)m2, st))) return rc;
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, B1, w.UB, (int)m1, N, w.Zb);
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, V1, w.UB, (int)m1, N, w.Zv1);
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, B1, w.UV, (int)m1, N, w.Zv2);
    // PT_d = B_d diag(|b_other|^2) B_d^T, partial traces of Sinv
    VGM_LAUNCH1D(vgm_scalecols_kernel, m1 * N, st, B1, w.nb2, (int)m1, N, w.B1s);
    VGM_LAUNCH1D(vgm_scalecols_kernel, m2 * N, st, B2, w.nb1, (int)m2, N, w.B2s);
    if ((rc = gemm_longk(w.B1s, N, 1, B1, 1, N, w.PT1, (int)m1, (int)m1, (int)N, w.T, st))) return rc;
    if ((rc = gemm_longk(w.B2s, N, 1, B2, 1, N, w.PT2, (int)m2, (int)m2, (int)N, w.T, st))) return rc;
    VGM_LAUNCH1D(vgm_ptrace_kernel, m1 * m1, st, w.Sinv, (int)m1, (int)m2, 1, w.PTS1);
    VGM_LAUNCH1D(vgm_ptrace_kernel, m2 * m2, st, w.Sinv, (int)m1, (int)m2, 2, w.PTS2);
    VgmRedArgs ra;
    ra.njobs = RJ_COUNT;
    ra.partial = w.partial;
    auto job = [&](int k, const double* a, const double* b, long n, long sa, long sb, int op) {
        ra.job[k] = VgmRedJob{a, b, nullptr, n, sa, sb, op};
    };
    job(RJ_LOGDET, w.Lg, nullptr, M, M + 1, 0, 1);
    job(RJ_Q, C0, w.a0, M, 1, 1, 0);
    job(RJ_AA, w.a0, w.a0, M, 1, 1, 0);
    job(RJ_TRS, w.Sinv, nullptr, M, M + 1, 0, 2);
    job(RJ_TRPHI, w.nb1, w.nb2, N, 1, 1, 0);
    job(RJ_MK1PTS, d1.Mk, w.PTS1, m1 * m1, 1, 1, 0);
    job(RJ_TRMK1, d1.Mk, nullptr, m1, m1 + 1, 0, 2);
    job(RJ_SPHI1, w.Sinv, w.Phip, M * M, 1, 1, 0);
    job(RJ_AC1, w.a0, C1, M, 1, 1, 0);
    job(RJ_MKA1, w.MkA1, w.a0, M, 1, 1, 0);
    job(RJ_Z1, w.Zb, w.Zv1, N, 1, 1, 0);
    job(RJ_HV1, w.hv1, w.nb2, N, 1, 1, 0);
    job(RJ_MK1PT, d1.Mk, w.PT1, m1 * m1, 1, 1, 0);
    job(RJ_MK2PTS, d2.Mk, w.PTS2, m2 * m2, 1, 1, 0);
    job(RJ_TRMK2, d2.Mk, nullptr, m2, m2 + 1, 0, 2);
    job(RJ_SPHI2, w.Sinv, w.Phip + M * M, M * M, 1, 1, 0);
    job(RJ_AC2, w.a0, C2, M, 1, 1, 0);
    job(RJ_MKA2, w.MkA2, w.a0, M, 1, 1, 0);
    job(RJ_Z2, w.Zb, w.Zv2, N, 1, 1, 0);
    job(RJ_HV2, w.hv2, w.nb1, N, 1, 1, 0);
    job(RJ_MK2PT, d2.Mk, w.PT2, m2 * m2, 1, 1, 0);
    hipLaunchKernelGGL(vgm_red_kernel, dim3(VG_MD_NPART, RJ_COUNT), dim3(256), 0, st, ra);
    // scalars that are sums over this rank's points (everything else is replicated): second, tiny collective
    const unsigned local_mask = (1u << RJ_TRPHI) | (1u << RJ_Z1) | (1u << RJ_HV1) | (1u << RJ_MK1PT) | (1u << RJ_Z2) | (1u << RJ_HV2) |
                                (1u << RJ_MK2PT);
    const bool multi = c->n_ranks > 1 || c->comm || c->cb;
    hipLaunchKernelGGL(vgm_sum_kernel, dim3(1), dim3(64), 0, st, w.partial, w.scal, local_mask, (!multi || c->rank == 0) ? 1 : 0);
    if ((rc = vg_allreduce(c, w.scal, RJ_COUNT, st))) return rc;
    VgmFinalArgs fa{c->theta, w.scal, w.out, (double)(c->desc.n_total > N ? c->desc.n_total : N), yy, (int)m1, (int)m2};
    hipLaunchKernelGGL(vgm_final_kernel, dim3(1), dim3(64), 0, st, fa);
    VG_HIP(hipGetLastError());
    VG_HIP(hipMemcpyAsync(c->h_out->out, w.out, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    for (int k = 0; k < 2; ++k) {
        VG_HIP(hipMemcpyAsync(&c->h_out->jitter[k], c->d[k].jitter, sizeof(double), hipMemcpyDeviceToHost, st));
        VG_HIP(hipMemcpyAsync(&c->h_out->status[k], c->d[k].status, sizeof(int), hipMemcpyDeviceToHost, st));
    }
    VG_HIP(hipMemcpyAsync(&c->h_out->counters[1][3], w.cholstatus, sizeof(int), hipMemcpyDeviceToHost, st));
    { const int wrc_ = vg_comm_wait(c, st); if (wrc_) return wrc_; }      // (polls the communicator: a dead peer is an error, not a hang)
    const int hs = c->h_out->counters[1][3];
    *elbo_out = c->h_out->out[0];
    for (int i = 0; i < 5; ++i) grad_out[i] = c->h_out->out[1 + i];
    const int status = c->h_out->status[0] ? c->h_out->status[0] : (c->h_out->status[1] ? c->h_out->status[1] : hs);
    if (info) {
        info->jitter1 = c->h_out->jitter[0]; info->jitter2 = c->h_out->jitter[1];
        info->sweeps1 = info->sweeps2 = info->rounds1 = info->rounds2 = 0;
        info->status = status; info->polished = 0;
    }
    if (status) { vg_set_error("scattered step: a factor is not positive definite"); return VGGP_ENOTPD; }
    c->have_masked = true;
    return VGGP_OK;
}

// ---- gradient of the scattered ELBO w.r.t. the inducing coordinates (SVGP's trainable Z on along-track data) -------------------
// G_B1[i1][k] = -rho sum_i2 U[(i1,i2)][k] B2[i2][k] + w_k UB1[i1][k] + (s1 s2 / v) |b2_k|^2 B1[i1][k],  U = Sigma~^-1 (B1 kr B2),
// w_k = (s1 s2 / v^2)(y_k - rho b1_k^T A0 b2_k), UB1 = A0 B2 (dimension 2: the mirror image, `other` = B1, UBs = A0^T B1).
__global__ void vgm_zg_kernel(const double* U, const double* Bs, const double* Bo, const double* UBs, const double* y, const double* zb,
                              const double* nbo, const double* theta, int ms, int mo, long N, int dim, double* G) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)ms * N) return;
    const int i = (int)(idx / N);
    const long k = idx - (long)i * N;
    const double s12 = theta[2] * theta[3], v = theta[4], rho = s12 / v;
    double u = 0.0;
    if (dim == 0) for (int j = 0; j < mo; ++j) u += U[((long)i * mo + j) * N + k] * Bo[(long)j * N + k];       // rows (i, j) of U
    else          for (int j = 0; j < mo; ++j) u += U[((long)j * ms + i) * N + k] * Bo[(long)j * N + k];       // rows (j, i)
    G[idx] = -rho * u + (s12 / (v * v)) * (y[k] - rho * zb[k]) * UBs[idx] + (s12 / v) * nbo[k] * Bs[idx];
}
__global__ void vgm_scal_kernel(double* x, long n, double a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= a;
}

// dELBO/dz after vggp_elbo_step_scattered on the same y (oracle/kron.py z_grad_scattered; the reference: autograd through _elbo()
// into the Z Parameter of its SVGP classes, kronecker_structure.py:303-304).  The ELBO is a function of A0^T K0^-1 A0 alone, so
// Abar = L0^-T G_B and Kbar = -1/2 L0^-T (G_B B^T) L0^-1; the contraction with d kappa / d z is vggp_zgrad's.  One extra
// M x M x N product (U) on top of the step's three; workspace 2 M N doubles.
extern "C" int vggp_zgrad_scattered(vggp_ctx* c, const double* y, double* gz1, double* gz2, void* stream) {
    if (!c || !c->have_masked || !c->masked || !(c->desc.flags & VGGP_FLAG_SCATTERED)) {
        vg_set_error("vggp_zgrad_scattered: no finished scattered step");
        return VGGP_ESTATE;
    }
    VG_REQUIRE(y && gz1 && gz2, "vggp_zgrad_scattered: null argument");
    // point-sharded job: every term is a sum over points (Kbar through G_B B^T), so each rank contracts its own points and ONE
    // all-reduce of m1 + m2 doubles adds the parts
    const bool multi = c->n_ranks > 1 || c->comm || c->cb;
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    const long m1 = w.m1, m2 = w.m2, M = w.M, N = c->desc.n1;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const bool pts[2] = {d1.basis == VGGP_BASIS_POINTS, d2.basis == VGGP_BASIS_POINTS};
    if (!pts[0]) VG_HIP(hipMemsetAsync(gz1, 0, sizeof(double) * m1, st));
    if (!pts[1]) VG_HIP(hipMemsetAsync(gz2, 0, sizeof(double) * m2, st));
    if (!pts[0] && !pts[1]) return VGGP_OK;
    VG_REQUIRE(M * N < (1L << 31), "vggp_zgrad_scattered: M N = %ld too large", M * N);
    int rc = vg_ensure_misc(c, sizeof(double) * (size_t)(2 * M * N + 2 * (m1 + m2) * N + 3 * N + m1 * m1 + m2 * m2 + m1 + m2 + 64));
    if (rc) return rc;
    double* p = reinterpret_cast<double*>(c->misc);
    double *Zt = p; p += M * N;
    double *U = p; p += M * N;
    double *G[2], *UBs[2];
    G[0] = p; p += m1 * N; G[1] = p; p += m2 * N; UBs[0] = p; p += m1 * N; UBs[1] = p; p += m2 * N;
    double *zb = p; p += N;
    double *nb1 = p; p += N;
    double *nb2 = p; p += N;
    double* WM[2] = {p, p + m1 * m1};
    double* gzbuf = p + m1 * m1 + m2 * m2;
    const double *B1 = d1.BV, *B2 = d2.BV;
    VGM_LAUNCH1D(vgm_pairprod_kernel, M * N, st, B1, B2, (int)m1, (int)m2, N, Zt);
    VG_HIP(hipGetLastError());
    if ((rc = gemm1(w.Sinv, M, 1, Zt, N, 1, U, (int)N, (int)M, (int)N, (int)M, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, B2, N, 1, UBs[0], (int)N, (int)m1, (int)N, (int)m2, st))) return rc;       // A0 B2
    if ((rc = gemm1(w.a0, 1, m2, B1, N, 1, UBs[1], (int)N, (int)m2, (int)N, (int)m1, st))) return rc;       // A0^T B1
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, B1, UBs[0], (int)m1, N, zb);
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, B1, B1, (int)m1, N, nb1);
    VGM_LAUNCH1D(vgm_coldot_kernel, N, st, B2, B2, (int)m2, N, nb2);
    VGM_LAUNCH1D(vgm_zg_kernel, m1 * N, st, U, B1, B2, UBs[0], y, zb, nb2, c->theta, (int)m1, (int)m2, N, 0, G[0]);
    VGM_LAUNCH1D(vgm_zg_kernel, m2 * N, st, U, B2, B1, UBs[1], y, zb, nb1, c->theta, (int)m2, (int)m1, N, 1, G[1]);
    VG_HIP(hipGetLastError());
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        if (!pts[k]) continue;
        if ((rc = gemm_longk(G[k], N, 1, d.BV, 1, N, WM[k], d.m, d.m, (int)N, w.T, st))) return rc;          // G_B B^T
        VGM_LAUNCH1D(vgm_scal_kernel, (long)d.m * d.m, st, WM[k], (long)d.m * d.m, -0.5);
    }
    for (int pass = 0; pass < 2; ++pass) {
        VgTrsmSpec q[4];
        int nq = 0;
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            if (!pts[k]) continue;
            const bool dv = d.m <= VG_TRSM_BLK && d.Dinv0 && c->dinv_valid;
            const double* dp = dv ? d.Dinv0 : d.Linv0;
            const long blk = dv ? 256 : 16L * d.m + 16, dld = dv ? 16 : d.m;
            if (pass == 0) {
                q[nq++] = VgTrsmSpec{d.L0, d.m, dp, blk, dld, G[k], N, 1, N, d.m, 1};                       // Abar = L0^-T G_B
                q[nq++] = VgTrsmSpec{d.L0, d.m, dp, blk, dld, WM[k], d.m, 1, d.m, d.m, 1};                  // L0^-T W_M
            } else {
                q[nq++] = VgTrsmSpec{d.L0, d.m, dp, blk, dld, WM[k], 1, d.m, d.m, d.m, 1};                  // (.) L0^-1
            }
        }
        if ((rc = vg_trsm_batch(q, nq, st))) return rc;
    }
    double *o1 = multi ? gzbuf : gz1, *o2 = multi ? gzbuf + m1 : gz2;
    if (multi) VG_HIP(hipMemsetAsync(gzbuf, 0, sizeof(double) * (m1 + m2), st));
    if (pts[0]) VG_HIP(vg_zdot_launch(c->theta, 0, d1.grid, d1.x, (int)m1, N, G[0], d1.AD + m1 * N, WM[0], d1.dK0, o1, st));
    if (pts[1]) VG_HIP(vg_zdot_launch(c->theta, 1, d2.grid, d2.x, (int)m2, N, G[1], d2.AD + m2 * N, WM[1], d2.dK0, o2, st));
    if (multi) {
        if ((rc = vg_allreduce(c, gzbuf, m1 + m2, st))) return rc;
        VG_HIP(hipMemcpyAsync(gz1, gzbuf, sizeof(double) * m1, hipMemcpyDeviceToDevice, st));
        VG_HIP(hipMemcpyAsync(gz2, gzbuf + m1, sizeof(double) * m2, hipMemcpyDeviceToDevice, st));
    }
    { const int wrc_ = vg_comm_wait(c, st); if (wrc_) return wrc_; }      // (polls the communicator: a dead peer is an error, not a hang)
    return VGGP_OK;
}

// q(v) of the last masked step: mean = (s1 s2 / v) L1 A0 L2^T, diag cov = s1 s2 rowdot((L1 (x) L2) Sinv, L1 (x) L2)
extern "C" int vggp_qv_masked(vggp_ctx* c, double* mean, double* var, void* stream) {
    if (!c || !c->have_masked || !c->masked) { vg_set_error("vggp_qv_masked: no finished masked step"); return VGGP_ESTATE; }
    VG_REQUIRE(mean && var, "vggp_qv_masked: null output");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    const long m1 = w.m1, m2 = w.m2, M = w.M;
    int rc;
    if ((rc = gemm1(c->d[0].L0, m1, 1, w.a0, m2, 1, w.MkA1, (int)m2, (int)m1, (int)m2, (int)m1, st))) return rc;      // L1 A0
    if ((rc = gemm1(w.MkA1, m2, 1, c->d[1].L0, 1, m2, mean, (int)m2, (int)m1, (int)m2, (int)m2, st))) return rc;     // . L2^T
    const int e1 = (c->d[0].basis == VGGP_BASIS_VFF || c->d[0].basis == VGGP_BASIS_B1) ? -1 : 1;
    const int e2 = (c->d[1].basis == VGGP_BASIS_VFF || c->d[1].basis == VGGP_BASIS_B1) ? -1 : 1;
    VGM_LAUNCH1D(vgm_scale_rho_kernel, M, st, mean, M, c->theta, e1, e2);
    VGM_LAUNCH1D(vgm_kron_kernel, M * M, st, c->d[0].L0, c->d[1].L0, (int)m1, (int)m2, w.R);
    if ((rc = gemm1(w.R, M, 1, w.Sinv, M, 1, w.Sg, (int)M, (int)M, (int)M, (int)M, st))) return rc;
    VGM_LAUNCH1D(vgm_rowdot_scale_kernel, M, st, w.Sg, w.R, M, c->theta, var, e1, e2);
    VG_HIP(hipGetLastError());
    VG_HIP(hipStreamSynchronize(st));
    return VGGP_OK;
}

// mean[p] = rho * sum_u T[u][p] a0[u] ;  var[p] = s1 s2 (1 - sum_u T[u][p]^2 + sum_u T[u][p] ST[u][p])
__global__ void vgm_post_kernel(const double* T, const double* ST, const double* a0, long M, long cn, const double* theta,
                                double* mean, double* var) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cn) return;
    double lin = 0.0, nrm = 0.0, quad = 0.0;
    for (long u = 0; u < M; ++u) {
        const double t = T[u * cn + p];
        lin += t * a0[u];
        nrm += t * t;
        quad += t * ST[u * cn + p];
    }
    const double ss = theta[2] * theta[3];
    mean[p] = (ss / theta[4]) * lin;
    var[p] = ss * (1.0 - nrm + quad);
}

// gridded read-out: cell p = a * mv2 + b; column p of T1x / T2x is column a of U1 / column b of U2
__global__ void vgm_expand_kernel(const double* U1, const double* U2, int m1, int m2, long mv1, long mv2, long p0, long cn,
                                  double* T1x, double* T2x) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)(m1 + m2) * cn) return;
    const long row = idx / cn, q = idx - row * cn, p = p0 + q;
    const long a = p / mv2, b = p - a * mv2;
    if (row < m1) T1x[row * cn + q] = U1[row * mv1 + a];
    else T2x[(row - m1) * cn + q] = U2[(row - m1) * mv2 + b];
}
// mean = (s1 s2 / v) t^T a0;  var = s1 s2 (kd1_a kd2_b - |t|^2 + quad), quad = t^T Sinv t (conditional) or |Lc^T t|^2 (literal:
// t^T Sigma~ t, i.e. X = S_u^-1 in the reference's expression)
__global__ void vgm_readout_kernel(const double* T, const double* ST, const double* a0, long M, long cn, const double* theta,
                                   const double* kd1, const double* kd2, long mv2, long p0, int literal, double* mean, double* var) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cn) return;
    double lin = 0.0, nrm = 0.0, quad = 0.0;
    for (long u = 0; u < M; ++u) {
        const double t = T[u * cn + q], s = ST[u * cn + q];
        lin += t * a0[u];
        nrm += t * t;
        quad += literal ? s * s : t * s;
    }
    const long p = p0 + q, a = p / mv2, b = p - a * mv2;
    const double ss = theta[2] * theta[3];
    mean[p] = (ss / theta[4]) * lin;
    var[p] = ss * (kd1[a] * kd2[b] - nrm + quad);
}

// Gridded read-out q(v) of B0 cell features from the M-space state of the last masked / scattered step (the Gridded* models of
// gridded_kronecker_structure.py:396-438 / :613-654 / :903-947 on data that is no full grid -- the along-track case of notebook
// 61): arguments and meaning as vggp_readout.  t = (L0_1^-1 C1^T)[:, a] (x) (L0_2^-1 C2^T)[:, b]: the point-wise posterior's algebra
// with the cross-covariances in place of the kernel columns and kd1_a kd2_b in place of the unit prior variance.
extern "C" int vggp_readout_masked(vggp_ctx* c, const double* C1, int64_t mv1, const double* C2, int64_t mv2, const double* kd1,
                                   const double* kd2, double* mean, double* var, int flags, void* stream) {
    if (!c || !c->have_masked || !c->masked) { vg_set_error("vggp_readout_masked: no finished masked / scattered step"); return VGGP_ESTATE; }
    VG_REQUIRE(C1 && C2 && kd1 && kd2 && mean && var && mv1 > 0 && mv2 > 0, "vggp_readout_masked: bad argument");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    const long m1 = w.m1, m2 = w.m2, M = w.M, ns = mv1 * mv2;
    const int literal = (flags & VGGP_READOUT_LITERAL) ? 1 : 0;
    const long chunk = std::min<long>(ns, M);            // T and (Sinv | Lc^T) T live in the two M x M scratch matrices
    int rc = vg_ensure_misc(c, (size_t)(m1 * mv1 + m2 * mv2 + chunk * (m1 + m2)) * sizeof(double));
    if (rc) return rc;
    double* p = (double*)c->misc;
    double* U1 = p; p += m1 * mv1;
    double* U2 = p; p += m2 * mv2;
    double* T1x = p; p += m1 * chunk;
    double* T2x = p;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    VgGemmBatch g;
    vg_gemm_init(&g);                                    // U_d = Linv0_d C_d^T   (m_d x mv_d)
    vg_gemm_add(&g, d1.Linv0, m1, 1, C1, 1, m1, U1, (int)mv1, (int)m1, (int)mv1, (int)m1);
    vg_gemm_add(&g, d2.Linv0, m2, 1, C2, 1, m2, U2, (int)mv2, (int)m2, (int)mv2, (int)m2);
    VG_HIP(vg_gemm_launch(&g, st));
    for (long off = 0; off < ns; off += chunk) {
        const long cn = std::min<long>(chunk, ns - off);
        VGM_LAUNCH1D(vgm_expand_kernel, (m1 + m2) * cn, st, U1, U2, (int)m1, (int)m2, (long)mv1, (long)mv2, off, cn, T1x, T2x);
        VGM_LAUNCH1D(vgm_pairprod_kernel, M * cn, st, T1x, T2x, (int)m1, (int)m2, cn, w.R);
        if (literal) { if ((rc = gemm1(w.Lg, 1, M, w.R, cn, 1, w.Sg, (int)cn, (int)M, (int)cn, (int)M, st))) return rc; }     // Lc^T T
        else if ((rc = gemm1(w.Sinv, M, 1, w.R, cn, 1, w.Sg, (int)cn, (int)M, (int)cn, (int)M, st))) return rc;
        VGM_LAUNCH1D(vgm_readout_kernel, cn, st, w.R, w.Sg, w.a0, M, cn, c->theta, kd1, kd2, (long)mv2, off, literal, mean, var);
    }
    VG_HIP(hipGetLastError());
    VG_HIP(hipStreamSynchronize(st));
    return VGGP_OK;
}

// posterior(x*) of the last masked step (kronecker_structure.py:199-230 on the observed subset):
//   t = (L1^{-1} a1(x*)) (x) (L2^{-1} a2(x*)),  mean = rho t^T a0,  var = s1 s2 (1 - |t|^2 + t^T Sigma~^{-1} t)
extern "C" int vggp_posterior_masked(vggp_ctx* c, const double* xs1, const double* xs2, int64_t ns, double* mean, double* var,
                                     void* stream) {
    if (!c || !c->have_masked || !c->masked) { vg_set_error("vggp_posterior_masked: no finished masked step"); return VGGP_ESTATE; }
    VG_REQUIRE(xs1 && xs2 && mean && var && ns >= 0, "vggp_posterior_masked: bad argument");
    if (ns == 0) return VGGP_OK;
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    const long m1 = w.m1, m2 = w.m2, M = w.M;
    const long chunk = std::min<long>(ns, M);            // T and Sigma~^{-1} T live in the two M x M scratch matrices
    int rc = vg_ensure_misc(c, (size_t)chunk * 2 * (m1 + m2) * sizeof(double));
    if (rc) return rc;
    double* p = (double*)c->misc;
    double* A1 = p; p += m1 * chunk;
    double* B1 = p; p += m1 * chunk;
    double* A2 = p; p += m2 * chunk;
    double* B2 = p;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    for (long off = 0; off < ns; off += chunk) {
        const int cn = (int)std::min<long>(chunk, ns - off);
        VgFactorJob fj[2] = {
            VgFactorJob{xs1 + off, d1.grid, A1, nullptr, nullptr, nullptr, cn, d1.m, d1.kind, d1.basis, 0, 0.0, c->desc.flags},
            VgFactorJob{xs2 + off, d2.grid, A2, nullptr, nullptr, nullptr, cn, d2.m, d2.kind, d2.basis, 1, 0.0, c->desc.flags}};
        VG_HIP(vg_factor_build_launch(fj, 2, c->theta, st));
        VgGemmBatch g;
        vg_gemm_init(&g);
        vg_gemm_add(&g, d1.Linv0, m1, 1, A1, cn, 1, B1, cn, (int)m1, cn, (int)m1);
        vg_gemm_add(&g, d2.Linv0, m2, 1, A2, cn, 1, B2, cn, (int)m2, cn, (int)m2);
        VG_HIP(vg_gemm_launch(&g, st));
        VGM_LAUNCH1D(vgm_pairprod_kernel, M * cn, st, B1, B2, (int)m1, (int)m2, (long)cn, w.R);
        if ((rc = gemm1(w.Sinv, M, 1, w.R, cn, 1, w.Sg, cn, (int)M, cn, (int)M, st))) return rc;
        VGM_LAUNCH1D(vgm_post_kernel, cn, st, w.R, w.Sg, w.a0, M, (long)cn, c->theta, mean + off, var + off);
    }
    VG_HIP(hipGetLastError());
    VG_HIP(hipStreamSynchronize(st));
    return VGGP_OK;
}

hipError_t vg_prior_cov_launch(const double* xs1, const double* xs2, long ns, int kind1, int kind2, const double* theta,
                               double* cov, hipStream_t st);

// dense covariance of posterior(x*) of the last masked step: cov = K** + s1 s2 (T^T Sigma~^{-1} T - T^T T), T = (L1^{-1} a1*) (x) (L2^{-1} a2*)
extern "C" int vggp_posterior_cov_masked(vggp_ctx* c, const double* xs1, const double* xs2, int64_t ns, double* cov, void* stream) {
    if (!c || !c->have_masked || !c->masked) { vg_set_error("vggp_posterior_cov_masked: no finished masked step"); return VGGP_ESTATE; }
    VG_REQUIRE(xs1 && xs2 && cov && ns >= 1, "vggp_posterior_cov_masked: bad argument");
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    const long m1 = w.m1, m2 = w.m2, M = w.M;
    VG_REQUIRE(ns <= M, "vggp_posterior_cov_masked: at most M = %ld points per call (the scratch is M x M)", M);
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    int rc = vg_ensure_misc(c, (size_t)ns * 2 * (m1 + m2) * sizeof(double));
    if (rc) return rc;
    double* p = (double*)c->misc;
    double* A1 = p; p += m1 * ns;
    double* B1 = p; p += m1 * ns;
    double* A2 = p; p += m2 * ns;
    double* B2 = p;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const int cn = (int)ns;
    VgFactorJob fj[2] = {
        VgFactorJob{xs1, d1.grid, A1, nullptr, nullptr, nullptr, cn, d1.m, d1.kind, d1.basis, 0, 0.0, c->desc.flags},
        VgFactorJob{xs2, d2.grid, A2, nullptr, nullptr, nullptr, cn, d2.m, d2.kind, d2.basis, 1, 0.0, c->desc.flags}};
    VG_HIP(vg_factor_build_launch(fj, 2, c->theta, st));
    VgGemmBatch g;
    vg_gemm_init(&g);
    vg_gemm_add(&g, d1.Linv0, m1, 1, A1, cn, 1, B1, cn, (int)m1, cn, (int)m1);
    vg_gemm_add(&g, d2.Linv0, m2, 1, A2, cn, 1, B2, cn, (int)m2, cn, (int)m2);
    VG_HIP(vg_gemm_launch(&g, st));
    VGM_LAUNCH1D(vgm_pairprod_kernel, M * cn, st, B1, B2, (int)m1, (int)m2, (long)cn, w.R);                 // T  (M x ns)
    if ((rc = gemm1(w.Sinv, M, 1, w.R, cn, 1, w.Sg, cn, (int)M, cn, (int)M, st))) return rc;                  // Sigma~^{-1} T
    const double ss = c->h_theta[2] * c->h_theta[3];
    if ((rc = gemm1(w.R, 1, cn, w.Sg, cn, 1, cov, cn, cn, cn, (int)M, st, ss, 0))) return rc;                  // + s1 s2 T^T Sigma~^{-1} T
    if ((rc = gemm1(w.R, 1, cn, w.R, cn, 1, cov, cn, cn, cn, (int)M, st, -ss, 1))) return rc;                  // - s1 s2 T^T T
    VG_HIP(vg_prior_cov_launch(xs1, xs2, ns, d1.kind, d2.kind, c->theta, cov, st));
    VG_HIP(hipStreamSynchronize(st));
    return VGGP_OK;
}

// dense M x M covariance of q(v) of the last masked step: S = s1^e1 s2^e2 (L1 (x) L2) Sigma~^{-1} (L1 (x) L2)^T
__global__ void vgm_scale_e_kernel(double* x, long n, const double* theta, int e1, int e2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= (e1 > 0 ? theta[2] : 1.0 / theta[2]) * (e2 > 0 ? theta[3] : 1.0 / theta[3]);
}
extern "C" int vggp_qv_cov_masked(vggp_ctx* c, double* cov, void* stream) {
    if (!c || !c->have_masked || !c->masked) { vg_set_error("vggp_qv_cov_masked: no finished masked step"); return VGGP_ESTATE; }
    VG_REQUIRE(cov, "vggp_qv_cov_masked: null output");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    const long m1 = w.m1, m2 = w.m2, M = w.M;
    int rc;
    const int e1 = (c->d[0].basis == VGGP_BASIS_VFF || c->d[0].basis == VGGP_BASIS_B1) ? -1 : 1;
    const int e2 = (c->d[1].basis == VGGP_BASIS_VFF || c->d[1].basis == VGGP_BASIS_B1) ? -1 : 1;
    VGM_LAUNCH1D(vgm_kron_kernel, M * M, st, c->d[0].L0, c->d[1].L0, (int)m1, (int)m2, w.R);
    if ((rc = gemm1(w.R, M, 1, w.Sinv, M, 1, w.Sg, (int)M, (int)M, (int)M, (int)M, st))) return rc;
    if ((rc = gemm1(w.Sg, M, 1, w.R, 1, M, cov, (int)M, (int)M, (int)M, (int)M, st))) return rc;
    VGM_LAUNCH1D(vgm_scale_e_kernel, M * M, st, cov, M * M, c->theta, e1, e2);
    VG_HIP(hipGetLastError());
    VG_HIP(hipStreamSynchronize(st));
    return VGGP_OK;
}
