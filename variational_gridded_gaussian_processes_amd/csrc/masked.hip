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
    bool iter = false;             // iterative step (vggp_elbo_step_masked_iter): none of the M x M / pair-product buffers exist
    void* imem = nullptr;          // its own workspace (VgIter, below)
    size_t ibytes = 0;
    // the iterative step's kept preconditioner basis: valid, steps since the cold solve, PCG iterations right after it / last step
    bool ib_valid = false, ib_fresh = false;
    int ib_age = 0, ib_ref_its = 0, ib_last_its = 0, ib_nbc = 0, ib_maxit = 0;
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
    const size_t M = w.M, MM = w.iter ? 0 : M * M, n1 = w.n1, n2 = w.n2, m1 = w.m1, m2 = w.m2;
    const size_t mmax2 = m1 > m2 ? m1 * m1 : m2 * m2;
    if (!w.iter) {
        w.Sg = take(MM); w.Lg = take(MM); w.Xg = take(MM); w.Sinv = take(MM); w.R = take(MM); w.Phip = take(2 * MM);
        w.DI = take((size_t)w.nblk * VG_MB * VG_MB); w.Tmp = take((size_t)VG_MB * M);
        w.PP1 = take(m1 * m1 * n1); w.PP1v = take(m1 * m1 * n1); w.PP2 = take(m2 * m2 * n2); w.PP2v = take(m2 * m2 * n2);
    }
    const size_t grid = w.scattered ? n1 : n1 * n2;            // scattered: zb, zv1, zv2 are per-point vectors
    // T: the assembly's n1 x m2^2 buffer, later slab scratch of the long-K products (scattered / iterative: only the latter)
    w.T = take((w.scattered || w.iter) ? std::max<size_t>(256 * mmax2, 64 * n1) : n1 * m2 * m2);
    w.Tv = take((w.scattered || w.iter) ? 256 : n1 * m2 * m2);
    w.UB = take(m1 * n2); w.UV = take(m1 * n2); w.Zb = take(grid); w.Zv1 = take(grid); w.Zv2 = take(grid);
    w.B1s = take(m1 * n1); w.B2s = take(m2 * n2);
    w.nb1 = take(n1); w.nb2 = take(n2); w.hv1 = take(n1); w.hv2 = take(n2); w.wn1 = take(n2);
    const size_t wn2len = w.scattered ? 0 : n1;                 // (scattered: the all-reduce payload is C, C1, C2 | R3, the same
    w.mpay = take(2 * m2 * m2 + 3 * M + wn2len + 3 * MM + 8);   //  length on every rank whatever its number of points)
    w.wn2 = base ? w.mpay + 2 * m2 * m2 + 3 * M : nullptr;
    w.R3 = base ? w.wn2 + wn2len : nullptr;
    w.scal = take(32);
    w.PT1 = take(m1 * m1); w.PT2 = take(m2 * m2); w.PTS1 = take(m1 * m1); w.PTS2 = take(m2 * m2);
    w.MkA1 = take(M); w.MkA2 = take(M); w.a0 = take(M);
    w.partial = take(VG_MD_MAXJOBS * VG_MD_NPART); w.out = take(8);
    w.cholscratch = take(VG_MB * (VG_MB + 1)); w.choljit = take(8);
    w.cholstatus = reinterpret_cast<int*>(take(8));
}

static int vgm_prepare(vggp_ctx* c, bool iter = false) {
    VgMasked* w = reinterpret_cast<VgMasked*>(c->masked);
    if (!w) { w = new VgMasked(); c->masked = w; }
    const long m1 = c->desc.m1, m2 = c->desc.m2;
    const bool sc = (c->desc.flags & VGGP_FLAG_SCATTERED) != 0;
    if (w->mem && w->M == m1 * m2 && w->n1 == c->desc.n1 && w->n2 == c->desc.n2 && w->m1 == m1 && w->scattered == sc && w->iter == iter)
        return VGGP_OK;
    if (w->mem) { VG_HIP(hipFree(w->mem)); w->mem = nullptr; }
    if (w->imem) { VG_HIP(hipFree(w->imem)); w->imem = nullptr; w->ibytes = 0; }
    w->ib_valid = false;
    w->M = m1 * m2; w->m1 = (int)m1; w->m2 = (int)m2; w->n1 = c->desc.n1; w->n2 = c->desc.n2; w->scattered = sc; w->iter = iter;
    w->nblk = (int)((w->M + VG_MB - 1) / VG_MB);
    size_t off = 0;
    vgm_layout(*w, nullptr, off);
    w->bytes = off + 4096;
    {   // say what does not fit instead of failing inside hipMalloc (the scattered / masked assemblies are not chunked over N)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && w->bytes > free_b) {
            vg_set_error("the dense M-space workspace of this %s step needs %.1f GiB (M = %ld, n1 = %ld, n2 = %ld, m1 = %ld, m2 = %ld; "
                         "~10 M^2 + (m1^2 + m2^2) N doubles) but only %.1f GiB of device memory are free%s",
                         sc ? "scattered" : "masked", (double)w->bytes / 1073741824.0, (long)(m1 * m2), (long)c->desc.n1, (long)c->desc.n2,
                         (long)m1, (long)m2, (double)free_b / 1073741824.0,
                         (sc || iter) ? "" : "; vggp_elbo_step_masked_iter needs no M x M matrix");
            w->M = 0;
            return VGGP_ENOMEM;
        }
    }
    VG_HIP(hipMalloc(&w->mem, w->bytes));
    VG_HIP(hipMemset(w->mem, 0, w->bytes));
    off = 0;
    vgm_layout(*w, reinterpret_cast<char*>(w->mem), off);
    return VGGP_OK;
}

void vg_masked_new_plan(vggp_ctx* c) {          // vggp_plan: the iterative step's kept preconditioner basis belongs to the old plan
    VgMasked* w = reinterpret_cast<VgMasked*>(c->masked);
    if (w) w->ib_valid = false;
}

void vg_masked_free(vggp_ctx* c) {
    VgMasked* w = reinterpret_cast<VgMasked*>(c->masked);
    if (!w) return;
    if (w->mem) (void)hipFree(w->mem);
    if (w->imem) (void)hipFree(w->imem);
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

// ---- pieces shared by the dense and the iterative masked step ----------------------------------------------------------------
// column statistics of the whitened factors: |b_i|^2, v_i . b_i, and their mask-weighted sums along the other axis
static int vgm_colstats(vggp_ctx* c, VgMasked& w, const double* W, hipStream_t st) {
    const long m1 = c->desc.m1, m2 = c->desc.m2, n1 = c->desc.n1, n2 = c->desc.n2;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const double *B1 = d1.BV, *V1 = d1.BV + m1 * n1, *B2 = d2.BV, *V2 = d2.BV + m2 * n2;
    VGM_LAUNCH1D(vgm_coldot_kernel, n1, st, B1, B1, (int)m1, n1, w.nb1);
    VGM_LAUNCH1D(vgm_coldot_kernel, n2, st, B2, B2, (int)m2, n2, w.nb2);
    VGM_LAUNCH1D(vgm_coldot_kernel, n1, st, V1, B1, (int)m1, n1, w.hv1);
    VGM_LAUNCH1D(vgm_coldot_kernel, n2, st, V2, B2, (int)m2, n2, w.hv2);
    {
        const long mm = m2 * m2;                                // w.T (written by the assembly later / slab scratch) is the scratch
        const int nslab = (int)(mm < 64 ? mm : 64);
        hipLaunchKernelGGL(vgm_wcol_part_kernel, dim3((unsigned)((n1 + 255) / 256), (unsigned)nslab), dim3(256), 0, st, W, w.nb2,
                           n1, n2, nslab, w.T);
        VGM_LAUNCH1D(vgm_wcol_sum_kernel, n1, st, w.T, n1, nslab, w.wn2);           // partial over this rank's rows
    }
    hipLaunchKernelGGL(vgm_wrow_kernel, dim3((unsigned)n2), dim3(256), 0, st, W, w.nb1, n1, n2, w.wn1);
    VG_HIP(hipGetLastError());
    return VGGP_OK;
}
// everything of the gradient that needs a0 = Sigma~^-1 c0 only (w.a0 as an m1 x m2 matrix) and the two PT matrices
static int vgm_a0_terms(vggp_ctx* c, VgMasked& w, hipStream_t st) {
    const long m1 = c->desc.m1, m2 = c->desc.m2, n1 = c->desc.n1, n2 = c->desc.n2;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const double *B1 = d1.BV, *V1 = d1.BV + m1 * n1, *B2 = d2.BV, *V2 = d2.BV + m2 * n2;
    int rc;
    // Mk1 A0, A0 Mk2 ; UB = A0 B2, UV = A0 V2 ; Zb^T = UB^T B1, Zv1^T = UB^T V1, Zv2^T = UV^T B1   ([n2][n1] like W)
    if ((rc = gemm1(d1.Mk, m1, 1, w.a0, m2, 1, w.MkA1, (int)m2, (int)m1, (int)m2, (int)m1, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, d2.Mk, m2, 1, w.MkA2, (int)m2, (int)m1, (int)m2, (int)m2, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, B2, n2, 1, w.UB, (int)n2, (int)m1, (int)n2, (int)m2, st))) return rc;
    if ((rc = gemm1(w.a0, m2, 1, V2, n2, 1, w.UV, (int)n2, (int)m1, (int)n2, (int)m2, st))) return rc;
    if ((rc = gemm1(w.UB, 1, n2, B1, n1, 1, w.Zb, (int)n1, (int)n2, (int)n1, (int)m1, st))) return rc;
    if ((rc = gemm1(w.UB, 1, n2, V1, n1, 1, w.Zv1, (int)n1, (int)n2, (int)n1, (int)m1, st))) return rc;
    if ((rc = gemm1(w.UV, 1, n2, B1, n1, 1, w.Zv2, (int)n1, (int)n2, (int)n1, (int)m1, st))) return rc;
    // PT_d = B_d diag(w) B_d^T
    VGM_LAUNCH1D(vgm_scalecols_kernel, m1 * n1, st, B1, w.wn2, (int)m1, n1, w.B1s);
    VGM_LAUNCH1D(vgm_scalecols_kernel, m2 * n2, st, B2, w.wn1, (int)m2, n2, w.B2s);
    // (short outputs, long reductions: split-K with the T buffer -- free by now -- as slab scratch)
    if ((rc = gemm_longk(w.B1s, n1, 1, B1, 1, n1, w.PT1, (int)m1, (int)m1, (int)n1, w.T, st))) return rc;
    if ((rc = gemm_longk(w.B2s, n2, 1, B2, 1, n2, w.PT2, (int)m2, (int)m2, (int)n2, w.T, st))) return rc;
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

    if ((rc = vgm_colstats(c, w, W, st))) return rc;
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
    if ((rc = vgm_a0_terms(c, w, st))) return rc;
    // partial traces of Sinv
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

// ================================================================================================================================
// Iterative masked step: what gpytorch does for the reference above max_cholesky_size = 800 --
// `inv_matmul` by conjugate gradients, `log_prob` by stochastic Lanczos quadrature (kronecker_structure.py:269, :273) -- rebuilt for
// the Kronecker structure, preconditioned and with FIXED probes, so that no M x M matrix exists and M = m1 m2 is not limited by
// O(M^2) memory or O(M^3) time.  Specification: oracle/kron.py elbo_step_masked_iter (tolerances against the dense step: ELBO 1e-5,
// gradient 1e-4 of its largest component).
//
//   operator        Sigma~ V = V + rho B1 (W^T o (B1^T V B2)) B2^T          V: m1 x m2; four MFMA GEMMs + one mask pass over n1 x n2
//   block vectors   nbc = 1 + n_probes right-hand sides, stored interleaved [m1][nbc][m2]: then every GEMM of the operator is ONE
//                   plain GEMM for the whole block ([m1][(nbc m2)] as B operand, [(m1 nbc)][m2] / [(n1 nbc)][.] as A operand)
//   preconditioner  P = I + rho p G1 (x) G2 (p = observed fraction): E[Phi~] for an unstructured mask, diagonal in the Kronecker
//                   eigenbasis Q1 (x) Q2 of the full-grid path (the step's own eigensolver), applied by four m x m x m GEMMs
//   log|Sigma~|     log|P| + mean_z M e1^T log(T_z) e1, T_z = Lanczos tridiagonal of P^-1/2 Sigma~ P^-1/2 from the PCG coefficients of
//                   the probe z = P^1/2 z0, z0 Rademacher from a counter-based hash (bitwise reproducible)
//   traces          tr(Sigma~^-1 D) = tr(P^-1 D) [closed form, two GEMMs over the mask] + mean_z (Sigma~^-1 z - P^-1 z)^T D P^-1 z
// ================================================================================================================================
struct VgIter {
    int nbc = 0, maxit = 0;
    double *Wt;                                   // [n1][n2] mask, transposed once per step
    double *X, *R, *Zp, *Pd, *AP, *Wz, *Tm, *Tm2; // block vectors [m1][nbc][m2]
    double *T1;                                   // [n1][nbc][max(m1, m2)]
    double *F0, *F1, *F2;                         // fields [n1][nbc][n2]
    double *alh, *beh;                            // [maxit][nbc] PCG coefficients
    double *col;                                  // per-column scalars: [0] rz, [1] pAp, [2] r0^2, [3] rr, [4] alpha, [5] beta, [6] active, [7] k
    double *Rr1, *Rr2, *Rv1, *Rv2, *RR1, *RRV1, *RR2, *RRV2, *TW, *TWv, *Ex;   // rotated factors and the exact traces' temporaries
    double *dg1, *dg2, *Tq;                       // diag(Q^T Mk Q), m x m temporary
    double *Qk1, *Qk2;                            // the preconditioner's eigenbasis of the last cold solve (rows = eigenvectors), kept across steps
    double *ts;                                   // [32] scalars
    int* nact;                                    // device word: number of active columns
    int* h_nact;                                  // pinned
};

__device__ __forceinline__ unsigned long long vgi_mix(unsigned long long x) {       // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
// block[a][c][b] = +-1 for probe columns c >= 1 (counter-based: the same probes on every device, every run), 0 for column 0
__global__ void vgi_probe_kernel(double* blk, int m1, int nbc, int m2, unsigned long long seed) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)m1 * nbc * m2) return;
    const int b = (int)(idx % m2), c = (int)((idx / m2) % nbc), a = (int)(idx / ((long)m2 * nbc));
    const unsigned long long h = vgi_mix(seed ^ vgi_mix(((unsigned long long)c << 40) ^ ((unsigned long long)a * m2 + b)));
    blk[idx] = c == 0 ? 0.0 : ((h >> 17) & 1ULL ? 1.0 : -1.0);
}
// out[a][c][b] = in[a][c][b] * f(dP[a][b]),  dP = 1 + rho p max(lam1,0) max(lam2,0);  mode 0: 1/dP, 1: sqrt(dP), 2: 1/sqrt(dP)
__global__ void vgi_scale_kernel(const double* in, double* out, const double* lam1, const double* lam2, const double* theta, double p,
                                 int m1, int nbc, int m2, int mode, int c_from) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)m1 * nbc * m2) return;
    const int b = (int)(idx % m2), c = (int)((idx / m2) % nbc), a = (int)(idx / ((long)m2 * nbc));
    const double rho = theta[2] * theta[3] / theta[4];
    const double d = 1.0 + rho * p * fmax(lam1[a], 0.0) * fmax(lam2[b], 0.0);
    const double f = mode == 0 ? 1.0 / d : (mode == 1 ? sqrt(d) : 1.0 / sqrt(d));
    out[idx] = c >= c_from ? in[idx] * f : 0.0;
}
__global__ void vgi_transpose_kernel(const double* W, long n1, long n2, double* Wt) {      // Wt[i][j] = W[j][i]
    __shared__ double tile[32][33];
    const long i0 = (long)blockIdx.x * 32, j0 = (long)blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const long j = j0 + r, i = i0 + threadIdx.x;
        tile[r][threadIdx.x] = (j < n2 && i < n1) ? W[j * n1 + i] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const long i = i0 + r, j = j0 + threadIdx.x;
        if (i < n1 && j < n2) Wt[i * n2 + j] = tile[threadIdx.x][r];
    }
}
// F[i][c][j] *= Wt[i][j]
__global__ void vgi_mask_kernel(double* F, const double* Wt, long n1, int nbc, long n2) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n1 * nbc * n2) return;
    const long j = idx % n2, i = idx / (n2 * nbc);
    F[idx] *= Wt[i * n2 + j];
}
// column c of the block vector <- the m1 x m2 matrix src (mode 0) / the matrix dst <- column c (mode 1)
__global__ void vgi_col_kernel(double* blk, double* mat, int m1, int nbc, int m2, int c, int mode) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)m1 * m2) return;
    const int a = (int)(idx / m2), b = (int)(idx - (long)a * m2);
    double* e = blk + ((long)a * nbc + c) * m2 + b;
    if (mode == 0) *e = mat[idx]; else mat[idx] = *e;
}
// per-column dot products of block vectors: out0[c] = <A, B>_c, out1[c] = <C, D>_c (second pair optional); one workgroup per
// column, fixed summation order
__global__ __launch_bounds__(256) void vgi_coldots_kernel(const double* A, const double* B, const double* C, const double* D, int m1, int nbc,
                                                          int m2, double* out0, double* out1) {
    __shared__ double red[8];
    const int c = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (long e = threadIdx.x; e < (long)m1 * m2; e += 256) {
        const long a = e / m2, b = e - a * m2, o = (a * nbc + c) * m2 + b;
        s0 += A[o] * B[o];
        if (C) s1 += C[o] * D[o];
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s0; red[4 + (threadIdx.x >> 6)] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out0[c] = (red[0] + red[1]) + (red[2] + red[3]);
        if (out1) out1[c] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}
// PCG scalar logic, one lane per column.  col: [0] rz, [1] pAp, [2] r0^2, [3] rr, [4] alpha, [5] beta, [6] active, [7] k (rows of nbc)
__global__ void vgi_pcg_scalars_kernel(double* col, int nbc, int phase, int it, double tol, double* alh, double* beh, int* nact) {
    const int c = threadIdx.x;
    double *rz = col, *pAp = col + nbc, *r02 = col + 2 * nbc, *rr = col + 3 * nbc, *al = col + 4 * nbc, *be = col + 5 * nbc,
           *act = col + 6 * nbc, *kc = col + 7 * nbc;
    if (c < nbc) {
        if (phase == 0) {                 // start: rz, rr hold <r, P^-1 r>, <r, r>
            r02[c] = rr[c]; act[c] = rr[c] > 0.0 ? 1.0 : 0.0; kc[c] = 0.0;
        } else if (phase == 1) {          // after pAp
            const double a = (act[c] != 0.0 && pAp[c] > 0.0) ? rz[c] / pAp[c] : 0.0;
            al[c] = a; alh[(long)it * nbc + c] = a;
        } else {                          // after the new <r, P^-1 r> (in pAp's slot is NOT touched: rz_new arrives in be's slot)
            const double rzn = be[c];
            const double b = (act[c] != 0.0 && rz[c] > 0.0) ? rzn / rz[c] : 0.0;
            be[c] = b; beh[(long)it * nbc + c] = b;
            rz[c] = rzn;
            if (act[c] != 0.0) kc[c] += 1.0;
            if (!(rr[c] > tol * tol * r02[c])) act[c] = 0.0;
        }
    }
    __syncthreads();
    if (phase != 1 && threadIdx.x == 0) { int n = 0; for (int k = 0; k < nbc; ++k) n += act[k] != 0.0 ? 1 : 0; *nact = n; }
}
// x += alpha_c p, r -= alpha_c ap    /    p = z + beta_c p
__global__ void vgi_update_kernel(double* X, double* R, double* Pd, const double* AP, const double* Zp, const double* col, int m1, int nbc,
                                  int m2, int phase) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)m1 * nbc * m2) return;
    const int c = (int)((idx / m2) % nbc);
    if (phase == 1) { const double a = col[4 * nbc + c]; X[idx] += a * Pd[idx]; R[idx] -= a * AP[idx]; }
    else Pd[idx] = Zp[idx] + col[5 * nbc + c] * Pd[idx];
}
// out = a + rho b   (AP = P + rho Phi~ P)
__global__ void vgi_axpy_rho_kernel(const double* a, const double* b, const double* theta, long n, double* out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = a[idx] + (theta[2] * theta[3] / theta[4]) * b[idx];
}
__global__ void vgi_sub_kernel(const double* a, const double* b, long n, double* out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = a[idx] - b[idx];
}
__global__ void vgi_mul_kernel(const double* a, const double* b, long n, double* out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = a[idx] * b[idx];
}
// out[a] = sum_k A[a][k] B[a][k]
__global__ void vgi_rowdot_kernel(const double* A, const double* B, int m, double* out) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += A[(long)a * m + k] * B[(long)a * m + k];
    out[a] = s;
}
// Gauss quadrature of log on the Lanczos tridiagonal of each probe column (implicit QL with shifts, first eigenvector components
// only): one lane per probe, k <= VGI_MAXIT.  out[c - 1] = M sum_j tau_j^2 log(theta_j)
#define VGI_MAXIT 128
__global__ void vgi_slq_kernel(const double* alh, const double* beh, const double* col, int nbc, double Mdim, double* out, int* fail) {
    const int c = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nbc) return;
    const int k = (int)col[7 * nbc + c];
    double d[VGI_MAXIT], e[VGI_MAXIT], z[VGI_MAXIT];
    for (int j = 0; j < k; ++j) {
        const double a = alh[(long)j * nbc + c];
        d[j] = 1.0 / a + (j > 0 ? beh[(long)(j - 1) * nbc + c] / alh[(long)(j - 1) * nbc + c] : 0.0);
        e[j] = j + 1 < k ? sqrt(beh[(long)j * nbc + c]) / a : 0.0;
        z[j] = j == 0 ? 1.0 : 0.0;
    }
    for (int l = 0; l < k; ++l) {
        int iter = 0, mm;
        do {
            for (mm = l; mm < k - 1; ++mm) {
                const double dd = fabs(d[mm]) + fabs(d[mm + 1]);
                if (fabs(e[mm]) <= 1e-16 * dd) break;
            }
            if (mm != l) {
                if (iter++ == 60) { atomicOr(fail, 1); break; }
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = hypot(g, 1.0);
                g = d[mm] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                double s = 1.0, cc = 1.0, p = 0.0;
                int i;
                for (i = mm - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = cc * e[i];
                    e[i + 1] = (r = hypot(f, g));
                    if (r == 0.0) { d[i + 1] -= p; e[mm] = 0.0; break; }
                    s = f / r; cc = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * cc * b;
                    d[i + 1] = g + (p = s * r);
                    g = cc * r - b;
                    f = z[i + 1];
                    z[i + 1] = s * z[i] + cc * f;
                    z[i] = cc * z[i] - s * f;
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[mm] = 0.0;
            }
        } while (mm != l);
    }
    double acc = 0.0;
    for (int j = 0; j < k; ++j) { if (!(d[j] > 0.0)) atomicOr(fail, 2); acc += z[j] * z[j] * log(d[j]); }
    out[c - 1] = Mdim * acc;
}
// scalars of the iterative step -> the slots of the dense step's reduction buffer (vgm_final_kernel reads them)
//   ts: [0] sum log dP, [1] sum of the quadratures, [2] exact tr(P^-1 Phi), [3] est BB, [4] exact V1, [5] est 1a, [6] est 1b,
//       [7] exact V2, [8] est 2a, [9] est 2b, [10] exact Mk1, [11] est Mk1, [12] exact Mk2, [13] est Mk2
__global__ void vgi_combine_kernel(const double* ts, const double* theta, double Mdim, int nz, double* scal) {
    if (threadIdx.x != 0) return;
    const double rho = theta[2] * theta[3] / theta[4], inz = 1.0 / nz;
    scal[RJ_LOGDET] = 0.5 * (ts[0] + ts[1] * inz);
    const double trSP = ts[2] + ts[3] * inz;
    scal[RJ_TRS] = Mdim - rho * trSP;
    scal[RJ_SPHI1] = 0.5 * (2.0 * ts[4] + (ts[5] + ts[6]) * inz);
    scal[RJ_SPHI2] = 0.5 * (2.0 * ts[7] + (ts[8] + ts[9]) * inz);
    scal[RJ_MK1PTS] = ts[10] + ts[11] * inz;
    scal[RJ_MK2PTS] = ts[12] + ts[13] * inz;
}
// out[0] = sum_{a,b} f(dP[a,b]) * E[a][b] (E null: 1) * (dg1 ? dg1[a] : 1) * (dg2 ? dg2[b] : 1);  mode 0: f = 1/dP, 1: f = log dP
__global__ __launch_bounds__(256) void vgi_dpsum_kernel(const double* lam1, const double* lam2, const double* theta, double p, int m1, int m2,
                                                        const double* E, const double* dg1, const double* dg2, int mode, double* out) {
    __shared__ double red[4];
    const double rho = theta[2] * theta[3] / theta[4];
    double s = 0.0;
    for (long e = threadIdx.x; e < (long)m1 * m2; e += 256) {
        const int a = (int)(e / m2), b = (int)(e - (long)a * m2);
        const double d = 1.0 + rho * p * fmax(lam1[a], 0.0) * fmax(lam2[b], 0.0);
        double v = mode == 0 ? 1.0 / d : log(d);
        if (E) v *= E[e];
        if (dg1) v *= dg1[a];
        if (dg2) v *= dg2[b];
        s += v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[0] = sum over i, c >= 1, j of Wt[i][j] Fa[i][c][j] Fb[i][c][j]   (two stages, fixed order)
__global__ __launch_bounds__(256) void vgi_fieldsum_part_kernel(const double* Fa, const double* Fb, const double* Wt, long n1, int nbc, long n2,
                                                                double* part) {
    __shared__ double red[4];
    const long total = n1 * nbc * n2;
    double s = 0.0;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long j = idx % n2, c = (idx / n2) % nbc, i = idx / (n2 * nbc);
        if (c >= 1) s += Wt[i * n2 + j] * Fa[idx] * Fb[idx];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void vgi_sum_kernel(const double* part, int n, double* out, int c_from) {           // out[0] = sum_{k >= c_from} part[k]
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int k = c_from; k < n; ++k) s += part[k];
    out[0] = s;
}

static int vgi_prepare(vggp_ctx* c, VgMasked& w, VgIter& it, int nbc, int maxit) {
    const size_t m1 = w.m1, m2 = w.m2, n1 = w.n1, n2 = w.n2, M = w.M, mx = std::max(m1, m2);
    size_t off = 0;
    char* base = nullptr;
    for (int pass = 0; pass < 2; ++pass) {
        off = 0;
        auto take = [&](size_t count) {
            off = (off + 255) & ~size_t(255);
            double* p = base ? reinterpret_cast<double*>(base + off) : nullptr;
            off += count * sizeof(double);
            return p;
        };
        it.Wt = take(n1 * n2);
        it.X = take(M * nbc); it.R = take(M * nbc); it.Zp = take(M * nbc); it.Pd = take(M * nbc); it.AP = take(M * nbc);
        it.Wz = take(M * nbc); it.Tm = take(M * nbc); it.Tm2 = take(M * nbc);
        it.T1 = take(n1 * nbc * mx);
        it.F0 = take(n1 * nbc * n2); it.F1 = take(n1 * nbc * n2); it.F2 = take(n1 * nbc * n2);
        it.alh = take((size_t)maxit * nbc); it.beh = take((size_t)maxit * nbc);
        it.col = take(8 * (size_t)nbc);
        it.Rr1 = take(m1 * n1); it.Rv1 = take(m1 * n1); it.RR1 = take(m1 * n1); it.RRV1 = take(m1 * n1);
        it.Rr2 = take(m2 * n2); it.Rv2 = take(m2 * n2); it.RR2 = take(m2 * n2); it.RRV2 = take(m2 * n2);
        it.TW = take(n1 * m2); it.TWv = take(n1 * m2); it.Ex = take(M);
        it.dg1 = take(m1); it.dg2 = take(m2); it.Tq = take(mx * mx);
        it.Qk1 = take(m1 * m1); it.Qk2 = take(m2 * m2);
        it.ts = take(64);
        it.nact = reinterpret_cast<int*>(take(8));
        if (pass == 0) {
            const size_t need = off + 4096;
            if (w.ibytes < need || w.ib_nbc != nbc || w.ib_maxit != maxit) w.ib_valid = false;      // (the layout moves: the kept basis is gone)
            w.ib_nbc = nbc; w.ib_maxit = maxit;
            if (w.ibytes < need) {
                if (w.imem) { VG_HIP(hipFree(w.imem)); w.imem = nullptr; w.ibytes = 0; }
                VG_HIP(hipMalloc(&w.imem, need));
                w.ibytes = need;
            }
            base = reinterpret_cast<char*>(w.imem);
        }
    }
    it.nbc = nbc; it.maxit = maxit;
    (void)c;
    return VGGP_OK;
}

// G = L^T-side product: block field F[i][c][j] = l_i^T V_c r_j for all columns (V block [m1][nbc][m2], L: m1 x n1, R: m2 x n2)
static int vgi_field(const VgMasked& w, const VgIter& it, const double* L, const double* V, const double* R, double* F, hipStream_t st) {
    const int m1 = w.m1, m2 = w.m2, nbc = it.nbc;
    const long n1 = w.n1, n2 = w.n2;
    int rc;
    // T1[i][(c,b)] = sum_a L[a][i] V[a][(c,b)]
    if ((rc = gemm1(L, 1, n1, V, (long)nbc * m2, 1, it.T1, nbc * m2, (int)n1, nbc * m2, m1, st))) return rc;
    // F[(i,c)][j] = sum_b T1[(i,c)][b] R[b][j]
    return gemm1(it.T1, m2, 1, R, n2, 1, F, (int)n2, (int)(n1 * nbc), (int)n2, m2, st);
}
// out[a][(c,b)] = sum_{i,j} L[a][i] F[i][c][j] R[b][j]
static int vgi_back(const VgMasked& w, const VgIter& it, const double* L, const double* F, const double* R, double* out, hipStream_t st) {
    const int m1 = w.m1, m2 = w.m2, nbc = it.nbc;
    const long n1 = w.n1, n2 = w.n2;
    int rc;
    // T1[(i,c)][b] = sum_j F[(i,c)][j] R[b][j]
    if ((rc = gemm1(F, n2, 1, R, 1, n2, it.T1, m2, (int)(n1 * nbc), m2, (int)n2, st))) return rc;
    return gemm1(L, n1, 1, it.T1, (long)nbc * m2, 1, out, nbc * m2, m1, nbc * m2, (int)n1, st);
}
// out = Q1 ((Q1^T V Q2) o f(dP)) Q2^T for the whole block (Qt rows = eigenvectors); mode as vgi_scale_kernel; c_from: columns below are zeroed
static int vgi_rot(vggp_ctx* c, const VgMasked& w, const VgIter& it, const double* V, double* out, int mode, int c_from, double p, hipStream_t st,
                   double* rotated_out = nullptr) {
    const int m1 = w.m1, m2 = w.m2, nbc = it.nbc;
    const long nb = (long)m1 * nbc * m2;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    int rc;
    if ((rc = gemm1(d1.Qt, m1, 1, V, (long)nbc * m2, 1, it.Tm, nbc * m2, m1, nbc * m2, m1, st))) return rc;            // Q1^T V
    if ((rc = gemm1(it.Tm, m2, 1, d2.Qt, 1, m2, it.Tm2, m2, m1 * nbc, m2, m2, st))) return rc;                      // (.) Q2
    if (rotated_out) VG_HIP(hipMemcpyAsync(rotated_out, it.Tm2, sizeof(double) * nb, hipMemcpyDeviceToDevice, st));
    VGM_LAUNCH1D(vgi_scale_kernel, nb, st, it.Tm2, it.Tm2, d1.lam0, d2.lam0, c->theta, p, m1, nbc, m2, mode, c_from);
    if ((rc = gemm1(it.Tm2, m2, 1, d2.Qt, m2, 1, it.Tm, m2, m1 * nbc, m2, m2, st))) return rc;                      // (.) Q2^T
    return gemm1(d1.Qt, 1, m1, it.Tm, (long)nbc * m2, 1, out, nbc * m2, m1, nbc * m2, m1, st);                     // Q1 (.)
}

#define VGI_ESTALE (-1001)      // internal: the kept preconditioner basis is too far off (the caller repeats the step with a cold solve)
static int masked_iter_once(vggp_ctx* c, const double* Ym, const double* W, double n_obs, double yy_obs, const double theta[5],
                            int n_probes, double tol, int max_iter, double* elbo_out, double grad_out[5], vggp_info* info, void* stream);
extern "C" int vggp_elbo_step_masked_iter(vggp_ctx* c, const double* Ym, const double* W, double n_obs, double yy_obs, const double theta[5],
                                          int n_probes, double tol, int max_iter, double* elbo_out, double grad_out[5], vggp_info* info,
                                          void* stream) {
    int rc = masked_iter_once(c, Ym, W, n_obs, yy_obs, theta, n_probes, tol, max_iter, elbo_out, grad_out, info, stream);
    if (rc == VGI_ESTALE) rc = masked_iter_once(c, Ym, W, n_obs, yy_obs, theta, n_probes, tol, max_iter, elbo_out, grad_out, info, stream);
    if (rc == VGI_ESTALE) { vg_set_error("vggp_elbo_step_masked_iter: internal (stale basis after a cold solve)"); rc = VGGP_ESTATE; }
    return rc;
}
static int masked_iter_once(vggp_ctx* c, const double* Ym, const double* W, double n_obs, double yy_obs, const double theta[5],
                            int n_probes, double tol, int max_iter, double* elbo_out, double grad_out[5], vggp_info* info, void* stream) {
    if (!c || !c->planned) { vg_set_error("vggp_elbo_step_masked_iter: context not planned"); return VGGP_ESTATE; }
    VG_REQUIRE(Ym && W && theta && elbo_out && grad_out, "vggp_elbo_step_masked_iter: null argument");
    c->have_masked = false;
    VG_REQUIRE(!(c->desc.flags & VGGP_FLAG_SCATTERED), "vggp_elbo_step_masked_iter: the context was planned for scattered points");
    VG_REQUIRE(!(c->n_ranks > 1 || c->comm || c->cb), "vggp_elbo_step_masked_iter: single-rank contexts only");
    if (n_probes <= 0) n_probes = 16;
    if (max_iter <= 0) max_iter = 100;
    if (!(tol > 0.0)) tol = 1e-10;
    VG_REQUIRE(n_probes <= 63 && max_iter <= VGI_MAXIT, "vggp_elbo_step_masked_iter: n_probes <= 63, max_iter <= %d", VGI_MAXIT);
    const long m1 = c->desc.m1, m2 = c->desc.m2, n1 = c->desc.n1, n2 = c->desc.n2, M = m1 * m2;
    const int nbc = n_probes + 1;
    VG_REQUIRE(n1 * nbc * n2 < (1L << 31) * 4 && n1 * nbc < (1L << 31) && M * nbc < (1L << 31), "vggp_elbo_step_masked_iter: problem too large");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    for (int i = 0; i < 5; ++i) {
        VG_REQUIRE(theta[i] > 0.0 && std::isfinite(theta[i]), "theta[%d]=%g must be positive and finite", i, theta[i]);
        c->h_theta[i] = theta[i];
    }
    int rc = vgm_prepare(c, /*iter=*/true);
    if (rc) return rc;
    VgMasked& w = *reinterpret_cast<VgMasked*>(c->masked);
    VgIter it;
    if ((rc = vgi_prepare(c, w, it, nbc, max_iter))) return rc;
    // factors, Cholesky, B|V, Mk, the projections C0, C1, C2 and the full-grid Gram matrices G1 (d1.GH), G2 (mpay)
    if ((rc = vg_partials_enqueue(c, Ym, w.mpay, st))) return rc;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const double *B1 = d1.BV, *V1 = d1.BV + m1 * n1, *B2 = d2.BV, *V2 = d2.BV + m2 * n2;
    double *C0 = w.mpay + 2 * m2 * m2, *C1 = C0 + M, *C2 = C1 + M;
    if ((rc = vgm_colstats(c, w, W, st))) return rc;
    hipLaunchKernelGGL(vgi_transpose_kernel, dim3((unsigned)((n1 + 31) / 32), (unsigned)((n2 + 31) / 32)), dim3(32, 8), 0, st, W, n1, n2, it.Wt);
    bool reused = false;
    // Eigenbasis of the preconditioner.  Everything below is exact in expectation for ANY orthonormal Q_d and any positive
    // dP = 1 + rho p l1 l2 -- P~ = (Q1 (x) Q2) diag(dP) (Q1 (x) Q2)^T is then simply another SPD preconditioner with a known
    // determinant: log|Sigma~| = sum log dP + tr log(P~^-1/2 Sigma~ P~^-1/2), the exact traces are traces against P~^-1 -- so the
    // cold Jacobi solve of G1, G2 (1.7 ms at m_d = 128, 74 ms at m_d = 256 where the solver works in global memory: 70 % of that
    // step) runs only on the first step of a plan, every 64 steps, and when the PCG needed 4 iterations more than right after the
    // last solve; in between the kept basis is reused with the Rayleigh quotients l_i = q_i^T G q_i of the CURRENT Gram matrices.
    // VGGP_ITER_COLD_BASIS=1: solve every step.
    {
        static const bool always = getenv("VGGP_ITER_COLD_BASIS") != nullptr;
        const double* G[2] = {d1.GH, w.mpay};
        // (not for RBF factors: rho lam1 lam2 spans eight decades there, and a basis that is 1e-2 off leaves P~^-1 Sigma~ with a
        //  condition number of 1e4 -- the PCG stalls; measured.  Matern spectra are flat enough: 8 -> 11 -> 13 iterations over 4 % drift)
        const bool reuse = !always && w.ib_valid && w.ib_age < 64 && w.ib_last_its <= w.ib_ref_its + 4 && d1.kind != VGGP_KIND_RBF &&
                           d2.kind != VGGP_KIND_RBF;
        reused = reuse;
        if (reuse) {
            double* Qk[2] = {it.Qk1, it.Qk2};
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                VG_HIP(hipMemcpyAsync(d.Qt, Qk[k], sizeof(double) * d.m * d.m, hipMemcpyDeviceToDevice, st));
                if ((rc = gemm1(d.Qt, d.m, 1, G[k], d.m, 1, it.Tq, d.m, d.m, d.m, d.m, st))) return rc;         // Q^T-rows times G
                VGM_LAUNCH1D(vgi_rowdot_kernel, d.m, st, it.Tq, d.Qt, d.m, d.lam0);
            }
            ++w.ib_age;
            w.ib_fresh = false;
        } else {
            VgEigJob ej[2];
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                ej[k] = VgEigJob{G[k], d.lam0, d.Qt, nullptr, d.gwork, d.rotlog, d.roundlog, d.counters, d.m, d.max_rounds,
                                 (long)vg_eigh_log_bytes(d.m), 0};
                ej[k].perm = d.perm;
                ej[k].err = d.status + 1;
            }
            VG_HIP(vg_eigh_launch(ej, 2, st));
            VG_HIP(hipMemcpyAsync(it.Qk1, d1.Qt, sizeof(double) * m1 * m1, hipMemcpyDeviceToDevice, st));
            VG_HIP(hipMemcpyAsync(it.Qk2, d2.Qt, sizeof(double) * m2 * m2, hipMemcpyDeviceToDevice, st));
            w.ib_valid = false;            // ... until this step has returned without an error
            w.ib_fresh = true;
        }
    }
    const double p = n_obs / ((double)n1 * (double)n2);
    const long nb = M * nbc;
    // right-hand sides: column 0 = c0, columns 1.. = z = P^1/2 z0;  Wz = P^-1 z = P^-1/2 z0
    VGM_LAUNCH1D(vgi_probe_kernel, nb, st, it.X, (int)m1, nbc, (int)m2, 0x5647475000000001ULL);
    if ((rc = vgi_rot(c, w, it, it.X, it.R, 1, 1, p, st))) return rc;
    if ((rc = vgi_rot(c, w, it, it.X, it.Wz, 2, 1, p, st))) return rc;
    VGM_LAUNCH1D(vgi_col_kernel, M, st, it.R, C0, (int)m1, nbc, (int)m2, 0, 0);
    VG_HIP(hipMemsetAsync(it.X, 0, sizeof(double) * nb, st));
    // PCG
    if ((rc = vgi_rot(c, w, it, it.R, it.Zp, 0, 0, p, st))) return rc;
    VG_HIP(hipMemcpyAsync(it.Pd, it.Zp, sizeof(double) * nb, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(vgi_coldots_kernel, dim3(nbc), dim3(256), 0, st, it.R, it.Zp, it.R, it.R, (int)m1, nbc, (int)m2, it.col, it.col + 3 * nbc);
    hipLaunchKernelGGL(vgi_pcg_scalars_kernel, dim3(1), dim3(64), 0, st, it.col, nbc, 0, 0, tol, it.alh, it.beh, it.nact);
    int iters = 0, nact = nbc;
    for (int k = 0; k < max_iter && nact > 0; ++k) {
        // AP = Pd + rho B1 (Wt o (B1^T Pd B2)) B2^T
        if ((rc = vgi_field(w, it, B1, it.Pd, B2, it.F0, st))) return rc;
        VGM_LAUNCH1D(vgi_mask_kernel, n1 * nbc * n2, st, it.F0, it.Wt, n1, nbc, n2);
        if ((rc = vgi_back(w, it, B1, it.F0, B2, it.AP, st))) return rc;
        VGM_LAUNCH1D(vgi_axpy_rho_kernel, nb, st, it.Pd, it.AP, c->theta, nb, it.AP);
        hipLaunchKernelGGL(vgi_coldots_kernel, dim3(nbc), dim3(256), 0, st, it.Pd, it.AP, (const double*)nullptr, (const double*)nullptr, (int)m1, nbc,
                           (int)m2, it.col + nbc, (double*)nullptr);
        hipLaunchKernelGGL(vgi_pcg_scalars_kernel, dim3(1), dim3(64), 0, st, it.col, nbc, 1, k, tol, it.alh, it.beh, it.nact);
        VGM_LAUNCH1D(vgi_update_kernel, nb, st, it.X, it.R, it.Pd, it.AP, it.Zp, it.col, (int)m1, nbc, (int)m2, 1);
        if ((rc = vgi_rot(c, w, it, it.R, it.Zp, 0, 0, p, st))) return rc;
        hipLaunchKernelGGL(vgi_coldots_kernel, dim3(nbc), dim3(256), 0, st, it.R, it.Zp, it.R, it.R, (int)m1, nbc, (int)m2, it.col + 5 * nbc,
                           it.col + 3 * nbc);
        hipLaunchKernelGGL(vgi_pcg_scalars_kernel, dim3(1), dim3(64), 0, st, it.col, nbc, 2, k, tol, it.alh, it.beh, it.nact);
        VGM_LAUNCH1D(vgi_update_kernel, nb, st, it.X, it.R, it.Pd, it.AP, it.Zp, it.col, (int)m1, nbc, (int)m2, 2);
        VG_HIP(hipMemcpyAsync(&c->h_out->counters[0][0], it.nact, sizeof(int), hipMemcpyDeviceToHost, st));
        VG_HIP(hipStreamSynchronize(st));
        nact = c->h_out->counters[0][0];
        iters = k + 1;
        if (reused && nact > 0 && iters > w.ib_ref_its + 12) {      // the kept basis has drifted too far: not worth iterating on
            w.ib_valid = false;
            return VGI_ESTALE;
        }
    }
    if (nact > 0) { vg_set_error("vggp_elbo_step_masked_iter: PCG did not reach %.1e in %d iterations (%d columns left)", tol, max_iter, nact); return VGGP_ENOCONV; }
    // log det: log|P| + the quadratures
    VG_HIP(hipMemsetAsync(it.ts, 0, 64 * sizeof(double), st));
    VG_HIP(hipMemsetAsync(w.cholstatus, 0, sizeof(int), st));
    hipLaunchKernelGGL(vgi_dpsum_kernel, dim3(1), dim3(256), 0, st, d1.lam0, d2.lam0, c->theta, p, (int)m1, (int)m2, (const double*)nullptr,
                       (const double*)nullptr, (const double*)nullptr, 1, it.ts + 0);
    hipLaunchKernelGGL(vgi_slq_kernel, dim3(1), dim3(64), 0, st, it.alh, it.beh, it.col, nbc, (double)M, it.ts + 32, w.cholstatus);
    hipLaunchKernelGGL(vgi_sum_kernel, dim3(1), dim3(64), 0, st, it.ts + 32, n_probes, it.ts + 1, 0);
    // a0 and the a0 terms of the gradient
    VGM_LAUNCH1D(vgi_col_kernel, M, st, it.X, w.a0, (int)m1, nbc, (int)m2, 0, 1);
    if ((rc = vgm_a0_terms(c, w, st))) return rc;
    // dU = Sigma~^-1 z - P^-1 z (probe columns)
    VGM_LAUNCH1D(vgi_sub_kernel, nb, st, it.X, it.Wz, nb, it.Zp);
    double* dU = it.Zp;
    // stochastic parts of the traces: fields of dU and of Wz with (B1,B2), (V1,B2), (B1,V2)
    auto fsum = [&](const double* Fa, const double* Fb, double* out) {
        hipLaunchKernelGGL(vgi_fieldsum_part_kernel, dim3(1024), dim3(256), 0, st, Fa, Fb, it.Wt, n1, nbc, n2, it.AP);      // (AP is free after the PCG)
        hipLaunchKernelGGL(vgi_sum_kernel, dim3(1), dim3(64), 0, st, it.AP, 1024, out, 0);
    };
    if ((rc = vgi_field(w, it, B1, it.Wz, B2, it.F0, st))) return rc;              // F0 = Fw^BB
    if ((rc = vgi_field(w, it, B1, dU, B2, it.F1, st))) return rc;                 // F1 = Fu^BB
    fsum(it.F1, it.F0, it.ts + 3);
    if ((rc = vgi_field(w, it, V1, dU, B2, it.F2, st))) return rc;                 // Fu^VB
    fsum(it.F2, it.F0, it.ts + 6);                                                 // (u-w)^T Phi'_1^T w
    if ((rc = vgi_field(w, it, B1, dU, V2, it.F2, st))) return rc;                 // Fu^BV
    fsum(it.F2, it.F0, it.ts + 9);
    if ((rc = vgi_field(w, it, V1, it.Wz, B2, it.F0, st))) return rc;              // Fw^VB
    fsum(it.F1, it.F0, it.ts + 5);
    if ((rc = vgi_field(w, it, B1, it.Wz, V2, it.F0, st))) return rc;              // Fw^BV
    fsum(it.F1, it.F0, it.ts + 8);
    // Mk terms: <dU, Mk1 Wz>, <dU, Wz Mk2^T>
    if ((rc = gemm1(d1.Mk, m1, 1, it.Wz, (long)nbc * m2, 1, it.Tm, nbc * (int)m2, (int)m1, nbc * (int)m2, (int)m1, st))) return rc;
    hipLaunchKernelGGL(vgi_coldots_kernel, dim3(nbc), dim3(256), 0, st, dU, it.Tm, (const double*)nullptr, (const double*)nullptr, (int)m1, nbc,
                       (int)m2, it.ts + 32, (double*)nullptr);
    hipLaunchKernelGGL(vgi_sum_kernel, dim3(1), dim3(64), 0, st, it.ts + 32, nbc, it.ts + 11, 1);
    if ((rc = gemm1(it.Wz, m2, 1, d2.Mk, 1, m2, it.Tm, (int)m2, (int)(m1 * nbc), (int)m2, (int)m2, st))) return rc;
    hipLaunchKernelGGL(vgi_coldots_kernel, dim3(nbc), dim3(256), 0, st, dU, it.Tm, (const double*)nullptr, (const double*)nullptr, (int)m1, nbc,
                       (int)m2, it.ts + 32, (double*)nullptr);
    hipLaunchKernelGGL(vgi_sum_kernel, dim3(1), dim3(64), 0, st, it.ts + 32, nbc, it.ts + 13, 1);
    // exact parts: factors rotated into the eigenbasis of P
    if ((rc = gemm1(d1.Qt, m1, 1, B1, n1, 1, it.Rr1, (int)n1, (int)m1, (int)n1, (int)m1, st))) return rc;
    if ((rc = gemm1(d1.Qt, m1, 1, V1, n1, 1, it.Rv1, (int)n1, (int)m1, (int)n1, (int)m1, st))) return rc;
    if ((rc = gemm1(d2.Qt, m2, 1, B2, n2, 1, it.Rr2, (int)n2, (int)m2, (int)n2, (int)m2, st))) return rc;
    if ((rc = gemm1(d2.Qt, m2, 1, V2, n2, 1, it.Rv2, (int)n2, (int)m2, (int)n2, (int)m2, st))) return rc;
    VGM_LAUNCH1D(vgi_mul_kernel, m1 * n1, st, it.Rr1, it.Rr1, m1 * n1, it.RR1);
    VGM_LAUNCH1D(vgi_mul_kernel, m1 * n1, st, it.Rr1, it.Rv1, m1 * n1, it.RRV1);
    VGM_LAUNCH1D(vgi_mul_kernel, m2 * n2, st, it.Rr2, it.Rr2, m2 * n2, it.RR2);
    VGM_LAUNCH1D(vgi_mul_kernel, m2 * n2, st, it.Rr2, it.Rv2, m2 * n2, it.RRV2);
    if ((rc = gemm1(it.Wt, n2, 1, it.RR2, 1, n2, it.TW, (int)m2, (int)n1, (int)m2, (int)n2, st))) return rc;        // Wt RR2^T  (n1 x m2)
    if ((rc = gemm1(it.Wt, n2, 1, it.RRV2, 1, n2, it.TWv, (int)m2, (int)n1, (int)m2, (int)n2, st))) return rc;
    auto exact = [&](const double* RRa, const double* TWb, double* out) -> int {
        int r2 = gemm_longk(RRa, n1, 1, TWb, m2, 1, it.Ex, (int)m1, (int)m2, (int)n1, w.T, st);
        if (r2) return r2;
        hipLaunchKernelGGL(vgi_dpsum_kernel, dim3(1), dim3(256), 0, st, d1.lam0, d2.lam0, c->theta, p, (int)m1, (int)m2, it.Ex,
                           (const double*)nullptr, (const double*)nullptr, 0, out);
        return VGGP_OK;
    };
    if ((rc = exact(it.RR1, it.TW, it.ts + 2))) return rc;
    if ((rc = exact(it.RRV1, it.TW, it.ts + 4))) return rc;
    if ((rc = exact(it.RR1, it.TWv, it.ts + 7))) return rc;
    // tr(P^-1 (Mk1 (x) I)) = sum_ab diag(Q1^T Mk1 Q1)_a / dP_ab, likewise dimension 2
    if ((rc = gemm1(d1.Qt, m1, 1, d1.Mk, m1, 1, it.Tq, (int)m1, (int)m1, (int)m1, (int)m1, st))) return rc;
    VGM_LAUNCH1D(vgi_rowdot_kernel, m1, st, it.Tq, d1.Qt, (int)m1, it.dg1);
    if ((rc = gemm1(d2.Qt, m2, 1, d2.Mk, m2, 1, it.Tq, (int)m2, (int)m2, (int)m2, (int)m2, st))) return rc;
    VGM_LAUNCH1D(vgi_rowdot_kernel, m2, st, it.Tq, d2.Qt, (int)m2, it.dg2);
    hipLaunchKernelGGL(vgi_dpsum_kernel, dim3(1), dim3(256), 0, st, d1.lam0, d2.lam0, c->theta, p, (int)m1, (int)m2, (const double*)nullptr, it.dg1,
                       (const double*)nullptr, 0, it.ts + 10);
    hipLaunchKernelGGL(vgi_dpsum_kernel, dim3(1), dim3(256), 0, st, d1.lam0, d2.lam0, c->theta, p, (int)m1, (int)m2, (const double*)nullptr,
                       (const double*)nullptr, it.dg2, 0, it.ts + 12);
    // the dense step's reductions that do not involve Sigma~^-1 as a matrix; the six that do come from the combine kernel
    VgmRedArgs ra;
    ra.njobs = RJ_COUNT;
    ra.partial = w.partial;
    auto job = [&](int k, const double* a, const double* b, long n, long sa, long sb, int op, const double* cc = nullptr) {
        ra.job[k] = VgmRedJob{a, b, cc, n, sa, sb, op};
    };
    for (int k = 0; k < RJ_COUNT; ++k) job(k, w.a0, w.a0, 0, 1, 1, 0);
    job(RJ_Q, C0, w.a0, M, 1, 1, 0);
    job(RJ_AA, w.a0, w.a0, M, 1, 1, 0);
    job(RJ_TRPHI, w.nb1, w.wn2, n1, 1, 1, 0);
    job(RJ_TRMK1, d1.Mk, nullptr, m1, m1 + 1, 0, 2);
    job(RJ_AC1, w.a0, C1, M, 1, 1, 0);
    job(RJ_MKA1, w.MkA1, w.a0, M, 1, 1, 0);
    job(RJ_Z1, W, w.Zb, n1 * n2, 1, 1, 3, w.Zv1);
    job(RJ_HV1, w.hv1, w.wn2, n1, 1, 1, 0);
    job(RJ_MK1PT, d1.Mk, w.PT1, m1 * m1, 1, 1, 0);
    job(RJ_TRMK2, d2.Mk, nullptr, m2, m2 + 1, 0, 2);
    job(RJ_AC2, w.a0, C2, M, 1, 1, 0);
    job(RJ_MKA2, w.MkA2, w.a0, M, 1, 1, 0);
    job(RJ_Z2, W, w.Zb, n1 * n2, 1, 1, 3, w.Zv2);
    job(RJ_HV2, w.hv2, w.wn1, n2, 1, 1, 0);
    job(RJ_MK2PT, d2.Mk, w.PT2, m2 * m2, 1, 1, 0);
    hipLaunchKernelGGL(vgm_red_kernel, dim3(VG_MD_NPART, RJ_COUNT), dim3(256), 0, st, ra);
    hipLaunchKernelGGL(vgm_sum_kernel, dim3(1), dim3(64), 0, st, w.partial, w.scal, 0u, 1);
    hipLaunchKernelGGL(vgi_combine_kernel, dim3(1), dim3(64), 0, st, it.ts, c->theta, (double)M, n_probes, w.scal);
    VgmFinalArgs fa{c->theta, w.scal, w.out, n_obs, yy_obs, (int)m1, (int)m2};
    hipLaunchKernelGGL(vgm_final_kernel, dim3(1), dim3(64), 0, st, fa);
    VG_HIP(hipGetLastError());
    VG_HIP(hipMemcpyAsync(c->h_out->out, w.out, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    for (int k = 0; k < 2; ++k) {
        VG_HIP(hipMemcpyAsync(&c->h_out->jitter[k], c->d[k].jitter, sizeof(double), hipMemcpyDeviceToHost, st));
        VG_HIP(hipMemcpyAsync(&c->h_out->status[k], c->d[k].status, sizeof(int), hipMemcpyDeviceToHost, st));
        VG_HIP(hipMemcpyAsync(&c->h_out->counters[k][2], c->d[k].counters + 2, sizeof(int), hipMemcpyDeviceToHost, st));
    }
    VG_HIP(hipMemcpyAsync(&c->h_out->counters[1][3], w.cholstatus, sizeof(int), hipMemcpyDeviceToHost, st));
    VG_HIP(hipStreamSynchronize(st));
    *elbo_out = c->h_out->out[0];
    for (int i = 0; i < 5; ++i) grad_out[i] = c->h_out->out[1 + i];
    int status = c->h_out->status[0] ? c->h_out->status[0] : (c->h_out->status[1] ? c->h_out->status[1] : 0);
    if (!status) status = c->h_out->counters[0][2] ? c->h_out->counters[0][2] : c->h_out->counters[1][2];
    if (info) {
        info->jitter1 = c->h_out->jitter[0]; info->jitter2 = c->h_out->jitter[1];
        info->sweeps1 = n_probes; info->sweeps2 = 0; info->rounds1 = iters; info->rounds2 = 0;
        info->status = status; info->polished = 0;
    }
    if (status == VGGP_ENOTPD) { vg_set_error("iterative masked step: a factor is not positive definite"); return VGGP_ENOTPD; }
    if (status) { vg_set_error("iterative masked step: the preconditioner's eigensolver failed (status %d)", status); return VGGP_ENOCONV; }
    if (c->h_out->counters[1][3]) { vg_set_error("iterative masked step: the Lanczos quadrature failed (code %d)", c->h_out->counters[1][3]); return VGGP_ENOCONV; }
    if (w.ib_fresh) { w.ib_valid = true; w.ib_age = 0; w.ib_ref_its = iters; }
    w.ib_last_its = iters;
    c->have_step = false; c->have_partials = false;
    return VGGP_OK;
}
