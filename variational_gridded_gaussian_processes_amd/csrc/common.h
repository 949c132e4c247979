// Shared declarations for the libvggp_hip.so translation units (gfx950 only).
#pragma once
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libvggp_hip is written for gfx950 (MI355X) only: kernels rely on its wave64 barrier semantics, MFMA shapes and LDS size"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vggp.h"

#define VG_WAVE 64

// ---- error plumbing (never throw across the C boundary) -----------------------
void vg_set_error(const char* fmt, ...);
#define VG_HIP(call)                                                              \
    do {                                                                          \
        hipError_t _e = (call);                                                   \
        if (_e != hipSuccess) {                                                   \
            vg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,            \
                         hipGetErrorString(_e));                                  \
            return VGGP_EHIP;                                                     \
        }                                                                         \
    } while (0)
#define VG_REQUIRE(cond, ...)                                                     \
    do {                                                                          \
        if (!(cond)) {                                                            \
            vg_set_error(__VA_ARGS__);                                            \
            return VGGP_EINVAL;                                                   \
        }                                                                         \
    } while (0)

// Every C entry point runs on the context's device and restores the caller's current device on return (the library never
// changes the calling thread's device as a side effect).
struct VgDeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev != dev) { e = hipSetDevice(dev); switched = (e == hipSuccess); }
        return e;
    }
    ~VgDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define VG_ENTER_DEVICE(dev) VgDeviceGuard vg_dev_guard_; VG_HIP(vg_dev_guard_.enter(dev))

// ---- batched strided GEMM descriptors (passed by value as kernel argument) -----
#define VG_GEMM_MAXP 12
#define VG_BM 64
#define VG_BN 64
#define VG_BK 16

struct VgGemmP {
    const double* A;   // element (i,k) at A[i*sa_m + k*sa_k]
    const double* B;   // element (k,j) at B[k*sb_k + j*sb_n]
    double* C;         // row-major, leading dim ldc; split-K slab s at C + s*c_slab
    long sa_m, sa_k, sb_k, sb_n;
    long c_slab;       // doubles between split-K output slabs
    long b_slab;       // B operand = sum of b_nslab slabs, this many doubles apart
    long a_slab;       // A operand likewise (a_nslab slabs)
    int a_nslab;
    int M, N, K;
    int ldc;
    int ksplit;        // >= 1
    int b_nslab;       // >= 1
    int tiles_m, tiles_n;
    int tile_start;    // first linear block index of this problem
    int kchunk;        // K elements per split (multiple of VG_BK)
    double alpha;      // C = (accum ? C : 0) + alpha * A B
    int accum;
    int xcd_group;     // 1: XCD-aware block order (vg_gemm_xcd_group): the tile rows that stream the same B block share an XCD
    int tri;           // VG_TRI_*: one operand is triangular, tiles skip the k-range where it is zero (at 128-granularity)
    // optional reduction epilogue: dot_out[tile] = sum over the tile of (alpha A B)_ij * dotw[i * dotw_ld + j] (fixed order:
    // bitwise reproducible); C may then be null (nothing stored).  Not combined with split-K.
    const double* dotw;
    double* dot_out;
    int dotw_ld;
    int deep;          // set by vg_gemm_launch: this problem runs on the deep-stage tile (gemm.hip vg_gemm_deep_body)
    int rev;           // 1: tiles in reverse order (deep tile only): a triangular problem's long tiles meet its batch partner's short ones on a CU
};
#define VG_TRI_NONE 0
#define VG_TRI_A_LOWER 1   // op(A)[i][k] = 0 for k > i:  k < roundup128(row0 + T)
#define VG_TRI_A_UPPER 2   // op(A)[i][k] = 0 for k < i:  k >= rounddown128(row0)
#define VG_TRI_B_UPPER 3   // op(B)[k][j] = 0 for k > j:  k < roundup128(col0 + T)
#define VG_TRI_B_LOWER 4   // op(B)[k][j] = 0 for k < j:  k >= rounddown128(col0)
#define VG_TRI_A_UPPER_B_LOWER 5   // both (X^T X of a lower-triangular X):  k >= rounddown128(max(row0, col0))
struct VgGemmBatch {
    int nprob;
    int total_tiles;
    VgGemmP p[VG_GEMM_MAXP];
};

// host helpers (gemm.hip)
void vg_gemm_init(VgGemmBatch* b);
// append one problem; returns index.  ksplit slabs land at C + s*c_slab.
int vg_gemm_add(VgGemmBatch* b, const double* A, long sa_m, long sa_k, const double* B, long sb_k,
                long sb_n, double* C, int ldc, int M, int N, int K, int ksplit = 1, long c_slab = 0,
                int b_nslab = 1, long b_slab = 0, double alpha = 1.0, int accum = 0);
#define VG_GEMM_TAG_GRAM_PROJECT 1
#define VG_GEMM_TAG_WIDE 2          // about one workgroup per CU and a long reduction: the 8-wave tile
void vg_gemm_xcd_group(VgGemmBatch* b, int prob);      // switch the XCD-aware block order on for a problem (if its shape allows)
hipError_t vg_gemm_launch(const VgGemmBatch* b, hipStream_t st, int tag = 0);
const char* vg_last_project_kernel();      // name of the kernel the last VG_GEMM_TAG_GRAM_PROJECT launch dispatched

// segment reduction: out[i] = sum_s in[s*slab + i]
#define VG_RED_MAXSEG 8
struct VgRedSeg { const double* in; double* out; long n; long slab; int nslab; int block_start; };
struct VgRedBatch { int nseg; int total_blocks; VgRedSeg s[VG_RED_MAXSEG]; };
void vg_red_init(VgRedBatch* b);
void vg_red_add(VgRedBatch* b, const double* in, double* out, long n, long slab, int nslab);
hipError_t vg_red_launch(const VgRedBatch* b, hipStream_t st);

// ---- factor build (factor_build.hip) ------------------------------------------
struct VgFactorJob {
    const double* x;      // [n] observation coordinates (device)
    const double* grid;   // mesh [m+1] or inducing coords [m] (device)
    double* A0;           // [m][n]   (may be null)
    double* dA0;          // [m][n]
    double* K0;           // [m][m]   (may be null)
    double* dK0;          // [m][m]
    int n, m, kind, basis;
    int theta_idx;        // index of ell in the device theta vector, or -1 -> use ell_imm
    double ell_imm;
    int flags;            // VGGP_FLAG_*
};
struct VgClearArgs;
// theta may be a device pointer or device-visible pinned host memory; when theta_copy is given the kernel also copies the
// 5 hyper-parameters there (for the later kernels of the step) and block 0 zeroes the words listed in clr -- this makes
// the factor build the FIRST node of the step graph (no H2D memcpy node, no separate clear launch).
hipError_t vg_factor_build_launch(const VgFactorJob* jobs, int njobs, const double* theta_dev, hipStream_t st,
                                  double* theta_copy = nullptr, const VgClearArgs* clr = nullptr);

// ---- Cholesky + explicit inverse (chol.hip) ------------------------------------
struct VgCholJob {
    const double* K;      // [m][ldk] (ldk = 0 -> m)
    double* L;            // [m][ldl] lower, zero above (ldl = 0 -> m)
    double* Linv;         // [m][m] lower, zero above
    double* scratch;      // [m][m+1] global work area (used when m does not fit LDS)
    double* jitter_out;   // device scalar
    int* status;          // device int (0 ok, VGGP_ENOTPD)
    int m;
    int ldk = 0, ldl = 0; // row strides of K and L when they are sub-blocks of larger matrices
    int only_level0 = 0;  // 1: factor as is (no jitter levels); used for the well-conditioned Sigma~ blocks
    double* Dinv_out = nullptr;   // optional [ceil(m/16)][16][16]: inverses of the diagonal blocks (m <= 128 path); with Linv = nullptr
                                  // the full inverse is not formed at all
};
struct VgGemmBatch;
hipError_t vg_chol_launch(const VgCholJob* jobs, int njobs, hipStream_t st, const VgGemmBatch* rider = nullptr);   // rider: m <= 128 path only
hipError_t vg_chol_setup();   // opt-in to large dynamic LDS
// 128 < m <= 256 (api.hip vg_chol_big_enqueue): four jittered copies of K; selection of the lowest level that survived both blocks
hipError_t vg_jitcopy_launch(const double* K, int m, double* Kc, hipStream_t st);
hipError_t vg_cholsel_launch(const double* Lc, const double* Dc, const int* st8, int m, double* L0, double* Dinv0, double* jitter_out,
                             int* status, hipStream_t st);

// ---- triangular solves by substitution (trsm.hip) -------------------------------------------------------------------
struct VgTrsmJob {
    const double* L;      // [m][ldl] lower-triangular Cholesky factor
    const double* Dinv;   // inverses of its 16 x 16 diagonal blocks: block b at Dinv + b * dinv_blk, row stride dinv_ld
                          // (the diagonal blocks of an explicit inverse [m][m]: dinv_blk = 16 * m + 16, dinv_ld = m)
    const double* R;      // right-hand sides: element (row k, column c) at R[k * r_sk + c * r_sc]
    double* X;            // solution, element (k, c) at X[k * x_sk + c * x_sc]  (may alias R)
    long ldl, dinv_blk, dinv_ld;
    long r_sk, r_sc, x_sk, x_sc;
    long ncols;
    int m;                // <= 128 (larger factors are blocked by the caller, api.hip trsm_batch)
    int trans;            // 0: L X = R, 1: L^T X = R
    int rhs_ident = 0;    // 1: R is the m x m identity (generated in registers, R is not read): X = L^{-1} resp. L^{-T}
};
hipError_t vg_trsm_launch(const VgTrsmJob* jobs, int njobs, hipStream_t st);
hipError_t vg_trsm_setup();
hipError_t vg_tri_diaginv_launch(const double* L, long ldl, int m, double* out /* [ceil(m/16)][16][16] */, hipStream_t st,
                                 const double* L2 = nullptr, long ldl2 = 0, int m2 = 0, double* out2 = nullptr);   // optional second matrix

// ---- Jacobi eigensolver (eigh.hip) ---------------------------------------------
#define VG_EIG_MAXSWEEP 60
struct VgEigJob {
    const double* G;      // [m][m] symmetric
    double* lam;          // [m]
    double* Qt;           // [m][m], row j = eigenvector j
    const double* Qt0;    // optional start basis (warm start): G is then Qt0 G Qt0^T already
    double* gwork;        // [m2][m2+1] global G work area when it does not fit LDS
    double2* rotlog;      // [max_rounds][m2/2] (c,s)
    int* roundlog;        // [max_rounds] round index r of each logged round
    int* counters;        // [4]: nlog, sweeps, status, pad
    int m;
    int max_rounds;
    long log_bytes;       // size of the rotlog buffer in bytes (vg_eigh_log_bytes(m))
    int block;            // 1: block-Jacobi variant (m <= 128), 0: scalar cyclic Jacobi
    double tol = 0.0;     // off-diagonal threshold relative to ||G||_F / m (0: VG_EIG_TOL)
    double* Qt2 = nullptr; // optional second copy of Qt (next step's warm-start basis; may alias Qt0)
    const double* cp_src = nullptr;  // optional: each replay workgroup first copies its column slice cp_src -> cp_dst
    double* cp_dst = nullptr;        // (keeps the basis before last for the extrapolated warm start; scalar variant only)
    int* perm = nullptr;   // [m] scratch: rank of eigenpair i in decreasing order (scalar variant; null = leave unsorted)
    int fast_switch = 112; // fixed-address dense sweeps (m2 <= 128) while >= fast_switch/256 of a sweep's pairs rotate; 0 = off
    int polish = 1;        // dense phase: replace the remaining sweeps by a first-order polish when its a-priori bound allows
    int* err = nullptr;    // optional device word the replay workgroups OR a 1 into when they give up waiting for the producer
    int polish0 = 0;       // also try the polish before the first sweep (the start basis was refined by vg_refine_launch)
    const double* Hl = nullptr;  // the matrix as a product G = Hl Hr^T (m x hk row-major factors; LDS variant only): formed by the
    const double* Hr = nullptr;  // producer workgroup itself on the matrix cores; `G` is then ignored
    int hk = 0;
    int null_from = 0;     // subspace start: rows >= null_from of G must be numerically null (else bit 1 of *err is raised)
    int newton = 0;        // nearly diagonal small problem (m <= 48, no start basis): try the Newton start first (vg_newton_diag)
    int sparse_first = 0;  // skip the dense phase: the start basis already block-diagonalises G (subspace start), a few elements remain
};
#define VG_EIG_RANK_CUT 1e-14   // eigenvalues above this fraction of the largest count towards the numerical rank (counters[1] >> 8)
// Row orthonormalisation for the subspace start (eigh.hip): V1 = rows of Z (r x m, r <= 64, m <= 128) orthonormalised in the
// given order (classical Gram-Schmidt, re-orthogonalised), one workgroup per job; also copies cp_src -> cp_dst (cp_n doubles).
struct VgRowQrJob { const double* Z; double* V1; int r; int m; const double* cp_src; double* cp_dst; long cp_n; };
// `rider` (both launchers): a GEMM batch executed by extra workgroups of the same launch, beside the single-workgroup jobs
struct VgGemmBatch;
hipError_t vg_rowqr_launch(const VgRowQrJob* jobs, int njobs, hipStream_t st, const VgGemmBatch* rider = nullptr);
hipError_t vg_identity_launch(double* A, int m, hipStream_t st);
// First-order refinement of a warm start (eigh.hip): from Gw = S G S^T, E_ij = g_ij / (g_ii - g_jj) for the elements above
// the eigensolver's threshold; outputs E and R1 = I + E (both [m][m]); E = 0, R1 = I when some |E_ij| > 1e-3.
struct VgRefineJob { const double* Gw; double* E; double* R1; int m; double tol;
                     double emax = 0.0;      // largest rotation accepted (0: VG_POLISH_EMAX = 1e-3, the polish's limit)
                     int* flag = nullptr;    // optional: bit 1 is OR-ed in when the start was rejected (some |E_ij| > emax, NaN)
                     double noise = 0.0;     // > 0: pairs whose two diagonal entries are both below noise * max diagonal are left alone
                                             // (a cluster at the rounding floor: any orthonormal basis of it serves -- what the ELBO sees of
                                             // it is its trace -- and its quotients g_ij / (g_jj - g_ii) are noise over noise)
                     const double* lam_prev = nullptr;   // optional: the previous step's eigenvalues in the row order of the start basis; pairs
                     double null_cut = 0.0;              // of rows that were BOTH below null_cut * largest then are left alone (a null space --
                                                         // B1 hats without data, the Fourier features' null combinations -- turns with the
                                                         // hyper-parameters, so its block of Gw is delta^2 lam_range with O(1) quotients inside,
                                                         // but only the range-null rotations matter: any basis of the null space serves)
};
// Newton chain (api.hip finish_enqueue): after the last iteration -- eigenvalues = diag(Gw), convergence check (largest
// off-diagonal element against tol * ||Gw||_F / m), numerical rank; counters [0] = 0, [1] = iterations | rank << 8, [2] = 0,
// [3] = 0; bit 1 of *err (-> VG_ESUBMISS: the host repeats the step on the regular chain) when not converged or rejected.
struct VgNewtonCheckJob { const double* Gw; double* lam; int* counters; int* err; const int* flag; int m; int iters; double tol; double noise;
                          double null_cut = 0.0;   // > 0: pairs of rows whose PREVIOUS eigenvalues (lam on entry) were both below null_cut * largest are not judged
};
hipError_t vg_newton_check_launch(const VgNewtonCheckJob* jobs, int njobs, hipStream_t st);
hipError_t vg_copy_if_launch(const int* const* err, const double* const* src_a, double* const* dst_a, const double* const* src_b,
                             double* const* dst_b, const long* n, int njobs, hipStream_t st);
hipError_t vg_refine_launch(const VgRefineJob* jobs, int njobs, hipStream_t st, const VgGemmBatch* rider = nullptr);
size_t vg_eigh_log_bytes(int m);     // log capacity needed for an m x m problem (scalar or block variant)
hipError_t vg_eigh_launch(const VgEigJob* jobs, int njobs, hipStream_t st, const VgGemmBatch* rider = nullptr);
hipError_t vg_eigh_setup();

// ---- m-space elementwise / reductions (mspace.hip) -----------------------------
#define VG_ESUBMISS (-100)   // internal status: the subspace start of the eigensolver chain missed part of the range (never returned
                            // to the caller: vggp_elbo_step / vggp_elbo_finish repeat the step with a cold start)
struct VgHostOut {          // pinned readback block (one 128-byte burst)
    double out[8];
    double jitter[2];
    int counters[2][4];
    int status[2];
    double seq;             // sequence number of the step that wrote the block (host checks it against the one it sent)
};
struct VgMspace {
    // inputs
    const double* theta;     // device [5]
    const double* lam1;      // unit-scale eigenvalues [m1]
    const double* lam2;      // [m2]
    const double* P3;        // [3][m1][m2] unit-scale P0, P1_0, P2_0
    const double* E1;        // [m1][m1]  Q1^T Mk1 Q1
    const double* F1;        // [m1][m1]  Q1^T H0_1 Q1 (unsymmetrised, unit scale)
    const double* E2;
    const double* F2;
    // produced by dstage
    double* beta;            // [m1][m2]
    double* bl2;             // beta * lam2[None,:]
    double* bl1;             // beta * lam1[:,None]
    double* invD;            // [m1][m2]
    double* rowpart;         // [m1][8] per-row partial scalars
    double* r1;              // [m1] sum_i2 1/D
    double* r1l;             // [m1] sum_i2 lam2/D
    double* ol;              // [2][m1]: ones; s1 lam1   (left by the D-stage for the column-sum product [r2; r2l] = ol invD)
    const double* dotp;      // [4][64] per-tile partial sums of the four dot products (reduction epilogue of the beta launch)
    double* r2;              // [m2] sum_i1 1/D        (beta launch: row 0 of ol invD)
    double* r2l;             // [m2] sum_i1 lam1/D
    double* dotpart;         // [64][4] slices of sum(E1 o X1), sum(F1 o X1l), sum(E2 o X2), sum(F2 o X2l)
    // outputs
    double* out;             // [8]: elbo, g_ell1, g_ell2, g_s1, g_s2, g_v, -, -
    int m1, m2;
    double n_total;
    double yy;
    // fused tail: the last workgroup of vg_partial_kernel (ticket) does the final combination and writes results and
    // diagnostics straight into the pinned host block (no D2H memcpy nodes in the step graph)
    int* ticket = nullptr;           // device int, zero between launches
    VgHostOut* hout = nullptr;       // device-visible address of the pinned readback block (may be null)
    const double* peer_fail = nullptr;   // multi-rank step: the failure word behind the all-reduced payload (-> VgHostOut::out[6])
    const double* jit[2] = {nullptr, nullptr};
    const int* status[2] = {nullptr, nullptr};
    const int* counters[2] = {nullptr, nullptr};
};
hipError_t vg_dstage_launch(const VgMspace* ms, hipStream_t st);
// "Thin" m-space stage of a subspace-start step (thin.hip): everything from the Ritz pairs to the pinned result block in ONE
// single-workgroup kernel, range directions only (r1, r2 <= 32); all inputs at unit outputscale.
#define VG_THIN_MAXR 32
#define VG_THIN_ELBO_TOL 1e-10 // ... and its first-order effect on the bound, relative to the number of observations
#define VG_THIN_MISS 1e-12     // admissible (tr G - sum of the Ritz values) per complement direction, relative to lam_max
struct VgThinTail {
    const double* theta;             // device [6]: ell1, ell2, s1, s2, v, sequence number
    double n_total, yy;
    int r1, r2, m1, m2;
    const double *W1, *W2;           // [r][r] row-major: row j = Ritz vector j in V1 coordinates (sorted by decreasing Ritz value)
    const double *lam1, *lam2;       // [r] Ritz values (unit scale, decreasing)
    const double *AM1, *AH1, *AM2, *AH2;   // [r][r]: V1 Mk0 V1^T, V1 H0 V1^T
    const double* AC;                // [3][r1][r2]: V1_1 {C, C1, C2} V1_2^T  (ac_nslab split-K slabs, ac_slab doubles apart)
    int ac_nslab; long ac_slab;
    const double *G1, *H1, *G2, *H2; // [m][m] reduced Gram pairs (only their diagonals are read: traces over ALL directions)
    double* out;                     // device [8] (may be null)
    double *beta_out, *invd_out;     // [r1][r2] (compact): beta and 1/D on range x range, for the read-outs (may be null)
    VgHostOut* hout;                 // pinned result block
    const double* peer_fail;         // see VgMspace
    const double* jit[2];
    const int* status[2];            // [2] per dimension: Cholesky status, eigensolver error bits
    const int* rcounters[2];         // the Ritz solves' counters [4]
    unsigned long long* stamps = nullptr;   // diagnostic builds (-DVGGP_DIAG): s_memrealtime stamps of the kernel's phases
};
hipError_t vg_thin_tail_launch(const VgThinTail* tt, hipStream_t st);
// cold range finder (thin.hip): r steps of diagonally pivoted Cholesky of G (m x m, m <= 256); the columns become the rows of V (r x m)
struct VgPivCholJob { const double* G; const double* Omega; double* V; int m; int r; };
struct VgPivCholArgs { VgPivCholJob job[2]; };
hipError_t vg_pivchol_launch(const VgPivCholJob* jobs, int njobs, hipStream_t st);
hipError_t vg_thin_tail_setup();
hipError_t vg_final_launch(const VgMspace* ms, hipStream_t st);
// inducing-point gradient (vggp_zgrad): weights of the gradient functional, and the row contraction with d kappa / d z
hipError_t vg_zw_launch(const double* theta, int self, const double* lam_self, const double* lam_other, int m, int m_other,
                        const double* X, const double* Xl, const double* r, const double* rl, double* WE, double* WFs, hipStream_t st);
hipError_t vg_zdot_launch(const double* theta, int self, const double* z, const double* x, int m, long n, const double* Abar,
                          const double* dA0, const double* Kbar, const double* dK0, double* out, hipStream_t st);

// zero a handful of small device buffers with ONE kernel (used instead of hipMemsetAsync: memset nodes of a
// captured graph were observed to replay with a wrong fill value after an unrelated hipMalloc on ROCm 7.2)
#define VG_CLEAR_MAX 12
struct VgClearArgs { int* ptr[VG_CLEAR_MAX]; int nwords[VG_CLEAR_MAX]; int n; };
hipError_t vg_clear_launch(const VgClearArgs* a, hipStream_t st);

// q(v) / posterior helpers
hipError_t vg_readout_weights_launch(const double* theta, const double* beta, const double* invD, double* w, long n, int literal,
                                     hipStream_t st);
hipError_t vg_readout_var_launch(const double* theta, const double* kd1, const double* kd2, long mv1, long mv2, double* var, hipStream_t st);
hipError_t vg_scale_sq_launch(const double* in, double* out_sq, long n, hipStream_t st);
hipError_t vg_qv_weights_launch(const double* theta, const double* beta, const double* invD, double* w_mean,
                                long n, hipStream_t st, int e1 = 1, int e2 = 1);
hipError_t vg_sumsq_launch(const double* y, long n, double* partial, double* out, hipStream_t st);
hipError_t vg_post_combine_launch(const double* theta, const double* T1, const double* T2, const double* beta,
                                  const double* invD, int m1, int m2, long ns, double* mean, double* var,
                                  hipStream_t st);
hipError_t vg_scale_launch(double* x, long n, const double* theta, int mode, hipStream_t st, int e1 = 1, int e2 = 1);
