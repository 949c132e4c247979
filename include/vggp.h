/*
 * vggp.h -- C-ABI of libvggp_hip.so: the MI355X (gfx950) engine for the
 * Kronecker-structured collapsed-ELBO hot path of
 * maxnorman569/Variational-Gridded-Gaussian-Processes.
 *
 * The reference has NO FFI / plugin boundary (it is plain Python classes on
 * gpytorch); this boundary is defined by the build.  Each entry
 * point cites the reference code it replaces (paths relative to the reference
 * checkout).  Host code (Python/ctypes) owns every data
 * buffer; pointers marked DEVICE are hipMalloc'ed (e.g. torch tensor.data_ptr()),
 * contiguous float64.  The library owns only the opaque context (workspace arena,
 * HIP-graph cache).  Every function returns 0 on success or a negative VGGP_E*
 * code; the message is available from vggp_last_error() (thread-local).  Nothing
 * throws across the boundary.  One context per (process, device); not re-entrant.
 * All work is enqueued on the hipStream_t passed as `stream` (void*, 0 = default).
 *
 * Conventions
 *   theta[5] = { ell_1, ell_2, s_1, s_2, sigma2 }  (constrained values:
 *              lengthscales, outputscales, likelihood.noise -- a variance,
 *              kronecker_structure.py:263)
 *   Y        : observations on the local grid shard, [n2][n1] row-major,
 *              Y[j][i] = y(x1[i], x2[j])  (x1 fastest: utils/datagenerators.py:70-72)
 *   inducing index u = i1*m2 + i2 (kronecker_structure.py:805, :822)
 *   multi-GPU: one process (one context) per GPU; the grid is sharded along the slow storage axis (rows j of Y,
 *              i.e. dimension 2); dimension 1 and all m-space algebra are replicated.  The context OWNS the collective
 *              (an RCCL communicator created in vggp_create, or a host callback): vggp_elbo_step on an n_ranks > 1
 *              context is  partials -> ONE sum all-reduce of the packed payload -> finish  on one stream with one host
 *              synchronisation, and every rank returns the identical value and gradient.
 */
#ifndef VGGP_H
#define VGGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGGP_VERSION 200          /* 0.2.0 */

/* error codes */
#define VGGP_OK        0
#define VGGP_EINVAL   -1          /* bad argument / shape                          */
#define VGGP_ENOTPD   -2          /* a factor is not PD after the jitter schedule  */
#define VGGP_EHIP     -3          /* HIP runtime error                             */
#define VGGP_ENOMEM   -4
#define VGGP_ESTATE   -5          /* call order (e.g. finish before partials)      */
#define VGGP_ENOCONV  -6          /* Jacobi eigensolver hit its sweep limit        */
#define VGGP_ERCCL    -7          /* the collective failed (RCCL error, RCCL not loadable, callback error) */

/* kernel family of one dimension (gpytorch MaternKernel(nu) / the build's RBF) */
#define VGGP_KIND_MATERN12 0
#define VGGP_KIND_MATERN32 1
#define VGGP_KIND_MATERN52 2
#define VGGP_KIND_RBF      3

/* inducing-feature basis of one dimension */
#define VGGP_BASIS_POINTS 0       /* pairwise k(z, x): Matern12SVGP, kronecker_structure.py:306-338 */
#define VGGP_BASIS_B0     1       /* B0-spline cell integrals (Matern-1/2 only):
                                     kronecker_structure.py:702-790 == gridded_kronecker_structure.py:1286-1374 */
#define VGGP_BASIS_ONE    2       /* trivial factor K=[1], A=1: turns the engine into the 1-D model,
                                     univariate_structure.py:234-263, :693-717 */
#define VGGP_BASIS_VFF    3       /* variational Fourier features (Matern-1/2 only): Matern12VFFGP,
                                     kronecker_structure.py:346-515.  grid = [a, b, omega_0 .. omega_M], m = 2M + 1.
                                     Kuu_d = K0(ell) / s_d, Kuf_d = features(x) (no s_d). */
#define VGGP_BASIS_B1     4       /* B1-spline (hat function) features (Matern-1/2 only): Matern12B1SplineASVGP,
                                     kronecker_structure.py:524-660.  grid = knot mesh (m knots).
                                     Kuu_d = (A ell + B / ell + BC) / (2 s_d), Kuf_d = hats(x). */

/* The reference keeps the B0 mesh and delta as float32 attributes (torch.linspace default
 * dtype; Module.to(float64) does not touch them), so `(k - 1) * delta` in _Kuu_along_dim
 * (kronecker_structure.py:731-733) is rounded to float32 before the float64 division by the
 * lengthscale.  With this flag the B0 Kuu builder reproduces that rounding (and the literal
 * three-exponential form) so results match the reference bit-for-bit in the inputs. */
#define VGGP_FLAG_B0_F32_KDELTA 1
/* Use the block-Jacobi eigensolver (16-wide index blocks, MFMA super-block updates) instead of the scalar cyclic
 * Jacobi for m <= 128.  Same results to rounding; measured slower on RBF factors and ~10 % faster on Matern factors
 * at m = 128, hence off by default. */
#define VGGP_FLAG_BLOCK_JACOBI 2
#define VGGP_FLAG_SCATTERED 4      /* x1[k], x2[k] are the coordinates of N = n1 = n2 scattered POINTS (not grid axes): only        */
                                   /* vggp_elbo_step_scattered and the *_masked read-outs apply                                      */

typedef struct vggp_ctx vggp_ctx;

/* Problem description (host struct; the small coordinate arrays are HOST pointers and
 * are copied into the context by vggp_plan). */
typedef struct vggp_desc {
    int32_t kind1, basis1;        /* dimension 1 (x1, fast axis of Y)                 */
    int32_t kind2, basis2;        /* dimension 2 (x2, slow axis of Y; sharded axis)   */
    int64_t n1, n2;               /* LOCAL grid: Y is [n2][n1]                        */
    int64_t m1, m2;               /* inducing features per dimension                  */
    int64_t n_total;              /* number of observations over ALL ranks (N)        */
    const double* x1;             /* HOST [n1]  unique coordinates along dim 1        */
    const double* x2;             /* HOST [n2]  local coordinates along dim 2         */
    const double* grid1;          /* HOST: mesh [m1+1] (B0) or inducing coords [m1]   */
    const double* grid2;          /* HOST: mesh [m2+1] (B0) or inducing coords [m2]   */
    int32_t warm_start;           /* 1: reuse the previous step's eigenvectors        */
    int32_t flags;                /* VGGP_FLAG_* bit mask                             */
} vggp_desc;

/* per-step diagnostics (host struct filled by vggp_elbo_finish / vggp_elbo_step) */
typedef struct vggp_info {
    double jitter1, jitter2;      /* jitter actually added to the unit-scale factors  */
    int32_t sweeps1, sweeps2;     /* Jacobi sweeps used                               */
    int32_t rounds1, rounds2;     /* Jacobi rotation rounds applied                   */
    int32_t status;               /* 0 or a VGGP_E* code detected on the device       */
    int32_t polished;             /* bit d-1: dimension d's eigensolver ended in a first-order polish */
} vggp_info;

int         vggp_version(void);
const char* vggp_last_error(void);

/* lifecycle ---------------------------------------------------------------- */
/* One context per (process, GPU).  n_ranks = 1: single GPU (rank, unique_id ignored).  n_ranks > 1: this process is rank
 * `rank` of a row-sharded job; `unique_id` (VGGP_UNIQUE_ID_BYTES bytes, generated on rank 0 by vggp_unique_id and
 * distributed to the other ranks by the host program) creates the context's RCCL communicator (collective call: every
 * rank must be inside vggp_create).  unique_id = NULL with n_ranks > 1 creates no communicator: the caller installs a host
 * transport with vggp_set_allreduce before the first step.  (A unique id with n_ranks = 1 creates a communicator of size
 * one and the step runs the multi-rank sequence through it: the RCCL path on a one-GPU box.) */
#define VGGP_UNIQUE_ID_BYTES 128
int vggp_create(vggp_ctx** out, int device, int n_ranks, int rank, const void* unique_id);
int vggp_destroy(vggp_ctx* ctx);
int vggp_unique_id(void* out /* VGGP_UNIQUE_ID_BYTES */);
/* Host-callback transport: fn sums `count` doubles at the HOST address `buf` over all ranks, in place, and returns 0.
 * The library stages the payload through pinned memory around the call.  Used by the multi-rank rehearsal on one GPU
 * (gloo; RCCL refuses several ranks on one device) and as the seam for other transports. */
typedef int (*vggp_allreduce_fn)(void* user, double* buf, int64_t count);
int vggp_set_allreduce(vggp_ctx* ctx, vggp_allreduce_fn fn, void* user);
/* The context's sum all-reduce on a DEVICE buffer (in place; returns after the result is complete). */
int vggp_allreduce(vggp_ctx* ctx, double* buf, int64_t count, void* stream);
/* n_ranks, rank and the transport in use (0 none, 1 RCCL, 2 host callback); any output may be NULL. */
int vggp_comm_info(const vggp_ctx* ctx, int* n_ranks, int* rank, int* transport);
/* (Re)plan the context for a problem: allocates the workspace arena (no allocation
 * happens per step afterwards) and uploads coordinates / meshes.
 * Replaces: KroneckerStructure.__init__ + Matern12GriddedGP.__init__ bookkeeping
 * (kronecker_structure.py:19-32, gridded_kronecker_structure.py:1259-1284). */
int vggp_plan(vggp_ctx* ctx, const vggp_desc* desc);
/* Number of doubles in the all-reduce payload {G2,H2,C,C1,C2} of the planned problem. */
int64_t vggp_payload_len(const vggp_ctx* ctx);
int64_t vggp_workspace_bytes(const vggp_ctx* ctx);

/* the hot path ------------------------------------------------------------- */
/* One ELBO step = value + gradient w.r.t. theta.  Replaces KroneckerStructure._elbo
 * (kronecker_structure.py:249-278) AND the autograd backward of the notebook loop
 * (5_gridded_kronecker_structure_models.ipynb cell 26).
 *   Y        DEVICE [n2][n1]  (this rank's row slab)
 *   yy_total sum of y^2 over ALL ranks (constant of the data; vggp_sumsq returns it)
 *   elbo_out, grad_out[5]  HOST outputs (the only host sync of the step)
 * Single-rank convenience = vggp_elbo_partials + vggp_elbo_finish. */
int vggp_elbo_step(vggp_ctx* ctx, const double* Y, double yy_total, const double theta[5],
                   double* elbo_out, double grad_out[5], vggp_info* info, void* stream);

/* The two halves of the step, for callers that carry the all-reduce themselves (vggp_elbo_step on a multi-rank context
 * does all three): partials fills `payload` (DEVICE, vggp_payload_len doubles) with this rank's contribution; the caller
 * sums it over ranks with ONE all-reduce and passes the reduced buffer to finish. */
int vggp_elbo_partials(vggp_ctx* ctx, const double* Y, const double theta[5],
                       double* payload, void* stream);
int vggp_elbo_finish(vggp_ctx* ctx, const double* payload, double yy_total, const double theta[5],
                     double* elbo_out, double grad_out[5], vggp_info* info, void* stream);

/* Masked / partially observed grid (BASELINE config 5).  Ym = W o Y and W (0/1 as float64) are DEVICE [n2][n1] (this rank's
 * row slab); n_obs = sum(W), yy_obs = sum(Ym^2) over ALL ranks.  Phi = Kuf W Kuf^T is assembled in M-space (M = m1 m2 <= 16384)
 * and factored densely; value + analytic gradient as for vggp_elbo_step.  Replaces KroneckerStructure._elbo
 * (kronecker_structure.py:249-278) called with the observed subset of the grid as X, y.
 * Multi-rank context: every rank assembles the partial Phi_r (and its two lengthscale derivatives, the projections and a
 * column statistic) of its rows, ONE all-reduce (3 M^2 + 3 M + n1 doubles) makes Sigma~ and its factorisation replicated,
 * and a second all-reduce of 21 scalars closes the gradient terms that are sums over grid rows. */
int vggp_elbo_step_masked(vggp_ctx* ctx, const double* Ym, const double* W, double n_obs, double yy_obs,
                          const double theta[5], double* elbo_out, double grad_out[5], vggp_info* info, void* stream);
/* The masked step WITHOUT the M x M matrices -- for M = m1 m2 beyond the dense solver (M > 16384) or when O(M^3) is too slow:
 * what gpytorch does for the reference above max_cholesky_size = 800 (`inv_matmul` by CG, `log_prob` by stochastic Lanczos
 * quadrature, kronecker_structure.py:269, :273), here with matricised Kronecker MVMs  Sigma~ V = V + rho B1 (W^T o (B1^T V B2)) B2^T,
 * the Kronecker-eigenbasis preconditioner P = I + rho p G1 (x) G2 (p = observed fraction), log|Sigma~| = log|P| + Lanczos
 * quadrature of the PCG coefficients of n_probes FIXED Rademacher probes, and derivative traces as closed form + control-variate
 * probe estimator.  Bitwise reproducible.  Stated tolerance against vggp_elbo_step_masked: ELBO 1e-5 relative, gradient 1e-4 of
 * its largest component (n_probes = 16).  n_probes <= 0: 16; tol <= 0: 1e-10 (PCG residual, relative); max_iter <= 0: 100.
 * Arguments otherwise as vggp_elbo_step_masked; single-rank contexts; info->rounds1 = PCG iterations, info->sweeps1 = probes.
 * The dense read-outs (vggp_qv_masked, ...) need the dense step. */
int vggp_elbo_step_masked_iter(vggp_ctx* ctx, const double* Ym, const double* W, double n_obs, double yy_obs,
                               const double theta[5], int n_probes, double tol, int max_iter, double* elbo_out, double grad_out[5],
                               vggp_info* info, void* stream);
/* q(v) of the last masked step: mean and covariance diagonal, DEVICE [m1][m2]. */
int vggp_qv_masked(vggp_ctx* ctx, double* mean, double* var, void* stream);
/* posterior(x*) of the last masked step (kronecker_structure.py:199-230); arguments as vggp_posterior. */
int vggp_posterior_masked(vggp_ctx* ctx, const double* xs1, const double* xs2, int64_t n_star, double* mean, double* var,
                          void* stream);

/* Gridded read-out q(v) of B0 cell features v from the posterior over the inducing features u of the last finished step
 * (q_u -> p(v|u) -> q_v; gridded_kronecker_structure.py:396-438 SVGP, :613-654 VFF, :903-947 ASVGP), Kronecker in the
 * per-dimension cross-covariances: C_d DEVICE [mv_d][m_d] = Cov(v, u) along dimension d at UNIT outputscale (Kvu_d / s_d
 * for kernel-evaluated features, Kvu_d itself for VFF / B1 features), kd_d DEVICE [mv_d] = diag(Kvv_d) / s_d.
 * mean, var DEVICE [mv1][mv2]:  mean = Kvu Kuu^-1 mu_u;  var = diag(Kvv - Kvu Kuu^-1 Kuv + Kvu X Kuv) with
 * X = S_u^-1 when flags & VGGP_READOUT_LITERAL (what the reference's q_v computes, :431) and X = Kuu^-1 S_u Kuu^-1
 * (the conditional variance of v under q(u)) otherwise. */
#define VGGP_READOUT_LITERAL 1
int vggp_readout(vggp_ctx* ctx, const double* C1, int64_t mv1, const double* C2, int64_t mv2, const double* kd1, const double* kd2,
                 double* mean, double* var, int flags, void* stream);

/* The same read-out from the M-space state of the last MASKED or SCATTERED step (the Gridded* models on data that is no full
 * grid: along-track observations, notebook 61); arguments as vggp_readout. */
int vggp_readout_masked(vggp_ctx* ctx, const double* C1, int64_t mv1, const double* C2, int64_t mv2, const double* kd1, const double* kd2,
                        double* mean, double* var, int flags, void* stream);

/* q(v) of the last finished step: mean and diagonal of the covariance, both DEVICE
 * [m1][m2] (flat index u = i1*m2+i2).  Replaces Matern12GriddedGP.q_v
 * (gridded_kronecker_structure.py:1409-1433 == kronecker_structure.py:825-849). */
int vggp_qv(vggp_ctx* ctx, double* mean, double* var, void* stream);
/* Dense M x M covariance Kuu Sigma^{-1} Kuu of q(v) (DEVICE, M = m1*m2; small M only). */
int vggp_qv_cov(vggp_ctx* ctx, double* cov, void* stream);

/* Gradient of the ELBO of the LAST vggp_elbo_step with respect to the inducing-point coordinates of the "points" basis
 * (Matern12SVGP and friends register Z as a trainable Parameter, kronecker_structure.py:303-304, and let autograd differentiate
 * through kernel(Z) :318-319 and kernel(cartesian_prod(Z), x) :336-337).  Y: the same device array the step was given.
 * gz1 [m1], gz2 [m2] (device): d ELBO / d z_d[i]; zeros for a dimension whose basis is not VGGP_BASIS_POINTS.
 * Analytic (no autograd): the sensitivities Kbar = L^-T W_M L^-1, Abar = L^-T W_V follow from the linearity of the lengthscale
 * gradient in (dK, dA), and d kappa(z, x)/dz = -(d kappa/d ell) ell / (z - x) for the stationary kernels.
 * Row-sharded contexts: Y is the rank's slab; the parts of the ranks' rows are summed by one all-reduce of m1 + m2 doubles. */
int vggp_zgrad(vggp_ctx* ctx, const double* Y, double* gz1, double* gz2, void* stream);

/* Collapsed ELBO and gradient for N SCATTERED observations (along-track points: the reference's _elbo(), kronecker_structure.py
 * :249-278, receives arbitrary (x1, x2) pairs in notebooks 6 / 61 / 7 and evaluates Kuf densely, :808-823).  The context must have
 * been planned with VGGP_FLAG_SCATTERED (x1[k], x2[k] = the coordinates of point k, n1 = n2 = N).  y [N] device, yy = sum y^2.
 * Kuf[:, k] = a1(x1_k) (x) a2(x2_k) is a Khatri-Rao product: Sigma~ = I + rho sum_k (b1_k (x) b2_k)(b1_k (x) b2_k)^T is assembled
 * in M-space (M = m1 m2 <= 16384) by one GEMM over the points and factored densely, as in the masked step; cost O(M^2 N + M^3).
 * vggp_qv_masked / vggp_posterior_masked / the *_cov_masked entries read the result. */
int vggp_elbo_step_scattered(vggp_ctx* ctx, const double* y, double yy, const double theta[5], double* elbo_out,
                             double grad_out[5], vggp_info* info, void* stream);

/* Gradient of the scattered ELBO w.r.t. the inducing coordinates (what the reference obtains from autograd through _elbo()
 * into the Z Parameter of its SVGP classes, kronecker_structure.py:303-304, when X holds scattered points), after
 * vggp_elbo_step_scattered on the same y: gz1 [m1], gz2 [m2] (DEVICE; zeros for a dimension that does not use the points
 * basis).  One more M x M x N product; workspace 2 M N doubles (M N < 2^31).  Point-sharded contexts: the parts of the ranks'
 * points are summed by one all-reduce of m1 + m2 doubles. */
int vggp_zgrad_scattered(vggp_ctx* ctx, const double* y, double* gz1, double* gz2, void* stream);

/* New inducing coordinates z[0..m) (host array) for dimension dim (0 or 1) of a planned context whose basis there is
 * VGGP_BASIS_POINTS, without re-planning: arena, captured graphs and the eigensolver's warm start are kept.  What an optimiser
 * that trains Z (kronecker_structure.py:303-304 registers it as a Parameter) calls between steps.  Read-outs need a new step. */
int vggp_set_inducing(vggp_ctx* ctx, int dim, const double* z, int64_t m);

/* Point-wise posterior at ns scattered test points (xs1[p], xs2[p]) (DEVICE inputs):
 * mean[ns], var[ns] (DEVICE).  Replaces KroneckerStructure.posterior mean and the
 * diagonal of its covariance (kronecker_structure.py:199-230). */
int vggp_posterior(vggp_ctx* ctx, const double* xs1, const double* xs2, int64_t ns,
                   double* mean, double* var, void* stream);

/* Dense ns x ns covariance of posterior(x*) (DEVICE [ns][ns]; ns <= 8192 and M ns <= 2^27): K** + Kuf*^T Sigma^-1 Kuf* -
 * Kuf*^T Kuu^-1 Kuf*, kronecker_structure.py:223-229, from the step's eigenbasis (never forming Sigma); the masked variant
 * from the dense Sigma~^-1 of the last masked step (ns <= M).  Callers that only need the diagonal use vggp_posterior. */
int vggp_posterior_cov(vggp_ctx* ctx, const double* xs1, const double* xs2, int64_t ns, double* cov, void* stream);
int vggp_posterior_cov_masked(vggp_ctx* ctx, const double* xs1, const double* xs2, int64_t ns, double* cov, void* stream);
/* Dense M x M covariance of q(v) of the last masked step (Kuu Sigma^-1 Kuu on the observed subset). */
int vggp_qv_cov_masked(vggp_ctx* ctx, double* cov, void* stream);

/* building blocks (exported for tests, benchmarks and re-use) ---------------- */
/* Unit-outputscale factor build for one dimension: A0[m][n], dA0/d ell [m][n],
 * K0[m][m], dK0/d ell [m][m] (any output pointer may be NULL).  x DEVICE [n];
 * grid DEVICE ([m+1] mesh for B0, [m] coords for POINTS).
 * Replaces _Kuu_along_dim/_Kuf_along_dim (kronecker_structure.py:702-790) and the
 * pairwise kernel_d(Z), kernel(Z, x) evaluations (:318-319, :336-337). */
int vggp_factor_build(vggp_ctx* ctx, int kind, int basis, const double* x, int64_t n,
                      const double* grid, int64_t m, double ell, int flags,
                      double* A0, double* dA0, double* K0, double* dK0, void* stream);

/* Cholesky K + jitter*I = L L^T with the psd_safe_cholesky jitter schedule (0, 1e-8,
 * 1e-7, 1e-6) and the explicit inverse of L.  K, L, Linv DEVICE [m][m] row-major.
 * Replaces the Cholesky hidden inside lazify(Kuu).inv_matmul (kronecker_structure.py:269).
 * jitter_out HOST (may be NULL). */
int vggp_cholesky_inverse(vggp_ctx* ctx, const double* K, int64_t m, double* L, double* Linv,
                          double* jitter_out, void* stream);

/* Symmetric eigendecomposition G = Q diag(lam) Q^T by parallel cyclic Jacobi.
 * G DEVICE [m][m]; lam DEVICE [m]; Qt DEVICE [m][m] with ROW j = eigenvector j. */
int vggp_eigh(vggp_ctx* ctx, const double* G, int64_t m, double* lam, double* Qt,
              int32_t* sweeps_out, int flags, void* stream);

/* Strided fp64 MFMA GEMM  C[M][N] = op(A) op(B)  with element (i,k) of op(A) at
 * A[i*sa_m + k*sa_k] and (k,j) of op(B) at B[k*sb_k + j*sb_n]; C row-major, ld = ldc. */
int vggp_gemm(vggp_ctx* ctx, const double* A, int64_t sa_m, int64_t sa_k,
              const double* B, int64_t sb_k, int64_t sb_n,
              double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream);

/* Triangular solve by substitution on the matrix cores: L X = R (trans = 0) or L^T X = R (trans = 1), L DEVICE [m][m]
 * lower-triangular (row-major, the upper part is not read), R, X DEVICE [m][ncols] row-major (X may alias R).
 * Blocked: 16 x 16 diagonal blocks inverted in a wave, 128 x 128 diagonal blocks solved from LDS, the rest by MFMA GEMM
 * updates (csrc/trsm.hip).  Replaces the triangular solves inside lazify(Kuu).inv_matmul (kronecker_structure.py:269). */
int vggp_trsm(vggp_ctx* ctx, const double* L, int64_t m, const double* R, int64_t ncols, double* X, int trans, void* stream);

/* Kronecker solve  X = K1^{-1} Y K2^{-T},  K_d = L_d L_d^T,  from the CHOLESKY FACTORS (BASELINE metric ii; nothing is
 * pre-inverted by the caller, the whole solve is inside this call), X = L1^{-T} (L1^{-1} Y L2^{-T}) L2^{-1}, never
 * materialising K1 (x) K2.  n1, n2 <= 128: four triangular solves by substitution (vggp_trsm's strip kernel).  Larger
 * factors: their 128 x 128 diagonal blocks are inverted by substitution on the identity, the rest of L^{-1} follows by
 * block doubling (two MFMA GEMM launches per level), and the four applications are triangular-aware MFMA GEMMs that skip
 * the zero half -- a substitution sweep over n / 128 block rows is a chain of 2 n / 128 dependent launches per solve
 * (1.3 ms at n = 1024 against 0.3 ms; VGGP_KRON_SUBST=1 selects it).  L1 [n1][n1], L2 [n2][n2] lower-triangular
 * (e.g. from vggp_cholesky_inverse), Y, X DEVICE [n1][n2] (X may alias Y).
 * Replaces Kuu.inv_matmul(.) with Kuu = torch.kron(Kuu_1, Kuu_2) (kronecker_structure.py:269, :805). */
int vggp_kron_solve(vggp_ctx* ctx, const double* L1, int64_t n1, const double* L2, int64_t n2,
                    const double* Y, double* X, void* stream);

/* Per-stage timing with HIP events on the stream the kernels are launched on (bench.py's
 * live roofline measurement).  When enabled, every ELBO step records one event after each
 * launch group; vggp_profile_read returns the accumulated milliseconds per stage. */
#define VGGP_NSTAGE 20
int         vggp_profile(vggp_ctx* ctx, int enable);
int         vggp_profile_read(vggp_ctx* ctx, double ms_out[VGGP_NSTAGE], int32_t* steps_out, int reset);
const char* vggp_stage_name(int stage);
/* Name of the kernel the last projection launch (S = [B2;V2] Y, the only pass over Y) dispatched: the roofline kernel. */
const char* vggp_project_kernel_name(void);

/* sum of squares of a DEVICE array, summed over all ranks of the context (yy_total); result to HOST. */
int vggp_sumsq(vggp_ctx* ctx, const double* y, int64_t n, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VGGP_H */
