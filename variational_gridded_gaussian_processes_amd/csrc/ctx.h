// Context and per-dimension workspace of libvggp_hip.so (shared by api.hip and masked.hip).
#pragma once
#include "common.h"

#define VG_MAXEV 40
#define VG_NFORK 4

struct VgDim {
    int kind = 0, basis = 0, n = 0, m = 0;
    double *x = nullptr, *grid = nullptr;
    double *K0 = nullptr, *dK0 = nullptr, *AD = nullptr, *L0 = nullptr, *Linv0 = nullptr, *BV = nullptr;
    double *Kc = nullptr, *Lc = nullptr, *Dc = nullptr;   // 128 < m <= 256: four jittered copies of K0, their factors, their diagonal-block inverses
    int* st8 = nullptr;               // ... and the status words [level][block] of the eight block factorisations
    double* Dinv0 = nullptr;          // [ceil(m/16)][16][16]: inverses of the diagonal blocks of L0 (left by the Cholesky launch)
    double *X = nullptr, *Mk = nullptr, *GH = nullptr, *GHslab = nullptr, *Gw = nullptr;
    double *lam0 = nullptr, *Qt = nullptr, *QtPrev = nullptr, *QtPrev2 = nullptr, *U = nullptr;
    double *Ep = nullptr, *Fp = nullptr, *Wp = nullptr;      // prediction of the next start basis (finish_enqueue tail)
    double *TM = nullptr, *TH = nullptr, *E = nullptr, *F = nullptr, *RQ = nullptr, *RQsq = nullptr;
    double *chol_scratch = nullptr, *gwork = nullptr, *jitter = nullptr;
    double2* rotlog = nullptr;
    int *roundlog = nullptr, *counters = nullptr, *status = nullptr, *perm = nullptr;
    int gh_split = 1, max_rounds = 0;
    // subspace start (numerically rank-deficient Gram matrices, e.g. RBF): scratch for the r leading rows and the small eigenproblem
    double* Id = nullptr;             // m x m identity ("Id G" through the slab-summing GEMM = the reduced G)
    double *Zs = nullptr, *V1s = nullptr, *Hs = nullptr, *Ws = nullptr, *lam_s = nullptr, *gwork2 = nullptr;
    double2* rotlog2 = nullptr;
    int *roundlog2 = nullptr, *counters2 = nullptr, *perm2 = nullptr;
    int sub_r = 0;                    // rows treated as the numerical range in the next step (0: subspace start off)
    // thin chain (thin.hip): V1 Mk0, V1 H0 (r x m) and V1 Mk0 V1^T, V1 H0 V1^T (r x r)
    double *tMV = nullptr, *tHV = nullptr, *tAM = nullptr, *tAH = nullptr;
    double* Omega = nullptr;          // [VG_THIN_MAXR][m] fixed pseudo-random rows: start of the thin chain's cold range finder
    int thin_rows = 0;                // leading rows of QtPrev that hold eigenvectors (m after a full solve, r after a thin step)
    bool have_prev = false, have_prev2 = false;     // QtPrev / QtPrev2 hold the bases of the last / the step before
};

struct VgGraphKey {
    const void* y = nullptr;
    const void* payload = nullptr;
    double yy = 0.0;
    bool operator==(const VgGraphKey& o) const { return y == o.y && payload == o.payload && yy == o.yy; }
};

typedef VgHostOut HostOut;   // pinned readback block (common.h)

struct vggp_ctx {
    int device = 0;
    bool planned = false;
    vggp_desc desc{};
    VgDim d[2];
    void* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    // cross-dimension buffers
    double* Sp = nullptr;             // [A2;dA2] Y (split-K slabs): the early projection of the thin chain
    double *St = nullptr, *CCslab = nullptr, *payload = nullptr, *GH1 = nullptr;
    double *T3 = nullptr, *P3 = nullptr, *beta = nullptr, *bl2 = nullptr, *bl1 = nullptr, *invD = nullptr;
    double *rowpart = nullptr, *r1 = nullptr, *r1l = nullptr, *r2 = nullptr, *r2l = nullptr, *dotpart = nullptr, *ol = nullptr;
    double *out = nullptr, *theta = nullptr, *wq = nullptr;
    int st_split = 1, cc_split = 1;
    int st_slabs = 1;                            // ... of S = [B2;V2] Y
    bool dinv_valid = false;                     // Dinv0 holds the diagonal-block inverses of the current L0 (m <= 128 Cholesky path)
    int ride_stage0 = 0;                         // first pending rider of the fused warm step: 0 = S, 1 = [C;C1;C2] (S came out of the early projection)
    int gh_slabs[2] = {1, 1}, cc_slabs = 1;      // split-K slab counts actually produced by the last partials launch
    long payload_len = 0;
    bool have_partials = false, have_step = false, have_masked = false;
    void* masked = nullptr;          // VgMasked workspace (masked.hip), allocated on first use
    // pinned host staging
    double* h_theta = nullptr;
    HostOut* h_out = nullptr;
    HostOut* d_hout = nullptr;       // device-visible address of h_out
    double* d_htheta = nullptr;      // device-visible address of h_theta
    int* ticket = nullptr;           // last-block ticket of the fused final reduction
    // scratch for the exported building blocks / posterior (lazy)
    void* misc = nullptr;
    size_t misc_bytes = 0;
    double* sumsq_partial = nullptr;
    double* sumsq_out = nullptr;
    // captured step graphs.  Capture is illegal on the legacy default stream, so when the caller passes stream 0 the
    // step runs on `own_stream`, a BLOCKING stream: it is implicitly ordered with the legacy default stream in both
    // directions (uploads / all-reduce issued by torch on stream 0 before, q(v) / posterior calls after).
    hipStream_t own_stream = nullptr;
    // side stream for the branches of a step that do not depend on each other (extrapolated basis || factor + Cholesky;
    // projection of Y || eigensolver chain); under capture the fork / join events become graph edges
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork[VG_NFORK] = {}, ev_join[VG_NFORK] = {};
    bool use_graph = true;
    hipGraphExec_t gexec[20] = {};
    VgGraphKey gkey[20];
    // what the accurate (cold) recompute of the read-outs needs from the last step (api.hip vg_accurate_state)
    bool last_warm = false, last_slabs = false, acc_valid = false;
    const double* last_payload = nullptr;
    double last_yy = 0.0;
    VgGemmBatch ride_proj, ride_cc;   // the projection launches of a fused warm step, deferred to the eigensolver chain as riders
    bool ride_pending = false;
    double last_ell[2] = {0.0, 0.0};  // lengthscales of the previous step (jump detection, vggp_elbo_step)
    bool pred_consumed = false;       // this prediction's Newton-Schulz step has been applied to Fp (it accumulates: once only)
    bool refine_next = false;
    bool sub_next = false;            // the last step's numerical ranks allow the subspace start
    int sub_r_cap[2] = {0, 0};        // ranks the _S graphs were captured with
    double *tCV = nullptr, *tAC = nullptr;   // thin chain: {C, C1, C2} V1_2^T  [3 m1][r2]  and  V1_1 (.)  [3][r1][r2]
    bool cur_thin = false;            // the step in flight runs the thin chain, with these ranks (vg_start_prepare -> finish_collect)
    int cur_r[2] = {0, 0};
    bool cur_extrap = false;          // the step in flight starts from the extrapolated basis
    int cur_newton = 0;               // Newton-chain iterations of the step in flight (0: another chain)
    bool last_newton = false, newton_next = false;
    int newton_block = 0;             // steps for which the Newton chain stays off after a miss
    int newton_iters = 3, newton_cap = 0;
    int newton_ok_run = 0;           // Newton-chain steps since the last miss (the iteration count decays after 64)
    bool last_thin = false;           // the last finished step ran the thin chain: QtPrev holds r rows, the m-space state (beta, 1/D, E, F) is not there
    int thin_block = 0;               // steps for which the thin chain stays off after it missed
    bool thin_off = false;            // a caller needed the full m-space state of a warm step (vggp_zgrad): keep to the full chain from now on
    bool sub_mode = false;            // the current step uses the subspace start (U then holds the identity)         // the last step ended in the polish in both dimensions: refine the next start basis
    // the context's collective (comm.hip)
    int n_ranks = 1, rank = 0;
    void* comm = nullptr;             // ncclComm_t
    bool comm_dead = false;           // the communicator was aborted (a peer failed or timed out): multi-rank steps return VGGP_ERCCL
    vggp_allreduce_fn cb = nullptr;   // host-callback transport (rehearsal / other transports)
    void* cb_user = nullptr;
    double* h_stage = nullptr;        // pinned staging of the callback transport
    long h_stage_count = 0;
    hipStream_t poll_stream = nullptr;   // the last step's completion was seen in the pinned result block, not through the runtime:
    bool poll_stream_valid = false;      // this stream may still be finishing its graph (api.hip vg_quiesce)
    long seq = 0;                     // step sequence number (h_theta[5] -> device theta[5] -> VgHostOut::seq)
    int warm_run = 0;                 // consecutive warm-started steps (periodic cold restart bounds orthogonality drift)
    // per-stage profiling (bench.py): event e[i] is recorded after stage i-1's launches
    bool prof = false;
    hipEvent_t ev[VG_MAXEV] = {};
    int ev_stage[VG_MAXEV] = {};      // stage charged with the time since the previous event (-1: clock restart)
    int nev = 0;
    double prof_ms[VGGP_NSTAGE] = {};
    int prof_steps = 0;
};


int vg_ensure_misc(vggp_ctx* c, size_t bytes);
int vg_comm_init(vggp_ctx* c, int n_ranks, int rank, const void* unique_id);
void vg_comm_destroy(vggp_ctx* c);
int vg_allreduce(vggp_ctx* c, double* buf, long count, hipStream_t st);
void vg_comm_abort(vggp_ctx* c);
int vg_comm_wait(vggp_ctx* c, hipStream_t st, const volatile double* seq = nullptr, double want = 0.0);
// blocked dense Cholesky + inverse for matrices beyond one workgroup (masked.hip); S is destroyed; status != 0 on failure
#define VG_DENSE_MB 128
struct VgDenseChol { double *S, *L, *X, *DI, *Tmp, *scratch, *jit; int* status; long M; double* Sinv; };
int vg_blocked_chol_inverse(const VgDenseChol& w, hipStream_t st);
int vg_partials_enqueue(vggp_ctx* c, const double* Y, double* payload, hipStream_t st, bool reduce = true, bool extrap = false, bool fused = false,
                        bool apply_ns = false, int early = 0);      // early: 1 = thin chain, 2 = regular warm chains ([C;C1;C2] is then the first rider)
void vg_masked_free(vggp_ctx* c);
void vg_masked_new_plan(vggp_ctx* c);
// batch of triangular solves, each in place on its X (api.hip trsm_batch: element (row k, column c) at X[k * sk + c * sc])
#define VG_TRSM_BLK 128
struct VgTrsmSpec {
    const double* L; long ldl;
    const double* Dinv; long dinv_blk, dinv_ld;
    double* X; long sk, sc, ncols;
    long m; int trans;
};
int vg_trsm_batch(const VgTrsmSpec* sp, int n, hipStream_t st);
