// C-ABI of libvggp_hip.so (include/vggp.h): context, workspace arena and the launch sequence
// of the ELBO step / q(v) / posterior.  Host logic only -- every number is produced by the HIP
// kernels in factor_build.hip, chol.hip, gemm.hip, eigh.hip and mspace.hip.  There is no CPU
// fallback: without a gfx950 device every entry point returns VGGP_EHIP.
#include "common.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <vector>

// ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vggp_last_error(void) { return g_err; }
extern "C" int vggp_version(void) { return VGGP_VERSION; }

#include "ctx.h"
#include "factor_elem.h"
#define VG_CHOL_MAXJOBS_HOST 8
#define VG_SP_SLABS 4             // split-K slabs of the early projection S' = [A2;dA2] Y (riding in the Cholesky launch)
#define VG_NEWTON_TOL 1e-12      // off-diagonal threshold of the Newton chain's rotations, relative to ||Gw||_F / m
#define VG_NEWTON_NULL 1e-12     // rows whose eigenvalue was below this fraction of the largest in the PREVIOUS step form the known null space:
                                 // pairs inside it are neither rotated nor judged (VgRefineJob::lam_prev)
#define VG_NEWTON_CLUSTER 1e-7   // ... and the rows below this fraction form the near-null cluster, whose internal pairs wait for the second iteration
#define VG_NEWTON_NOISE 1e-13    // diagonal entries below this fraction of the largest are "at the rounding floor" (VgRefineJob::noise)
#define VG_NEWTON_ACCEPT 1e-11   // ... of its convergence check: Gw = S G S^T comes out of two GEMMs with ~ eps sqrt(m) ||G|| of rounding noise per
                                 // element, 2e-15 ||G|| at m = 256 against 4e-15 ||G||_F for 1e-12 -- the chain converged TO the threshold and
                                 // missed by a hair (measured: 10 log10(offdiag / thr) = 0 at every miss); the ELBO sees such elements at second order

static void graphs_clear(vggp_ctx* c);
static int vg_quiesce(vggp_ctx* c);

static const char* VG_STAGE_NAMES[VGGP_NSTAGE] = {
    "factor_build", "cholesky_inverse", "trsm_BV(L^-1[A|dA|dK])", "extrap_basis(3 small gemms)", "gemm_project(S=[B2;V2]Y)",
    "gemm_C(B1*S)", "gemm_gram(G,H,Mk)", "reduce_slabs", "warm_first_gemm([Z,G]|TM)", "rowqr|refine", "warm_mid_gemms",
    "ritz_eigh", "warm_last_gemms(->Gw)", "jacobi_eigh+replay(fused)", "gemm_rotate_right", "gemm_rotate_left", "dstage",
    "gemm_betaGram", "final_reduce", "(unused)"};
extern "C" const char* vggp_stage_name(int i) { return (i >= 0 && i < VGGP_NSTAGE) ? VG_STAGE_NAMES[i] : ""; }

// Per-stage profiling (bench.py): in profiling mode the step runs as plain launches on ONE stream (no graph, no side stream)
// and an event is recorded after every launch group; stage `id` is charged the time since the previous event.
// VG_MARK(-1) only (re)starts the clock (start of the step, start of the finish half after the caller's all-reduce).
#define VG_MARK(id)                                                                                     \
    do {                                                                                                \
        if (c->prof && c->nev < VG_MAXEV) { VG_HIP(hipEventRecord(c->ev[c->nev], st)); c->ev_stage[c->nev++] = (id); } \
    } while (0)

// Fork / join onto the context's side stream (works eagerly and under stream capture, where it becomes a graph branch).
// OFF by default: measured on ROCm 7.2 / MI355X (gpurun_out/r2c), running the projection of Y (2 launches, 40 us) beside
// the eigensolver chain as a second graph branch made the 1024^2 step SLOWER (0.478 vs 0.453 ms) -- every cross-stream
// edge of a replayed graph costs more than the launches it hides; independent work is batched into shared launches
// instead (see the prediction GEMMs in the tail of finish_enqueue).  VGGP_FORK=1 switches the branch on for A/B runs.
// Profiling mode keeps everything on the one stream so that the per-stage events mean what they say.
static hipStream_t vg_side(vggp_ctx* c, hipStream_t st) {
    static const bool on = getenv("VGGP_FORK") != nullptr;
    return (c->prof || !on) ? st : c->side_stream;
}
#define VG_FORK(i)                                                                                      \
    do { if (sx != st) { VG_HIP(hipEventRecord(c->ev_fork[i], st)); VG_HIP(hipStreamWaitEvent(sx, c->ev_fork[i], 0)); } } while (0)
#define VG_JOIN_RECORD(i) do { if (sx != st) VG_HIP(hipEventRecord(c->ev_join[i], sx)); } while (0)
#define VG_JOIN_WAIT(i) do { if (sx != st) VG_HIP(hipStreamWaitEvent(st, c->ev_join[i], 0)); } while (0)

int vg_ensure_misc(vggp_ctx* c, size_t bytes) {
    if (c->misc_bytes >= bytes) return VGGP_OK;
    if (c->misc) { VG_HIP(hipFree(c->misc)); c->misc = nullptr; c->misc_bytes = 0; }
    VG_HIP(hipMalloc(&c->misc, bytes));
    c->misc_bytes = bytes;
    return VGGP_OK;
}

extern "C" int vggp_create(vggp_ctx** out, int device, int n_ranks, int rank, const void* unique_id) {
    if (!out) { vg_set_error("vggp_create: null out"); return VGGP_EINVAL; }
    VG_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "vggp_create: rank %d of %d", rank, n_ranks);
    int ndev = 0;
    VG_HIP(hipGetDeviceCount(&ndev));
    VG_REQUIRE(device >= 0 && device < ndev, "vggp_create: device %d out of range (%d devices)", device, ndev);
    VG_ENTER_DEVICE(device);
    hipDeviceProp_t prop;
    VG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vg_set_error("vggp_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return VGGP_EHIP;
    }
    VG_HIP(vg_chol_setup());
    VG_HIP(vg_eigh_setup());
    VG_HIP(vg_trsm_setup());
    VG_HIP(vg_thin_tail_setup());
    vggp_ctx* c = new (std::nothrow) vggp_ctx();
    if (!c) { vg_set_error("out of host memory"); return VGGP_ENOMEM; }
    c->device = device;
    const char* ng = getenv("VGGP_NO_GRAPH");
    c->use_graph = !(ng && ng[0] == '1');
    VG_HIP(hipStreamCreate(&c->own_stream));
    VG_HIP(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    for (int i = 0; i < VG_NFORK; ++i) {
        VG_HIP(hipEventCreateWithFlags(&c->ev_fork[i], hipEventDisableTiming));
        VG_HIP(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    VG_HIP(hipHostMalloc((void**)&c->h_theta, 8 * sizeof(double), hipHostMallocDefault));
    VG_HIP(hipHostMalloc((void**)&c->h_out, sizeof(HostOut), hipHostMallocDefault));
    VG_HIP(hipHostGetDevicePointer((void**)&c->d_hout, c->h_out, 0));
    VG_HIP(hipHostGetDevicePointer((void**)&c->d_htheta, c->h_theta, 0));
    VG_HIP(hipMalloc((void**)&c->ticket, 4 * sizeof(int)));
    VG_HIP(hipMemset(c->ticket, 0, 4 * sizeof(int)));
    VG_HIP(hipMalloc((void**)&c->sumsq_partial, 1024 * sizeof(double)));
    VG_HIP(hipMalloc((void**)&c->sumsq_out, 8 * sizeof(double)));
    const int rcc = vg_comm_init(c, n_ranks, rank, unique_id);
    if (rcc) { (void)vggp_destroy(c); return rcc; }
    *out = c;
    return VGGP_OK;
}

extern "C" int vggp_destroy(vggp_ctx* c) {
    if (!c) return VGGP_OK;
    VgDeviceGuard guard;
    (void)guard.enter(c->device);
    (void)vg_quiesce(c);
    for (int i = 0; i < 20; ++i) if (c->gexec[i]) (void)hipGraphExecDestroy(c->gexec[i]);
    for (int i = 0; i < VG_MAXEV; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < VG_NFORK; ++i) {
        if (c->ev_fork[i]) (void)hipEventDestroy(c->ev_fork[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    vg_masked_free(c);
    vg_comm_destroy(c);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->arena) (void)hipFree(c->arena);
    if (c->misc) (void)hipFree(c->misc);
    if (c->h_theta) (void)hipHostFree(c->h_theta);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->ticket) (void)hipFree(c->ticket);
    if (c->sumsq_partial) (void)hipFree(c->sumsq_partial);
    if (c->sumsq_out) (void)hipFree(c->sumsq_out);
    delete c;
    return VGGP_OK;
}

// bump allocator over the arena (256-B aligned)
struct Bump {
    char* base;
    size_t off = 0;
    bool dry;
    template <typename T>
    T* take(size_t count) {
        off = (off + 255) & ~size_t(255);
        T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
        off += count * sizeof(T);
        return p;
    }
};

static int pick_split(int tiles, int K, int target) {
    int s = (target + tiles - 1) / tiles;
    int maxs = std::max(1, K / (4 * VG_BK));
    return std::max(1, std::min(s, maxs));
}

// number of doubles in the host/device `grid` array of a dimension
static long vg_grid_len(int basis, long m) {
    if (basis == VGGP_BASIS_B0) return m + 1;                 // mesh knots
    if (basis == VGGP_BASIS_VFF) return 2 + (m - 1) / 2 + 1;  // a, b, omega_0 .. omega_M (m = 2M + 1)
    return m;                                                 // points / B1 knots / trivial
}
// +1: Kuu_d = s_d K0, Kuf_d = s_d A0 (kernel-evaluated bases);  -1: Kuu_d = K0 / s_d, Kuf_d = A0 (inter-domain VFF / B1).
// B = L^{-1} Kuf is sqrt(s) L0^{-1} A0 either way, so only u-space results (q(v)) need the sign.
static int vg_uexp(int basis) { return (basis == VGGP_BASIS_VFF || basis == VGGP_BASIS_B1) ? -1 : 1; }

static void layout(vggp_ctx* c, Bump& b) {
    // Allocation ORDER matters: the small kernels of a step are latency chains whose first loads miss everything
    // (the producer ran on another XCD), and each distinct 2 MB region they touch adds an address-translation miss on
    // top.  So the arrays are grouped by the stage that reads them -- tail (m-space) arrays together, eigen-stage arrays
    // together, factor-stage arrays together -- and the large streaming buffers (operands over the grid, split-K slabs,
    // rotation log) come last.
    const vggp_desc& D = c->desc;
    const long n1 = D.n1, n2 = D.n2, m1 = D.m1, m2 = D.m2;
    // ---- tail: m-space stage and final reduction
    c->out = b.take<double>(8);
    c->theta = b.take<double>(8);
    c->rowpart = b.take<double>(m1 * 8);
    c->r1 = b.take<double>(m1);
    c->r1l = b.take<double>(m1);
    c->r2 = b.take<double>(2 * m2);          // [r2; r2l]: the two rows of one GEMM output
    c->r2l = c->r2 + m2;
    c->dotpart = b.take<double>(64 * 4);    // [4][64] per-tile partial sums (unused slots stay zero: the arena is cleared at plan time)
    c->ol = b.take<double>(2 * m1);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        const long m = d.m, m2e = m + (m & 1);
        d.lam0 = b.take<double>(m);
        d.jitter = b.take<double>(2);
        d.counters = b.take<int>(8);
        d.counters2 = d.counters + 4;
        d.perm = b.take<int>(m2e);
        d.perm2 = b.take<int>(m2e);
        d.lam_s = b.take<double>(m);
        d.status = b.take<int>(2);
    }
    c->invD = b.take<double>(m1 * m2);
    c->beta = b.take<double>(m1 * m2);
    c->bl2 = b.take<double>(m1 * m2);
    c->bl1 = b.take<double>(m1 * m2);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        d.E = b.take<double>((long)d.m * d.m);
        d.F = b.take<double>((long)d.m * d.m);
    }
    c->P3 = b.take<double>(3 * m1 * m2);
    c->T3 = b.take<double>(3 * m1 * m2);
    // ---- eigen stage
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        const long m = d.m;
        d.TM = b.take<double>(m * m);
        d.TH = b.take<double>(m * m);
        d.Qt = b.take<double>(m * m);
        d.QtPrev = b.take<double>(m * m);
        d.QtPrev2 = b.take<double>(m * m);
        d.U = b.take<double>(m * m);
        d.Ep = b.take<double>(m * m);
        d.Fp = b.take<double>(m * m);
        d.Wp = b.take<double>(m * m);
        d.Id = b.take<double>(m * m);
        d.Gw = b.take<double>(m * m);
        d.GH = b.take<double>(2 * m * m);
        d.Mk = b.take<double>(m * m);
    }
    c->payload_len = 2 * m2 * m2 + 3 * m1 * m2;
    c->payload = b.take<double>(c->payload_len + 8);      // + the peer-failure word of a multi-rank step (payload[payload_len])
    // ---- factor stage
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        const long m = d.m;
        const int glen = (int)vg_grid_len(d.basis, m);
        d.grid = b.take<double>(glen);
        d.K0 = b.take<double>(m * m);
        d.dK0 = b.take<double>(m * m);
        d.L0 = b.take<double>(m * m);
        d.Linv0 = b.take<double>(m * m);
        d.Dinv0 = b.take<double>(((m + 15) / 16) * 256);
        if (m > 128) {                   // blocked Cholesky of the step (vg_chol_big_enqueue)
            d.Kc = b.take<double>(4 * m * m); d.Lc = b.take<double>(4 * m * m); d.Dc = b.take<double>(4 * ((m + 15) / 16) * 256);
            d.st8 = b.take<int>(8);
        }
        d.X = b.take<double>(m * m);
        d.chol_scratch = b.take<double>(m * (m + 1));
        d.RQ = b.take<double>(m * m);
        d.RQsq = b.take<double>(m * m);
    }
    c->wq = b.take<double>(2 * m1 * m2);
    // ---- large streaming buffers
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        const long m = d.m, n = d.n, m2e = m + (m & 1);
        d.x = b.take<double>(n);
        d.AD = b.take<double>(2 * m * n);
        d.BV = b.take<double>(2 * m * n);
        static const char* ghe = getenv("VGGP_GH_TARGET");
        d.gh_split = pick_split((int)(((m + 63) / 64) * ((m + 63) / 64)), (int)n, ghe ? atoi(ghe) : 32);
        d.GHslab = b.take<double>((size_t)d.gh_split * 2 * m * m);
        d.gwork = b.take<double>(m2e * (m2e + 1));
        d.max_rounds = (int)(VG_EIG_MAXSWEEP * (m2e - 1));
        d.rotlog = b.take<double2>(vg_eigh_log_bytes((int)m) / sizeof(double2) + 1);
        d.roundlog = b.take<int>(d.max_rounds);
        if (m >= 24 && m <= 256) {       // subspace start: the small problem has at most m/2 rows (m > 128: the thin chain only)
            d.gwork2 = b.take<double>(m2e * (m2e + 1));
            d.rotlog2 = b.take<double2>(vg_eigh_log_bytes((int)m) / sizeof(double2) + 1);
            d.roundlog2 = b.take<int>(d.max_rounds);
            d.Zs = b.take<double>(m * m); d.V1s = b.take<double>(m * m); d.Hs = b.take<double>(m * m); d.Ws = b.take<double>(m * m);
            d.tMV = b.take<double>(VG_THIN_MAXR * m); d.tHV = b.take<double>(VG_THIN_MAXR * m);
            d.tAM = b.take<double>(VG_THIN_MAXR * VG_THIN_MAXR); d.tAH = b.take<double>(VG_THIN_MAXR * VG_THIN_MAXR);
            d.Omega = b.take<double>(VG_THIN_MAXR * m);
        }
    }
    const int st_tiles = (int)(((2 * m2 + 63) / 64) * ((n1 + 63) / 64));
    static const char* ste = getenv("VGGP_ST_TARGET");
    c->st_split = pick_split(st_tiles, (int)n2, ste ? atoi(ste) : 256);
    c->St = b.take<double>((size_t)std::max(c->st_split, VG_SP_SLABS) * 2 * m2 * n1);   // (the early projection leaves up to VG_SP_SLABS slabs of S here)
    c->Sp = b.take<double>((size_t)VG_SP_SLABS * 2 * m2 * n1);       // [A2;dA2] Y of the thin chain's early projection (split-K slabs)
    const int cc_tiles = (int)(((2 * m1 + 63) / 64) * ((m2 + 63) / 64));
    const char* cce = getenv("VGGP_CC_TARGET");
    c->cc_split = pick_split(cc_tiles, (int)n1, cce ? atoi(cce) : 64);
    c->CCslab = b.take<double>((size_t)c->cc_split * 3 * m1 * m2);
    c->tCV = b.take<double>(3 * m1 * VG_THIN_MAXR);
    c->tAC = b.take<double>(3 * VG_THIN_MAXR * VG_THIN_MAXR + 64);      // (+ 64 words of phase stamps in diagnostic builds)
}

static int check_dim(int kind, int basis, long n, long m, const char* which) {
    VG_REQUIRE(kind >= 0 && kind <= 3, "vggp_plan: bad kind for %s", which);
    VG_REQUIRE(basis >= 0 && basis <= 4, "vggp_plan: bad basis for %s", which);
    VG_REQUIRE(!((basis == VGGP_BASIS_VFF || basis == VGGP_BASIS_B1) && kind != VGGP_KIND_MATERN12),
               "vggp_plan: the VFF and B1 features exist for Matern-1/2 only (%s)", which);
    VG_REQUIRE(!(basis == VGGP_BASIS_VFF && (m < 3 || (m & 1) == 0)), "vggp_plan: VFF needs m = 2M + 1 >= 3 (%s)", which);
    VG_REQUIRE(!(basis == VGGP_BASIS_B1 && m < 2), "vggp_plan: B1 needs at least 2 knots (%s)", which);
    VG_REQUIRE(!(basis == VGGP_BASIS_B0 && kind != VGGP_KIND_MATERN12),
               "vggp_plan: the B0 basis exists for Matern-1/2 only (%s)", which);
    VG_REQUIRE(n >= 1, "vggp_plan: %s has no observations", which);
    VG_REQUIRE(m >= 1 && m <= 256, "vggp_plan: m=%ld for %s outside [1, 256]", m, which);
    VG_REQUIRE(!(basis == VGGP_BASIS_ONE && m != 1), "vggp_plan: BASIS_ONE needs m=1 (%s)", which);
    return VGGP_OK;
}

extern "C" int vggp_plan(vggp_ctx* c, const vggp_desc* desc) {
    if (!c || !desc) { vg_set_error("vggp_plan: null argument"); return VGGP_EINVAL; }
    VG_ENTER_DEVICE(c->device);
    int rc;
    if ((rc = check_dim(desc->kind1, desc->basis1, desc->n1, desc->m1, "dimension 1"))) return rc;
    if ((rc = check_dim(desc->kind2, desc->basis2, desc->n2, desc->m2, "dimension 2"))) return rc;
    VG_REQUIRE(desc->x1 && desc->x2, "vggp_plan: null coordinate arrays");
    VG_REQUIRE((desc->basis1 == VGGP_BASIS_ONE || desc->grid1) && (desc->basis2 == VGGP_BASIS_ONE || desc->grid2),
               "vggp_plan: null grid arrays");
    if (desc->flags & VGGP_FLAG_SCATTERED)
        VG_REQUIRE(desc->n1 == desc->n2 && desc->basis1 != VGGP_BASIS_ONE && desc->basis2 != VGGP_BASIS_ONE,
                   "vggp_plan: scattered points need n1 == n2 (one coordinate pair per point) and two real dimensions");
    else
        VG_REQUIRE(desc->n_total >= desc->n1 * desc->n2, "vggp_plan: n_total smaller than the local grid");
    VG_REQUIRE(desc->n1 < (1L << 24) && desc->n2 < (1L << 24), "vggp_plan: grid axis too long");
    c->planned = false;
    graphs_clear(c);
    VG_HIP(hipDeviceSynchronize());
    c->warm_run = 0;
    c->refine_next = false;
    c->sub_next = false; c->sub_mode = false; c->sub_r_cap[0] = c->sub_r_cap[1] = 0;
    c->pred_consumed = false;
    c->acc_valid = false; c->last_warm = false; c->last_slabs = false; c->last_payload = nullptr;
    c->last_thin = false; c->thin_off = false; c->thin_block = 0;
    c->cur_newton = 0; c->last_newton = false; c->newton_next = false; c->newton_block = 0; c->newton_iters = 3; c->newton_ok_run = 0;
    c->desc = *desc;
    c->d[0] = VgDim();
    c->d[1] = VgDim();
    c->d[0].kind = desc->kind1; c->d[0].basis = desc->basis1; c->d[0].n = (int)desc->n1; c->d[0].m = (int)desc->m1;
    c->d[1].kind = desc->kind2; c->d[1].basis = desc->basis2; c->d[1].n = (int)desc->n2; c->d[1].m = (int)desc->m2;
    Bump dry{nullptr, 0, true};
    layout(c, dry);
    const size_t need = dry.off + 4096;
    if (need > c->arena_bytes) {
        if (c->arena) { VG_HIP(hipFree(c->arena)); c->arena = nullptr; c->arena_bytes = 0; }
        VG_HIP(hipMalloc(&c->arena, need));
        c->arena_bytes = need;
    }
    Bump wet{reinterpret_cast<char*>(c->arena), 0, false};
    layout(c, wet);
    c->arena_used = wet.off;
    VG_HIP(hipMemset(c->arena, 0, c->arena_used));
    const double one = 1.0;
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        const double* hx = k == 0 ? desc->x1 : desc->x2;
        const double* hg = k == 0 ? desc->grid1 : desc->grid2;
        VG_HIP(hipMemcpy(d.x, hx, sizeof(double) * d.n, hipMemcpyHostToDevice));
        if (d.basis == VGGP_BASIS_ONE) {
            VG_HIP(hipMemcpy(d.grid, &one, sizeof(double), hipMemcpyHostToDevice));
        } else {
            const int glen = (int)vg_grid_len(d.basis, d.m);
            VG_HIP(hipMemcpy(d.grid, hg, sizeof(double) * glen, hipMemcpyHostToDevice));
        }
    }
    for (int k = 0; k < 2; ++k) VG_HIP(vg_identity_launch(c->d[k].Id, c->d[k].m, nullptr));
    for (int k = 0; k < 2; ++k) {          // start rows of the cold range finder (vg_cold_thin_prepass): fixed pseudo-random numbers in (-1, 1)
        VgDim& d = c->d[k];
        if (!d.Omega) continue;
        std::vector<double> om((size_t)VG_THIN_MAXR * d.m);
        unsigned long long z = 0x9E3779B97F4A7C15ull * (unsigned long long)(k + 1);
        for (double& v : om) {
            z += 0x9E3779B97F4A7C15ull;                          // splitmix64
            unsigned long long x = z;
            x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
            v = (double)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
        }
        VG_HIP(hipMemcpy(d.Omega, om.data(), sizeof(double) * om.size(), hipMemcpyHostToDevice));
    }
    VG_HIP(hipDeviceSynchronize());
    c->desc.x1 = c->desc.x2 = c->desc.grid1 = c->desc.grid2 = nullptr;   // host pointers not retained
    c->have_partials = c->have_step = c->have_masked = false;
    vg_masked_new_plan(c);
    c->planned = true;
    return VGGP_OK;
}

// New coordinates for the inducing points of a "points" basis WITHOUT re-planning: the device array is overwritten in place, so
// the arena, the captured graphs (they hold the pointer, not the values) and the warm start all stay -- a small move of Z changes
// the Gram matrices as little as a small move of the lengthscale does, and the subspace start checks itself (VG_ESUBMISS).  This
// is what an optimiser that trains Z (kronecker_structure.py:303-304) calls between steps.
extern "C" int vggp_set_inducing(vggp_ctx* c, int dim, const double* z, int64_t m) {
    if (!c || !c->planned) { vg_set_error("vggp_set_inducing: context not planned"); return VGGP_ESTATE; }
    VG_REQUIRE(dim == 0 || dim == 1, "vggp_set_inducing: dim must be 0 or 1");
    VgDim& d = c->d[dim];
    VG_REQUIRE(d.basis == VGGP_BASIS_POINTS, "vggp_set_inducing: dimension %d does not use the points basis", dim + 1);
    VG_REQUIRE(z && m == d.m, "vggp_set_inducing: expected %d coordinates, got %lld", d.m, (long long)m);
    for (int64_t i = 0; i < m; ++i) VG_REQUIRE(std::isfinite(z[i]), "vggp_set_inducing: z[%lld] is not finite", (long long)i);
    VG_ENTER_DEVICE(c->device);
    VG_HIP(hipStreamSynchronize(c->own_stream));              // no step of this context is reading the old coordinates
    VG_HIP(hipMemcpy(d.grid, z, sizeof(double) * m, hipMemcpyHostToDevice));
    c->have_step = false; c->have_partials = false; c->have_masked = false; c->acc_valid = false;      // read-outs need a new step
    return VGGP_OK;
}

extern "C" int64_t vggp_payload_len(const vggp_ctx* c) { return (c && c->planned) ? c->payload_len : 0; }
extern "C" int64_t vggp_workspace_bytes(const vggp_ctx* c) { return c ? (int64_t)c->arena_used : 0; }

// ---------------------------------------------------------------------------------
// ---- triangular solves for any m: blocked substitution ------------------------------------------------------------------
// A batch of solves, each in place on its X (element (row k, column c) at X[k * sk + c * sc], one of the two strides is 1).
// m <= 128: one launch of the strip kernel (trsm.hip).  Larger m: block rows of VG_TRSM_BLK = 128; per block row ONE MFMA
// GEMM launch subtracts the contribution of the block rows already solved (L[b, :b] X[:b], all workgroups of the chip, all
// jobs of the batch) and ONE strip-kernel launch solves the 128 x 128 diagonal blocks (factor staged in LDS).
// Dinv: inverses of the 16 x 16 diagonal blocks of L, block b16 at Dinv + b16 * dinv_blk, row stride dinv_ld.
static int trsm_batch(const VgTrsmSpec* sp, int n, hipStream_t st) {
    int max_nblk = 0;
    for (int j = 0; j < n; ++j) max_nblk = std::max(max_nblk, (int)((sp[j].m + VG_TRSM_BLK - 1) / VG_TRSM_BLK));
    for (int s = 0; s < max_nblk; ++s) {
        VgGemmBatch g;
        vg_gemm_init(&g);
        VgTrsmJob tj[12];
        int nt = 0;
        for (int j = 0; j < n; ++j) {
            const VgTrsmSpec& q = sp[j];
            const int nblk = (int)((q.m + VG_TRSM_BLK - 1) / VG_TRSM_BLK);
            if (s >= nblk) continue;
            const int b = q.trans ? nblk - 1 - s : s;
            const long r0 = (long)b * VG_TRSM_BLK, rb = std::min<long>(VG_TRSM_BLK, q.m - r0), r1 = r0 + rb;
            // rows of X already solved: [0, r0) for L X = R, [r1, m) for L^T X = R
            const long k0 = q.trans ? r1 : 0, kn = q.trans ? q.m - r1 : r0;
            if (kn > 0) {
                // operator block T (rb x kn): T[i][k] = L[r0 + i][k0 + k] (no trans) or L[k0 + k][r0 + i] (trans)
                const double* Tp = q.trans ? q.L + k0 * q.ldl + r0 : q.L + r0 * q.ldl + k0;
                const long t_i = q.trans ? 1 : q.ldl, t_k = q.trans ? q.ldl : 1;
                if (q.sc == 1)        // X row-major: X[r0:r1, :] -= T X[k0:k0+kn, :]
                    vg_gemm_add(&g, Tp, t_i, t_k, q.X + k0 * q.sk, q.sk, 1, q.X + r0 * q.sk, (int)q.sk, (int)rb, (int)q.ncols, (int)kn,
                                1, 0, 1, 0, -1.0, 1);
                else                  // X holds the transposed right-hand sides: X'[:, r0:r1] -= X'[:, k0:k0+kn] T^T
                    vg_gemm_add(&g, q.X + k0, q.sc, 1, Tp, t_k, t_i, q.X + r0, (int)q.sc, (int)q.ncols, (int)rb, (int)kn, 1, 0, 1, 0,
                                -1.0, 1);
            }
            if (nt >= 12) { vg_set_error("trsm_batch: too many jobs"); return VGGP_EINVAL; }
            tj[nt++] = VgTrsmJob{q.L + r0 * q.ldl + r0, q.Dinv + (r0 / 16) * q.dinv_blk, q.X + r0 * q.sk, q.X + r0 * q.sk, q.ldl,
                                 q.dinv_blk, q.dinv_ld, q.sk, q.sc, q.sk, q.sc, q.ncols, (int)rb, q.trans};
        }
        if (g.nprob) VG_HIP(vg_gemm_launch(&g, st));
        if (nt) VG_HIP(vg_trsm_launch(tj, nt, st));
    }
    return VGGP_OK;
}
static int trsm_inplace(const double* L, long m, long ldl, const double* Dinv, double* X, long x_sk, long x_sc, long ncols,
                        int trans, hipStream_t st) {
    VgTrsmSpec q{L, ldl, Dinv, 256, 16, X, x_sk, x_sc, ncols, m, trans};
    return trsm_batch(&q, 1, st);
}

// Enqueue-only halves of the step (no host synchronisation, no host-side reads): they run either directly on the
// caller's stream (profiling mode) or once under stream capture, after which the step is a single graph launch.
// riders: see vg_partials_enqueue.  Off in profiling mode (every stage is then its own launch and can be timed) and with VGGP_NO_RIDE=1.
static bool vg_ride(const vggp_ctx* c) {
    static const bool off = getenv("VGGP_NO_RIDE") != nullptr;
    return !off && !c->prof && c->desc.m1 <= 128 && c->desc.m2 <= 128;
}

// Cholesky of a factor with 128 < m <= 256 inside the step: [[A11, .], [A21, A22]] with a 128-block A11, all four jitter levels
// side by side as independent plain factorisations of K + jitter I (the step's graph cannot walk the jitter ladder on the host):
// per level  L11 = chol(A11), L21 = A21 L11^-T (substitution), S22 = A22 - L21 L21^T (MFMA), L22 = chol(S22); the lowest level
// whose two blocks both survive is selected on the device.  Leaves L0, the inverses of its 16 x 16 diagonal blocks (Dinv0), the
// jitter value and the status word -- exactly what the m <= 128 launch leaves.  (The single-workgroup generic kernel this replaces
// took 3.0 ms at m = 256: profiles/r2_md256_kernel_stats.csv.)
static int vg_chol_big_enqueue(vggp_ctx* c, const int* dims, int ndims, hipStream_t st) {
    VgCholJob cj[VG_CHOL_MAXJOBS_HOST];
    int nj = 0;
    for (int t = 0; t < ndims; ++t) {
        VgDim& d = c->d[dims[t]];
        VG_HIP(vg_jitcopy_launch(d.K0, d.m, d.Kc, st));
    }
    const long NB = 128;
    for (int stage = 0; stage < 2; ++stage) {
        if (stage == 1) {
            // L21^T = L11^-1 A21^T by substitution (transposed views), then S22 = A22 + jitter I - L21 L21^T in place
            VgTrsmJob tj[8];
            int nt = 0;
            VgGemmBatch g;
            vg_gemm_init(&g);
            for (int t = 0; t < ndims; ++t) {
                VgDim& d = c->d[dims[t]];
                const long m = d.m, mm = m * m, rest = m - NB, nb16 = (m + 15) / 16;
                for (int l = 0; l < 4; ++l) {
                    double* Kl = d.Kc + l * mm;
                    double* Ll = d.Lc + l * mm;
                    // R(k, c) = A21[c][k], X(k, c) -> L21[c][k]: both with element (k, c) at base[k + c * m]
                    tj[nt++] = VgTrsmJob{Ll, d.Dc + l * nb16 * 256, Kl + NB * m, Ll + NB * m, m, 256, 16, 1, m, 1, m, rest, (int)NB, 0};
                    vg_gemm_add(&g, Ll + NB * m, m, 1, Ll + NB * m, 1, m, Kl + NB * m + NB, (int)m, (int)rest, (int)rest, (int)NB, 1, 0, 1, 0, -1.0, 1);
                }
            }
            VG_HIP(vg_trsm_launch(tj, nt, st));
            VG_HIP(vg_gemm_launch(&g, st));
        }
        nj = 0;
        for (int t = 0; t < ndims; ++t) {
            VgDim& d = c->d[dims[t]];
            const long m = d.m, mm = m * m, nb16 = (m + 15) / 16;
            const long o = stage ? NB : 0;
            const int mb = stage ? (int)(m - NB) : (int)NB;
            for (int l = 0; l < 4; ++l) {
                VgCholJob j{d.Kc + l * mm + o * m + o, d.Lc + l * mm + o * m + o, nullptr, d.chol_scratch, nullptr, d.st8 + 2 * l + stage, mb};
                j.ldk = (int)m; j.ldl = (int)m; j.only_level0 = 1;
                j.Dinv_out = d.Dc + l * nb16 * 256 + (o / 16) * 256;
                cj[nj++] = j;
            }
        }
        VG_HIP(vg_chol_launch(cj, nj, st));
    }
    for (int t = 0; t < ndims; ++t) {
        VgDim& d = c->d[dims[t]];
        VG_HIP(vg_cholsel_launch(d.Lc, d.Dc, d.st8, d.m, d.L0, d.Dinv0, d.jitter, d.status, st));
    }
    return VGGP_OK;
}

int vg_partials_enqueue(vggp_ctx* c, const double* Y, double* payload, hipStream_t st, bool reduce, bool extrap, bool fused, bool apply_ns,
                        int early) {
    // early (1: thin chain of a fused single-rank step, finish_enqueue; 2: the regular warm chains, where [C;C1;C2] then is the first
    // rider of the eigensolver chain instead of S): the pass over Y is taken BEFORE the whitening --
    // S' = [A2;dA2] Y needs nothing but the factor build, so it rides in the Cholesky launch (250 idle CUs for 43 us), and
    // S = L2^-1 S' is a handful of extra strips of the substitution launch (L^-1 (A Y) instead of (L^-1 A) Y: the same
    // substitution, applied to other right-hand sides).  [C;C1;C2] is then complete before the Ritz solve and nothing that touches
    // Y or C is left behind it.
    const long n1 = c->desc.n1, n2 = c->desc.n2, m1 = c->desc.m1, m2 = c->desc.m2;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    hipStream_t sx = vg_side(c, st);
    VgGemmBatch g;
    VG_MARK(-1);

    // 1. factor build (unit outputscale): A0|dA0, K0, dK0 for both dimensions.  First node of the step: it reads the
    //    hyper-parameters from the pinned host block (and leaves a device copy for the later kernels), and its block 0
    //    zeroes the status words, the jitter-level flags and the Jacobi progress words of the step.
    VgFactorJob fj[2];
    VgCholJob cj[2];
    // m <= 128: the Cholesky launch only leaves L and the inverses of its 16 x 16 diagonal blocks; L^{-1} itself (wanted by
    // Mk = X L^{-T} and by the read-outs) comes out of the substitution launch as one more right-hand side, the identity
    static const bool chol_legacy = getenv("VGGP_CHOL_LEGACY") != nullptr;
    const bool dinv_path = d1.m <= VG_TRSM_BLK && d2.m <= VG_TRSM_BLK && !chol_legacy;
    VgClearArgs clr;
    clr.n = 0;
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        fj[k] = VgFactorJob{d.x, d.grid, d.AD, d.AD + (long)d.m * d.n, d.K0, d.dK0, d.n, d.m, d.kind, d.basis, k, 0.0, c->desc.flags};
        clr.ptr[clr.n] = d.status; clr.nwords[clr.n++] = 2;
        clr.ptr[clr.n] = reinterpret_cast<int*>(d.chol_scratch); clr.nwords[clr.n++] = 16;     // jitter-level flags
        clr.ptr[clr.n] = d.counters; clr.nwords[clr.n++] = 8;                                  // Jacobi progress words (counters, counters2)
        if (k == 0) { clr.ptr[clr.n] = reinterpret_cast<int*>(c->payload + c->payload_len); clr.nwords[clr.n++] = 2; }   // peer-failure word
        if (d.st8) { clr.ptr[clr.n] = d.st8; clr.nwords[clr.n++] = 8; }                        // block statuses of the m > 128 Cholesky
        cj[k] = VgCholJob{d.K0, d.L0, dinv_path ? nullptr : d.Linv0, d.chol_scratch, d.jitter, d.status, d.m};
        cj[k].Dinv_out = d.Dinv0;
    }
    VG_HIP(vg_factor_build_launch(fj, 2, c->d_htheta, st, c->theta, &clr));
    VG_MARK(0);

    // 2. Cholesky (+ jitter schedule) and explicit inverse of both factors
    // The Newton-Schulz step of the predicted start basis, Fp += -0.5 Wp Ep (operands left by the previous step's tail, see
    // finish_enqueue), needs nothing of this step: it rides in the Cholesky launch (8 workgroups on 256 CUs) when that is the
    // MFMA kernel, otherwise in a GEMM launch further down.
    const bool ns_on_chol = extrap && apply_ns && dinv_path;
    VgGemmBatch gns;
    vg_gemm_init(&gns);
    if (ns_on_chol)
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&gns, d.Wp, d.m, 1, d.Ep, d.m, 1, d.Fp, d.m, d.m, d.m, d.m, 1, 0, 1, 0, -0.5, 1);
        }
    static const bool chol_big_off = getenv("VGGP_NO_CHOL_BIG") != nullptr;
    const bool any_big = (d1.m > VG_TRSM_BLK || d2.m > VG_TRSM_BLK) && !chol_legacy && !chol_big_off;
    VgGemmBatch gsp;
    vg_gemm_init(&gsp);
    int sp_slabs = 1;
    if (early) {
        // (two callers: the fused single-rank step -- no reduction, [C;C1;C2] becomes a rider of the eigensolver chain -- and the
        //  partials half of a multi-rank step -- reduce: [C;C1;C2] shares the Gram launch and the slab reduction fills the payload)
        const bool ok_fused = fused && !reduce && (vg_ride(c) || c->prof);
        const bool ok_partials = reduce;                 // (also the cold thin step: fused, reduced operands)
        if (!(dinv_path && !any_big && (ok_fused || ok_partials) && Y && sx == st)) {
            vg_set_error("internal: the early projection was requested where it cannot run");
            return VGGP_ESTATE;
        }
        // split-K only as far as the chip needs it: the plan's split of S (256 workgroups) capped at VG_SP_SLABS -- 4 slabs at 1024^2, ONE
        // for a 1024 x 4096 slab, whose 64 x 64 tiles fill the chip by themselves (no slab traffic, two substitution jobs instead of eight)
        vg_gemm_add(&gsp, d2.AD, n2, 1, Y, n1, 1, c->Sp, (int)n1, (int)(2 * m2), (int)n1, (int)n2, std::min(c->st_split, VG_SP_SLABS), 2L * m2 * n1);
        sp_slabs = gsp.p[0].ksplit;
        if (c->prof) {                 // profiling mode: every launch group by itself -- the pass over Y as a launch of its own
            vg_gemm_xcd_group(&gsp, 0);
            VG_HIP(vg_gemm_launch(&gsp, st, VG_GEMM_TAG_GRAM_PROJECT));
            VG_MARK(4);
        }
    }
    if (!any_big) {
        // riders of the Cholesky launch: the early pass over Y and / or the Newton-Schulz step of the predicted start basis
        VgGemmBatch gr2 = gns;
        if (early && !c->prof) {
            gr2 = gsp;
            if (ns_on_chol)
                for (int k = 0; k < 2; ++k) {
                    VgDim& d = c->d[k];
                    vg_gemm_add(&gr2, d.Wp, d.m, 1, d.Ep, d.m, 1, d.Fp, d.m, d.m, d.m, d.m, 1, 0, 1, 0, -0.5, 1);
                }
        }
        VG_HIP(vg_chol_launch(cj, 2, st, ((early && !c->prof) || ns_on_chol) ? &gr2 : nullptr));
    } else {
        // a factor beyond one 128-block: blocked (vg_chol_big_enqueue); a small partner keeps the one-launch MFMA kernel.  Either
        // way the launch leaves L0 and the diagonal-block inverses only; L0^-1 comes out of the substitution below.
        int big[2], nbig = 0;
        for (int k = 0; k < 2; ++k) {
            if (c->d[k].m > VG_TRSM_BLK) big[nbig++] = k;
            else { cj[k].Linv = nullptr; VG_HIP(vg_chol_launch(&cj[k], 1, st)); }
        }
        const int rcb = vg_chol_big_enqueue(c, big, nbig, st);
        if (rcb) return rcb;
    }
    c->dinv_valid = dinv_path || any_big;
    VG_MARK(1);

    // 3. B|V = L0^{-1} [A0|dA0],  X = L0^{-1} dK0  by blocked substitution on the matrix cores (trsm.hip; the 16 x 16
    //    diagonal-block inverses are the diagonal blocks of Linv0).  Not "Linv0 times A0": on RBF factors that product
    //    carries cond(L0) into every element of B and shows in the posterior variance of large grids.
    {
        const bool big = d1.m > VG_TRSM_BLK || d2.m > VG_TRSM_BLK;
        if (!big) {
            VgTrsmJob tj[16];
            int nt = 0;
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                const long mn = (long)d.m * d.n;
                const double* dv = dinv_path ? d.Dinv0 : d.Linv0;
                const long dblk = dinv_path ? 256 : 16L * d.m + 16, dld = dinv_path ? 16 : d.m;
                for (int b = 0; b < 2; ++b)
                    tj[nt++] = VgTrsmJob{d.L0, dv, d.AD + b * mn, d.BV + b * mn, d.m, dblk, dld, d.n, 1, d.n, 1, d.n, d.m, 0};
                tj[nt++] = VgTrsmJob{d.L0, dv, d.dK0, d.X, d.m, dblk, dld, d.m, 1, d.m, 1, d.m, d.m, 0};
                if (dinv_path) {
                    tj[nt] = VgTrsmJob{d.L0, dv, d.L0, d.Linv0, d.m, dblk, dld, d.m, 1, d.m, 1, d.m, d.m, 0};      // Linv0 = L0^{-1} I
                    tj[nt++].rhs_ident = 1;
                }
            }
            if (early)               // S = L2^-1 S', slab by slab (the substitution is linear: the consumer sums the slabs)
                for (int sl = 0; sl < sp_slabs; ++sl)
                    for (int b = 0; b < 2; ++b) {
                        const long o = (long)sl * 2 * m2 * n1 + (long)b * m2 * n1;
                        tj[nt++] = VgTrsmJob{d2.L0, d2.Dinv0, c->Sp + o, c->St + o, m2, 256, 16, n1, 1, n1, 1, n1, (int)m2, 0};
                    }
            VG_HIP(vg_trsm_launch(tj, nt, st));
        } else {
            // 128 < m <= 256: blocked (128-row diagonal solves + one GEMM update between them), in place on copies
            VgTrsmSpec q[8];
            int nq = 0;
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                const long mn = (long)d.m * d.n;
                // diagonal-block inverses: Dinv0 when the Cholesky left them there, else the diagonal blocks of its explicit inverse
                const double* dv = any_big ? d.Dinv0 : d.Linv0;
                const long dblk = any_big ? 256 : 16L * d.m + 16, dld = any_big ? 16 : d.m;
                VG_HIP(hipMemcpyAsync(d.BV, d.AD, sizeof(double) * 2 * mn, hipMemcpyDeviceToDevice, st));
                VG_HIP(hipMemcpyAsync(d.X, d.dK0, sizeof(double) * d.m * d.m, hipMemcpyDeviceToDevice, st));
                for (int b = 0; b < 2; ++b) q[nq++] = VgTrsmSpec{d.L0, d.m, dv, dblk, dld, d.BV + b * mn, d.n, 1, d.n, d.m, 0};
                q[nq++] = VgTrsmSpec{d.L0, d.m, dv, dblk, dld, d.X, d.m, 1, d.m, d.m, 0};
                if (any_big) {           // L0^-1 = L0^-1 I, as in the m <= 128 launch
                    VG_HIP(hipMemcpyAsync(d.Linv0, d.Id, sizeof(double) * d.m * d.m, hipMemcpyDeviceToDevice, st));
                    q[nq++] = VgTrsmSpec{d.L0, d.m, dv, dblk, dld, d.Linv0, d.m, 1, d.m, d.m, 0};
                }
            }
            const int rc = trsm_batch(q, nq, st);
            if (rc) return rc;
        }
    }
    VG_MARK(2);

    if (!Y) {          // factors only (scattered step, masked.hip): B|V, X are done; Mk = X Linv0^T and nothing of the grid path
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, d.X, d.m, 1, d.Linv0, 1, d.m, d.Mk, d.m, d.m, d.m, d.m);
        }
        VG_HIP(vg_gemm_launch(&g, st));
        return VGGP_OK;
    }
    // 4./5. (side stream in the fused step: overlaps the Gram products and the whole eigensolver chain, which need only G, H)
    //    S = [B2;V2] Y (split-K slabs): the only pass over Y; then [C;C1] = [B1;V1] S_B, C2 = B1 S_V (S slabs summed on
    //    load; split-K over n1).
    hipStream_t sp = (fused && !extrap) ? sx : st;
    // "Riders" (fused warm step): the two projection launches are not enqueued here at all -- their batches are handed to
    // finish_enqueue, which attaches them to the single-workgroup launches of the eigensolver chain (row QR / Ritz solve /
    // main solve) as extra workgroups, so they run BESIDE that chain on the 250 idle CUs instead of in front of it.
    const bool ride = fused && !reduce && sp == st && vg_ride(c);
    if (sp != st) VG_FORK(1);
    VgGemmBatch gp, gc;
    vg_gemm_init(&gp);
    vg_gemm_add(&gp, d2.BV, n2, 1, Y, n1, 1, c->St, (int)n1, (int)(2 * m2), (int)n1, (int)n2, c->st_split, 2L * m2 * n1);
    const int st_slabs = early ? sp_slabs : gp.p[0].ksplit;
    vg_gemm_xcd_group(&gp, 0);
    // predicted start basis of this step's eigensolvers: the previous step's tail left Qpred (Ep), 1.5 Qpred (Fp) and
    // W = Qpred Qpred^T (Wp); the Newton-Schulz step Fp += -0.5 W Qpred rides in this launch (see the tail of finish_enqueue)
    // -- or in the Gram launch when the projection itself is deferred
    auto add_ns = [&](VgGemmBatch* b) {
        if (extrap && apply_ns && !ns_on_chol)
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                vg_gemm_add(b, d.Wp, d.m, 1, d.Ep, d.m, 1, d.Fp, d.m, d.m, d.m, d.m, 1, 0, 1, 0, -0.5, 1);
            }
    };
    if (!ride && !early) {
        add_ns(&gp);
        VG_HIP(vg_gemm_launch(&gp, sp, VG_GEMM_TAG_GRAM_PROJECT));
        if (sp == st) VG_MARK(4);
    }
    vg_gemm_init(&gc);
    const long cc_slab = 3L * m1 * m2;
    vg_gemm_add(&gc, d1.BV, n1, 1, c->St, 1, n1, c->CCslab, (int)m2, (int)(2 * m1), (int)m2, (int)n1, c->cc_split, cc_slab,
                st_slabs, 2L * m2 * n1);
    vg_gemm_add(&gc, d1.BV, n1, 1, c->St + m2 * n1, 1, n1, c->CCslab + 2 * m1 * m2, (int)m2, (int)m1, (int)m2, (int)n1,
                c->cc_split, cc_slab, st_slabs, 2L * m2 * n1);
    const int cc_slabs = gc.p[0].ksplit;
    // 6. Gram pairs [G0;H0] = [B;V] B^T (split-K slabs), Mk = X Linv0^T: independent of the projection, so they share ITS
    //    second launch (whose own products fill less than half the chip) unless the projection branch runs on the side stream
    int gh_slabs[2] = {1, 1};
    auto add_gram = [&](VgGemmBatch* b) {
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const int ig = vg_gemm_add(b, d.BV, d.n, 1, d.BV, 1, d.n, d.GHslab, d.m, 2 * d.m, d.m, d.n, d.gh_split, 2L * d.m * d.m);
            gh_slabs[k] = b->p[ig].ksplit;
            vg_gemm_add(b, d.X, d.m, 1, d.Linv0, 1, d.m, d.Mk, d.m, d.m, d.m, d.m);
        }
    };
    if (ride) {
        vg_gemm_init(&g);
        add_gram(&g);
        add_ns(&g);
        VG_HIP(vg_gemm_launch(&g, st));
        c->ride_proj = gp; c->ride_cc = gc;
        c->ride_pending = early != 1;      // (thin chain: S exists already and finish_enqueue places [C;C1;C2] itself)
        c->ride_stage0 = early == 2 ? 1 : 0;
    } else {
        if (sp == st) add_gram(&gc);
        VG_HIP(vg_gemm_launch(&gc, sp));
        if (sp == st) { VG_MARK(5); VG_MARK(6); }
        if (sp != st) {
            VG_JOIN_RECORD(1);
            vg_gemm_init(&g);
            add_gram(&g);
            VG_HIP(vg_gemm_launch(&g, st));
            VG_MARK(6);
        }
    }

    c->gh_slabs[0] = gh_slabs[0]; c->gh_slabs[1] = gh_slabs[1]; c->cc_slabs = cc_slabs; c->st_slabs = st_slabs;
    // 7. deterministic slab reduction into {G1,H1} (local) and the payload {G2,H2,C,C1,C2}.  Skipped inside a fused
    //    warm step: its consumers (three small GEMMs) then sum the slabs on load, one launch less on the critical path.
    if (!reduce) return VGGP_OK;          // fused warm step: the caller joins the projection branch before the rotations
    if (sp != st) VG_JOIN_WAIT(1);
    VgRedBatch r;
    vg_red_init(&r);
    vg_red_add(&r, d1.GHslab, d1.GH, 2L * m1 * m1, 2L * m1 * m1, gh_slabs[0]);
    vg_red_add(&r, d2.GHslab, payload, 2L * m2 * m2, 2L * m2 * m2, gh_slabs[1]);
    vg_red_add(&r, c->CCslab, payload + 2 * m2 * m2, 3L * m1 * m2, cc_slab, cc_slabs);
    VG_HIP(vg_red_launch(&r, st));
    VG_MARK(7);
    return VGGP_OK;
}

// The subspace start is fed the previous basis itself (U = I: the partials' extrapolation products reduce to its Newton-Schulz
// clean-up).  VGGP_SUB_EXTRAP=1 feeds it the extrapolated basis instead: 5 us faster at m = 128, but at m = 64 the main
// eigensolver was seen to need 55 rounds instead of ~10, so it is not the default.
static bool vg_sub_ident() { static const bool ex = getenv("VGGP_SUB_EXTRAP") != nullptr; return !ex; }

static int finish_enqueue(vggp_ctx* c, const double* payload, double yy_total, bool warm, hipStream_t st, bool copy_theta,
                          bool from_slabs = false, bool extrap = false, bool refine = false, bool subspace = false, bool thin = false,
                          int newton = 0, bool thin_early = false) {
    // stand-alone finish (multi-rank seam): refresh the device copy of the hyper-parameters; inside a fused step the
    // factor kernel already did
    if (copy_theta) VG_HIP(hipMemcpyAsync(c->theta, c->h_theta, 6 * sizeof(double), hipMemcpyHostToDevice, st));
    const long m1 = c->desc.m1, m2 = c->desc.m2;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    // from_slabs (fused warm step): G, H, C are still split-K slabs; every consumer below is a GEMM that sums them on load
    const double* G0[2] = {from_slabs ? d1.GHslab : d1.GH, from_slabs ? d2.GHslab : payload};
    const double* H0[2] = {G0[0] + m1 * m1, G0[1] + m2 * m2};
    const double* C3 = from_slabs ? c->CCslab : payload + 2 * m2 * m2;
    const int ghn[2] = {from_slabs ? c->gh_slabs[0] : 1, from_slabs ? c->gh_slabs[1] : 1};
    const long ghs[2] = {2L * m1 * m1, 2L * m2 * m2};
    const int ccn = from_slabs ? c->cc_slabs : 1;
    const long ccs = 3L * m1 * m2;
    VgGemmBatch g;
    const bool ride = from_slabs && c->ride_pending;      // the projection launches were deferred to this chain (vg_partials_enqueue)
    int ride_stage = ride ? c->ride_stage0 : 0;           // 0: S pending, 1: [C;C1;C2] pending (S came out of the early projection), 2: done

    // 7. eigendecompositions (optionally warm-started from the previous step's basis)
    hipStream_t sx = vg_side(c, st);
    VG_MARK(-1);                  // (re)start the stage clock: the caller's all-reduce sits between the two halves
    const double* Hr[2] = {H0[0], H0[1]};      // H for the rotation below: a reduced copy when a warm chain's first launch made one
    int hrn[2] = {ghn[0], ghn[1]};
    VgEigJob ej[2];
    if (warm && subspace && thin) {
        // ---- THIN chain (thin.hip): numerically rank-deficient Gram matrices, r = numerical rank + margin <= 32 per dimension.
        // V = the r leading eigenvectors of the previous step (rows of QtPrev); Z = V G (one step of subspace iteration: G
        // annihilates every null component); V1 = orth(Z) spans range(G); Rayleigh-Ritz on V1 G V1^T gives the range eigenpairs
        // (W, theta) -- and that is all the ELBO and its gradient need: every rotated quantity is W (V1 . V1^T) W^T of an
        // r x r matrix, formed by the tail kernel itself.  No complement basis, no full eigensolve, no m x m rotation.
        // early (vg_partials_enqueue took the pass over Y before the whitening: S is there): [C;C1;C2] rides beside the row QR,
        // {C,C1,C2} V1_2^T joins the launch after it, V1_1 (.) rides in the Ritz launch, the tail kernel follows the Ritz solve directly
        // and the new range basis E_r = W V1 -- which only the NEXT step reads -- trails behind the tail (the host has its results by then)
        // (multi-rank step: the partials half took the early pass and [C;C1;C2] arrives reduced in the payload -- same placement,
        //  no rider beside the row QR)
        const bool early = thin_early && !ride;
        const double* Gr[2] = {G0[0], G0[1]};
        int grn[2] = {ghn[0], ghn[1]};
        double* gdst[2] = {d1.GH, c->payload};
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, d.QtPrev, d.m, 1, G0[k], d.m, 1, d.TM, d.m, d.sub_r, d.m, d.m, 1, 0, ghn[k], ghs[k]);            // Z = V G
            if (from_slabs && ghn[k] > 1) {                  // reduced copies of G and H for the later readers
                vg_gemm_add(&g, d.Id, d.m, 1, G0[k], d.m, 1, gdst[k], d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
                vg_gemm_add(&g, d.Id, d.m, 1, H0[k], d.m, 1, gdst[k] + (long)d.m * d.m, d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
            }
        }
        VG_HIP(vg_gemm_launch(&g, st));
        VG_MARK(8);
        for (int k = 0; k < 2; ++k)
            if (from_slabs && ghn[k] > 1) { Gr[k] = gdst[k]; grn[k] = 1; Hr[k] = gdst[k] + (long)c->d[k].m * c->d[k].m; hrn[k] = 1; }
        VgRowQrJob qj[2];
        for (int k = 0; k < 2; ++k) { VgDim& d = c->d[k]; qj[k] = VgRowQrJob{d.TM, d.V1s, d.sub_r, d.m, nullptr, nullptr, 0}; }
        // + rider: [C;C1;C2] (early; profiling mode has launched it by itself) / S = [B2;V2] Y
        VG_HIP(vg_rowqr_launch(qj, 2, st, early ? ((c->prof || !from_slabs) ? nullptr : &c->ride_cc) : (ride ? &c->ride_proj : nullptr)));
        if (ride) ride_stage = 1;
        VG_MARK(9);
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const int r = d.sub_r;
            vg_gemm_add(&g, d.V1s, d.m, 1, Gr[k], d.m, 1, d.Zs, d.m, r, d.m, d.m, 1, 0, grn[k], ghs[k]);        // T = V1 G
            vg_gemm_add(&g, d.V1s, d.m, 1, d.Mk, d.m, 1, d.tMV, d.m, r, d.m, d.m);                              // V1 Mk0
            vg_gemm_add(&g, d.V1s, d.m, 1, Hr[k], d.m, 1, d.tHV, d.m, r, d.m, d.m, 1, 0, hrn[k], ghs[k]);       // V1 H0
        }
        if (early) {
            const int ic = vg_gemm_add(&g, C3, m2, 1, d2.V1s, 1, m2, c->tCV, d2.sub_r, (int)(3 * m1), d2.sub_r, (int)m2);   // {C,C1,C2} V1_2^T
            g.p[ic].a_nslab = ccn; g.p[ic].a_slab = ccs;
        }
        VG_HIP(vg_gemm_launch(&g, st));
        VG_MARK(10);
        // Ritz solve (forms H = T V1^T itself, Newton iteration); riders: the r x r sandwiches, and [C;C1;C2] of the fused step
        VgGemmBatch gx;
        vg_gemm_init(&gx);
        if (ride) gx = c->ride_cc;
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const int r = d.sub_r;
            vg_gemm_add(&gx, d.tMV, d.m, 1, d.V1s, 1, d.m, d.tAM, r, r, r, d.m);                                // V1 Mk0 V1^T
            vg_gemm_add(&gx, d.tHV, d.m, 1, d.V1s, 1, d.m, d.tAH, r, r, r, d.m);                                // V1 H0 V1^T
        }
        if (early)
            for (int q = 0; q < 3; ++q)                                                                         // V1_1 ({C,C1,C2} V1_2^T)
                vg_gemm_add(&gx, d1.V1s, m1, 1, c->tCV + (long)q * m1 * d2.sub_r, d2.sub_r, 1, c->tAC + (long)q * d1.sub_r * d2.sub_r,
                            d2.sub_r, d1.sub_r, d2.sub_r, (int)m1);
        VgEigJob sj[2];
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            sj[k] = VgEigJob{nullptr, d.lam_s, d.Ws, nullptr, d.gwork2, d.rotlog2, d.roundlog2, d.counters2, d.sub_r, d.max_rounds,
                             (long)vg_eigh_log_bytes(d.m), 0};
            sj[k].Hl = d.Zs; sj[k].Hr = d.V1s; sj[k].hk = d.m;
            sj[k].perm = d.perm2;
            sj[k].err = d.status + 1;
            sj[k].newton = 1;
        }
        VG_HIP(vg_eigh_launch(sj, 2, st, &gx));
        if (ride) ride_stage = 2;
        c->ride_pending = false;
        VG_MARK(11);
        const int ac_nslab = 1;
        const long ac_slab = 3L * d1.sub_r * d2.sub_r;
        if (!early) {
            // {C, C1, C2} V1_2^T, and the new range basis E_r = W V1 into the leading rows of QtPrev (next step's V)
            vg_gemm_init(&g);
            {
                const int ic = vg_gemm_add(&g, C3, m2, 1, d2.V1s, 1, m2, c->tCV, d2.sub_r, (int)(3 * m1), d2.sub_r, (int)m2);
                g.p[ic].a_nslab = ccn; g.p[ic].a_slab = ccs;
            }
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                vg_gemm_add(&g, d.Ws, d.sub_r, 1, d.V1s, d.m, 1, d.QtPrev, d.m, d.sub_r, d.m, d.sub_r);
            }
            VG_HIP(vg_gemm_launch(&g, st));
            VG_MARK(14);
            vg_gemm_init(&g);
            for (int q = 0; q < 3; ++q)
                vg_gemm_add(&g, d1.V1s, m1, 1, c->tCV + (long)q * m1 * d2.sub_r, d2.sub_r, 1, c->tAC + (long)q * d1.sub_r * d2.sub_r, d2.sub_r,
                            d1.sub_r, d2.sub_r, (int)m1);
            VG_HIP(vg_gemm_launch(&g, st));
            VG_MARK(15);
        }
        VgThinTail tt{};
        tt.theta = c->theta; tt.n_total = (double)c->desc.n_total; tt.yy = yy_total;
        tt.r1 = d1.sub_r; tt.r2 = d2.sub_r; tt.m1 = (int)m1; tt.m2 = (int)m2;
        tt.W1 = d1.Ws; tt.W2 = d2.Ws; tt.lam1 = d1.lam_s; tt.lam2 = d2.lam_s;
        tt.AM1 = d1.tAM; tt.AH1 = d1.tAH; tt.AM2 = d2.tAM; tt.AH2 = d2.tAH; tt.AC = c->tAC;
        tt.ac_nslab = ac_nslab; tt.ac_slab = ac_slab;
        tt.G1 = Gr[0]; tt.H1 = Hr[0]; tt.G2 = Gr[1]; tt.H2 = Hr[1];
        tt.out = c->out; tt.hout = c->d_hout;
        tt.beta_out = c->beta; tt.invd_out = c->invD;
        tt.peer_fail = (payload == c->payload && (c->n_ranks > 1 || c->comm || c->cb)) ? c->payload + c->payload_len : nullptr;
        for (int k = 0; k < 2; ++k) { tt.jit[k] = c->d[k].jitter; tt.status[k] = c->d[k].status; tt.rcounters[k] = c->d[k].counters2; }
#ifdef VGGP_DIAG
        tt.stamps = reinterpret_cast<unsigned long long*>(c->tAC + 3 * VG_THIN_MAXR * VG_THIN_MAXR);
#endif
        VG_HIP(vg_thin_tail_launch(&tt, st));
        VG_MARK(18);
        if (early) {           // E_r = W V1 -> leading rows of QtPrev: behind the tail, whose pinned result block the host polls
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                vg_gemm_add(&g, d.Ws, d.sub_r, 1, d.V1s, d.m, 1, d.QtPrev, d.m, d.sub_r, d.m, d.sub_r);
            }
            VG_HIP(vg_gemm_launch(&g, st));
        }
        return VGGP_OK;
    }
    if (warm && subspace) {
        // ---- subspace start (numerically rank-deficient G, e.g. RBF: rank ~20 of 128).  S = d.Fp is the previous basis after
        // one Newton-Schulz step (partials ran with U = I), rows sorted by decreasing eigenvalue; its r leading rows V span
        // the previous numerical range.  One step of subspace iteration, Z = V G, removes every null-space component
        // exactly (G annihilates them); V1 = orth(Z); Rayleigh-Ritz on H = V1 G V1^T gives the r leading eigenpairs;
        // the other rows are the previous ones projected off span(V1) (the next step's Newton-Schulz restores their
        // orthonormality, 1e-5 off after the projection).  Q G Q^T is then diagonal up to a handful of elements
        // (tools/studies/subspace_rbf_study.py: 1-3 rotations left instead of ~12000), which the eigensolver finds by scan.
        VgRowQrJob qj[2];
        // G is read three times in this chain; in the fused step it is still split-K slabs, so the first launch also leaves a
        // reduced copy ("Id G" through the slab-summing GEMM) for the other two
        const double* Gr[2] = {G0[0], G0[1]};
        int grn[2] = {ghn[0], ghn[1]};
        double* gdst[2] = {d1.GH, c->payload};
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, d.Fp, d.m, 1, G0[k], d.m, 1, d.TM, d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);           // S G (rows < r: Z = V G)
            if (from_slabs && ghn[k] > 1) {
                vg_gemm_add(&g, d.Id, d.m, 1, G0[k], d.m, 1, gdst[k], d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
                vg_gemm_add(&g, d.Id, d.m, 1, H0[k], d.m, 1, gdst[k] + (long)d.m * d.m, d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
            }
        }
        VG_HIP(vg_gemm_launch(&g, st));
        VG_MARK(8);
        for (int k = 0; k < 2; ++k)
            if (from_slabs && ghn[k] > 1) { Gr[k] = gdst[k]; grn[k] = 1; Hr[k] = gdst[k] + (long)c->d[k].m * c->d[k].m; hrn[k] = 1; }
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const long r = d.sub_r;
            qj[k] = VgRowQrJob{d.TM, d.V1s, d.sub_r, d.m, d.Fp + r * d.m, d.E + r * d.m, (long)(d.m - r) * d.m};   // + E[r:] <- S[r:]
        }
        VG_HIP(vg_rowqr_launch(qj, 2, st, ride ? &c->ride_proj : nullptr));      // + rider: S = [B2;V2] Y
        if (ride) ride_stage = 1;
        VG_MARK(9);
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const long r = d.sub_r;
            vg_gemm_add(&g, d.V1s, d.m, 1, Gr[k], d.m, 1, d.Zs, d.m, (int)r, d.m, d.m, 1, 0, grn[k], ghs[k]);        // T = V1 G (Z is spent)
            vg_gemm_add(&g, d.Fp + r * d.m, d.m, 1, d.V1s, 1, d.m, d.TH, (int)r, (int)(d.m - r), (int)r, d.m);           // P = S[r:] V1^T
        }
        VG_HIP(vg_gemm_launch(&g, st));
        VG_MARK(10);
        // The Ritz matrix H = T V1^T is formed by the Ritz launch's producer workgroups themselves (VgEigJob::Hl/Hr).  The
        // complement rows do not wait for the Ritz vectors: E[r:] = S[r:] - P V1 and, by linearity,
        // (E G)[r:] = (S G)[r:] - P T -- both ride in the Ritz launch as extra workgroups (with [C;C1;C2] in the fused step).
        VgGemmBatch gx;
        vg_gemm_init(&gx);
        if (ride) gx = c->ride_cc;
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const long r = d.sub_r;
            vg_gemm_add(&gx, d.TH, r, 1, d.V1s, d.m, 1, d.E + r * d.m, d.m, (int)(d.m - r), d.m, (int)r, 1, 0, 1, 0, -1.0, 1);    // E[r:] -= P V1
            vg_gemm_add(&gx, d.TH, r, 1, d.Zs, d.m, 1, d.TM + r * d.m, d.m, (int)(d.m - r), d.m, (int)r, 1, 0, 1, 0, -1.0, 1);   // (S G)[r:] -= P T
        }
        VgEigJob sj[2];
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            sj[k] = VgEigJob{nullptr, d.lam_s, d.Ws, nullptr, d.gwork2, d.rotlog2, d.roundlog2, d.counters2, d.sub_r, d.max_rounds,
                             (long)vg_eigh_log_bytes(d.m), 0};
            sj[k].Hl = d.Zs; sj[k].Hr = d.V1s; sj[k].hk = d.m;
            sj[k].perm = d.perm2;
            sj[k].err = d.status + 1;
            sj[k].newton = 1;
        }
        VG_HIP(vg_eigh_launch(sj, 2, st, &gx));                              // Ritz pairs (+ riders)
        if (ride) ride_stage = 2;
        VG_MARK(11);
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const long r = d.sub_r;
            vg_gemm_add(&g, d.Ws, r, 1, d.V1s, d.m, 1, d.E, d.m, (int)r, d.m, (int)r);                                // E[:r] = W V1
            vg_gemm_add(&g, d.Ws, r, 1, d.Zs, d.m, 1, d.TM, d.m, (int)r, d.m, (int)r);                                 // (E G)[:r] = W T
        }
        VG_HIP(vg_gemm_launch(&g, st));
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, d.TM, d.m, 1, d.E, 1, d.m, d.Gw, d.m, d.m, d.m, d.m);
        }
        VG_HIP(vg_gemm_launch(&g, st));
    } else if (warm && newton > 0) {
        // ---- Newton chain: full-rank Gram matrices whose warm start is too far off for the first-order refinement (or too large
        // for the LDS eigensolver, m > 128).  From the start basis S (rows), Gw = S G S^T; then `newton` times
        //     E_ij = g_ij / (g_jj - g_ii) (skew, elements above the threshold),  S <- (I + E + E^2 / 2) S,
        //     one Newton-Schulz step (not after the last iteration),  Gw = S G S^T
        // -- quadratic: |E| 1e-1 -> 1e-2 -> 1e-4 -> 1e-8 -- every product a batched MFMA GEMM over the whole chip, no single-workgroup
        // sweep.  The last look at Gw (vg_newton_check_kernel) takes the eigenvalues from its diagonal and raises the retry bit when
        // an off-diagonal element is still above the threshold or a rotation was too large (a crossing, a degenerate pair): the
        // host then repeats the step on the regular chain.  Eigenpairs stay in the ORDER of the start basis (no sort).
        const double* Gr[2] = {G0[0], G0[1]};
        int grn[2] = {ghn[0], ghn[1]};
        double* gdst[2] = {d1.GH, c->payload};
        const double* Scur[2] = {extrap ? d1.Fp : d1.QtPrev, extrap ? d2.Fp : d2.QtPrev};
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, Scur[k], d.m, 1, G0[k], d.m, 1, d.TM, d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
            if (from_slabs && ghn[k] > 1) {                  // reduced copies: G is read once per iteration, H in the rotation
                vg_gemm_add(&g, d.Id, d.m, 1, G0[k], d.m, 1, gdst[k], d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
                vg_gemm_add(&g, d.Id, d.m, 1, H0[k], d.m, 1, gdst[k] + (long)d.m * d.m, d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
            }
        }
        VG_HIP(vg_gemm_launch(&g, st));
        VG_MARK(8);
        for (int k = 0; k < 2; ++k)
            if (from_slabs && ghn[k] > 1) { Gr[k] = gdst[k]; grn[k] = 1; Hr[k] = gdst[k] + (long)c->d[k].m * c->d[k].m; hrn[k] = 1; }
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) { VgDim& d = c->d[k]; vg_gemm_add(&g, d.TM, d.m, 1, Scur[k], 1, d.m, d.Gw, d.m, d.m, d.m, d.m); }
        VG_HIP(vg_gemm_launch(&g, st));
        VG_MARK(12);
        static const bool newton_null = getenv("VGGP_NO_NULL_SKIP") == nullptr;
        int bi = 0;                                                           // rotation of the three basis buffers E, X, F
        for (int it = 0; it < newton; ++it) {
            VgRefineJob rj[2];
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                rj[k] = VgRefineJob{d.Gw, d.U, d.TH, d.m, VG_NEWTON_TOL};
                rj[k].emax = 0.3; rj[k].flag = d.counters2; rj[k].noise = VG_NEWTON_NOISE;
                // first iteration: the whole near-null cluster is left alone (its block is dominated by delta^2 lam_range until the
                // range-cluster rotations have been applied); afterwards only the exactly-null rows
                if (newton_null) { rj[k].lam_prev = d.lam0; rj[k].null_cut = it == 0 ? VG_NEWTON_CLUSTER : VG_NEWTON_NULL; }
            }
            const VgGemmBatch* rd = !ride ? nullptr : (ride_stage == 0 ? &c->ride_proj : (ride_stage == 1 ? &c->ride_cc : nullptr));
            VG_HIP(vg_refine_launch(rj, 2, st, rd));                           // E -> U, I + E -> TH  (+ riders: S, then [C;C1;C2])
            if (rd) ++ride_stage;
            VG_MARK(9);
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) { VgDim& d = c->d[k]; vg_gemm_add(&g, d.U, d.m, 1, d.U, d.m, 1, d.TH, d.m, d.m, d.m, d.m, 1, 0, 1, 0, 0.5, 1); }   // TH += E^2 / 2
            VG_HIP(vg_gemm_launch(&g, st));
            double* Snew[2];
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                double* bufs[3] = {d.E, d.X, d.F};
                Snew[k] = bufs[bi];
                vg_gemm_add(&g, d.TH, d.m, 1, Scur[k], d.m, 1, Snew[k], d.m, d.m, d.m, d.m);                 // (I + E + E^2/2) S
            }
            VG_HIP(vg_gemm_launch(&g, st));
            bi = (bi + 1) % 3;
            if (it + 1 < newton) {                                            // Newton-Schulz: S <- 1.5 S - 0.5 (S S^T) S
                double* Sns[2];
                vg_gemm_init(&g);
                for (int k = 0; k < 2; ++k) {
                    VgDim& d = c->d[k];
                    double* bufs[3] = {d.E, d.X, d.F};
                    Sns[k] = bufs[bi];
                    vg_gemm_add(&g, Snew[k], d.m, 1, Snew[k], 1, d.m, d.TM, d.m, d.m, d.m, d.m);             // W = S S^T
                    vg_gemm_add(&g, d.Id, d.m, 1, Snew[k], d.m, 1, Sns[k], d.m, d.m, d.m, d.m, 1, 0, 1, 0, 1.5, 0);
                }
                VG_HIP(vg_gemm_launch(&g, st));
                vg_gemm_init(&g);
                for (int k = 0; k < 2; ++k) { VgDim& d = c->d[k]; vg_gemm_add(&g, d.TM, d.m, 1, Snew[k], d.m, 1, Sns[k], d.m, d.m, d.m, d.m, 1, 0, 1, 0, -0.5, 1); }
                VG_HIP(vg_gemm_launch(&g, st));
                bi = (bi + 1) % 3;
                for (int k = 0; k < 2; ++k) Scur[k] = Sns[k];
            } else {
                for (int k = 0; k < 2; ++k) Scur[k] = Snew[k];
            }
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) { VgDim& d = c->d[k]; vg_gemm_add(&g, Scur[k], d.m, 1, Gr[k], d.m, 1, d.TM, d.m, d.m, d.m, d.m, 1, 0, grn[k], ghs[k]); }
            VG_HIP(vg_gemm_launch(&g, st));
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) { VgDim& d = c->d[k]; vg_gemm_add(&g, d.TM, d.m, 1, Scur[k], 1, d.m, d.Gw, d.m, d.m, d.m, d.m); }
            VG_HIP(vg_gemm_launch(&g, st));
        }
        VgNewtonCheckJob nj[2];
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            nj[k] = VgNewtonCheckJob{d.Gw, d.lam0, d.counters, d.status + 1, d.counters2, d.m, newton, VG_NEWTON_ACCEPT, VG_NEWTON_NOISE};
            if (newton_null) nj[k].null_cut = VG_NEWTON_NULL;
            VG_HIP(hipMemcpyAsync(d.Qt, Scur[k], sizeof(double) * d.m * d.m, hipMemcpyDeviceToDevice, st));
        }
        VG_HIP(vg_newton_check_launch(nj, 2, st));
        {   // converged: QtPrev2 <- QtPrev (the basis before last, for the next extrapolation), QtPrev <- the new basis; a miss leaves
            // both alone, and the host repeats the step on the regular chain from the same warm start
            const int* er[2] = {d1.status + 1, d2.status + 1};
            const double* sa[2] = {Scur[0], Scur[1]};
            double* da[2] = {d1.QtPrev, d2.QtPrev};
            const double* sb[2] = {d1.QtPrev, d2.QtPrev};
            double* db[2] = {d1.QtPrev2, d2.QtPrev2};
            const long nn[2] = {m1 * m1, m2 * m2};
            VG_HIP(vg_copy_if_launch(er, sa, da, sb, db, nn, 2, st));
        }
        VG_MARK(13);
        while (ride && ride_stage < 2) {                                      // (fewer than two iterations: what is still pending)
            VG_HIP(vg_gemm_launch(ride_stage == 0 ? &c->ride_proj : &c->ride_cc, st));
            ++ride_stage;
        }
    } else if (warm) {
        const double* Gr[2] = {G0[0], G0[1]};
        int grn[2] = {ghn[0], ghn[1]};
        double* gdst[2] = {d1.GH, c->payload};
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, extrap ? d.Fp : d.QtPrev, d.m, 1, G0[k], d.m, 1, d.TM, d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
            if (from_slabs && ghn[k] > 1) {                  // reduced copies for the later readers (G after a refinement, H in the rotation)
                if (refine) {
                    vg_gemm_add(&g, d.Id, d.m, 1, G0[k], d.m, 1, gdst[k], d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
                    Gr[k] = gdst[k]; grn[k] = 1;
                }
                vg_gemm_add(&g, d.Id, d.m, 1, H0[k], d.m, 1, gdst[k] + (long)d.m * d.m, d.m, d.m, d.m, d.m, 1, 0, ghn[k], ghs[k]);
                Hr[k] = gdst[k] + (long)d.m * d.m; hrn[k] = 1;
            }
        }
        VG_HIP(vg_gemm_launch(&g, st));
        VG_MARK(8);
        vg_gemm_init(&g);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, d.TM, d.m, 1, extrap ? d.Fp : d.QtPrev, 1, d.m, d.Gw, d.m, d.m, d.m, d.m);
        }
        VG_HIP(vg_gemm_launch(&g, st));
        VG_MARK(12);
        // First-order refinement of the start basis (used while the previous step ended in the polish, i.e. on
        // well-separated spectra): S' = (I + E + E^2/2) S with E from Gw = S G S^T, then Gw' = S' G S'^T.  Five short
        // launches that replace the one dense Jacobi sweep (127 rounds) the polish still needed: the eigensolver
        // then only scans, and the replay workgroups apply the final polish.  A start that is too far off (|E| > 1e-3)
        // gets E = 0 from the kernel and the eigensolver proceeds as usual.
        if (refine) {
            VgRefineJob rj[2];
            static const bool null_skip = getenv("VGGP_NO_NULL_SKIP") == nullptr;
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                rj[k] = VgRefineJob{d.Gw, d.U, d.TH, d.m, 0.0};
                if (null_skip) { rj[k].lam_prev = d.lam0; rj[k].null_cut = 1e-12; }
            }
            {   // E -> U, I + E -> TH   (+ rider: S = [B2;V2] Y, or [C;C1;C2] when S came out of the early projection)
                const VgGemmBatch* rr = !ride ? nullptr : (ride_stage == 0 ? &c->ride_proj : (ride_stage == 1 ? &c->ride_cc : nullptr));
                VG_HIP(vg_refine_launch(rj, 2, st, rr));
                if (rr) ++ride_stage;
            }
            VG_MARK(9);
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                const double* S0 = extrap ? d.Fp : d.QtPrev;
                vg_gemm_add(&g, d.TH, d.m, 1, S0, d.m, 1, d.E, d.m, d.m, d.m, d.m);    // (I + E) S -> E
                vg_gemm_add(&g, d.U, d.m, 1, d.U, d.m, 1, d.X, d.m, d.m, d.m, d.m);     // E E -> X
            }
            VG_HIP(vg_gemm_launch(&g, st));
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                const double* S0 = extrap ? d.Fp : d.QtPrev;
                vg_gemm_add(&g, d.X, d.m, 1, S0, d.m, 1, d.E, d.m, d.m, d.m, d.m, 1, 0, 1, 0, 0.5, 1);   // += E^2 S / 2
            }
            VG_HIP(vg_gemm_launch(&g, st));
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                vg_gemm_add(&g, d.E, d.m, 1, Gr[k], d.m, 1, d.TM, d.m, d.m, d.m, d.m, 1, 0, grn[k], ghs[k]);
            }
            VG_HIP(vg_gemm_launch(&g, st));
            vg_gemm_init(&g);
            for (int k = 0; k < 2; ++k) {
                VgDim& d = c->d[k];
                vg_gemm_add(&g, d.TM, d.m, 1, d.E, 1, d.m, d.Gw, d.m, d.m, d.m, d.m);
            }
            VG_HIP(vg_gemm_launch(&g, st));
        }
    }
    VG_MARK(12);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        ej[k] = VgEigJob{warm ? d.Gw : G0[k], d.lam0, d.Qt, warm ? ((refine || subspace) ? d.E : (extrap ? d.Fp : d.QtPrev)) : nullptr, d.gwork, d.rotlog, d.roundlog,
                         d.counters, d.m, d.max_rounds, (long)vg_eigh_log_bytes(d.m),
                         (c->desc.flags & VGGP_FLAG_BLOCK_JACOBI) ? 1 : 0};
        ej[k].Qt2 = d.QtPrev;        // the replay workgroups leave the new basis in both places (next warm start, q(v))
        ej[k].perm = d.perm;
        ej[k].cp_src = d.QtPrev; ej[k].cp_dst = d.QtPrev2;      // the basis before last, for the next extrapolation
        ej[k].err = d.status + 1;      // replay timeout flag (folded into the step status by the final kernel)
        ej[k].polish0 = (warm && refine) ? 1 : 0;
        ej[k].sparse_first = (warm && subspace) ? 1 : 0;
        ej[k].null_from = (warm && subspace) ? d.sub_r : 0;
    }
    // (counters were zeroed by the clear kernel at the start of the step)
    // riders of the main solve: whichever of the two projection launches is next (S when no earlier launch of the chain took
    // it; [C;C1;C2] when the refinement launch carried S); what is still pending afterwards runs as a launch of its own
    const VgGemmBatch* mr = !ride ? nullptr : (ride_stage == 0 ? &c->ride_proj : (ride_stage == 1 ? &c->ride_cc : nullptr));
    if (!(warm && newton > 0)) {              // (the Newton chain has left lam0, Qt, QtPrev itself)
        VG_HIP(vg_eigh_launch(ej, 2, st, mr));
        VG_MARK(13);
        if (mr) ++ride_stage;
    }
    if (ride && ride_stage == 1) { VG_HIP(vg_gemm_launch(&c->ride_cc, st)); ride_stage = 2; }
    c->ride_pending = false;
    if (from_slabs) VG_JOIN_WAIT(1);          // fused step: the projection branch (S, C slabs) ran beside the eigensolver chain

    // 8. rotate into the eigenbasis: first the right factors ...
    vg_gemm_init(&g);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        vg_gemm_add(&g, d.Mk, d.m, 1, d.Qt, 1, d.m, d.TM, d.m, d.m, d.m, d.m);      // Mk Q
        const int ih = vg_gemm_add(&g, Hr[k], d.m, 1, d.Qt, 1, d.m, d.TH, d.m, d.m, d.m, d.m);     // H0 Q
        g.p[ih].a_nslab = hrn[k]; g.p[ih].a_slab = ghs[k];
    }
    const int ic = vg_gemm_add(&g, C3, m2, 1, d2.Qt, 1, m2, c->T3, (int)m2, (int)(3 * m1), (int)m2, (int)m2);   // [C;C1;C2] Q2
    g.p[ic].a_nslab = ccn; g.p[ic].a_slab = ccs;
    // Prediction of the NEXT step's start basis, riding in the three GEMM launches of this tail (and one of the next step's
    // head): the basis moved from Q(t-1) to Q(t) by the rotation U = Q(t) Q(t-1)^T; applying it once more predicts the next
    // basis, Qpred = U Q(t) (rows = eigenvectors).  A product of three bases triples their departure from orthogonality and
    // feeds it back into the next bases -- it would grow ~2.4x per step -- so one Newton-Schulz step follows:
    // Q' = 1.5 Qpred - 0.5 (Qpred Qpred^T) Qpred.  Here: U; next launch: Ep = Qpred, Fp = 1.5 Qpred; the beta-Gram launch:
    // Wp = Qpred Qpred^T; the next step's projection launch: Fp += -0.5 Wp Ep.  The subspace start is fed the previous basis
    // itself (U = I: only the Newton-Schulz clean-up), see vg_sub_ident().  (The eigensolver's replay workgroups have
    // already left Q(t) in QtPrev and Q(t-1) in QtPrev2.)
    const bool predict = c->desc.warm_start && !(c->desc.flags & VGGP_FLAG_BLOCK_JACOBI);
    const bool pred_ident = c->sub_mode && vg_sub_ident();
    if (predict && !pred_ident)
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, d.QtPrev, d.m, 1, d.QtPrev2, 1, d.m, d.U, d.m, d.m, d.m, d.m);
        }
    VG_HIP(vg_gemm_launch(&g, st));
    VG_MARK(14);
    //    ... then the left factors
    vg_gemm_init(&g);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        vg_gemm_add(&g, d.Qt, d.m, 1, d.TM, d.m, 1, d.E, d.m, d.m, d.m, d.m);       // E = Q^T Mk Q
        vg_gemm_add(&g, d.Qt, d.m, 1, d.TH, d.m, 1, d.F, d.m, d.m, d.m, d.m);       // F = Q^T H0 Q
    }
    for (int q = 0; q < 3; ++q)
        vg_gemm_add(&g, d1.Qt, m1, 1, c->T3 + q * m1 * m2, m2, 1, c->P3 + q * m1 * m2, (int)m2, (int)m1, (int)m2, (int)m1);
    if (predict)
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const double* Uk = pred_ident ? d.Id : d.U;
            vg_gemm_add(&g, Uk, d.m, 1, d.QtPrev, d.m, 1, d.Ep, d.m, d.m, d.m, d.m);
            vg_gemm_add(&g, Uk, d.m, 1, d.QtPrev, d.m, 1, d.Fp, d.m, d.m, d.m, d.m, 1, 0, 1, 0, 1.5, 0);
        }
    VG_HIP(vg_gemm_launch(&g, st));
    VG_MARK(15);

    // 9. D-stage, the four beta Gram matrices, final reduction
    VgMspace ms;
    ms.theta = c->theta; ms.lam1 = d1.lam0; ms.lam2 = d2.lam0; ms.P3 = c->P3;
    ms.E1 = d1.E; ms.F1 = d1.F; ms.E2 = d2.E; ms.F2 = d2.F;
    ms.beta = c->beta; ms.bl2 = c->bl2; ms.bl1 = c->bl1; ms.invD = c->invD;
    ms.rowpart = c->rowpart; ms.r1 = c->r1; ms.r1l = c->r1l; ms.out = c->out;
    ms.r2 = c->r2; ms.r2l = c->r2l; ms.dotp = c->dotpart; ms.ol = c->ol;
    ms.m1 = (int)m1; ms.m2 = (int)m2; ms.n_total = (double)c->desc.n_total; ms.yy = yy_total;
    ms.ticket = c->ticket; ms.hout = c->d_hout;
    // multi-rank step: the word behind the payload travelled through the all-reduce; non-zero = some rank failed its partials
    ms.peer_fail = (payload == c->payload && (c->n_ranks > 1 || c->comm || c->cb)) ? c->payload + c->payload_len : nullptr;
    for (int k = 0; k < 2; ++k) { ms.jit[k] = c->d[k].jitter; ms.status[k] = c->d[k].status; ms.counters[k] = c->d[k].counters; }
    VG_HIP(vg_dstage_launch(&ms, st));
    VG_MARK(16);
    vg_gemm_init(&g);
    // The four dot products of the gradient, sum(E1 o beta beta^T), sum(F1 o (beta lam2) beta^T), sum(E2 o beta^T beta),
    // sum(F2 o (lam1 beta)^T beta), never materialise the m x m Gram matrices: sum_ij E_ij (beta beta^T)_ij = sum((E^T beta) o beta),
    // so each is an m1 x m2 product whose tiles are multiplied by beta and summed in the GEMM's reduction epilogue (one word per
    // tile, fixed order).  The column sums [r2; r2l] = [1; s1 lam1] (1/D) ride along as a 2-row product.
    {
        const int idot[4] = {
            vg_gemm_add(&g, d1.E, 1, m1, c->beta, m2, 1, nullptr, (int)m2, (int)m1, (int)m2, (int)m1),     // E1^T beta
            vg_gemm_add(&g, d1.F, 1, m1, c->bl2, m2, 1, nullptr, (int)m2, (int)m1, (int)m2, (int)m1),      // F1^T (beta lam2)
            vg_gemm_add(&g, c->beta, m2, 1, d2.E, m2, 1, nullptr, (int)m2, (int)m1, (int)m2, (int)m2),     // beta E2
            vg_gemm_add(&g, c->bl1, m2, 1, d2.F, m2, 1, nullptr, (int)m2, (int)m1, (int)m2, (int)m2)};     // (lam1 beta) F2
        for (int q = 0; q < 4; ++q) { g.p[idot[q]].dotw = c->beta; g.p[idot[q]].dotw_ld = (int)m2; g.p[idot[q]].dot_out = c->dotpart + 64 * q; }
        vg_gemm_add(&g, c->ol, m1, 1, c->invD, m2, 1, c->r2, (int)m2, 2, (int)m2, (int)m1);
    }
    if (predict)
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            vg_gemm_add(&g, d.Ep, d.m, 1, d.Ep, 1, d.m, d.Wp, d.m, d.m, d.m, d.m);      // W = Qpred Qpred^T
        }
    VG_HIP(vg_gemm_launch(&g, st));
    VG_MARK(17);
    VG_HIP(vg_final_launch(&ms, st));
    VG_MARK(18);

    // 10. nothing to copy: the replay workgroups already left the new basis in QtPrev (next warm start, q(v), posterior)
    //     and the last workgroup of the final reduction wrote results + diagnostics into the pinned host block
    return VGGP_OK;
}

// ---- HIP-graph cache: the launch sequence of a step is fixed for a plan, so it is captured once per
// (variant, data pointers) and replayed; hyper-parameters travel through the pinned theta buffer.
enum { VG_G_PARTIALS = 0, VG_G_PARTIALS_X, VG_G_FINISH_COLD, VG_G_FINISH_WARM, VG_G_FINISH_WARM_X, VG_G_STEP_COLD, VG_G_STEP_WARM,
       VG_G_STEP_WARM_X, VG_G_FINISH_WARM_XR, VG_G_STEP_WARM_XR, VG_G_FINISH_WARM_S, VG_G_STEP_WARM_S, VG_G_FINISH_WARM_T, VG_G_STEP_WARM_T,
       VG_G_FINISH_WARM_N, VG_G_STEP_WARM_N, VG_G_PARTIALS_E, VG_G_FINISH_WARM_TE, VG_G_STEP_COLD_T, VG_G_PARTIALS_XE, VG_G_COUNT };
static_assert(VG_G_COUNT <= 20, "vggp_ctx::gexec");
// _T: thin chain (subspace start without a complement basis, thin.hip); _N: Newton chain
// _S: subspace start (see finish_enqueue)
// _X: warm start from the extrapolated basis; _XR: ... refined to first order before the eigensolver (see finish_enqueue)

static void graphs_clear(vggp_ctx* c) {
    (void)vg_quiesce(c);
    for (int i = 0; i < VG_G_COUNT; ++i) {
        if (c->gexec[i]) { (void)hipGraphExecDestroy(c->gexec[i]); c->gexec[i] = nullptr; }
        c->gkey[i] = VgGraphKey();
    }
}

template <typename F>
static int run_graph(vggp_ctx* c, int which, const VgGraphKey& key, hipStream_t st, F enqueue, bool bypass = false) {
    if (!c->use_graph || c->prof || bypass) return enqueue();
    if (!c->gexec[which] || !(c->gkey[which] == key)) {
        if (c->gexec[which]) { (void)vg_quiesce(c); (void)hipGraphExecDestroy(c->gexec[which]); c->gexec[which] = nullptr; }
        VG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue();
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) { vg_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return VGGP_EHIP; }
        const hipError_t ei = hipGraphInstantiate(&c->gexec[which], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { c->gexec[which] = nullptr; vg_set_error("hipGraphInstantiate -> %s", hipGetErrorString(ei)); return VGGP_EHIP; }
        c->gkey[which] = key;
    }
    VG_HIP(hipGraphLaunch(c->gexec[which], st));
    return VGGP_OK;
}

// extrapolated warm start: needs the bases of the last two steps (scalar Jacobi variant; VGGP_NO_EXTRAP=1 switches it off)
static bool vg_refine(const vggp_ctx* c, bool extrap) {
    static const bool off = getenv("VGGP_NO_REFINE") != nullptr;
    return !off && extrap && c->refine_next;
}

static bool vg_extrapolate(const vggp_ctx* c) {
    static const bool off = getenv("VGGP_NO_EXTRAP") != nullptr;
    return !off && c->desc.warm_start && !(c->desc.flags & VGGP_FLAG_BLOCK_JACOBI) && c->d[0].have_prev && c->d[1].have_prev &&
           c->d[0].have_prev2 && c->d[1].have_prev2;
}

// start-basis strategy of this step; switching the subspace mode on puts the identity into U (the partials then run a plain
// Newton-Schulz clean-up of QtPrev through the extrapolation products) and drops stale _S graphs when the ranks moved
struct VgStart { bool extrap, refine, subspace, thin; int newton; };
// The thin chain needs: ranks known and small, the leading rows of QtPrev valid, the step's payload owned by the context (its
// read-outs re-run the finish half cold on the resident G, H, C), and no caller that wanted the full m-space state of a warm step.
static bool vg_thin_ok(const vggp_ctx* c, bool own_payload) {
    static const bool off = getenv("VGGP_NO_THIN") != nullptr;
    if (off || c->thin_off || c->thin_block > 0 || !own_payload || !c->desc.warm_start || (c->desc.flags & VGGP_FLAG_BLOCK_JACOBI) || !c->sub_next) return false;
    for (int k = 0; k < 2; ++k) {
        const VgDim& d = c->d[k];
        if (!d.have_prev || !d.tMV || d.sub_r < 1 || d.sub_r > VG_THIN_MAXR || d.sub_r > d.thin_rows) return false;
    }
    return true;
}
// warm start possible?  After a thin step QtPrev holds range rows only: a step that cannot run thin then starts cold.
static bool vg_warm(vggp_ctx* c, bool own_payload) {
    bool warm = c->desc.warm_start && c->d[0].have_prev && c->d[1].have_prev;
    if (warm && (c->d[0].thin_rows < c->d[0].m || c->d[1].thin_rows < c->d[1].m) && !vg_thin_ok(c, own_payload)) {
        c->warm_run = 0;
        for (int k = 0; k < 2; ++k) c->d[k].have_prev = c->d[k].have_prev2 = false;
        warm = false;
    }
    return warm;
}
static int vg_start_prepare(vggp_ctx* c, bool warm, hipStream_t st, VgStart* out, bool own_payload) {
    out->thin = warm && vg_thin_ok(c, own_payload);
    out->extrap = !out->thin && vg_extrapolate(c);
    // (the full subspace chain -- complement basis, sparse-first main solve -- lives in the LDS eigensolver body: m <= 128)
    const bool lds_sized = c->d[0].m <= 128 && c->d[1].m <= 128;
    out->subspace = out->thin || (lds_sized && warm && out->extrap && c->sub_next && c->d[0].sub_r > 0 && c->d[1].sub_r > 0);
    // Newton chain: full-rank warm starts beyond the LDS eigensolver (m > 128) or where the last warm step still needed Jacobi rounds
    {
        static const bool off = getenv("VGGP_NO_NEWTON_CHAIN") != nullptr;
        static const char* ite = getenv("VGGP_NEWTON_ITERS");
        const bool big = c->d[0].m > 128 || c->d[1].m > 128;
        // (below m_d = 96 a sweep of the LDS solver is cheaper than an iteration of this chain -- seven launches whatever the size)
        const bool roomy = c->d[0].m >= 96 && c->d[1].m >= 96;
        const bool want = warm && !out->subspace && out->extrap && !off && c->newton_block == 0 && (big || (c->newton_next && roomy)) &&
                          !(c->desc.flags & VGGP_FLAG_BLOCK_JACOBI) && c->d[0].m >= 8 && c->d[1].m >= 8;
        out->newton = want ? (ite ? atoi(ite) : c->newton_iters) : 0;
        if (c->newton_block > 0) --c->newton_block;
    }
    out->refine = warm && !out->subspace && !out->newton && vg_refine(c, out->extrap);
    c->sub_mode = out->subspace;
    if (out->subspace && (c->sub_r_cap[0] != c->d[0].sub_r || c->sub_r_cap[1] != c->d[1].sub_r)) {
        for (int v : {(int)VG_G_FINISH_WARM_S, (int)VG_G_STEP_WARM_S, (int)VG_G_FINISH_WARM_T, (int)VG_G_STEP_WARM_T, (int)VG_G_FINISH_WARM_TE})
            if (c->gexec[v]) { (void)vg_quiesce(c); (void)hipGraphExecDestroy(c->gexec[v]); c->gexec[v] = nullptr; c->gkey[v] = VgGraphKey(); }
        c->sub_r_cap[0] = c->d[0].sub_r; c->sub_r_cap[1] = c->d[1].sub_r;
    }
    if (out->newton && out->newton != c->newton_cap) {          // the _N graphs hold a fixed number of iterations
        for (int v : {(int)VG_G_FINISH_WARM_N, (int)VG_G_STEP_WARM_N})
            if (c->gexec[v]) { (void)vg_quiesce(c); (void)hipGraphExecDestroy(c->gexec[v]); c->gexec[v] = nullptr; c->gkey[v] = VgGraphKey(); }
        c->newton_cap = out->newton;
    }
    c->cur_thin = out->thin;
    c->cur_extrap = out->extrap;
    c->cur_newton = out->newton;
    c->cur_r[0] = c->d[0].sub_r; c->cur_r[1] = c->d[1].sub_r;
    return VGGP_OK;
}

static int set_theta(vggp_ctx* c, const double theta[5]) {
    for (int i = 0; i < 5; ++i) {
        VG_REQUIRE(theta[i] > 0.0 && std::isfinite(theta[i]), "theta[%d]=%g must be positive and finite", i, theta[i]);
        c->h_theta[i] = theta[i];
    }
    c->h_theta[5] = (double)(++c->seq);        // travels through the step and comes back in the pinned result block
    return VGGP_OK;
}

// host side of the end of a step: the only synchronisation, then unpack the pinned block
// Completion of a single-rank step WITHOUT a HIP call: the last kernel of the step writes the 128-byte result block into pinned
// host memory as one burst whose last word is the step's sequence number, so the host polls that word.  hipStreamSynchronize
// returns only after the end-of-graph bookkeeping of the runtime (measured: ~8 us later); the stream itself keeps the order of
// whatever is enqueued next.  Anything that must not overlap with the tail of the graph (destroying a graph exec, re-planning)
// calls vg_quiesce first.  Falls back to the real synchronisation after 20 ms (a fault would otherwise spin for ever).
static int vg_quiesce(vggp_ctx* c) {
    if (c->poll_stream_valid) { c->poll_stream_valid = false; VG_HIP(hipStreamSynchronize(c->poll_stream)); }
    return VGGP_OK;
}
static int vg_wait_step(vggp_ctx* c, hipStream_t st) {
    static const bool nopoll = getenv("VGGP_NO_POLL") != nullptr;
    const volatile double* seq = &c->h_out->seq;
    const double want = c->h_theta[5];
    if (nopoll || c->prof || (c->n_ranks > 1 && !c->comm && !c->cb)) { c->poll_stream_valid = false; return vg_comm_wait(c, st); }
    if (c->comm && !c->cb) {
        // RCCL transport: the same word, polled together with the communicator's health (a dead peer is an error, not a hang)
        const int wrc = vg_comm_wait(c, st, seq, want);
        c->poll_stream = st; c->poll_stream_valid = (wrc == VGGP_OK);
        return wrc;
    }
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long spin = 0;; ++spin) {
        if (*seq == want) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            c->poll_stream = st; c->poll_stream_valid = true;
            return VGGP_OK;
        }
        __builtin_ia32_pause();
        if ((spin & 4095) == 4095) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > 0.02) break;
        }
    }
    c->poll_stream_valid = false;
    VG_HIP(hipStreamSynchronize(st));
    return VGGP_OK;
}

static int finish_collect(vggp_ctx* c, double* elbo_out, double grad_out[5], vggp_info* info, hipStream_t st) {
    {
        const int wrc = vg_wait_step(c, st);          // (single rank: polls the pinned result block; multi-rank: vg_comm_wait)
        if (wrc) {
            c->warm_run = 0;
            for (int k = 0; k < 2; ++k) c->d[k].have_prev = c->d[k].have_prev2 = false;
            c->have_step = false;
            return wrc;
        }
    }
    c->pred_consumed = false;              // the tail of this step left a fresh prediction in Ep / Fp / Wp
    c->acc_valid = false;
    if (c->prof && c->nev > 1) {
        for (int i = 1; i < c->nev; ++i) {
            const int id = c->ev_stage[i];
            float ms = 0.f;
            if (id >= 0 && id < VGGP_NSTAGE && hipEventElapsedTime(&ms, c->ev[i - 1], c->ev[i]) == hipSuccess) c->prof_ms[id] += ms;
        }
        c->prof_steps++;
    }
    c->nev = 0;
    if (c->h_out->seq != c->h_theta[5]) {       // the block was not written by THIS step (a launch was lost or reordered)
        vg_set_error("step %.0f: the result block carries sequence number %.0f (stale results)", c->h_theta[5], c->h_out->seq);
        c->warm_run = 0;
        for (int k = 0; k < 2; ++k) c->d[k].have_prev = c->d[k].have_prev2 = false;
        c->have_step = false;
        return VGGP_ESTATE;
    }
    if (c->h_out->out[6] != 0.0) {          // the peer-failure word of the all-reduced payload: the sums miss a rank's contribution
        vg_set_error("rank %d / %d: %.0f rank(s) of the job failed their half of this step; the result is discarded", c->rank,
                     c->n_ranks, c->h_out->out[6]);
        c->warm_run = 0;
        for (int k = 0; k < 2; ++k) c->d[k].have_prev = c->d[k].have_prev2 = false;
        c->have_step = false;
        return VGGP_ERCCL;
    }
    *elbo_out = c->h_out->out[0];
    for (int i = 0; i < 5; ++i) grad_out[i] = c->h_out->out[1 + i];
    int status = 0;
    for (int k = 0; k < 2; ++k) {
        if (c->h_out->status[k]) status = c->h_out->status[k];
        else if (c->h_out->counters[k][2]) status = c->h_out->counters[k][2];
    }
    if (info) {
        info->jitter1 = c->h_out->jitter[0]; info->jitter2 = c->h_out->jitter[1];
        info->sweeps1 = c->h_out->counters[0][1] & 0xff; info->sweeps2 = c->h_out->counters[1][1] & 0xff;
        info->rounds1 = c->h_out->counters[0][0]; info->rounds2 = c->h_out->counters[1][0];
        info->status = status;
        info->polished = ((c->h_out->counters[0][3] >> 28) & 1) | (((c->h_out->counters[1][3] >> 28) & 1) << 1);
    }
    // numerical ranks -> subspace start for the next step (rank-deficient Gram matrices only; hysteresis keeps the graph stable)
    {
        static const bool off = getenv("VGGP_NO_SUBSPACE") != nullptr;
        bool ok = !status && !off && c->desc.warm_start && !(c->desc.flags & VGGP_FLAG_BLOCK_JACOBI);
        for (int k = 0; k < 2; ++k) {
            VgDim& d = c->d[k];
            const int rank = (c->h_out->counters[k][1] >> 8) & 0x1ff;
            int want = 0;
            static const bool relax = getenv("VGGP_NO_SUB_RELAX") == nullptr;
            if (d.Zs && rank >= 1 && 3 * rank <= d.m) {
                // numerical rank (eigenvalues above 1e-14 of the largest) + 3 rows, rounded up to an even count (odd counts run a padded,
                // slower path: 21 rows 0.207 ms, 22 rows 0.190): at 1024^2, m_d = 128 that is 20 rows (rank 17), 22 when the rank reaches 18; the earlier rule (rank + 4 rounded to 8 = 24 rows) cost 11 us per step in the row QR, the
                // Ritz solve and the r-row products.  An RBF spectrum drops by ~6x per index and a fit-loop step moves it by a few
                // per cent, so the margin is consumed one row per many steps -- and the tail kernel's miss check catches the rest.
                // VGGP_SUB_MARGIN8=1: the earlier rule.
                static const bool m8 = getenv("VGGP_SUB_MARGIN8") != nullptr;
                want = m8 ? ((rank + 4 + 7) / 8) * 8 : ((rank + 3 + 1) / 2) * 2;
                if (want < 16) want = 16;
                if (want > 64 || 2 * want > d.m) want = 0;
            } else if (relax && d.Zs && rank >= 1 && d.m <= 64 && rank + 2 <= d.m - 8) {
                // small inducing counts (m_d = 32: numerical rank ~22): not "rank-deficient" by the rule above, but the near-null
                // cluster is what costs the warm Jacobi its 3-5 sweeps -- deflating the range part still pays as long as the Ritz
                // problem fits the Newton start (<= 48) and some rows are left over (195 -> 171 us at m_d = 32, 253 -> 203 at 48)
                want = ((rank + 2 + 3) / 4) * 4;
                if (want > d.m - 8) want = d.m - 8;
                if (want < 8 || want > 48) want = 0;
            }
            if (d.m > 128 && want > VG_THIN_MAXR) want = 0;      // beyond one LDS-resident matrix only the thin chain exists
            if (want == 0) d.sub_r = 0;
            else if (want > d.sub_r || want < d.sub_r - 8) d.sub_r = want;
            ok = ok && d.sub_r > 0;
        }
        c->sub_next = ok;
    }
    // refine the next start only where it can replace the last sweep: both dimensions polished after at most one sweep
    c->refine_next = !status && ((c->h_out->counters[0][3] >> 28) & 1) && ((c->h_out->counters[1][3] >> 28) & 1) &&
                     (c->h_out->counters[0][1] & 0xff) <= 1 && (c->h_out->counters[1][1] & 0xff) <= 1;
    if (status == VG_ESUBMISS && c->cur_newton) {
        // the Newton chain missed: it has not touched the stored bases -- repeat on the regular chain from the same warm start
        static const bool dbg = getenv("VGGP_NEWTON_DBG") != nullptr;
        if (dbg)
            fprintf(stderr, "[vggp] step %ld: Newton chain (%d iterations) missed: dim1 flags %d, 10 log10(offdiag / thr) = %d; dim2 flags %d, %d\n",
                    c->seq, c->cur_newton, c->h_out->counters[0][3] & 0xff, (c->h_out->counters[0][3] >> 8) - 100,
                    c->h_out->counters[1][3] & 0xff, (c->h_out->counters[1][3] >> 8) - 100);
        // (beyond the LDS eigensolver the regular chain is the global-memory Jacobi -- 18 ms per step at m = 256: there the chain is
        //  tried again as soon as BOTH stored bases come from the regular chain -- the eigensolver sorts, the Newton chain keeps its
        //  start order, and a pair of bases from different chains extrapolates badly; otherwise it rests for 16 steps)
        c->newton_block = (c->d[0].m > 128 || c->d[1].m > 128) ? 2 : 16; c->newton_next = false;
        if (c->newton_iters < 5) ++c->newton_iters;
        c->newton_ok_run = 0;
        c->have_step = false;
        return VG_ESUBMISS;
    }
    if (status) {          // the bases written by this step are not trustworthy: the next step starts cold
        c->warm_run = 0;
        for (int k = 0; k < 2; ++k) c->d[k].have_prev = c->d[k].have_prev2 = false;
        c->have_step = false;
    }
    if (status == VG_ESUBMISS) {            // (the caller repeats the step; the warm start was reset above)
        c->sub_next = false;
        if (c->cur_thin) c->thin_block = 32;      // the directions the thin chain leaves out mattered: full chains for a while
        return VG_ESUBMISS;
    }
    if (status == VGGP_ENOTPD) { vg_set_error("a Kuu factor is not positive definite after jitter 1e-6"); return VGGP_ENOTPD; }
    if (status == VGGP_ENOCONV) { vg_set_error("Jacobi eigensolver did not converge"); return VGGP_ENOCONV; }
    for (int k = 0; k < 2; ++k) {
        c->d[k].have_prev2 = c->d[k].have_prev;                // QtPrev2 <- previous basis (copied by the replay workgroups)
        c->d[k].have_prev = true;                              // QtPrev holds this step's basis
        c->d[k].thin_rows = c->cur_thin ? c->cur_r[k] : c->d[k].m;      // ... all of it, or the range rows of a thin step
    }
    c->last_thin = c->cur_thin;
    if (c->thin_block > 0) --c->thin_block;
    // (the Newton chain keeps the eigenpairs in the order of its start basis, the eigensolver sorts: after a crossing the two stored
    //  bases of a chain switch may not correspond row by row -- the extrapolated start is then poor, the chain that receives it
    //  notices: the regular one sweeps, the Newton chain misses and the step is repeated)
    c->last_newton = c->cur_newton > 0;
    if (c->cur_newton > 0 && ++c->newton_ok_run >= 64) {          // a long run without a miss: try one iteration less again
        c->newton_ok_run = 0;
        if (c->newton_iters > 3) --c->newton_iters;
    }
    // next step: stay on the Newton chain while it works; move to it when a warm step of the regular chain still needed rounds
    {
        const bool polished = ((c->h_out->counters[0][3] >> 28) & 1) && ((c->h_out->counters[1][3] >> 28) & 1);
        c->newton_next = c->cur_newton > 0 ||
                         (c->last_warm && c->cur_extrap && !c->cur_thin && !polished && (c->h_out->counters[0][0] + c->h_out->counters[1][0]) > 0) ||
                         // ... or ended in the polish only after more than a sweep and a half of rotations (B1 hats on a padded mesh: two
                         // dense sweeps per step, then the polish; an occasional single sweep of a Matern-3/2 step does not count)
                         // (m_d >= 96 only: an iteration of the chain is seven launches whatever the size, a sweep of a small matrix is cheap --
                         //  127 Fourier features gain, 63 lose)
                         (c->last_warm && c->cur_extrap && !c->cur_thin && c->d[0].m >= 96 && c->d[1].m >= 96 &&
                          (2 * c->h_out->counters[0][0] >= 3 * c->d[0].m || 2 * c->h_out->counters[1][0] >= 3 * c->d[1].m));
    }
    if (++c->warm_run >= 512) {                                // periodic cold restart: bounds the drift of orthogonality
        c->warm_run = 0;
        for (int k = 0; k < 2; ++k) c->d[k].have_prev = c->d[k].have_prev2 = false;
    }
    c->have_step = true;
    return VGGP_OK;
}

extern "C" int vggp_elbo_partials(vggp_ctx* c, const double* Y, const double theta[5], double* payload, void* stream) {
    if (!c || !c->planned) { vg_set_error("vggp_elbo_partials: context not planned"); return VGGP_ESTATE; }
    VG_REQUIRE(Y && theta && payload, "vggp_elbo_partials: null argument");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    int rc = set_theta(c, theta);
    if (rc) return rc;
    c->nev = 0;
    const VgGraphKey key{Y, payload, 0.0};
    VgStart sp;
    if ((rc = vg_start_prepare(c, vg_warm(c, false), st, &sp, false))) return rc;
    const bool extrap = sp.extrap;
    // the Newton-Schulz step of the predicted basis accumulates into Fp: exactly once per prediction (a repeated partials
    // call without a finish in between runs un-captured without it)
    const bool apply_ns = extrap && !c->pred_consumed;
    rc = run_graph(c, extrap ? VG_G_PARTIALS_X : VG_G_PARTIALS, key, st,
                   [&] { return vg_partials_enqueue(c, Y, payload, st, true, extrap, false, apply_ns); }, extrap && !apply_ns);
    if (rc) return rc;
    if (extrap) c->pred_consumed = true;
    c->have_partials = true;
    return VGGP_OK;
}

static int elbo_finish_once(vggp_ctx* c, const double* payload, double yy_total, const double theta[5],
                            double* elbo_out, double grad_out[5], vggp_info* info, void* stream);
extern "C" int vggp_elbo_finish(vggp_ctx* c, const double* payload, double yy_total, const double theta[5],
                                double* elbo_out, double grad_out[5], vggp_info* info, void* stream) {
    int rc = elbo_finish_once(c, payload, yy_total, theta, elbo_out, grad_out, info, stream);
    if (rc == VG_ESUBMISS) rc = elbo_finish_once(c, payload, yy_total, theta, elbo_out, grad_out, info, stream);   // cold, same payload
    if (rc == VG_ESUBMISS) { vg_set_error("the eigensolver chain failed twice on the same step"); rc = VGGP_ENOCONV; }
    return rc;
}

static int elbo_finish_once(vggp_ctx* c, const double* payload, double yy_total, const double theta[5],
                            double* elbo_out, double grad_out[5], vggp_info* info, void* stream) {
    if (!c || !c->planned || !c->have_partials) { vg_set_error("vggp_elbo_finish: call vggp_elbo_partials first"); return VGGP_ESTATE; }
    VG_REQUIRE(payload && theta && elbo_out && grad_out, "vggp_elbo_finish: null argument");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    int rc = set_theta(c, theta);
    if (rc) return rc;
    const bool warm = vg_warm(c, false);
    const VgGraphKey key{nullptr, payload, yy_total};
    VgStart sp;                                   // same state as at the matching vggp_elbo_partials call
    if ((rc = vg_start_prepare(c, warm, st, &sp, false))) return rc;
    const bool extrap = sp.extrap, refine = sp.refine, subspace = sp.subspace;
    const int newton = sp.newton;
    rc = run_graph(c, warm ? (newton ? VG_G_FINISH_WARM_N : subspace ? VG_G_FINISH_WARM_S : extrap ? (refine ? VG_G_FINISH_WARM_XR : VG_G_FINISH_WARM_X) : VG_G_FINISH_WARM)
                            : VG_G_FINISH_COLD, key, st,
                   [&] { return finish_enqueue(c, payload, yy_total, warm, st, true, false, extrap, refine, subspace, false, newton); });
    if (rc) return rc;
    c->last_warm = warm; c->last_slabs = false; c->last_payload = payload; c->last_yy = yy_total;
    return finish_collect(c, elbo_out, grad_out, info, st);
}

// ---- cold start of the thin chain (RBF factors): a range finder instead of the full Jacobi solve.
// G = B B^T of an RBF factor has numerical rank ~17 of 128, and "Z = V G" maps ANY r-row start V onto range(G) exactly (G annihilates
// the null components) -- but the row-by-row orthonormalisation of Z keeps the small range directions only when the rows of V are
// GRADED (row i mostly within the i leading eigen-directions); a random sketch is not.  So:
//   pass 1:  r0 = 32 steps of diagonally pivoted Cholesky of G (thin.hip, vg_pivchol_kernel): graded columns spanning range(G),
//            orthonormalised in pivot order into the leading rows of QtPrev;
//   pass 2:  the warm thin chain itself from those rows;
// and the tail kernel's miss check (tr G - sum theta) decides as in every thin step: a spectrum that does not fit 32 rows is refused,
// the step is repeated with the full solve and the thin chain stays off for 32 steps.  VGGP_NO_COLD_THIN=1: always the full solve.
#define VG_COLD_THIN_R 32
static bool vg_cold_thin_ok(const vggp_ctx* c) {
    static const bool off = getenv("VGGP_NO_COLD_THIN") != nullptr || getenv("VGGP_NO_THIN") != nullptr || getenv("VGGP_NO_SUBSPACE") != nullptr;
    if (off || c->thin_off || c->thin_block > 0 || c->prof || (c->desc.flags & (VGGP_FLAG_BLOCK_JACOBI | VGGP_FLAG_SCATTERED))) return false;
    if (c->n_ranks > 1 || c->comm || c->cb) return false;
    for (int k = 0; k < 2; ++k) {
        const VgDim& d = c->d[k];
        if (!d.Omega || !d.tMV || d.kind != VGGP_KIND_RBF || d.m < 3 * VG_COLD_THIN_R || d.m > 256) return false;
    }
    return true;
}
static int vg_cold_thin_prepass(vggp_ctx* c, const double* payload, hipStream_t st) {
    const double* Gk[2] = {c->d[0].GH, payload};                  // reduced Gram matrices (the partials half ran with reduce)
    VgPivCholJob pj[2];
    for (int k = 0; k < 2; ++k) { VgDim& d = c->d[k]; pj[k] = VgPivCholJob{Gk[k], d.Omega, d.TM, d.m, VG_COLD_THIN_R}; }
    VG_HIP(vg_pivchol_launch(pj, 2, st));
    VgRowQrJob qj[2];
    for (int k = 0; k < 2; ++k) { VgDim& d = c->d[k]; qj[k] = VgRowQrJob{d.TM, d.QtPrev, VG_COLD_THIN_R, d.m, nullptr, nullptr, 0}; }
    VG_HIP(vg_rowqr_launch(qj, 2, st, nullptr));
    return VGGP_OK;
}

static int elbo_step_once(vggp_ctx* c, const double* Y, double yy_total, const double theta[5], double* elbo_out,
                          double grad_out[5], vggp_info* info, void* stream);
// A step whose subspace start turns out to have missed part of the range (VG_ESUBMISS: the hyper-parameters jumped) is repeated
// once, cold -- the failed attempt has already reset the warm start.  A jump the host can see beforehand (> 5 % in a lengthscale
// since the last step; the Gram matrices depend on nothing else) skips the attempt.
extern "C" int vggp_elbo_step(vggp_ctx* c, const double* Y, double yy_total, const double theta[5], double* elbo_out,
                              double grad_out[5], vggp_info* info, void* stream) {
    if (c && c->planned && theta) {
        bool jump = false;
        for (int k = 0; k < 2; ++k) jump = jump || (c->last_ell[k] > 0.0 && std::fabs(theta[k] / c->last_ell[k] - 1.0) > 0.05);
        if (jump && c->sub_next) { c->warm_run = 0; for (int k = 0; k < 2; ++k) c->d[k].have_prev = c->d[k].have_prev2 = false; }
        c->last_ell[0] = theta[0]; c->last_ell[1] = theta[1];
    }
    int rc = elbo_step_once(c, Y, yy_total, theta, elbo_out, grad_out, info, stream);
    if (rc == VG_ESUBMISS) rc = elbo_step_once(c, Y, yy_total, theta, elbo_out, grad_out, info, stream);
    if (rc == VG_ESUBMISS) { vg_set_error("the eigensolver chain failed twice on the same step"); rc = VGGP_ENOCONV; }
    return rc;
}

static int elbo_step_once(vggp_ctx* c, const double* Y, double yy_total, const double theta[5], double* elbo_out,
                          double grad_out[5], vggp_info* info, void* stream) {
    if (!c || !c->planned) { vg_set_error("vggp_elbo_step: context not planned"); return VGGP_ESTATE; }
    if (c->desc.flags & VGGP_FLAG_SCATTERED) { vg_set_error("vggp_elbo_step: the context was planned for scattered points (use vggp_elbo_step_scattered)"); return VGGP_ESTATE; }
    VG_REQUIRE(Y && theta && elbo_out && grad_out, "vggp_elbo_step: null argument");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    int rc = set_theta(c, theta);
    if (rc) return rc;
    c->nev = 0;
    const bool warm = vg_warm(c, true);
    VgStart sp;
    if ((rc = vg_start_prepare(c, warm, st, &sp, true))) return rc;
    const bool extrap = sp.extrap, refine = sp.refine, subspace = sp.subspace, thin = sp.thin;
    const int newton = sp.newton;
    const bool apply_ns = extrap && !c->pred_consumed;
    if (c->n_ranks > 1 || c->comm || c->cb) {
        // row-sharded job: partials graph -> the context's all-reduce of the packed payload (RCCL: enqueued on this stream)
        // -> finish graph; the one host synchronisation is in finish_collect.  Every rank finishes redundantly, so all ranks
        // hold the identical value and gradient without a broadcast.
        const VgGraphKey kp{Y, c->payload, 0.0}, kf{nullptr, c->payload, yy_total};
        // thin chain: the pass over the rank's slab of Y rides beside the Cholesky (vg_partials_enqueue, early), the finish half places
        // the small products of the reduced [C;C1;C2] as the fused single-rank step does
        static const bool no_early_mr = getenv("VGGP_NO_EARLY") != nullptr;
        const bool early_ok = !no_early_mr && !c->prof && c->desc.m1 <= 128 && c->desc.m2 <= 128 && vg_side(c, st) == st &&
                              getenv("VGGP_CHOL_LEGACY") == nullptr;
        const bool thin_early = thin && early_ok && !extrap;
        // (the regular warm chains take the early association too, as in the fused single-rank step: the two must agree to rounding)
        static const bool no_early_reg_mr = getenv("VGGP_NO_EARLY_REG") != nullptr;
        const bool early_mr = thin_early || (warm && !thin && !subspace && early_ok && !no_early_reg_mr);
        rc = run_graph(c, early_mr ? (extrap ? VG_G_PARTIALS_XE : VG_G_PARTIALS_E) : extrap ? VG_G_PARTIALS_X : VG_G_PARTIALS, kp, st,
                       [&] { return vg_partials_enqueue(c, Y, c->payload, st, true, extrap, false, apply_ns, early_mr ? 1 : 0); }, extrap && !apply_ns);
        {
            // fault injection for the failure-path test (tests/test_gpu_dist.py): VGGP_FAULT_PARTIALS_AT=<step number>
            static const long fault_at = [] { const char* e = getenv("VGGP_FAULT_PARTIALS_AT"); return e ? atol(e) : -1L; }();
            if (!rc && fault_at >= 0 && c->seq == fault_at) { vg_set_error("injected fault in the partials of step %ld", c->seq); rc = VGGP_EHIP; }
        }
        if (rc) {
            // This rank cannot contribute, but its peers are about to enter the all-reduce: join it with a zero payload and the
            // failure word set, so that every rank leaves the collective and returns an error instead of hanging in it.
            char msg[512];
            snprintf(msg, sizeof(msg), "%s", vggp_last_error());
            static const double one = 1.0;
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { hipGraph_t gdrop = nullptr; (void)hipStreamEndCapture(st, &gdrop); if (gdrop) (void)hipGraphDestroy(gdrop); }
            (void)hipMemsetAsync(c->payload, 0, sizeof(double) * c->payload_len, st);
            (void)hipMemcpyAsync(c->payload + c->payload_len, &one, sizeof(double), hipMemcpyHostToDevice, st);
            if (vg_allreduce(c, c->payload, c->payload_len + 1, st) == VGGP_OK) (void)vg_comm_wait(c, st);
            else vg_comm_abort(c);
            c->warm_run = 0;
            for (int k = 0; k < 2; ++k) c->d[k].have_prev = c->d[k].have_prev2 = false;
            c->have_step = false;
            vg_set_error("%s", msg);
            return rc;
        }
        if (extrap) c->pred_consumed = true;
        c->have_partials = true;
        if ((rc = vg_allreduce(c, c->payload, c->payload_len + 1, st))) return rc;
        rc = run_graph(c, warm ? (thin ? (thin_early ? VG_G_FINISH_WARM_TE : VG_G_FINISH_WARM_T) : newton ? VG_G_FINISH_WARM_N : subspace ? VG_G_FINISH_WARM_S : extrap ? (refine ? VG_G_FINISH_WARM_XR : VG_G_FINISH_WARM_X) : VG_G_FINISH_WARM)
                                : VG_G_FINISH_COLD, kf, st,
                       [&] { return finish_enqueue(c, c->payload, yy_total, warm, st, false, false, extrap, refine, subspace, thin, newton, thin_early); });
        if (rc) return rc;
        c->last_warm = warm; c->last_slabs = false; c->last_payload = c->payload; c->last_yy = yy_total;
        return finish_collect(c, elbo_out, grad_out, info, st);
    }
    const VgGraphKey key{Y, c->payload, yy_total};
    // thin chain with the early projection (vg_partials_enqueue): where the step defers its projection launches to the eigensolver
    // chain anyway, and both factors are single 128-blocks
    static const bool no_early = getenv("VGGP_NO_EARLY") != nullptr;
    static const bool no_ride = getenv("VGGP_NO_RIDE") != nullptr;
    if (!warm && vg_cold_thin_ok(c)) {
        // cold step of a rank-deficient plan: range finder + thin chain (see vg_cold_thin_prepass) instead of the full Jacobi solve
        const bool early = !no_early && c->desc.m1 <= 128 && c->desc.m2 <= 128 && vg_side(c, st) == st && getenv("VGGP_CHOL_LEGACY") == nullptr;
        const int keep[2] = {c->d[0].sub_r, c->d[1].sub_r};
        c->d[0].sub_r = c->d[1].sub_r = VG_COLD_THIN_R;          // (sizes of the captured thin chain; the host's rank bookkeeping keeps its own)
        rc = run_graph(c, VG_G_STEP_COLD_T, key, st, [&] {
            int r1 = vg_partials_enqueue(c, Y, c->payload, st, /*reduce=*/true, false, /*fused=*/true, false, early);
            if (!r1) r1 = vg_cold_thin_prepass(c, c->payload, st);
            return r1 ? r1 : finish_enqueue(c, c->payload, yy_total, /*warm=*/true, st, false, /*from_slabs=*/false, false, false, /*subspace=*/true,
                                            /*thin=*/true, 0, early);
        });
        c->d[0].sub_r = keep[0]; c->d[1].sub_r = keep[1];
        if (rc) return rc;
        c->cur_thin = true; c->cur_extrap = false; c->cur_newton = 0;
        c->cur_r[0] = c->cur_r[1] = VG_COLD_THIN_R;
        c->have_partials = true;
        c->last_warm = true;             // (no full m-space state behind this step: the read-outs that need one rebuild it)
        c->last_slabs = false; c->last_payload = c->payload; c->last_yy = yy_total;
        return finish_collect(c, elbo_out, grad_out, info, st);
    }
    const bool thin_early = thin && !no_early && !no_ride && c->desc.m1 <= 128 && c->desc.m2 <= 128 && vg_side(c, st) == st &&
                            getenv("VGGP_CHOL_LEGACY") == nullptr;
    // the regular warm chains (refinement + polish, Newton chain) take the early pass over Y as well: [C;C1;C2] then rides where S
    // used to (beside the refinement kernel) and the main solve carries nothing.  VGGP_NO_EARLY_REG=1: riders as in round 2.
    static const bool no_early_reg = getenv("VGGP_NO_EARLY_REG") != nullptr;
    const bool early_reg = warm && !thin && !subspace && !no_early && !no_early_reg && !no_ride && c->desc.m1 <= 128 && c->desc.m2 <= 128 &&
                           vg_side(c, st) == st && getenv("VGGP_CHOL_LEGACY") == nullptr && (vg_ride(c) || c->prof);
    const int early_mode = thin_early ? 1 : (early_reg ? 2 : 0);
    rc = run_graph(c, warm ? (thin ? VG_G_STEP_WARM_T : newton ? VG_G_STEP_WARM_N : subspace ? VG_G_STEP_WARM_S : extrap ? (refine ? VG_G_STEP_WARM_XR : VG_G_STEP_WARM_X) : VG_G_STEP_WARM)
                            : VG_G_STEP_COLD, key, st, [&] {
        const int r1 = vg_partials_enqueue(c, Y, c->payload, st, /*reduce=*/!warm, extrap, /*fused=*/true, apply_ns, early_mode);
        return r1 ? r1 : finish_enqueue(c, c->payload, yy_total, warm, st, false, /*from_slabs=*/warm, extrap, refine, subspace, thin, newton,
                                        thin_early);
    }, extrap && !apply_ns);
    if (rc) return rc;
    if (extrap) c->pred_consumed = true;
    c->have_partials = true;
    c->last_warm = warm; c->last_slabs = warm; c->last_payload = c->payload; c->last_yy = yy_total;
    return finish_collect(c, elbo_out, grad_out, info, st);
}

// ---------------------------------------------------------------------------------
// q(v): mean = R1 (beta/v) R2^T, diag cov = (R1 o R1)(1/D)(R2 o R2)^T, R_d = sqrt(s_d) L0_d Q_d
// Read-outs after a WARM-started step.  The warm paths (subspace start, first-order refinement) diagonalise G to the
// eigensolver's ABSOLUTE threshold, which is all the ELBO and its gradient need; but the numerically-null block of an RBF Gram
// matrix is then only block-diagonalised -- its rows are the previous step's rows projected off the new range -- so the tiny
// eigenvalues lam_i (1e-13 lam_max) are mixtures.  The posterior VARIANCE sees that through D = 1 + lam1 lam2 / sigma^2 with
// lam1 ~ 1e3: 5e-4 relative at 1024 x 1024 RBF (measured; the cold solve's cyclic sweeps resolve that block to high RELATIVE
// accuracy and agree with the oracle to 6e-9).  So the first read-out after a warm step re-runs the finish half COLD on the
// step's own G, H, C (still resident): one cold eigensolve (~1 ms) per prediction request, nothing in the fit loop.
// VGGP_FAST_READOUT=1 skips it (read-outs then carry the warm basis' accuracy).
static int vg_accurate_state(vggp_ctx* c, hipStream_t st) {
    static const bool skip = getenv("VGGP_FAST_READOUT") != nullptr;
    if ((skip && !c->last_thin) || !c->last_warm || c->acc_valid || !c->have_step) return VGGP_OK;     // (a thin step leaves no m-space state at all)
    if (c->last_payload != c->payload) return VGGP_OK;       // the caller owned the payload buffer (partials / finish API): not retained
    const long m1 = c->desc.m1, m2 = c->desc.m2;
    if (c->last_slabs) {                                       // fused warm step: G, H, C are still split-K slabs
        VgRedBatch r;
        vg_red_init(&r);
        vg_red_add(&r, c->d[0].GHslab, c->d[0].GH, 2L * m1 * m1, 2L * m1 * m1, c->gh_slabs[0]);
        vg_red_add(&r, c->d[1].GHslab, c->payload, 2L * m2 * m2, 2L * m2 * m2, c->gh_slabs[1]);
        vg_red_add(&r, c->CCslab, c->payload + 2 * m2 * m2, 3L * m1 * m2, 3L * m1 * m2, c->cc_slabs);
        VG_HIP(vg_red_launch(&r, st));
    }
    VgClearArgs clr;                                           // the step's flag words (normally zeroed by the factor kernel)
    clr.n = 0;
    for (int k = 0; k < 2; ++k) { clr.ptr[clr.n] = c->d[k].counters; clr.nwords[clr.n++] = 8; clr.ptr[clr.n] = c->d[k].status + 1; clr.nwords[clr.n++] = 1; }
    VG_HIP(vg_clear_launch(&clr, st));
    const bool prof = c->prof;
    c->prof = false;
    const int rc = finish_enqueue(c, c->payload, c->last_yy, /*warm=*/false, st, /*copy_theta=*/false);
    c->prof = prof;
    if (rc) return rc;
    VG_HIP(hipStreamSynchronize(st));
    int status = 0;
    for (int k = 0; k < 2; ++k) {
        if (c->h_out->status[k]) status = c->h_out->status[k];
        else if (c->h_out->counters[k][2]) status = c->h_out->counters[k][2];
    }
    if (status) { vg_set_error("accurate read-out: the cold eigensolve failed (status %d)", status); return status; }
    // QtPrev now holds the cold basis and QtPrev2 the warm basis of the SAME step: no extrapolation across that pair
    for (int k = 0; k < 2; ++k) { c->d[k].have_prev2 = false; c->d[k].thin_rows = c->d[k].m; }
    c->last_thin = false;
    c->pred_consumed = false;
    c->acc_valid = true;
    return VGGP_OK;
}

// Read-outs straight from a THIN step (no cold recompute): beta and 1/D - 1 vanish outside range x range, so
//   q(v) mean      = R1r (beta sqrt(s1 s2) / v) R2r^T,                              R_dr = L0_d E_r^T  (m_d x r_d: range columns only)
//   q(v) variance  = s1 s2 [ rowsq(L0_1) rowsq(L0_2)^T + (R1r o R1r)(1/D - 1)(R2r o R2r)^T ]   (sum over ALL directions of R^2 = diag L0 L0^T)
//   posterior(x*)  = the full formulas with t_d = E_r L0_d^-1 a_d(x*)  (r_d x n*)
// and the range eigenpairs of a thin step are exact Ritz pairs of G on span(V1) = range(G) -- consistent (eigenvalue, eigenvector)
// pairs, unlike the projected-off complement rows of the full warm chain whose mixtures made the cold recompute necessary
// (DESIGN.md section 2).  VGGP_NO_THIN_READOUT=1: recompute cold as after any warm step.
static bool vg_thin_readout(const vggp_ctx* c) {
    static const bool off = getenv("VGGP_NO_THIN_READOUT") != nullptr;
    return !off && c->have_step && c->last_thin && !c->acc_valid;
}
__global__ void vg_rowsq_kernel(const double* L, int m, double* out) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m) return;
    double s = 0.0;
    for (int k = 0; k <= a; ++k) s += L[(long)a * m + k] * L[(long)a * m + k];
    out[a] = s;
}
__global__ void vg_add_outer_kernel(double* x, const double* u, const double* v, int m1, int m2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)m1 * m2) x[i] += u[i / m2] * v[i % m2];
}
static int vg_qv_thin(vggp_ctx* c, double* mean, double* var, hipStream_t st) {
    const long m1 = c->desc.m1, m2 = c->desc.m2;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const int r1 = d1.thin_rows, r2 = d2.thin_rows;
    VgGemmBatch g;
    vg_gemm_init(&g);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        vg_gemm_add(&g, d.L0, d.m, 1, d.QtPrev, 1, d.m, d.RQ, d.thin_rows, d.m, d.thin_rows, d.m);           // L0 E_r^T  (m x r)
    }
    VG_HIP(vg_gemm_launch(&g, st));
    for (int k = 0; k < 2; ++k) VG_HIP(vg_scale_sq_launch(c->d[k].RQ, c->d[k].RQsq, (long)c->d[k].m * c->d[k].thin_rows, st));
    VG_HIP(vg_qv_weights_launch(c->theta, c->beta, c->invD, c->wq, (long)r1 * r2, st, vg_uexp(d1.basis), vg_uexp(d2.basis)));
    vg_gemm_init(&g);
    vg_gemm_add(&g, d1.RQ, r1, 1, c->wq, r2, 1, c->T3, r2, (int)m1, r2, r1);
    vg_gemm_add(&g, d1.RQsq, r1, 1, c->wq + (long)r1 * r2, r2, 1, c->T3 + m1 * m2, r2, (int)m1, r2, r1);
    VG_HIP(vg_gemm_launch(&g, st));
    vg_gemm_init(&g);
    vg_gemm_add(&g, c->T3, r2, 1, d2.RQ, 1, r2, mean, (int)m2, (int)m1, (int)m2, r2);
    vg_gemm_add(&g, c->T3 + m1 * m2, r2, 1, d2.RQsq, 1, r2, var, (int)m2, (int)m1, (int)m2, r2);
    VG_HIP(vg_gemm_launch(&g, st));
    double* rs1 = c->rowpart;            // (m1 * 8 doubles of scratch, free outside a step)
    double* rs2 = c->r2;                 // (2 * m2)
    hipLaunchKernelGGL(vg_rowsq_kernel, dim3((unsigned)((m1 + 127) / 128)), dim3(128), 0, st, d1.L0, (int)m1, rs1);
    hipLaunchKernelGGL(vg_rowsq_kernel, dim3((unsigned)((m2 + 127) / 128)), dim3(128), 0, st, d2.L0, (int)m2, rs2);
    hipLaunchKernelGGL(vg_add_outer_kernel, dim3((unsigned)((m1 * m2 + 255) / 256)), dim3(256), 0, st, var, rs1, rs2, (int)m1, (int)m2);
    VG_HIP(hipGetLastError());
    VG_HIP(vg_scale_launch(var, m1 * m2, c->theta, 0, st, vg_uexp(d1.basis), vg_uexp(d2.basis)));
    return VGGP_OK;
}

static int build_RQ(vggp_ctx* c, hipStream_t st) {
    VgGemmBatch g;
    vg_gemm_init(&g);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        vg_gemm_add(&g, d.L0, d.m, 1, d.QtPrev, 1, d.m, d.RQ, d.m, d.m, d.m, d.m);   // L0 Q (QtPrev = last basis)
    }
    VG_HIP(vg_gemm_launch(&g, st));
    for (int k = 0; k < 2; ++k) VG_HIP(vg_scale_sq_launch(c->d[k].RQ, c->d[k].RQsq, (long)c->d[k].m * c->d[k].m, st));
    return VGGP_OK;
}

extern "C" int vggp_qv(vggp_ctx* c, double* mean, double* var, void* stream) {
    if (!c || !c->have_step) { vg_set_error("vggp_qv: no finished ELBO step"); return VGGP_ESTATE; }
    VG_REQUIRE(mean && var, "vggp_qv: null output");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const long m1 = c->desc.m1, m2 = c->desc.m2;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    if (vg_thin_readout(c)) return vg_qv_thin(c, mean, var, st);
    int rc = vg_accurate_state(c, st);
    if (rc) return rc;
    if ((rc = build_RQ(c, st))) return rc;
    VG_HIP(vg_qv_weights_launch(c->theta, c->beta, c->invD, c->wq, m1 * m2, st, vg_uexp(d1.basis), vg_uexp(d2.basis)));
    VgGemmBatch g;
    vg_gemm_init(&g);
    vg_gemm_add(&g, d1.RQ, m1, 1, c->wq, m2, 1, c->T3, (int)m2, (int)m1, (int)m2, (int)m1);
    vg_gemm_add(&g, d1.RQsq, m1, 1, c->invD, m2, 1, c->T3 + m1 * m2, (int)m2, (int)m1, (int)m2, (int)m1);
    VG_HIP(vg_gemm_launch(&g, st));
    vg_gemm_init(&g);
    vg_gemm_add(&g, c->T3, m2, 1, d2.RQ, 1, m2, mean, (int)m2, (int)m1, (int)m2, (int)m2);
    vg_gemm_add(&g, c->T3 + m1 * m2, m2, 1, d2.RQsq, 1, m2, var, (int)m2, (int)m1, (int)m2, (int)m2);
    VG_HIP(vg_gemm_launch(&g, st));
    VG_HIP(vg_scale_launch(var, m1 * m2, c->theta, 0, st, vg_uexp(d1.basis), vg_uexp(d2.basis)));
    return VGGP_OK;
}

__global__ void vg_kron_rows_kernel(const double* R1, const double* R2, const double* w, int m1, int m2, double* Rk,
                                    double* Rs) {
    // Rk[(a,b)][(i1,i2)] = R1[a][i1] R2[b][i2];  Rs = Rk * w[(i1,i2)]
    const long M = (long)m1 * m2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const long row = idx / M, col = idx - row * M;
    const int a = (int)(row / m2), b = (int)(row - (long)a * m2);
    const int i1 = (int)(col / m2), i2 = (int)(col - (long)i1 * m2);
    const double v = R1[a * m1 + i1] * R2[b * m2 + i2];
    Rk[idx] = v;
    Rs[idx] = v * w[col];
}

extern "C" int vggp_qv_cov(vggp_ctx* c, double* cov, void* stream) {
    if (!c || !c->have_step) { vg_set_error("vggp_qv_cov: no finished ELBO step"); return VGGP_ESTATE; }
    VG_REQUIRE(cov, "vggp_qv_cov: null output");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const long m1 = c->desc.m1, m2 = c->desc.m2, M = m1 * m2;
    VG_REQUIRE(M <= 8192, "vggp_qv_cov: M=%ld too large for a dense covariance (use vggp_qv for mean/variance)", M);
    int rc = vg_accurate_state(c, st);
    if (rc) return rc;
    if ((rc = build_RQ(c, st))) return rc;
    rc = vg_ensure_misc(c, 2 * M * M * sizeof(double));
    if (rc) return rc;
    double* Rk = (double*)c->misc;
    double* Rs = Rk + M * M;
    hipLaunchKernelGGL(vg_kron_rows_kernel, dim3((unsigned)((M * M + 255) / 256)), dim3(256), 0, st, c->d[0].RQ,
                       c->d[1].RQ, c->invD, (int)m1, (int)m2, Rk, Rs);
    VG_HIP(hipGetLastError());
    VgGemmBatch g;
    vg_gemm_init(&g);
    vg_gemm_add(&g, Rs, M, 1, Rk, 1, M, cov, (int)M, (int)M, (int)M, (int)M);
    VG_HIP(vg_gemm_launch(&g, st));
    VG_HIP(vg_scale_launch(cov, M * M, c->theta, 0, st, vg_uexp(c->d[0].basis), vg_uexp(c->d[1].basis)));
    return VGGP_OK;
}

// posterior at scattered points, processed in chunks so the workspace stays bounded
extern "C" int vggp_posterior(vggp_ctx* c, const double* xs1, const double* xs2, int64_t ns, double* mean, double* var,
                              void* stream) {
    if (!c || !c->have_step) { vg_set_error("vggp_posterior: no finished ELBO step"); return VGGP_ESTATE; }
    VG_REQUIRE(xs1 && xs2 && mean && var && ns >= 0, "vggp_posterior: bad argument");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const long m1 = c->desc.m1, m2 = c->desc.m2;
    // chunks of 64 Ki points: 6 launches each; 8 Ki-point chunks left the 1 M-point prediction launch-bound (15.4 ms, 13 TFLOP/s)
    const long chunk = std::min<long>(ns, 65536);
    if (ns == 0) return VGGP_OK;
    // per chunk: A(m x c), T(m x c) for both dims, T2sq, U, Uv (m1 x c); once: W_d = Q_d^T L0_d^-1 (m x m)
    const size_t per = (size_t)chunk * (2 * m1 + 3 * m2 + 2 * m1) + (size_t)(m1 * m1 + m2 * m2);
    // after a thin step: the same formulas on the range directions only (q1 x q2 instead of m1 x m2), no cold recompute
    const bool thin_ro = vg_thin_readout(c);
    int rc = thin_ro ? VGGP_OK : vg_accurate_state(c, st);
    if (rc) return rc;
    const int q1 = thin_ro ? c->d[0].thin_rows : (int)m1, q2 = thin_ro ? c->d[1].thin_rows : (int)m2;
    if ((rc = vg_ensure_misc(c, per * sizeof(double)))) return rc;
    double* p = (double*)c->misc;
    double* A1 = p; p += m1 * chunk;
    double* T1 = p; p += m1 * chunk;
    double* A2 = p; p += m2 * chunk;
    double* T2 = p; p += m2 * chunk;
    double* T2sq = p; p += m2 * chunk;
    double* U = p; p += m1 * chunk;
    double* Uv = p; p += m1 * chunk;
    double* W1 = p; p += m1 * m1;
    double* W2 = p;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    VG_HIP(vg_qv_weights_launch(c->theta, c->beta, c->invD, c->wq, (long)q1 * q2, st));   // wq = [beta rs / v | invD - 1]  (q1 x q2, compact)
    {
        VgGemmBatch g;                                                              // whitening and rotation in one operand
        vg_gemm_init(&g);
        vg_gemm_add(&g, d1.QtPrev, m1, 1, d1.Linv0, m1, 1, W1, (int)m1, q1, (int)m1, (int)m1);
        vg_gemm_add(&g, d2.QtPrev, m2, 1, d2.Linv0, m2, 1, W2, (int)m2, q2, (int)m2, (int)m2);
        VG_HIP(vg_gemm_launch(&g, st));
    }
    for (long off = 0; off < ns; off += chunk) {
        const int cn = (int)std::min<long>(chunk, ns - off);
        VgFactorJob fj[2] = {
            VgFactorJob{xs1 + off, d1.grid, A1, nullptr, nullptr, nullptr, cn, d1.m, d1.kind, d1.basis, 0, 0.0, c->desc.flags},
            VgFactorJob{xs2 + off, d2.grid, A2, nullptr, nullptr, nullptr, cn, d2.m, d2.kind, d2.basis, 1, 0.0, c->desc.flags}};
        VG_HIP(vg_factor_build_launch(fj, 2, c->theta, st));
        VgGemmBatch g;
        vg_gemm_init(&g);
        vg_gemm_add(&g, W1, m1, 1, A1, cn, 1, T1, cn, q1, cn, (int)m1);
        vg_gemm_add(&g, W2, m2, 1, A2, cn, 1, T2, cn, q2, cn, (int)m2);
        VG_HIP(vg_gemm_launch(&g, st));
        VG_HIP(vg_scale_sq_launch(T2, T2sq, (long)q2 * cn, st));
        vg_gemm_init(&g);
        vg_gemm_add(&g, c->wq, q2, 1, T2, cn, 1, U, cn, q1, cn, q2);
        vg_gemm_add(&g, c->wq + (long)q1 * q2, q2, 1, T2sq, cn, 1, Uv, cn, q1, cn, q2);
        VG_HIP(vg_gemm_launch(&g, st));
        VG_HIP(vg_post_combine_launch(c->theta, T1, U, Uv, nullptr, q1, q2, cn, mean + off, var + off, st));
    }
    return VGGP_OK;
}

// ---- dense posterior covariance at scattered points (kronecker_structure.py:223-229) ---------------------------------------
// T[(i1, i2)][p] = U1[i1][p] U2[i2][p]  (column-wise Khatri-Rao product), Tw = T * w[(i1, i2)]
__global__ void vg_khatri_rao_kernel(const double* U1, const double* U2, const double* w, int m1, int m2, long ns, double* T,
                                     double* Tw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)m1 * m2 * ns;
    if (idx >= total) return;
    const long u = idx / ns, p = idx - u * ns;
    const int i1 = (int)(u / m2), i2 = (int)(u - (long)i1 * m2);
    const double v = U1[(long)i1 * ns + p] * U2[(long)i2 * ns + p];
    T[idx] = v;
    if (Tw) Tw[idx] = v * w[u];
}
// cov[p][q] += s1 s2 kappa1(|x1_p - x1_q| / ell1) kappa2(|x2_p - x2_q| / ell2): the prior term K** of the product kernel
__global__ void vg_prior_cov_kernel(const double* xs1, const double* xs2, long ns, int kind1, int kind2, const double* theta,
                                    double* cov) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ns * ns) return;
    const long p = idx / ns, q = idx - p * ns;
    double k1, k2, d;
    vg_kappa(kind1, fabs(xs1[p] - xs1[q]), 1.0 / theta[0], k1, d);
    vg_kappa(kind2, fabs(xs2[p] - xs2[q]), 1.0 / theta[1], k2, d);
    cov[idx] += theta[2] * theta[3] * k1 * k2;
}
hipError_t vg_prior_cov_launch(const double* xs1, const double* xs2, long ns, int kind1, int kind2, const double* theta,
                               double* cov, hipStream_t st) {
    hipLaunchKernelGGL(vg_prior_cov_kernel, dim3((unsigned)((ns * ns + 255) / 256)), dim3(256), 0, st, xs1, xs2, ns, kind1,
                       kind2, theta, cov);
    return hipGetLastError();
}

// whitened, rotated cross-covariances at x*: U_d = Q_d^T L0_d^{-1} A0_d(x*)  (m_d x ns, unit outputscale), by substitution
static int posterior_factors(vggp_ctx* c, const double* xs1, const double* xs2, int ns, double* A1, double* A2, double* U1,
                             double* U2, hipStream_t st, bool rotate) {
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    VgFactorJob fj[2] = {
        VgFactorJob{xs1, d1.grid, A1, nullptr, nullptr, nullptr, ns, d1.m, d1.kind, d1.basis, 0, 0.0, c->desc.flags},
        VgFactorJob{xs2, d2.grid, A2, nullptr, nullptr, nullptr, ns, d2.m, d2.kind, d2.basis, 1, 0.0, c->desc.flags}};
    VG_HIP(vg_factor_build_launch(fj, 2, c->theta, st));
    VgTrsmSpec q[2] = {{d1.L0, d1.m, d1.Linv0, 16L * d1.m + 16, d1.m, A1, ns, 1, ns, d1.m, 0},
                       {d2.L0, d2.m, d2.Linv0, 16L * d2.m + 16, d2.m, A2, ns, 1, ns, d2.m, 0}};
    int rc = trsm_batch(q, 2, st);
    if (rc || !rotate) return rc;
    VgGemmBatch g;
    vg_gemm_init(&g);
    vg_gemm_add(&g, d1.QtPrev, d1.m, 1, A1, ns, 1, U1, ns, d1.m, ns, d1.m);
    vg_gemm_add(&g, d2.QtPrev, d2.m, 1, A2, ns, 1, U2, ns, d2.m, ns, d2.m);
    VG_HIP(vg_gemm_launch(&g, st));
    return VGGP_OK;
}

extern "C" int vggp_posterior_cov(vggp_ctx* c, const double* xs1, const double* xs2, int64_t ns, double* cov, void* stream) {
    if (!c || !c->have_step) { vg_set_error("vggp_posterior_cov: no finished ELBO step"); return VGGP_ESTATE; }
    VG_REQUIRE(xs1 && xs2 && cov && ns >= 1, "vggp_posterior_cov: bad argument");
    const long m1 = c->desc.m1, m2 = c->desc.m2, M = m1 * m2;
    VG_REQUIRE(ns <= 8192 && M * ns <= (1L << 27), "vggp_posterior_cov: ns=%ld points x M=%ld is too large for a dense covariance "
               "(use vggp_posterior for mean / variance)", (long)ns, M);
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const size_t need = (size_t)(2 * (m1 + m2) * ns + 2 * M * ns + M) * sizeof(double);
    int rc = vg_accurate_state(c, st);
    if (rc) return rc;
    if ((rc = vg_ensure_misc(c, need))) return rc;
    double* p = (double*)c->misc;
    double* A1 = p; p += m1 * ns;
    double* A2 = p; p += m2 * ns;
    double* U1 = p; p += m1 * ns;
    double* U2 = p; p += m2 * ns;
    double* T = p; p += M * ns;
    double* Tw = p; p += M * ns;
    if ((rc = posterior_factors(c, xs1, xs2, (int)ns, A1, A2, U1, U2, st, true))) return rc;
    VG_HIP(vg_qv_weights_launch(c->theta, c->beta, c->invD, c->wq, M, st));             // wq[M ..] = 1/D - 1
    hipLaunchKernelGGL(vg_khatri_rao_kernel, dim3((unsigned)((M * ns + 255) / 256)), dim3(256), 0, st, U1, U2, c->wq + M, (int)m1,
                       (int)m2, (long)ns, T, Tw);
    VG_HIP(hipGetLastError());
    // cov = s1 s2 Tw^T T  (t_p = sqrt(s1 s2) u1_p (x) u2_p for both Kuu scalings)  +  K**
    VgGemmBatch g;
    vg_gemm_init(&g);
    vg_gemm_add(&g, Tw, 1, ns, T, ns, 1, cov, (int)ns, (int)ns, (int)ns, (int)M, 1, 0, 1, 0, c->h_theta[2] * c->h_theta[3], 0);
    VG_HIP(vg_gemm_launch(&g, st));
    VG_HIP(vg_prior_cov_launch(xs1, xs2, ns, c->d[0].kind, c->d[1].kind, c->theta, cov, st));
    return VGGP_OK;
}

// Gridded read-out (include/vggp.h): t_d = sqrt(s_d) Q_d^T L0_d^{-1} C_d^T; mean = T1^T (beta / v) T2;
// var = s1 s2 (kd1 kd2^T + (T1 o T1)^T W (T2 o T2)), W = D - 1 (literal) or 1/D - 1.
extern "C" int vggp_readout(vggp_ctx* c, const double* C1, int64_t mv1, const double* C2, int64_t mv2, const double* kd1,
                            const double* kd2, double* mean, double* var, int flags, void* stream) {
    if (!c || !c->have_step) { vg_set_error("vggp_readout: no finished ELBO step"); return VGGP_ESTATE; }
    VG_REQUIRE(C1 && C2 && kd1 && kd2 && mean && var && mv1 >= 1 && mv2 >= 1, "vggp_readout: bad argument");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const long m1 = c->desc.m1, m2 = c->desc.m2;
    const size_t need = (size_t)(3 * (m1 * mv1 + m2 * mv2) + 2 * m1 * mv2) * sizeof(double);
    int rc = vg_accurate_state(c, st);
    if (rc) return rc;
    if ((rc = vg_ensure_misc(c, need))) return rc;
    double* p = (double*)c->misc;
    double* X1 = p; p += m1 * mv1;
    double* T1 = p; p += m1 * mv1;
    double* S1 = p; p += m1 * mv1;
    double* X2 = p; p += m2 * mv2;
    double* T2 = p; p += m2 * mv2;
    double* S2 = p; p += m2 * mv2;
    double* U = p; p += m1 * mv2;
    double* Uv = p;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    VG_HIP(vg_readout_weights_launch(c->theta, c->beta, c->invD, c->wq, m1 * m2, (flags & VGGP_READOUT_LITERAL) ? 1 : 0, st));
    VgGemmBatch g;
    vg_gemm_init(&g);                                    // X_d = Linv0_d C_d^T            (m_d x mv_d)
    vg_gemm_add(&g, d1.Linv0, m1, 1, C1, 1, m1, X1, (int)mv1, (int)m1, (int)mv1, (int)m1);
    vg_gemm_add(&g, d2.Linv0, m2, 1, C2, 1, m2, X2, (int)mv2, (int)m2, (int)mv2, (int)m2);
    VG_HIP(vg_gemm_launch(&g, st));
    vg_gemm_init(&g);                                    // T_d = Q_d^T X_d  (rows of QtPrev are the eigenvectors)
    vg_gemm_add(&g, d1.QtPrev, m1, 1, X1, mv1, 1, T1, (int)mv1, (int)m1, (int)mv1, (int)m1);
    vg_gemm_add(&g, d2.QtPrev, m2, 1, X2, mv2, 1, T2, (int)mv2, (int)m2, (int)mv2, (int)m2);
    VG_HIP(vg_gemm_launch(&g, st));
    VG_HIP(vg_scale_sq_launch(T1, S1, m1 * mv1, st));
    VG_HIP(vg_scale_sq_launch(T2, S2, m2 * mv2, st));
    vg_gemm_init(&g);                                    // U = Wm T2, Uv = Wv (T2 o T2)   (m1 x mv2)
    vg_gemm_add(&g, c->wq, m2, 1, T2, mv2, 1, U, (int)mv2, (int)m1, (int)mv2, (int)m2);
    vg_gemm_add(&g, c->wq + m1 * m2, m2, 1, S2, mv2, 1, Uv, (int)mv2, (int)m1, (int)mv2, (int)m2);
    VG_HIP(vg_gemm_launch(&g, st));
    vg_gemm_init(&g);                                    // mean = T1^T U, var' = (T1 o T1)^T Uv   (mv1 x mv2)
    vg_gemm_add(&g, T1, 1, mv1, U, mv2, 1, mean, (int)mv2, (int)mv1, (int)mv2, (int)m1);
    vg_gemm_add(&g, S1, 1, mv1, Uv, mv2, 1, var, (int)mv2, (int)mv1, (int)mv2, (int)m1);
    VG_HIP(vg_gemm_launch(&g, st));
    VG_HIP(vg_readout_var_launch(c->theta, kd1, kd2, mv1, mv2, var, st));
    return VGGP_OK;
}

// ---------------------------------------------------------------------------------
// exported building blocks
extern "C" int vggp_factor_build(vggp_ctx* c, int kind, int basis, const double* x, int64_t n, const double* grid,
                                 int64_t m, double ell, int flags, double* A0, double* dA0, double* K0, double* dK0, void* stream) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_REQUIRE(kind >= 0 && kind <= 3 && basis >= 0 && basis <= 4, "vggp_factor_build: bad kind/basis");
    VG_REQUIRE(!((basis == VGGP_BASIS_VFF || basis == VGGP_BASIS_B1) && kind != VGGP_KIND_MATERN12),
               "vggp_factor_build: VFF / B1 are Matern-1/2 only");
    VG_REQUIRE(!(basis == VGGP_BASIS_B0 && kind != VGGP_KIND_MATERN12), "vggp_factor_build: B0 is Matern-1/2 only");
    VG_REQUIRE(n >= 0 && m >= 1 && ell > 0.0, "vggp_factor_build: bad sizes / lengthscale");
    VG_REQUIRE(grid || basis == VGGP_BASIS_ONE, "vggp_factor_build: null grid");
    VG_REQUIRE((x || !(A0 || dA0)), "vggp_factor_build: null x");
    VG_ENTER_DEVICE(c->device);
    VgFactorJob j{x, grid, n > 0 ? A0 : nullptr, n > 0 ? dA0 : nullptr, K0, dK0, (int)n, (int)m, kind, basis, -1, ell, flags};
    VG_HIP(vg_factor_build_launch(&j, 1, nullptr, stream ? (hipStream_t)stream : c->own_stream));
    return VGGP_OK;
}

// adds jit to the diagonal of the m x m copy W of K (one launch per jitter attempt of the blocked path)
__global__ void vg_copy_jitter_kernel(const double* K, double* W, long m, double jit) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * m) return;
    W[idx] = K[idx] + ((idx / m == idx % m) ? jit : 0.0);
}

extern "C" int vggp_cholesky_inverse(vggp_ctx* c, const double* K, int64_t m, double* L, double* Linv, double* jitter_out,
                                     void* stream) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_REQUIRE(K && L && Linv && m >= 1 && m <= 8192, "vggp_cholesky_inverse: bad argument (1 <= m <= 8192)");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    if (m > 128) {
        // beyond one workgroup: blocked factorisation (128-wide panels by the single-workgroup kernel, the rest by MFMA
        // GEMMs); the jitter schedule is walked on the host, one attempt per level
        const size_t nblk = (size_t)(m + VG_DENSE_MB - 1) / VG_DENSE_MB;
        const size_t need = ((size_t)m * m + nblk * VG_DENSE_MB * VG_DENSE_MB + (size_t)VG_DENSE_MB * m + VG_DENSE_MB * (VG_DENSE_MB + 1) + 64) * sizeof(double);
        int rc = vg_ensure_misc(c, need);
        if (rc) return rc;
        double* p = (double*)c->misc;
        VgDenseChol w{};
        w.S = p; p += (size_t)m * m;
        w.DI = p; p += nblk * VG_DENSE_MB * VG_DENSE_MB;
        w.Tmp = p; p += (size_t)VG_DENSE_MB * m;
        w.scratch = p; p += VG_DENSE_MB * (VG_DENSE_MB + 1);
        w.jit = p; p += 8;
        w.status = reinterpret_cast<int*>(p);
        w.L = L; w.X = Linv; w.M = m; w.Sinv = nullptr;
        static const double levels[4] = {0.0, 1e-8, 1e-7, 1e-6};
        for (int lvl = 0; lvl < 4; ++lvl) {
            VG_HIP(hipMemsetAsync(w.status, 0, 2 * sizeof(int), st));
            hipLaunchKernelGGL(vg_copy_jitter_kernel, dim3((unsigned)(((size_t)m * m + 255) / 256)), dim3(256), 0, st, K, w.S, (long)m, levels[lvl]);
            VG_HIP(hipGetLastError());
            if ((rc = vg_blocked_chol_inverse(w, st))) return rc;
            int hs = 0;
            VG_HIP(hipMemcpyAsync(&hs, w.status, sizeof(int), hipMemcpyDeviceToHost, st));
            VG_HIP(hipStreamSynchronize(st));
            if (!hs) { if (jitter_out) *jitter_out = levels[lvl]; return VGGP_OK; }
        }
        if (jitter_out) *jitter_out = -1.0;
        vg_set_error("vggp_cholesky_inverse: not positive definite after jitter 1e-6");
        return VGGP_ENOTPD;
    }
    int rc = vg_ensure_misc(c, (size_t)(m * (m + 1) + 16) * sizeof(double));
    if (rc) return rc;
    double* scratch = (double*)c->misc;
    double* jit = scratch + m * (m + 1);
    int* status = (int*)(jit + 2);
    VgClearArgs clr;
    clr.n = 2;
    clr.ptr[0] = reinterpret_cast<int*>(jit); clr.nwords[0] = 16;
    clr.ptr[1] = reinterpret_cast<int*>(scratch); clr.nwords[1] = 16;
    VG_HIP(vg_clear_launch(&clr, st));
    VgCholJob j{K, L, Linv, scratch, jit, status, (int)m};
    VG_HIP(vg_chol_launch(&j, 1, st));
    double hj = 0.0;
    int hs = 0;
    VG_HIP(hipMemcpyAsync(&hj, jit, sizeof(double), hipMemcpyDeviceToHost, st));
    VG_HIP(hipMemcpyAsync(&hs, status, sizeof(int), hipMemcpyDeviceToHost, st));
    VG_HIP(hipStreamSynchronize(st));
    if (jitter_out) *jitter_out = hj;
    if (hs) { vg_set_error("vggp_cholesky_inverse: not positive definite after jitter 1e-6"); return VGGP_ENOTPD; }
    return VGGP_OK;
}

extern "C" int vggp_eigh(vggp_ctx* c, const double* G, int64_t m, double* lam, double* Qt, int32_t* sweeps_out, int flags,
                         void* stream) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_REQUIRE(G && lam && Qt && m >= 1 && m <= 256, "vggp_eigh: bad argument (1 <= m <= 256)");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const long m2e = m + (m & 1);
    const int max_rounds = (int)(VG_EIG_MAXSWEEP * (m2e - 1));
    const size_t logb = (vg_eigh_log_bytes((int)m) + 15) & ~size_t(15);
    const size_t need = (size_t)m2e * (m2e + 1) * 8 + logb + (size_t)max_rounds * 4 + 256 + (size_t)m2e * 4 + 64;
    int rc = vg_ensure_misc(c, need);
    if (rc) return rc;
    char* p = (char*)c->misc;
    double* gwork = (double*)p; p += (size_t)m2e * (m2e + 1) * 8;
    double2* rotlog = (double2*)p; p += logb;
    int* roundlog = (int*)p; p += (size_t)max_rounds * 4;
    p = (char*)(((uintptr_t)p + 63) & ~uintptr_t(63));
    int* counters = (int*)p;
    int* perm = counters + 16;
    VgClearArgs clr;
    clr.n = 1;
    clr.ptr[0] = counters; clr.nwords[0] = 8;
    VG_HIP(vg_clear_launch(&clr, st));
    VgEigJob j{G, lam, Qt, nullptr, gwork, rotlog, roundlog, counters, (int)m, max_rounds, (long)logb,
               (flags & VGGP_FLAG_BLOCK_JACOBI) ? 1 : 0};
    j.perm = perm;
    j.err = counters + 4;
    VG_HIP(vg_eigh_launch(&j, 1, st));
    int hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    VG_HIP(hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, st));
    VG_HIP(hipStreamSynchronize(st));
    if (sweeps_out) *sweeps_out = hc[1] & 0xff;
    if (hc[2] || hc[4]) { vg_set_error("vggp_eigh: no convergence"); return VGGP_ENOCONV; }
    return VGGP_OK;
}

extern "C" int vggp_gemm(vggp_ctx* c, const double* A, int64_t sa_m, int64_t sa_k, const double* B, int64_t sb_k,
                         int64_t sb_n, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_REQUIRE(A && B && C && M >= 1 && N >= 1 && K >= 1 && ldc >= N, "vggp_gemm: bad argument");
    VG_REQUIRE(M < (1L << 30) && N < (1L << 30) && K < (1L << 30), "vggp_gemm: dimension too large");
    VG_ENTER_DEVICE(c->device);
    VgGemmBatch g;
    vg_gemm_init(&g);
    vg_gemm_add(&g, A, sa_m, sa_k, B, sb_k, sb_n, C, (int)ldc, (int)M, (int)N, (int)K);
    VG_HIP(vg_gemm_launch(&g, stream ? (hipStream_t)stream : c->own_stream));
    return VGGP_OK;
}

extern "C" int vggp_trsm(vggp_ctx* c, const double* L, int64_t m, const double* R, int64_t ncols, double* X, int trans,
                         void* stream) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_REQUIRE(L && R && X && m >= 1 && m <= 16384 && ncols >= 1, "vggp_trsm: bad argument (1 <= m <= 16384)");
    VG_REQUIRE(m * ncols < (1L << 31) && m * m < (1L << 31), "vggp_trsm: matrix too large");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const long nb16 = (m + 15) / 16;
    int rc = vg_ensure_misc(c, (size_t)nb16 * 256 * sizeof(double));
    if (rc) return rc;
    double* Dinv = (double*)c->misc;
    VG_HIP(vg_tri_diaginv_launch(L, m, (int)m, Dinv, st));
    if (X != R) VG_HIP(hipMemcpyAsync(X, R, sizeof(double) * m * ncols, hipMemcpyDeviceToDevice, st));
    return trsm_inplace(L, m, m, Dinv, X, ncols, 1, ncols, trans ? 1 : 0, st);
}

// Explicit inverse of a lower-triangular factor (n > 128) without ever inverting more than a 16 x 16 block directly:
// the 128 x 128 diagonal blocks by substitution on the identity (strip kernel, one launch for all blocks of all factors),
// then block doubling  inv([A 0; B C]) = [A^-1 0; -C^-1 B A^-1, C^-1]  with two MFMA GEMM launches per level
// (128 -> 256 -> 512 -> ...; every pair of every factor in the same launch).  The 128-blocks of Xinv above the diagonal are
// neither written nor read (triangular-aware products): consumers must skip them as well.
struct VgTriInvSpec { const double* L; long n; const double* Dinv16; double* Xinv; double* tmp; };
static int tri_inverse_batch(const VgTriInvSpec* sp, int nf, hipStream_t st) {
    VgTrsmJob tj[16];
    int nt = 0;
    for (int f = 0; f < nf; ++f)
        for (long r0 = 0; r0 < sp[f].n; r0 += VG_TRSM_BLK) {
            const long rb = std::min<long>(VG_TRSM_BLK, sp[f].n - r0);
            if (nt == 16) { VG_HIP(vg_trsm_launch(tj, nt, st)); nt = 0; }
            VgTrsmJob j{sp[f].L + r0 * sp[f].n + r0, sp[f].Dinv16 + (r0 / 16) * 256, nullptr, sp[f].Xinv + r0 * sp[f].n + r0, sp[f].n, 256, 16,
                        sp[f].n, 1, sp[f].n, 1, rb, (int)rb, 0};
            j.rhs_ident = 1;
            tj[nt++] = j;
        }
    if (nt) VG_HIP(vg_trsm_launch(tj, nt, st));
    long nmax = 0;
    for (int f = 0; f < nf; ++f) nmax = std::max(nmax, sp[f].n);
    for (long b = VG_TRSM_BLK; b < nmax; b *= 2) {
        for (int pass = 0; pass < 2; ++pass) {
            VgGemmBatch g;
            vg_gemm_init(&g);
            for (int f = 0; f < nf; ++f) {
                const long n = sp[f].n;
                for (long a0 = 0; a0 + b < n; a0 += 2 * b) {
                    const long c0 = a0 + b, cb = std::min<long>(b, n - c0);           // A = [a0, a0+b), C = [c0, c0+cb)
                    if (g.nprob == VG_GEMM_MAXP) { VG_HIP(vg_gemm_launch(&g, st)); vg_gemm_init(&g); }
                    int ip;
                    if (pass == 0) {      // T = B A^-1        (cb x b; A^-1 lower triangular: its upper 128-blocks are never read)
                        ip = vg_gemm_add(&g, sp[f].L + c0 * n + a0, n, 1, sp[f].Xinv + a0 * n + a0, n, 1, sp[f].tmp + c0 * n + a0, (int)n, (int)cb,
                                         (int)b, (int)b);
                        g.p[ip].tri = VG_TRI_B_LOWER;
                    } else {              // W = -C^-1 T
                        ip = vg_gemm_add(&g, sp[f].Xinv + c0 * n + c0, n, 1, sp[f].tmp + c0 * n + a0, n, 1, sp[f].Xinv + c0 * n + a0, (int)n,
                                         (int)cb, (int)b, (int)cb, 1, 0, 1, 0, -1.0, 0);
                        g.p[ip].tri = VG_TRI_A_LOWER;
                    }
                }
            }
            if (g.nprob) VG_HIP(vg_gemm_launch(&g, st));
        }
    }
    return VGGP_OK;
}

extern "C" int vggp_kron_solve(vggp_ctx* c, const double* L1, int64_t n1, const double* L2, int64_t n2, const double* Y,
                               double* X, void* stream) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_REQUIRE(L1 && L2 && Y && X && n1 >= 1 && n2 >= 1 && n1 <= 16384 && n2 <= 16384, "vggp_kron_solve: bad argument");
    VG_REQUIRE(n1 * n2 < (1L << 31) && n1 * n1 < (1L << 31) && n2 * n2 < (1L << 31), "vggp_kron_solve: matrix too large");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const long nb1 = (n1 + 15) / 16, nb2 = (n2 + 15) / 16;
    static const bool subst_only = getenv("VGGP_KRON_SUBST") != nullptr;
    const bool small = (n1 <= VG_TRSM_BLK && n2 <= VG_TRSM_BLK) || subst_only;
    const size_t need = (size_t)(nb1 + nb2) * 256 + (small ? 0 : (size_t)(2 * (n1 * n1 + n2 * n2) + 4 * n1 * n2));
    int rc = vg_ensure_misc(c, need * sizeof(double));
    if (rc) return rc;
    double* D1 = (double*)c->misc;
    double* D2 = D1 + nb1 * 256;
    VG_HIP(vg_tri_diaginv_launch(L1, n1, (int)n1, D1, st, L2, n2, (int)n2, D2));
    if (small) {
        // X = L1^{-T} ( L1^{-1} Y L2^{-T} ) L2^{-1}: four solves by substitution, in place on X ([n1][n2] row-major);
        // left solves see X as it is (row k = row of X), right solves see its transpose (row k = column k of X)
        if (X != Y) VG_HIP(hipMemcpyAsync(X, Y, sizeof(double) * n1 * n2, hipMemcpyDeviceToDevice, st));
        if ((rc = trsm_inplace(L1, n1, n1, D1, X, n2, 1, n2, 0, st))) return rc;        // L1 T = Y
        if ((rc = trsm_inplace(L2, n2, n2, D2, X, 1, n2, n1, 0, st))) return rc;        // L2 T'^T = T^T      (T' = T L2^{-T})
        if ((rc = trsm_inplace(L1, n1, n1, D1, X, n2, 1, n2, 1, st))) return rc;        // L1^T T'' = T'
        if ((rc = trsm_inplace(L2, n2, n2, D2, X, 1, n2, n1, 1, st))) return rc;        // L2^T X^T = T''^T   (X = T'' L2^{-1})
        return VGGP_OK;
    }
    // Larger factors: a substitution sweep over n / 128 block rows is a chain of 2 n / 128 dependent launches per solve (8 of
    // them: 1.3 ms at n = 1024, measured), so the factors are inverted -- 128 x 128 diagonal blocks by substitution, the
    // rest by block doubling (tri_inverse_batch) -- and applied as four triangular-aware MFMA GEMMs.  Everything is inside
    // this call (BASELINE metric ii is timed from the Cholesky factors).
    double* Li1 = D2 + nb2 * 256;
    double* Li2 = Li1 + n1 * n1;
    double* tmp1 = Li2 + n2 * n2;
    double* tmp2 = tmp1 + n1 * n1;
    double* T1 = tmp2 + n2 * n2;                 // two split-K slabs each
    double* T2 = T1 + 2 * n1 * n2;
    // (no clearing of Li: the substitution writes whole diagonal 128-blocks, and every product below skips the 128-blocks above them)
    VgTriInvSpec sp[2] = {{L1, (long)n1, D1, Li1, tmp1}, {L2, (long)n2, D2, Li2, tmp2}};
    if ((rc = tri_inverse_batch(sp, 2, st))) return rc;            // both factors ride in the same launches
    // X = P1 Y P2 with P_d = K_d^-1 = Linv_d^T Linv_d: one launch forms both P_d (X^T X of a lower-triangular X: tiles skip the
    // k-range where either operand vanishes), then TWO dense products instead of four triangular-aware ones -- a 1024^3 product is
    // 256 tiles = one workgroup per CU, and the triangular skip leaves the longest tile (full K) on the critical path, so each of
    // the four cost what a dense product costs (40 us at n = 1024).  The error is that of the explicit inverses either way
    // (eps cond(K_d) per side).  VGGP_KRON_FOUR=1 keeps the four-product form reachable (A/B).
    static const bool four = getenv("VGGP_KRON_FOUR") != nullptr;
    VgGemmBatch g;
    if (!four) {
        double* P1 = tmp1;                       // (the block-doubling scratch is free again)
        double* P2 = tmp2;
        vg_gemm_init(&g);
        { const int i = vg_gemm_add(&g, Li1, 1, n1, Li1, n1, 1, P1, (int)n1, (int)n1, (int)n1, (int)n1); g.p[i].tri = VG_TRI_A_UPPER_B_LOWER; }
        { const int i = vg_gemm_add(&g, Li2, 1, n2, Li2, n2, 1, P2, (int)n2, (int)n2, (int)n2, (int)n2); g.p[i].tri = VG_TRI_A_UPPER_B_LOWER; g.p[i].rev = 1; }
        VG_HIP(vg_gemm_launch(&g, st, VG_GEMM_TAG_WIDE));
        vg_gemm_init(&g);
        vg_gemm_add(&g, P1, n1, 1, Y, n2, 1, T1, (int)n2, (int)n1, (int)n2, (int)n1);           // P1 Y
        VG_HIP(vg_gemm_launch(&g, st, VG_GEMM_TAG_WIDE));
        vg_gemm_init(&g);
        vg_gemm_add(&g, T1, n2, 1, P2, n2, 1, X, (int)n2, (int)n1, (int)n2, (int)n2);           // . P2   (symmetric)
        VG_HIP(vg_gemm_launch(&g, st, VG_GEMM_TAG_WIDE));
        return VGGP_OK;
    }
    // Four triangular-aware GEMMs.  Splitting the reduction in two
    // (slabs summed on load by the next product, two workgroups per CU) was measured SLOWER (397 vs 336 us for the whole
    // solve: the slab-summing operand path); VGGP_KRON_KSPLIT=2 keeps it reachable.
    const long slab = (long)n1 * n2;
    static const char* kse = getenv("VGGP_KRON_KSPLIT");
    const int ks = kse ? atoi(kse) : 1;
    vg_gemm_init(&g);
    { const int i = vg_gemm_add(&g, Li1, n1, 1, Y, n2, 1, T1, (int)n2, (int)n1, (int)n2, (int)n1, ks, slab); g.p[i].tri = VG_TRI_A_LOWER; }  // L1inv Y
    const int s1 = g.p[0].ksplit;
    VG_HIP(vg_gemm_launch(&g, st, VG_GEMM_TAG_WIDE));
    vg_gemm_init(&g);
    { const int i = vg_gemm_add(&g, T1, n2, 1, Li2, 1, n2, T2, (int)n2, (int)n1, (int)n2, (int)n2, ks, slab); g.p[i].tri = VG_TRI_B_UPPER;  // . L2inv^T
      g.p[i].a_nslab = s1; g.p[i].a_slab = slab; }
    const int s2 = g.p[0].ksplit;
    VG_HIP(vg_gemm_launch(&g, st, VG_GEMM_TAG_WIDE));
    vg_gemm_init(&g);
    { const int i = vg_gemm_add(&g, Li1, 1, n1, T2, n2, 1, T1, (int)n2, (int)n1, (int)n2, (int)n1, ks, slab, s2, slab); g.p[i].tri = VG_TRI_A_UPPER; }  // L1inv^T .
    const int s3 = g.p[0].ksplit;
    VG_HIP(vg_gemm_launch(&g, st, VG_GEMM_TAG_WIDE));
    vg_gemm_init(&g);
    { const int i = vg_gemm_add(&g, T1, n2, 1, Li2, n2, 1, X, (int)n2, (int)n1, (int)n2, (int)n2); g.p[i].tri = VG_TRI_B_LOWER;            // . L2inv
      g.p[i].a_nslab = s3; g.p[i].a_slab = slab; }
    VG_HIP(vg_gemm_launch(&g, st, VG_GEMM_TAG_WIDE));
    return VGGP_OK;
}

// diagnostic builds only (-DVG_EIG_STAMP): copy the head of the scratch buffer to the host
// ---------------------------------------------------------------------------------
// Gradient of the ELBO with respect to the inducing-point coordinates (SVGP's trainable Z: kronecker_structure.py:303-304
// registers Z as a Parameter and autograd differentiates through kernel(Z), kernel(Z, x)).  Spec: oracle/kron.py z_grad.
int vg_trsm_batch(const VgTrsmSpec* sp, int n, hipStream_t st) { return trsm_batch(sp, n, st); }      // (masked.hip: vggp_zgrad_scattered)

// The analytic lengthscale gradient is linear in the perturbation (dK, dA) it is fed, dELBO = <W_M, L^-1 dK L^-T> + <W_V, L^-1 dA>,
// so Kbar = L^-T W_M L^-1 and Abar = L^-T W_V are the sensitivities, and for a stationary kernel
// d kappa(z_i, x) / d z_i = -(d kappa / d ell) ell / (z_i - x): the derivative factors dA0, dK0 of the step are reused.
// Uses the resident state of the last vggp_elbo_step on the same Y (warm-basis accuracy, like the lengthscale gradient); one
// extra pass over Y (B1 Y^T), ~25 small launches.
extern "C" int vggp_zgrad(vggp_ctx* c, const double* Y, double* gz1, double* gz2, void* stream) {
    if (!c || !c->have_step) { vg_set_error("vggp_zgrad: no finished ELBO step"); return VGGP_ESTATE; }
    VG_REQUIRE(Y && gz1 && gz2, "vggp_zgrad: null argument");
    // Row-sharded job: every term of g is a sum over observations, so each rank forms the part of ITS rows -- the columns of dimension
    // 2 it owns, and for dimension 1 (whose n1 columns every rank holds) the projection term through its rows of Y -- while the terms
    // that only involve replicated state (Kbar, and s1 W_H B1 of dimension 1) are added by rank 0 alone; ONE all-reduce of m1 + m2 doubles.
    const bool multi = c->n_ranks > 1 || c->comm || c->cb;
    const bool lead = !multi || c->rank == 0;
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    if (c->last_thin) {
        // the inducing-point gradient contracts the FULL m-space state (beta, 1/D, Q): a thin step leaves none.  Rebuild it from
        // the resident G, H, C (cold finish half) and keep this plan on the full chain from now on -- its caller trains Z.
        c->thin_off = true;
        const int rca = vg_accurate_state(c, st);
        if (rca) return rca;
    }
    const long n1 = c->desc.n1, n2 = c->desc.n2, m1 = c->desc.m1, m2 = c->desc.m2;
    VgDim &d1 = c->d[0], &d2 = c->d[1];
    const bool pts[2] = {d1.basis == VGGP_BASIS_POINTS, d2.basis == VGGP_BASIS_POINTS};
    if (!pts[0]) VG_HIP(hipMemsetAsync(gz1, 0, sizeof(double) * m1, st));
    if (!pts[1]) VG_HIP(hipMemsetAsync(gz2, 0, sizeof(double) * m2, st));
    if (!pts[0] && !pts[1]) return VGGP_OK;
    const double s1 = c->h_theta[2], s2 = c->h_theta[3], v = c->h_theta[4];
    // workspace
    const long mm1 = m1 * m1, mm2 = m2 * m2, m12 = m1 * m2;
    const size_t need = sizeof(double) * (size_t)(6 * mm1 + 6 * mm2 + 2 * m12 + m1 * n1 + m2 * n2 + m1 * n2 + m1 + m2);
    int rc = vg_ensure_misc(c, need);
    if (rc) return rc;
    double* w = reinterpret_cast<double*>(c->misc);
    double *Xa[2], *Xl[2], *WE[2], *WF[2], *TA[2], *TB[2];
    for (int k = 0; k < 2; ++k) {
        const long mm = k ? mm2 : mm1;
        Xa[k] = w; w += mm; Xl[k] = w; w += mm; WE[k] = w; w += mm; WF[k] = w; w += mm; TA[k] = w; w += mm; TB[k] = w; w += mm;
    }
    double* U = w; w += m12;
    double* T1 = w; w += m12;
    double* WV[2] = {w, w + m1 * n1}; w += m1 * n1 + m2 * n2;
    double* R1 = w; w += m1 * n2;
    double* gzbuf = w;                                   // [m1 + m2]: the all-reduce buffer of a row-sharded job
    VgGemmBatch g;
    // 1. the four beta Gram matrices
    vg_gemm_init(&g);
    vg_gemm_add(&g, c->beta, m2, 1, c->beta, 1, m2, Xa[0], (int)m1, (int)m1, (int)m1, (int)m2);    // beta beta^T
    vg_gemm_add(&g, c->bl2, m2, 1, c->beta, 1, m2, Xl[0], (int)m1, (int)m1, (int)m1, (int)m2);     // (beta lam2) beta^T
    vg_gemm_add(&g, c->beta, 1, m2, c->beta, m2, 1, Xa[1], (int)m2, (int)m2, (int)m2, (int)m1);    // beta^T beta
    vg_gemm_add(&g, c->bl1, 1, m2, c->beta, m2, 1, Xl[1], (int)m2, (int)m2, (int)m2, (int)m1);     // (lam1 beta)^T beta
    // ... and R1 = B1 Y^T (m1 x n2): the projection the step itself does not form
    if (pts[1]) vg_gemm_add(&g, d1.BV, n1, 1, Y, 1, n1, R1, (int)n2, (int)m1, (int)n2, (int)n1);
    VG_HIP(vg_gemm_launch(&g, st));
    // 2. weights
    VG_HIP(vg_zw_launch(c->theta, 0, d1.lam0, d2.lam0, (int)m1, (int)m2, Xa[0], Xl[0], c->r1, c->r1l, WE[0], WF[0], st));
    VG_HIP(vg_zw_launch(c->theta, 1, d2.lam0, d1.lam0, (int)m2, (int)m1, Xa[1], Xl[1], c->r2, c->r2l, WE[1], WF[1], st));
    // 3. back to the original basis: W_M = Q W_E Q^T, W_H = Q (W_F + W_F^T) Q^T (Q = Qt^T), T1 = Q1 (beta / v^2) Q2^T
    vg_gemm_init(&g);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        vg_gemm_add(&g, WE[k], d.m, 1, d.Qt, d.m, 1, TA[k], d.m, d.m, d.m, d.m);
        vg_gemm_add(&g, WF[k], d.m, 1, d.Qt, d.m, 1, TB[k], d.m, d.m, d.m, d.m);
    }
    vg_gemm_add(&g, c->beta, m2, 1, d2.Qt, m2, 1, U, (int)m2, (int)m1, (int)m2, (int)m2, 1, 0, 1, 0, 1.0 / (v * v));
    VG_HIP(vg_gemm_launch(&g, st));
    vg_gemm_init(&g);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        vg_gemm_add(&g, d.Qt, 1, d.m, TA[k], d.m, 1, WE[k], d.m, d.m, d.m, d.m, 1, 0, 1, 0, lead ? 1.0 : 0.0);       // W_M -> WE (replicated: rank 0 only)
        vg_gemm_add(&g, d.Qt, 1, d.m, TB[k], d.m, 1, WF[k], d.m, d.m, d.m, d.m);       // W_H -> WF
    }
    vg_gemm_add(&g, d1.Qt, 1, m1, U, m2, 1, T1, (int)m2, (int)m1, (int)m2, (int)m1);
    VG_HIP(vg_gemm_launch(&g, st));
    // 4. W_V = s W_H B0 (+ the projection term in the next launch)
    vg_gemm_init(&g);
    for (int k = 0; k < 2; ++k) {
        VgDim& d = c->d[k];
        if (!pts[k]) continue;
        vg_gemm_add(&g, WF[k], d.m, 1, d.BV, d.n, 1, WV[k], d.n, d.m, d.n, d.m, 1, 0, 1, 0, k ? s2 : (lead ? s1 : 0.0));       // (dimension 1: replicated)
    }
    VG_HIP(vg_gemm_launch(&g, st));
    vg_gemm_init(&g);
    const double rs = std::sqrt(s1 * s2);
    if (pts[0]) {
        const int ip = vg_gemm_add(&g, T1, m2, 1, c->St, n1, 1, WV[0], (int)n1, (int)m1, (int)n1, (int)m2, 1, 0, c->st_slabs, 2L * m2 * n1, rs, 1);
        (void)ip;
    }
    if (pts[1]) {
        vg_gemm_add(&g, T1, 1, m2, R1, n2, 1, WV[1], (int)n2, (int)m2, (int)n2, (int)m1, 1, 0, 1, 0, rs, 1);
    }
    VG_HIP(vg_gemm_launch(&g, st));
    // 5. Abar = L0^-T W_V and Kbar = L0^-T W_M L0^-1 by substitution, in place (W_M sits in WE: first from the left, then -- on the
    //    transposed view -- from the right).  Not through the explicit inverse: on RBF factors that costs digits (section 2 of
    //    DESIGN.md), and here they are multiplied by 1 / (z_i - z_j).
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
                q[nq++] = VgTrsmSpec{d.L0, d.m, dp, blk, dld, WV[k], d.n, 1, d.n, d.m, 1};
                q[nq++] = VgTrsmSpec{d.L0, d.m, dp, blk, dld, WE[k], d.m, 1, d.m, d.m, 1};          // L^-T W_M
            } else {
                q[nq++] = VgTrsmSpec{d.L0, d.m, dp, blk, dld, WE[k], 1, d.m, d.m, d.m, 1};          // (.) L^-1: L^-T on the transpose
            }
        }
        if ((rc = trsm_batch(q, nq, st))) return rc;
    }
    // 6. contraction with d kappa / d z
    double *o1 = multi ? gzbuf : gz1, *o2 = multi ? gzbuf + m1 : gz2;
    if (multi) VG_HIP(hipMemsetAsync(gzbuf, 0, sizeof(double) * (m1 + m2), st));
    if (pts[0]) VG_HIP(vg_zdot_launch(c->theta, 0, d1.grid, d1.x, (int)m1, n1, WV[0], d1.AD + m1 * n1, WE[0], d1.dK0, o1, st));
    if (pts[1]) VG_HIP(vg_zdot_launch(c->theta, 1, d2.grid, d2.x, (int)m2, n2, WV[1], d2.AD + m2 * n2, WE[1], d2.dK0, o2, st));
    if (multi) {
        if ((rc = vg_allreduce(c, gzbuf, m1 + m2, st))) return rc;
        VG_HIP(hipMemcpyAsync(gz1, gzbuf, sizeof(double) * m1, hipMemcpyDeviceToDevice, st));
        VG_HIP(hipMemcpyAsync(gz2, gzbuf + m1, sizeof(double) * m2, hipMemcpyDeviceToDevice, st));
    }
    { const int wrc_ = vg_comm_wait(c, st); if (wrc_) return wrc_; }
    return VGGP_OK;
}

// Diagnostic builds only (tools/: -DVGGP_DIAG, or the stamp builds -DVG_EIG_RT / -DVG_CHOL_STAMP): the shipped library exports
// exactly the symbols include/vggp.h declares (tests/test_cabi.py checks the dynamic symbol table).
#if defined(VGGP_DIAG) || defined(VG_EIG_RT) || defined(VG_CHOL_STAMP)
extern "C" int vggp_debug_read_out(vggp_ctx* c, double* host8) {
    if (!c || !c->out) return VGGP_EINVAL;
    VG_HIP(hipMemcpy(host8, c->out, 8 * sizeof(double), hipMemcpyDeviceToHost));
    return VGGP_OK;
}

// diagnostic builds only (-DVG_EIG_RT): the eigensolver's global work area of dimension `dim` (which = 1: the Ritz problem's)
extern "C" int vggp_debug_read_gwork(vggp_ctx* c, int dim, int which, void* host, int64_t offset_doubles, int64_t bytes) {
    if (!c || !c->planned || dim < 0 || dim > 1) return VGGP_EINVAL;
    // which = 2: Gw (the matrix the main eigensolver started from), 3: the Ritz matrix Hs, 4: lam0, 5: the Ritz solve's counters (ints)
    if (which == 6) { VG_HIP(hipMemcpy(host, c->tAC + offset_doubles, bytes, hipMemcpyDeviceToHost)); return VGGP_OK; }
    const double* src = which == 5 ? reinterpret_cast<const double*>(c->d[dim].counters2) : which == 2 ? c->d[dim].Gw : which == 3 ? c->d[dim].Hs : which == 4 ? c->d[dim].lam0 : which ? c->d[dim].gwork2 : c->d[dim].gwork;
    if (!src) return VGGP_EINVAL;
    VG_HIP(hipMemcpy(host, src + offset_doubles, bytes, hipMemcpyDeviceToHost));
    return VGGP_OK;
}

extern "C" int vggp_debug_read_misc(vggp_ctx* c, void* host, int64_t bytes) {
    if (!c || !c->misc || (size_t)bytes > c->misc_bytes) return VGGP_EINVAL;
    VG_HIP(hipMemcpy(host, c->misc, bytes, hipMemcpyDeviceToHost));
    return VGGP_OK;
}
#endif

extern "C" const char* vggp_project_kernel_name(void) { return vg_last_project_kernel(); }

extern "C" int vggp_profile(vggp_ctx* c, int enable) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_ENTER_DEVICE(c->device);
    if (enable && !c->ev[0])
        for (int i = 0; i < VG_MAXEV; ++i) VG_HIP(hipEventCreate(&c->ev[i]));
    c->prof = enable != 0;
    return VGGP_OK;
}

extern "C" int vggp_profile_read(vggp_ctx* c, double ms_out[VGGP_NSTAGE], int32_t* steps_out, int reset) {
    if (!c || !ms_out) { vg_set_error("null argument"); return VGGP_EINVAL; }
    for (int i = 0; i < VGGP_NSTAGE; ++i) ms_out[i] = c->prof_ms[i];
    if (steps_out) *steps_out = c->prof_steps;
    if (reset) { for (int i = 0; i < VGGP_NSTAGE; ++i) c->prof_ms[i] = 0.0; c->prof_steps = 0; }
    return VGGP_OK;
}

extern "C" int vggp_sumsq(vggp_ctx* c, const double* y, int64_t n, double* out, void* stream) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_REQUIRE(y && out && n >= 0, "vggp_sumsq: bad argument");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    VG_HIP(vg_sumsq_launch(y, n, c->sumsq_partial, c->sumsq_out, st));
    const int rca = vg_allreduce(c, c->sumsq_out, 1, st);          // total over the ranks of the context (no-op for one rank)
    if (rca) return rca;
    VG_HIP(hipMemcpyAsync(out, c->sumsq_out, sizeof(double), hipMemcpyDeviceToHost, st));
    { const int wrc_ = vg_comm_wait(c, st); if (wrc_) return wrc_; }
    return VGGP_OK;
}
