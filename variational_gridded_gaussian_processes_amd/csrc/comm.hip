// The context's collective: ONE sum all-reduce of the packed payload per ELBO step, owned by the
// library (include/vggp.h: vggp_create(..., n_ranks, rank, unique_id), vggp_allreduce).
//
// Transport 1 -- RCCL over xGMI: the communicator is created from the unique id rank 0 generated (vggp_unique_id) and the
// all-reduce is enqueued on the step's stream between the partials graph and the finish graph: no host synchronisation,
// one host sync per step.  librccl.so.1 is dlopen'ed when the first communicator is created, so the library loads (and the
// C-ABI symbol tests run) on hosts without RCCL and never carries a second copy next to PyTorch's.
// Transport 2 -- host callback (vggp_set_allreduce): the payload is staged through pinned host memory and handed to the
// caller's function.  This is the rehearsal path of the tests (gloo, several ranks sharing the one GPU of the test box,
// where RCCL refuses duplicate devices) and a seam for other transports (MPI).
//
// The reference has no multi-device path: this file replaces nothing, it implements the exchange
// step that the sum structure of Kuf Kuf^T = sum over grid rows (kronecker_structure.py:249-278) admits.
#include "ctx.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct VgRccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // optional (present in every RCCL this was built against)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;   // optional
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static VgRccl g_rccl;

static int rccl_load() {
    if (g_rccl.lib) return VGGP_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { vg_set_error("RCCL is not loadable (librccl.so.1): %s", dlerror()); return VGGP_ERCCL; }
    VgRccl r;
    r.lib = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
    r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
        vg_set_error("librccl.so.1 lacks an expected symbol");
        return VGGP_ERCCL;
    }
    g_rccl = r;
    return VGGP_OK;
}

#define VG_NCCL(call)                                                                           \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            vg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
            return VGGP_ERCCL;                                                                  \
        }                                                                                       \
    } while (0)

static_assert(VGGP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int vggp_unique_id(void* out) {
    if (!out) { vg_set_error("vggp_unique_id: null output"); return VGGP_EINVAL; }
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    VG_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return VGGP_OK;
}

// called by vggp_create (the context's device is current)
int vg_comm_init(vggp_ctx* c, int n_ranks, int rank, const void* unique_id) {
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->comm = nullptr;
    if (!unique_id) return VGGP_OK;       // single rank, or the caller installs a callback transport
    // (a unique id with n_ranks = 1 creates a communicator of size one: the step then runs the multi-rank sequence --
    //  partials, all-reduce on the stream, finish -- which is how the RCCL path is exercised on a one-GPU box)
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    VG_NCCL(g_rccl.CommInitRank(&comm, n_ranks, id, rank));
    c->comm = comm;
    return VGGP_OK;
}

// Give up on the communicator: ncclCommAbort raises the abort flag the collective kernels poll, so an all-reduce that waits
// for a peer that will never arrive leaves the GPU instead of spinning for ever.  The context stays usable for nothing
// multi-rank afterwards (every later step returns VGGP_ERCCL).
void vg_comm_abort(vggp_ctx* c) {
    if (c->comm) {
        if (g_rccl.CommAbort) (void)g_rccl.CommAbort(reinterpret_cast<ncclComm_t>(c->comm));
        else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(c->comm));
        c->comm = nullptr;
    }
    c->comm_dead = true;
}

// Host side of the end of a multi-rank step: wait for the stream WITHOUT trusting the peers.  A plain hipStreamSynchronize
// blocks for ever when a peer died before its ncclAllReduce; here the stream is polled together with the communicator's
// asynchronous error state, and after VGGP_COMM_TIMEOUT_S seconds (default 120) without completion the communicator is
// aborted and the step returns VGGP_ERCCL.  Single-rank contexts and the callback transport use the plain synchronisation.
// seq / want: when given, completion is the arrival of the step's sequence number in the pinned result block (the last word the
// step's last kernel writes) instead of the stream running empty -- the same test the single-rank step polls (api.hip, vg_wait_step).
int vg_comm_wait(vggp_ctx* c, hipStream_t st, const volatile double* seq, double want) {
    if (!c->comm || !g_rccl.CommGetAsyncError) { VG_HIP(hipStreamSynchronize(st)); return VGGP_OK; }
    static const double limit_s = [] { const char* e = getenv("VGGP_COMM_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 120.0; }();
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long spin = 0;; ++spin) {
        if (seq) {
            if (*seq == want) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return VGGP_OK; }
            __builtin_ia32_pause();
        } else {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return VGGP_OK;
            if (q != hipErrorNotReady) { vg_set_error("hipStreamQuery -> %s", hipGetErrorString(q)); vg_comm_abort(c); return VGGP_EHIP; }
        }
        if ((spin & (seq ? 16383 : 1023)) == (seq ? 16383 : 1023)) {            // every so many polls: the communicator's health and the clock
            if (seq) {                           // a fault on the stream would leave the word unwritten for ever
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) { if (*seq == want) return VGGP_OK; vg_set_error("the step's stream ran empty without its result block"); return VGGP_ESTATE; }
                if (q != hipErrorNotReady) { vg_set_error("hipStreamQuery -> %s", hipGetErrorString(q)); vg_comm_abort(c); return VGGP_EHIP; }
            }
            ncclResult_t ae = ncclSuccess;
            if (g_rccl.CommGetAsyncError(reinterpret_cast<ncclComm_t>(c->comm), &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) {
                vg_set_error("RCCL reports an asynchronous error on rank %d / %d: %s (communicator aborted)", c->rank, c->n_ranks,
                             g_rccl.GetErrorString(ae));
                vg_comm_abort(c);
                (void)hipStreamSynchronize(st);
                return VGGP_ERCCL;
            }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double el = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            if (el > limit_s) {
                vg_set_error("rank %d / %d: the step's all-reduce did not complete within %.0f s (a peer rank is gone?); communicator "
                             "aborted", c->rank, c->n_ranks, limit_s);
                vg_comm_abort(c);
                (void)hipStreamSynchronize(st);
                return VGGP_ERCCL;
            }
            if (el > 0.002) sched_yield();      // a step takes ~0.3 ms: past 2 ms something is slow, stop burning the core
        }
    }
}

void vg_comm_destroy(vggp_ctx* c) {
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->h_stage_count = 0;
}

// Sum all-reduce of `count` doubles at the DEVICE address `buf`, in place, ordered on `st`.  RCCL: enqueued, no host
// synchronisation.  Callback: D2H into pinned memory, stream sync, the caller's function, H2D.
int vg_allreduce(vggp_ctx* c, double* buf, long count, hipStream_t st) {
    if (count <= 0 || (c->n_ranks <= 1 && !c->comm && !c->cb)) return VGGP_OK;
    if (c->cb) {
        if (c->h_stage_count < count) {
            if (c->h_stage) { VG_HIP(hipHostFree(c->h_stage)); c->h_stage = nullptr; c->h_stage_count = 0; }
            VG_HIP(hipHostMalloc((void**)&c->h_stage, sizeof(double) * count, hipHostMallocDefault));
            c->h_stage_count = count;
        }
        VG_HIP(hipMemcpyAsync(c->h_stage, buf, sizeof(double) * count, hipMemcpyDeviceToHost, st));
        VG_HIP(hipStreamSynchronize(st));
        const int rc = c->cb(c->cb_user, c->h_stage, (int64_t)count);
        if (rc) { vg_set_error("the all-reduce callback returned %d", rc); return VGGP_ERCCL; }
        VG_HIP(hipMemcpyAsync(buf, c->h_stage, sizeof(double) * count, hipMemcpyHostToDevice, st));
        return VGGP_OK;
    }
    if (c->comm_dead) { vg_set_error("rank %d / %d: the communicator was aborted after an earlier failure", c->rank, c->n_ranks); return VGGP_ERCCL; }
    if (!c->comm) {
        vg_set_error("context of rank %d / %d has no transport: pass the unique id to vggp_create or call vggp_set_allreduce",
                     c->rank, c->n_ranks);
        return VGGP_ESTATE;
    }
    VG_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(c->comm), st));
    return VGGP_OK;
}

extern "C" int vggp_set_allreduce(vggp_ctx* c, vggp_allreduce_fn fn, void* user) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    c->cb = fn;
    c->cb_user = user;
    return VGGP_OK;
}

extern "C" int vggp_allreduce(vggp_ctx* c, double* buf, int64_t count, void* stream) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    VG_REQUIRE(buf && count >= 0, "vggp_allreduce: bad argument");
    VG_ENTER_DEVICE(c->device);
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const int rc = vg_allreduce(c, buf, (long)count, st);
    if (rc) return rc;
    { const int wrc_ = vg_comm_wait(c, st); if (wrc_) return wrc_; }      // (polls the communicator: a dead peer is an error, not a hang)
    return VGGP_OK;
}

extern "C" int vggp_comm_info(const vggp_ctx* c, int* n_ranks, int* rank, int* transport) {
    if (!c) { vg_set_error("null context"); return VGGP_EINVAL; }
    if (n_ranks) *n_ranks = c->n_ranks;
    if (rank) *rank = c->rank;
    if (transport) *transport = c->cb ? 2 : (c->comm ? 1 : 0);      // 0 none, 1 RCCL, 2 host callback
    return VGGP_OK;
}
