// comm_rccl.hip -- the one collective of the path behind the C ABI: the sum of the
// flow-histogram counters hist[50] | hist2d[36*50] | histsum | histsum2d[36] = RC_HIST_WORDS int32 (7548 B)
// over the ranks of a node, RCCL over xGMI, so that a C++ host like the reference's (one process per GPU, each
// running ripcurrents.cpp:194-511 on its own video segment) derives the GLOBAL thresholds of
// ripcurrents.cpp:319-366 from the same integers on every rank.  No other data crosses GPUs.
//
// librccl is opened at run time (rcflow_comm_unique_id / rcflow_comm_init), so librcflow.so itself has no
// link-time dependency on it and single-GPU hosts never load it.  A world of one rank needs no RCCL at all:
// with a null id the collective is then the identity (with an id a one-rank RCCL communicator is built like any
// other: what the one-GPU test box can exercise of the real thing).  The collective runs on its own stream: it is ordered after the slot's
// histogram kernels by an event, and the slot's stream only waits for it where the caller says so
// (rcflow_allreduce_hist_join) -- the 7.5 KB all-reduce is latency-bound and hides beside the next batch.

#include <dlfcn.h>
#include <cstring>

#include <rccl/rccl.h>

#include "rc_host.h"

#define RC_COUNT_UNIT 2048ll
#define RC_COMM_WORDS (RC_HIST_WORDS + 1)

struct RcComm {
    int rank = 0, world = 1;
    void* lib = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    // RC_HIST_WORDS counters + one word every rank fills with the pixels it has counted, in units of RC_COUNT_UNIT,
    // rounded up: the reduced word bounds the summed histsum on every rank alike (rcflow_allreduce_hist_status)
    int32_t* staging = nullptr;       // the rank's counters as sent
    int32_t* reduced = nullptr;       // the reduced words + the count word, as the collective leaves them
    int32_t* result = nullptr;        // context-owned output when the caller passes none (RC_HIST_WORDS)
    int pending = 0;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static void* open_rccl() {
    // the soname first: a host framework that already loaded RCCL (PyTorch ships its own copy) shares it
    static const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
        if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) return h;
    rc_set_error("librccl not found: %s", dlerror());
    return nullptr;
}

extern "C" int rcflow_comm_unique_id(void* id_out) {
    if (!id_out) return RC_EINVAL;
    void* lib = open_rccl();
    if (!lib) return RC_ECOMM;
    auto get = (ncclResult_t(*)(ncclUniqueId*))dlsym(lib, "ncclGetUniqueId");
    if (!get) { rc_set_error("ncclGetUniqueId not exported by librccl"); return RC_ECOMM; }
    ncclUniqueId id;
    ncclResult_t r = get(&id);
    if (r != ncclSuccess) { rc_set_error("ncclGetUniqueId failed (%d)", (int)r); return RC_ECOMM; }
    static_assert(sizeof(id) == RC_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return RC_OK;
}

extern "C" int rcflow_comm_destroy(rc_ctx* ctx) {
    if (!ctx) return RC_EINVAL;
    RcComm* c = (RcComm*)ctx->comm;
    if (!c) return RC_OK;
    (void)hipSetDevice(ctx->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && c->CommDestroy) (void)c->CommDestroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->staging) (void)hipFree(c->staging);
    if (c->result) (void)hipFree(c->result);
    if (c->reduced) (void)hipFree(c->reduced);
    delete c;
    ctx->comm = nullptr;
    return RC_OK;
}

extern "C" int rcflow_comm_init(rc_ctx* ctx, const void* unique_id, int rank, int world) {
    if (!ctx || world < 1 || rank < 0 || rank >= world || (world > 1 && !unique_id)) {
        rc_set_error("rcflow_comm_init: bad arguments (rank %d of %d)", rank, world);
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    rcflow_comm_destroy(ctx);
    RcComm* c = new RcComm();
    c->rank = rank; c->world = world;
    ctx->comm = c;
    auto fail = [&](int code) { rcflow_comm_destroy(ctx); return code; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess ||
        hipMalloc(&c->staging, RC_COMM_WORDS * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&c->reduced, RC_COMM_WORDS * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&c->result, RC_HIST_WORDS * sizeof(int32_t)) != hipSuccess) {
        rc_set_error("rcflow_comm_init: stream / event / buffer creation failed");
        return fail(RC_EHIP);
    }
    if (world == 1 && !unique_id) return RC_OK;         // identity: no RCCL involved
    if (!(c->lib = open_rccl())) return fail(RC_ECOMM);
    auto init = (ncclResult_t(*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(c->lib, "ncclCommInitRank");
    c->AllReduce = (decltype(c->AllReduce))dlsym(c->lib, "ncclAllReduce");
    c->CommDestroy = (decltype(c->CommDestroy))dlsym(c->lib, "ncclCommDestroy");
    c->GetErrorString = (decltype(c->GetErrorString))dlsym(c->lib, "ncclGetErrorString");
    if (!init || !c->AllReduce || !c->CommDestroy) { rc_set_error("librccl lacks an entry point"); return fail(RC_ECOMM); }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = init(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        rc_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, c->GetErrorString ? c->GetErrorString(r) : "?");
        c->comm = nullptr;
        return fail(RC_ECOMM);
    }
    return RC_OK;
}

extern "C" int rcflow_comm_rank(rc_ctx* ctx, int* rank, int* world) {
    if (!ctx || !ctx->comm) { rc_set_error("collective layer not initialised (rcflow_comm_init)"); return RC_ECOMM; }
    RcComm* c = (RcComm*)ctx->comm;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return RC_OK;
}

extern "C" int rcflow_allreduce_hist(rc_ctx* ctx, int stream, int32_t* d_words_out) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    RcComm* c = (RcComm*)ctx->comm;
    if (!c) { rc_set_error("collective layer not initialised (rcflow_comm_init)"); return RC_ECOMM; }
    if (!s->an.hist.p) { rc_set_error("no histogram state: call rcflow_analysis_reset first"); return RC_ESTATE; }
    // No rank may decide on its own to skip the collective (the others would wait in it for ever): whether the summed
    // histsum fits the int32 payload is decided AFTER the reduction, from a reduced word that is the same on every
    // rank (rcflow_allreduce_hist_status).  Every rank must call this function the same number of times.
    RC_HIP(hipSetDevice(ctx->device));
    // snapshot of the counters in the slot's stream order; the collective stream picks it up from there.
    // (a collective still in flight reads the previous snapshot: order the copy after it)
    if (c->pending) RC_HIP(hipStreamWaitEvent(s->cur, c->done, 0));
    RC_HIP(hipMemcpyAsync(c->staging, s->an.hist.p, RC_HIST_WORDS * sizeof(int32_t), hipMemcpyDeviceToDevice, s->cur));
    const int units = (int)((s->an.hist_added + RC_COUNT_UNIT - 1) / RC_COUNT_UNIT);       // <= 2^20 (hist_added <= INT32_MAX)
    RC_HIP(hipMemsetD32Async((hipDeviceptr_t)(c->staging + RC_HIST_WORDS), units, 1, s->cur));
    RC_HIP(hipEventRecord(c->ready, s->cur));
    RC_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    if (!c->comm) {                                     // world of one without RCCL
        RC_HIP(hipMemcpyAsync(c->reduced, c->staging, RC_COMM_WORDS * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    } else {
        ncclResult_t r = c->AllReduce(c->staging, c->reduced, RC_COMM_WORDS, ncclInt32, ncclSum, c->comm, c->stream);
        if (r != ncclSuccess) {
            rc_set_error("ncclAllReduce failed: %s", c->GetErrorString ? c->GetErrorString(r) : "?");
            return RC_ECOMM;
        }
    }
    RC_HIP(hipMemcpyAsync(d_words_out ? d_words_out : c->result, c->reduced, RC_HIST_WORDS * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    RC_HIP(hipEventRecord(c->done, c->stream));
    c->pending = 1;
    return RC_OK;
}

extern "C" int rcflow_allreduce_hist_join(rc_ctx* ctx, int stream) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    RcComm* c = (RcComm*)ctx->comm;
    if (!c) { rc_set_error("collective layer not initialised (rcflow_comm_init)"); return RC_ECOMM; }
    if (!c->pending) return RC_OK;
    RC_HIP(hipSetDevice(ctx->device));
    RC_HIP(hipStreamWaitEvent(s->cur, c->done, 0));
    return RC_OK;
}

// The verdict on the collective started last, identical on every rank because it is computed from a reduced word:
// waits for the collective (host wait), then RC_ESTATE when the ranks together counted more pixels than an int32
// histsum can hold (an upper bound: every rank rounds its count up to RC_COUNT_UNIT), RC_OK otherwise.
extern "C" int rcflow_allreduce_hist_status(rc_ctx* ctx, long long* pixels_counted) {
    if (!ctx) return RC_EINVAL;
    RcComm* c = (RcComm*)ctx->comm;
    if (!c) { rc_set_error("collective layer not initialised (rcflow_comm_init)"); return RC_ECOMM; }
    if (pixels_counted) *pixels_counted = 0;
    if (!c->pending) return RC_OK;
    RC_HIP(hipSetDevice(ctx->device));
    RC_HIP(hipEventSynchronize(c->done));
    int32_t units = 0;
    RC_HIP(hipMemcpy(&units, c->reduced + RC_HIST_WORDS, sizeof(units), hipMemcpyDeviceToHost));
    const long long px = (long long)units * RC_COUNT_UNIT;
    if (pixels_counted) *pixels_counted = px;
    if (px > 0x7fffffffll) {
        rc_set_error("the %d ranks counted up to %lld pixels: the summed flow histogram may have wrapped int32 -- "
                     "all-reduce per shorter segment (rcflow_histogram_reset_dev)", c->world, px);
        return RC_ESTATE;
    }
    return RC_OK;
}

extern "C" int rcflow_allreduce_hist_result(rc_ctx* ctx, int32_t** d_words) {
    if (!ctx || !d_words) return RC_EINVAL;
    RcComm* c = (RcComm*)ctx->comm;
    if (!c) { rc_set_error("collective layer not initialised (rcflow_comm_init)"); return RC_ECOMM; }
    *d_words = c->result;
    return RC_OK;
}
