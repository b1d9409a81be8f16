// avd_comm.cpp -- the one exchange step of the path: reassembling the per-frame records of all ranks.
//
// Frames and clips shard across GPUs with no data-path collective; the only exchange is one all-gather of the 32-byte
// avd_frame_record per sampled frame (a few KB per clip: latency-bound over xGMI).  RCCL is bound at RUN TIME (dlopen of
// librccl.so: the copy already loaded by the process -- e.g. PyTorch's -- or /opt/rocm's), so libavd_hip.so has no
// link-time dependency on it and single-GPU deployments never load it.  The unique id is created by one rank
// (avd_comm_unique_id) and distributed by the caller's own means (a file, an environment variable, torch's store).
#include <dlfcn.h>
#include <cstring>
#include "avd_internal.h"

namespace {

struct UniqueId { char internal[128]; };                       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_id)(UniqueId*);
typedef int (*fn_init)(void**, int, UniqueId, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* so = nullptr;
    fn_get_id get_id = nullptr;
    fn_init init = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
};

Rccl* rccl(std::string& err)
{
    static Rccl r;
    if (r.so) return &r;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.so) break;
    }
    if (!r.so) { err = "librccl.so cannot be loaded (multi-GPU record exchange needs RCCL)"; return nullptr; }
    r.get_id = (fn_get_id)dlsym(r.so, "ncclGetUniqueId");
    r.init = (fn_init)dlsym(r.so, "ncclCommInitRank");
    r.allgather = (fn_allgather)dlsym(r.so, "ncclAllGather");
    r.destroy = (fn_destroy)dlsym(r.so, "ncclCommDestroy");
    r.errstr = (fn_errstr)dlsym(r.so, "ncclGetErrorString");
    if (!r.get_id || !r.init || !r.allgather || !r.destroy) { err = "librccl.so lacks the expected entry points"; dlclose(r.so); r.so = nullptr; return nullptr; }
    return &r;
}

int fail(avd_ctx* ctx, Rccl* r, const char* what, int rc)
{
    ctx->err = std::string(what) + ": " + (r && r->errstr ? r->errstr(rc) : "RCCL error");
    return AVD_ERR_DEVICE;
}

}  // namespace

int comm_unique_id(std::string& err, void* id128)
{
    Rccl* r = rccl(err);
    if (!r) return AVD_ERR_DEVICE;
    UniqueId id;
    const int rc = r->get_id(&id);
    if (rc) { err = std::string("ncclGetUniqueId: ") + (r->errstr ? r->errstr(rc) : "RCCL error"); return AVD_ERR_DEVICE; }
    std::memcpy(id128, id.internal, sizeof(id.internal));
    return AVD_OK;
}

int comm_init(avd_ctx* ctx, int rank, int world, const void* id128)
{
    if (world < 1 || rank < 0 || rank >= world || !id128) { ctx->err = "bad rank / world size"; return AVD_ERR_ARG; }
    Rccl* r = rccl(ctx->err);
    if (!r) return AVD_ERR_DEVICE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) { (void)r->destroy(ctx->comm); ctx->comm = nullptr; }
    UniqueId id;
    std::memcpy(id.internal, id128, sizeof(id.internal));
    const int rc = r->init(&ctx->comm, world, id, rank);
    if (rc) { ctx->comm = nullptr; return fail(ctx, r, "ncclCommInitRank", rc); }
    ctx->comm_rank = rank; ctx->comm_world = world;
    return AVD_OK;
}

void comm_destroy(avd_ctx* ctx)
{
    std::string ignore;
    if (ctx->comm) { if (Rccl* r = rccl(ignore)) (void)r->destroy(ctx->comm); ctx->comm = nullptr; }
    if (ctx->d_comm) { (void)hipFree(ctx->d_comm); ctx->d_comm = nullptr; ctx->comm_bytes = 0; }
}

// every rank contributes `count` records; all = world * count records in rank order (host pointers)
int comm_allgather_records(avd_ctx* ctx, const avd_frame_record* local, int count, avd_frame_record* all)
{
    if (count < 0 || (count > 0 && (!local || !all))) { ctx->err = "bad arguments"; return AVD_ERR_ARG; }
    if (!ctx->comm) { ctx->err = "avd_comm_init has not been called on this context"; return AVD_ERR_ARG; }
    if (count == 0) return AVD_OK;
    Rccl* r = rccl(ctx->err);
    if (!r) return AVD_ERR_DEVICE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t mine = sizeof(avd_frame_record) * (size_t)count, total = mine * (size_t)ctx->comm_world;
    if (ctx->comm_bytes < mine + total) {
        if (ctx->d_comm) { (void)hipFree(ctx->d_comm); ctx->d_comm = nullptr; }
        ctx->comm_bytes = 0;
        if (hipMalloc(&ctx->d_comm, mine + total) != hipSuccess) { ctx->err = "hipMalloc (record exchange)"; return AVD_ERR_NOMEM; }
        ctx->comm_bytes = mine + total;
    }
    char* d_send = static_cast<char*>(ctx->d_comm);
    char* d_recv = d_send + mine;
    HIP_TRY(ctx, hipMemcpyAsync(d_send, local, mine, hipMemcpyHostToDevice, ctx->stream));
    const int rc = r->allgather(d_send, d_recv, mine, /*ncclUint8*/ 1, ctx->comm, ctx->stream);
    if (rc) return fail(ctx, r, "ncclAllGather", rc);
    HIP_TRY(ctx, hipMemcpyAsync(all, d_recv, total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AVD_OK;
}

// The same exchange straight from the device: the records of the context's LAST analysis call (ws.d_rec, written by
// k_records on this stream) are the send buffer, the collective is enqueued behind them on the same stream, and ONE copy
// brings the gathered block to the host.  No host round trip and no extra synchronisation before the collective.
int comm_allgather_last_records(avd_ctx* ctx, int count, avd_frame_record* all)
{
    if (count < 0 || (count > 0 && !all)) { ctx->err = "bad arguments"; return AVD_ERR_ARG; }
    if (!ctx->comm) { ctx->err = "avd_comm_init has not been called on this context"; return AVD_ERR_ARG; }
    if (count > ctx->rec_n || !ctx->ws.d_rec) { ctx->err = "the last call on this context left fewer records than asked for (only avd_analyze_* calls leave records)"; return AVD_ERR_ARG; }
    if (count == 0) return AVD_OK;
    Rccl* r = rccl(ctx->err);
    if (!r) return AVD_ERR_DEVICE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t mine = sizeof(avd_frame_record) * (size_t)count, total = mine * (size_t)ctx->comm_world;
    if (ctx->comm_bytes < mine + total) {
        if (ctx->d_comm) { (void)hipFree(ctx->d_comm); ctx->d_comm = nullptr; }
        ctx->comm_bytes = 0;
        if (hipMalloc(&ctx->d_comm, mine + total) != hipSuccess) { ctx->err = "hipMalloc (record exchange)"; return AVD_ERR_NOMEM; }
        ctx->comm_bytes = mine + total;
    }
    char* d_recv = static_cast<char*>(ctx->d_comm) + mine;
    const int rc = r->allgather(ctx->ws.d_rec, d_recv, mine, /*ncclUint8*/ 1, ctx->comm, ctx->stream);
    if (rc) return fail(ctx, r, "ncclAllGather", rc);
    HIP_TRY(ctx, hipMemcpyAsync(all, d_recv, total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AVD_OK;
}

