// q3_dp.cpp — the data-parallel boundary for hosts that do not carry torch.distributed (the reference's Rust host):
// one process per GPU, ONE collective on the data path — the broadcast of rank 0's weight arena over RCCL/xGMI
//  — plus an all-gather of a few doubles for end-of-job timing. Utterances are independent
// (lib.rs:744-756: per-call KV caches, RNG and masks), so steady state has no inter-GPU traffic.
//
// RCCL is resolved at run time (dlopen): a process that already carries an RCCL (PyTorch bundles its own next to its
// own HIP runtime) keeps using that one, and hosts that never call q3_dp_* never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "q3_internal.h"

namespace {
struct UniqueId { char internal[Q3_DP_ID_BYTES]; };          // ncclUniqueId (rccl.h:40-43)
typedef void* Comm;
typedef int Result;                                          // ncclResult_t, ncclSuccess = 0
enum { kUint8 = 1, kFloat64 = 8 };                           // ncclDataType_t (rccl.h:459-467)
struct Api {
    void* h = nullptr;
    Result (*GetUniqueId)(UniqueId*) = nullptr;
    Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    Result (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(Result) = nullptr;
    bool ok = false;
} api;
std::once_flag api_once;

void load_api() {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // an RCCL the process already has
    for (const char* n : names) if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!api.h) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
    api.Broadcast = (decltype(api.Broadcast))dlsym(api.h, "ncclBroadcast");
    api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Broadcast && api.AllGather && api.GetErrorString;
}
q3_status need_api() {
    std::call_once(api_once, load_api);
    if (!api.ok) {
        const char* why = dlerror();      // one call: dlerror() clears its state, a second call returns NULL
        return q3i_set_err(Q3_RCCL_ERROR, "RCCL not available: librccl.so(.1) could not be loaded (%s)", why ? why : "missing symbols");
    }
    return Q3_OK;
}
}  // namespace

struct q3_dp_comm { Comm comm = nullptr; int rank = 0, world = 1, device = 0; hipStream_t st = nullptr; };

#define DP_NCCL(expr)                                                                                            \
    do {                                                                                                         \
        Result r_ = (expr);                                                                                      \
        if (r_ != 0) return q3i_set_err(Q3_RCCL_ERROR, "%s: %s", #expr, api.GetErrorString(r_));                 \
    } while (0)
#define DP_HIP(expr)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return q3i_set_err(Q3_HIP_ERROR, "%s: %s", #expr, hipGetErrorString(e_));          \
    } while (0)

extern "C" q3_status q3_dp_unique_id(void* id_out) {
    if (!id_out) return q3i_set_err(Q3_INVALID_ARG, "q3_dp_unique_id: null");
    Q3I_CHECK(need_api());
    UniqueId id;
    DP_NCCL(api.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return Q3_OK;
}

extern "C" q3_status q3_dp_init(int rank, int world, const void* id, int device, q3_dp_comm** out) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return q3i_set_err(Q3_INVALID_ARG, "q3_dp_init: bad argument");
    Q3I_CHECK(need_api());
    DP_HIP(hipSetDevice(device));
    auto* c = new q3_dp_comm();
    c->rank = rank; c->world = world; c->device = device;
    UniqueId uid; memcpy(&uid, id, sizeof uid);
    Result r = api.CommInitRank(&c->comm, world, uid, rank);
    if (r != 0) { delete c; return q3i_set_err(Q3_RCCL_ERROR, "ncclCommInitRank(rank %d of %d): %s", rank, world, api.GetErrorString(r)); }
    hipError_t he = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
    if (he != hipSuccess) { api.CommDestroy(c->comm); delete c; return q3i_set_err(Q3_HIP_ERROR, "q3_dp_init: %s", hipGetErrorString(he)); }
    *out = c;
    return Q3_OK;
}

extern "C" void q3_dp_free(q3_dp_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->st) { (void)hipStreamSynchronize(c->st); (void)hipStreamDestroy(c->st); }
    if (c->comm && api.ok) api.CommDestroy(c->comm);
    delete c;
}

extern "C" q3_status q3_dp_info(const q3_dp_comm* c, int* rank, int* world) {
    if (!c) return q3i_set_err(Q3_INVALID_ARG, "q3_dp_info: null");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return Q3_OK;
}

// The arena layout is a pure function of q3_config, so every rank's arena has the same size and offsets: the whole
// checkpoint moves as one byte range (chunked at 1 GiB so that a count never nears 2^31 elements of a wider type
// inside RCCL, and the ring pipelines over the 7 xGMI links).
extern "C" q3_status q3_dp_broadcast_weights(q3_dp_comm* c, q3_model* m, int root) {
    if (!c || !m || root < 0 || root >= c->world) return q3i_set_err(Q3_INVALID_ARG, "q3_dp_broadcast_weights: bad argument");
    void* ptr = nullptr; size_t nbytes = 0;
    Q3I_CHECK(q3_model_arena(m, &ptr, &nbytes));
    {   // world == 1 still goes through RCCL (an in-place self-broadcast): one code path, and single-GPU tests cover it
        DP_HIP(hipSetDevice(c->device));
        const size_t chunk = (size_t)1 << 30;
        for (size_t off = 0; off < nbytes; off += chunk) {
            const size_t n = nbytes - off < chunk ? nbytes - off : chunk;
            DP_NCCL(api.Broadcast((char*)ptr + off, (char*)ptr + off, n, kUint8, root, c->comm, c->st));
        }
        DP_HIP(hipStreamSynchronize(c->st));
    }
    if (c->rank != root) Q3I_CHECK(q3_model_mark_loaded(m));
    return Q3_OK;
}

extern "C" q3_status q3_dp_allgather_f64(q3_dp_comm* c, const double* in_host, int n, double* out_host) {
    if (!c || !in_host || !out_host || n < 1) return q3i_set_err(Q3_INVALID_ARG, "q3_dp_allgather_f64: bad argument");
    DP_HIP(hipSetDevice(c->device));
    double* d = nullptr;
    DP_HIP(hipMalloc((void**)&d, (size_t)n * (c->world + 1) * sizeof(double)));
    q3_status st = Q3_OK;
    hipError_t he = hipMemcpyAsync(d, in_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->st);
    Result r = he == hipSuccess ? api.AllGather(d, d + n, (size_t)n, kFloat64, c->comm, c->st) : 0;
    if (he == hipSuccess && r == 0) he = hipMemcpyAsync(out_host, d + n, (size_t)n * c->world * sizeof(double), hipMemcpyDeviceToHost, c->st);
    if (he == hipSuccess) he = hipStreamSynchronize(c->st);
    if (r != 0) st = q3i_set_err(Q3_RCCL_ERROR, "ncclAllGather: %s", api.GetErrorString(r));
    else if (he != hipSuccess) st = q3i_set_err(Q3_HIP_ERROR, "q3_dp_allgather_f64: %s", hipGetErrorString(he));
    (void)hipFree(d);
    return st;
}
