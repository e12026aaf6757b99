// Multi-GPU candidate exchange behind the C-ABI (include/comorag_hip.h, "row-shard exchange"): one RCCL all-gather of
// the per-shard top-k candidates, packed as ONE u64 per candidate, and the final merge — so that a host without
// torch.distributed can shard too (north_star: "an RCCL all-gather over xGMI of per-shard top-k candidates").
// The reference has no distributed path; nothing here mirrors reference code.
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: inside a PyTorch process the library must use the RCCL
// (and with it the HIP runtime) PyTorch already loaded — linking /opt/rocm's librccl would pull a second HIP runtime
// into the process (comorag_amd/_lib.py) — and a host that never shards needs no RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "../../include/comorag_hip.h"
#include "cmr_device.h"
#include "cmr_kernels.h"

int cmr_fail(int code, const char* fmt, ...);   // api.hip

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so", "librccl.so.1"}) {           // already in the process (PyTorch's)?
            r.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.h) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.h) break;
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.h) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
        r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) r.h = nullptr;
    });
    return r.h ? &r : nullptr;
}

}  // namespace

struct cmr_comm {
    int world = 1, rank = 0, device = 0;
    ncclComm_t nccl = nullptr;
    u64* send = nullptr;      // [nq*k] packed candidates of this shard
    u64* recv = nullptr;      // [world][nq*k]
    size_t cap = 0;           // candidates the buffers hold per shard
};

// ------------------------------------------------------------------------------------------ kernels
// (score, global row) -> one u64 whose unsigned order is the exported order (the candidate key of cmr_device.h with the
// GLOBAL row in the low word); empty slots (id < 0) -> 0
__global__ __launch_bounds__(256) void pack_candidates_kernel(const int64_t* __restrict__ ids, const float* __restrict__ scores, long long n,
                                                              u64* __restrict__ keys) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t id = ids[i];
    keys[i] = id < 0 ? 0ull : cmr_make_key(scores[i], (unsigned)id);
}

// keys [S][nq][k] -> per query the k best (rank by counting over S*k <= 4096 keys; keys of distinct rows are distinct)
__global__ __launch_bounds__(256) void merge_keys_kernel(const u64* __restrict__ keys, int S, int nq, int k, int64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    u64* pool = reinterpret_cast<u64*>(sm);
    const int q = blockIdx.x, n = S * k;
    for (int i = threadIdx.x; i < n; i += 256) pool[i] = keys[((size_t)(i / k) * nq + q) * k + (i % k)];
    for (int i = threadIdx.x; i < k; i += 256) { out_ids[(size_t)q * k + i] = -1; out_scores[(size_t)q * k + i] = -__builtin_inff(); }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const u64 me = pool[i];
        if (!me) continue;
        int rank = 0;
        for (int o = 0; o < n; ++o) rank += pool[o] > me ? 1 : 0;
        if (rank < k) { out_ids[(size_t)q * k + rank] = (int64_t)cmr_key_row(me); out_scores[(size_t)q * k + rank] = cmr_key_score(me); }
    }
}

static hipError_t launch_pack(const int64_t* ids, const float* scores, long long n, u64* keys, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_candidates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ids, scores, n, keys);
    return hipGetLastError();
}
static hipError_t launch_merge_keys(const u64* keys, int S, int nq, int k, int64_t* out_ids, float* out_scores, hipStream_t s) {
    const size_t lds = (size_t)S * k * 8;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(merge_keys_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // a constant: per function, not per launch
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_keys_kernel, dim3(nq), dim3(256), lds, s, keys, S, nq, k, out_ids, out_scores);
    return hipGetLastError();
}

#define COMM_HIP_TRY(expr)                                                                                             \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return cmr_fail(e_ == hipErrorOutOfMemory ? CMR_ERR_OOM : CMR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------ C-ABI
extern "C" {

int32_t cmr_pack_candidates_dev(const int64_t* ids_dev, const float* scores_dev, int64_t n, uint64_t* keys_dev, void* stream) {
    if (!ids_dev || !scores_dev || !keys_dev) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (n < 0) return cmr_fail(CMR_ERR_INVALID, "n < 0");
    COMM_HIP_TRY(launch_pack(ids_dev, scores_dev, n, (u64*)keys_dev, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_merge_keys_dev(const uint64_t* keys_dev, int32_t n_shards, int32_t nq, int32_t k, int64_t* out_ids_dev, float* out_scores_dev,
                           void* stream) {
    if (!keys_dev || !out_ids_dev || !out_scores_dev) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (n_shards <= 0 || nq <= 0 || k <= 0) return cmr_fail(CMR_ERR_INVALID, "n_shards, nq, k must be > 0");
    if ((long long)n_shards * k > 4096) return cmr_fail(CMR_ERR_UNSUPPORTED, "n_shards * k = %lld > 4096", (long long)n_shards * k);
    COMM_HIP_TRY(launch_merge_keys((const u64*)keys_dev, n_shards, nq, k, out_ids_dev, out_scores_dev, (hipStream_t)stream));
    return CMR_OK;
}

int32_t cmr_comm_unique_id(uint8_t* out_id128) {
    if (!out_id128) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    Rccl* r = rccl();
    if (!r) return cmr_fail(CMR_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded: %s", dlerror() ? dlerror() : "not found");
    static_assert(sizeof(ncclUniqueId) == CMR_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return cmr_fail(CMR_ERR_HIP, "ncclGetUniqueId: %s", r->GetErrorString ? r->GetErrorString(rc) : "error");
    memcpy(out_id128, &id, sizeof(id));
    return CMR_OK;
}

int32_t cmr_comm_create(int32_t world, int32_t rank, const uint8_t* id128, int32_t device_id, cmr_comm_t** out) {
    if (!out || !id128) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (world <= 0 || rank < 0 || rank >= world) return cmr_fail(CMR_ERR_INVALID, "rank %d outside [0, %d)", rank, world);
    Rccl* r = rccl();
    if (!r) return cmr_fail(CMR_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");
    COMM_HIP_TRY(hipSetDevice(device_id));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    cmr_comm* c = new cmr_comm();
    c->world = world; c->rank = rank; c->device = device_id;
    ncclResult_t rc = r->CommInitRank(&c->nccl, world, id, rank);
    if (rc != ncclSuccess) { delete c; return cmr_fail(CMR_ERR_HIP, "ncclCommInitRank: %s", r->GetErrorString ? r->GetErrorString(rc) : "error"); }
    *out = c;
    return CMR_OK;
}

int32_t cmr_comm_info(cmr_comm_t* c, int32_t* world, int32_t* rank, int32_t* rccl_ranks) {
    if (!c) return cmr_fail(CMR_ERR_INVALID, "NULL communicator");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (rccl_ranks) {
        *rccl_ranks = 0;
        Rccl* r = rccl();
        int n = 0;
        if (r && r->CommCount && c->nccl && r->CommCount(c->nccl, &n) == ncclSuccess) *rccl_ranks = n;
    }
    return CMR_OK;
}

int32_t cmr_comm_destroy(cmr_comm_t* c) {
    if (!c) return CMR_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    Rccl* r = rccl();
    if (r && c->nccl) (void)r->CommDestroy(c->nccl);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    delete c;
    return CMR_OK;
}

int32_t cmr_comm_allgather_merge(cmr_comm_t* c, const int64_t* ids_dev, const float* scores_dev, int32_t nq, int32_t k, int64_t* out_ids_dev,
                                 float* out_scores_dev, void* stream) {
    if (!c || !ids_dev || !scores_dev || !out_ids_dev || !out_scores_dev) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    if (nq <= 0 || k <= 0) return cmr_fail(CMR_ERR_INVALID, "nq, k must be > 0");
    if ((long long)c->world * k > 4096) return cmr_fail(CMR_ERR_UNSUPPORTED, "world * k = %lld > 4096", (long long)c->world * k);
    Rccl* r = rccl();
    if (!r) return cmr_fail(CMR_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");
    COMM_HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)nq * k;
    if (n > c->cap) {     // grow (rare; the previous exchange on another stream may still read the old buffers)
        COMM_HIP_TRY(hipDeviceSynchronize());
        if (c->send) COMM_HIP_TRY(hipFree(c->send));
        if (c->recv) COMM_HIP_TRY(hipFree(c->recv));
        c->send = c->recv = nullptr; c->cap = 0;
        COMM_HIP_TRY(hipMalloc((void**)&c->send, n * 8));
        COMM_HIP_TRY(hipMalloc((void**)&c->recv, n * 8 * c->world));
        c->cap = n;
    }
    COMM_HIP_TRY(launch_pack(ids_dev, scores_dev, (long long)n, c->send, s));
    ncclResult_t rc = r->AllGather(c->send, c->recv, n, ncclUint64, c->nccl, s);          // ONE collective: nq*k*8 bytes per rank
    if (rc != ncclSuccess) return cmr_fail(CMR_ERR_HIP, "ncclAllGather: %s", r->GetErrorString ? r->GetErrorString(rc) : "error");
    COMM_HIP_TRY(launch_merge_keys(c->recv, c->world, nq, k, out_ids_dev, out_scores_dev, s));
    return CMR_OK;
}

}  // extern "C"
