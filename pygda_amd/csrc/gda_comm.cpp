// RCCL exchange steps of data-parallel training behind the C ABI:
//   gda_allreduce_f32  -- the flat gradient all-reduce (one per step)
//   gda_allgather_f32  -- the all-gather of the MMD sample rows
// on a communicator this library owns, enqueued on the caller's stream (so the collectives
// order with the kernels like any other launch, without a ProcessGroup hop).  The reference has
// no multi-GPU path; these are the RCCL calls its DistributedDataParallel port
// would make.
//
// librccl is bound at RUN time with dlopen, from the path the host hands over -- the copy
// PyTorch-ROCm already loaded -- so exactly one RCCL lives in the process and libgda_hip.so keeps
// no link-time dependency on it (it loads on a box without RCCL; only these entry points then
// return GDA_E_UNSUPPORTED).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "../../include/gda_hip.h"

namespace {

struct Api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok() const { return handle && GetUniqueId && CommInitRank && CommDestroy && AllReduce && AllGather; }
};

Api g_api;
std::mutex g_mu;

// RCCL failures are reported as GDA_E_RCCL - code (status strings: gda_status_string)
int rccl_status(ncclResult_t r) { return r == ncclSuccess ? GDA_OK : GDA_E_RCCL - (int)r; }

}  // namespace

extern "C" int gda_rccl_load(const char* path) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_api.ok()) return GDA_OK;
    void* h = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return GDA_E_UNSUPPORTED;
    Api a;
    a.handle = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    if (!a.ok()) { dlclose(h); return GDA_E_UNSUPPORTED; }
    g_api = a;
    return GDA_OK;
}

extern "C" int gda_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out) return GDA_E_NULL;
    if (bytes < NCCL_UNIQUE_ID_BYTES) return GDA_E_SIZE;
    if (!g_api.ok()) return GDA_E_UNSUPPORTED;
    ncclUniqueId id;
    const int st = rccl_status(g_api.GetUniqueId(&id));
    if (st == GDA_OK) std::memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return st;
}

extern "C" int gda_comm_init_rank(const void* id, size_t bytes, int nranks, int rank, gda_comm_t* comm_out) {
    if (!id || !comm_out) return GDA_E_NULL;
    if (bytes < NCCL_UNIQUE_ID_BYTES || nranks < 1 || rank < 0 || rank >= nranks) return GDA_E_SIZE;
    if (!g_api.ok()) return GDA_E_UNSUPPORTED;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    const int st = rccl_status(g_api.CommInitRank(&c, nranks, uid, rank));
    if (st == GDA_OK) *comm_out = (gda_comm_t)c;
    return st;
}

extern "C" int gda_comm_destroy(gda_comm_t comm) {
    if (!comm) return GDA_OK;
    if (!g_api.ok()) return GDA_E_UNSUPPORTED;
    return rccl_status(g_api.CommDestroy((ncclComm_t)comm));
}

extern "C" int gda_allreduce_f32(float* buf, int64_t count, gda_comm_t comm, gda_stream_t stream) {
    if (count < 0) return GDA_E_SIZE;
    if (count == 0) return GDA_OK;
    if (!buf || !comm) return GDA_E_NULL;
    if (!g_api.ok()) return GDA_E_UNSUPPORTED;
    return rccl_status(g_api.AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)comm,
                                       (hipStream_t)stream));
}

extern "C" int gda_allgather_f32(const float* send, float* recv, int64_t count_per_rank, gda_comm_t comm,
                                 gda_stream_t stream) {
    if (count_per_rank < 0) return GDA_E_SIZE;
    if (count_per_rank == 0) return GDA_OK;
    if (!send || !recv || !comm) return GDA_E_NULL;
    if (!g_api.ok()) return GDA_E_UNSUPPORTED;
    return rccl_status(g_api.AllGather(send, recv, (size_t)count_per_rank, ncclFloat, (ncclComm_t)comm,
                                       (hipStream_t)stream));
}
