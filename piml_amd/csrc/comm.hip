// RCCL side of the C ABI: the per-step exchange of agent-block sharding as plain calls on
// an ncclComm_t -- all-gather of the owners' (p, v, a) records, reduce-scatter (sum) of the partial d/d(state), and
// an all-reduce for the flat [state gradient | weight gradients] bucket.
//
// The reference has no multi-process path (only nn.DataParallel, src/models/simulators.py:64-67).  RCCL is bound at
// RUN TIME (dlopen): libpiml_hip.so carries no link-time dependency on it, so it still loads on hosts without a GPU,
// and inside a PyTorch process it binds to the librccl.so torch already loaded (never a second copy).
#include <dlfcn.h>
#include <string.h>

#include "common.hpp"
#include "../../include/piml_hip.h"

namespace piml {

struct RcclApi {
    // signatures of rccl.h (ncclResult_t / ncclComm_t / ncclDataType_t / ncclRedOp_t as int / void* / int / int)
    int (*GetUniqueId)(void* id);
    int (*CommInitRank)(void** comm, int nranks, piml_comm_id id, int rank);
    int (*CommDestroy)(void* comm);
    int (*AllGather)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t s);
    int (*ReduceScatter)(const void* send, void* recv, size_t recvcount, int dtype, int op, void* comm, hipStream_t s);
    int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s);
    bool ok;
};

constexpr int kNcclFloat32 = 7, kNcclSum = 0;          // rccl.h: ncclFloat32 = 7, ncclSum = 0
constexpr int kErrNoRccl = 801;                        // hipErrorNotSupported
constexpr int kErrRcclBase = 10000;                    // + ncclResult_t

static const RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a = {};
        void* h = nullptr;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {                  // the copy already in the process (torch's), if any
            h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        for (int i = 0; !h && i < 2; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
        a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.ReduceScatter && a.AllReduce;
        return a;
    }();
    return api;
}

static int rc(int nccl_result) { return nccl_result == 0 ? 0 : kErrRcclBase + nccl_result; }

}  // namespace piml

using namespace piml;

PIML_API int piml_comm_available(void) { return rccl().ok ? 1 : 0; }

PIML_API int piml_comm_unique_id(piml_comm_id* id) {
    if (!id) return hipErrorInvalidValue;
    if (!rccl().ok) return kErrNoRccl;
    return rc(rccl().GetUniqueId(id));
}

PIML_API int piml_comm_init(void** comm, int world, int rank, const piml_comm_id* id) {
    if (!comm || !id || world < 1 || rank < 0 || rank >= world) return hipErrorInvalidValue;
    if (!rccl().ok) return kErrNoRccl;
    return rc(rccl().CommInitRank(comm, world, *id, rank));
}

PIML_API int piml_comm_destroy(void* comm) {
    if (!comm) return hipSuccess;
    if (!rccl().ok) return kErrNoRccl;
    return rc(rccl().CommDestroy(comm));
}

PIML_API int piml_allgather_state(void* comm, const float* own, size_t floats_per_rank, float* full, void* stream) {
    if (!comm || !own || !full) return hipErrorInvalidValue;
    if (!rccl().ok) return kErrNoRccl;
    return rc(rccl().AllGather(own, full, floats_per_rank, kNcclFloat32, comm, as_stream(stream)));
}

PIML_API int piml_reducescatter_grad(void* comm, const float* full, float* own, size_t floats_per_rank, void* stream) {
    if (!comm || !own || !full) return hipErrorInvalidValue;
    if (!rccl().ok) return kErrNoRccl;
    return rc(rccl().ReduceScatter(full, own, floats_per_rank, kNcclFloat32, kNcclSum, comm, as_stream(stream)));
}

PIML_API int piml_allreduce_sum(void* comm, float* buf, size_t count, void* stream) {
    if (!comm || !buf) return hipErrorInvalidValue;
    if (!rccl().ok) return kErrNoRccl;
    return rc(rccl().AllReduce(buf, buf, count, kNcclFloat32, kNcclSum, comm, as_stream(stream)));
}
