// P2P-store exchange of agent-block sharding: every rank writes its block of (p, v, a) records
// straight into every peer's receive buffer -- stores over xGMI between the GPUs of a node, plain device stores when two
// ranks share a GPU -- and raises one flag word per (receiver, sender); no collective library in the data path.  The
// reference has nothing here (its only multi-GPU mechanism is nn.DataParallel, src/models/simulators.py:64-67); the
// default exchange of this package is the RCCL all-gather of comm.hip, this is the latency-bound alternative SURVEY 8e
// names for 49 KB messages.
//
// Protocol (per rank: `recv` = [parity 2][sender world][floats_per_rank] floats, `flags` = [parity 2][sender world] dwords,
// zeroed once; both exported to the peers with hipIpc handles):
//   step `seq` (1, 2, ...), parity = seq & 1:
//     block r of rank s copies s's rows into recv_r[parity][s] (16-byte stores), every storing wave drains its stores,
//     the block's barrier, one lane: system-scope release, then flags_r[parity][s] = seq (relaxed system-scope store);
//     block 0 then polls its OWN flags[parity][0 .. world) (one lane per sender, relaxed system-scope loads, s_sleep between
//     polls, bounded) until all equal seq, one system-scope acquire, done: kernels behind it on the stream read recv[parity].
//   Two parities: a rank that is a step ahead writes the other half; it cannot be two steps ahead of a peer, because passing
//   step seq + 1 needs that peer's block of step seq + 1, which the peer writes only after it has finished reading step seq.
//   A poll that runs out (`spin_limit` rounds of ~4 us) sets status[0] = 1 and returns: a lost peer is an error code, not a hang.
#include <string.h>

#include "common.hpp"
#include "../../include/piml_hip.h"

namespace piml {

constexpr int kP2pMaxWorld = 8;

struct P2pArgs {
    const float* own;
    float* recv[kP2pMaxWorld];
    unsigned* flags[kP2pMaxWorld];
    unsigned long long n4;          // float4 per rank
    int rank, world, parity;
    unsigned seq, spin_limit;
    int* status;
};

__global__ __launch_bounds__(256) void p2p_allgather_kernel(P2pArgs A) {
    const int r = blockIdx.x;                              // the receiver this block serves
    const float4* src = reinterpret_cast<const float4*>(A.own);
    float4* dst = reinterpret_cast<float4*>(A.recv[r]) + ((size_t)A.parity * A.world + A.rank) * A.n4;
    for (unsigned long long e = threadIdx.x; e < A.n4; e += 256) dst[e] = src[e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores ...
    __syncthreads();                                       // ... before the one lane that signals for all of them
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope: the payload is visible to the peer before the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(A.flags[r] + A.parity * A.world + A.rank, A.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (r != 0) return;
    // block 0: wait for this rank's own flags, one lane per sender
    bool ok = true;
    if ((int)threadIdx.x < A.world) {
        const unsigned* f = A.flags[A.rank] + A.parity * A.world + threadIdx.x;
        unsigned spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != A.seq) {
            if (++spins > A.spin_limit) { ok = false; break; }
            __builtin_amdgcn_s_sleep(127);
        }
    }
    if (!ok) *A.status = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // what the senders released is what the next kernels read
}

}  // namespace piml

using namespace piml;

PIML_API int piml_p2p_alloc(size_t bytes, void** devptr) {
    if (!devptr || bytes == 0) return hipErrorInvalidValue;
    if (hipError_t e = hipMalloc(devptr, bytes)) return e;
    return hipMemset(*devptr, 0, bytes);
}

PIML_API int piml_p2p_free(void* devptr) { return devptr ? (int)hipFree(devptr) : (int)hipSuccess; }

PIML_API int piml_p2p_export(void* devptr, piml_ipc_handle* out) {
    if (!devptr || !out) return hipErrorInvalidValue;
    static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(piml_ipc_handle), "the handle fits");
    hipIpcMemHandle_t h;
    if (hipError_t e = hipIpcGetMemHandle(&h, devptr)) return e;
    memset(out, 0, sizeof(*out));
    memcpy(out, &h, sizeof(h));
    return hipSuccess;
}

PIML_API int piml_p2p_open(const piml_ipc_handle* in, void** devptr) {
    if (!in || !devptr) return hipErrorInvalidValue;
    hipIpcMemHandle_t h;
    memcpy(&h, in, sizeof(h));
    return hipIpcOpenMemHandle(devptr, h, hipIpcMemLazyEnablePeerAccess);
}

PIML_API int piml_p2p_close(void* devptr) { return devptr ? (int)hipIpcCloseMemHandle(devptr) : (int)hipSuccess; }

PIML_API int piml_p2p_copy(void* dst, const void* src, size_t bytes, void* stream) {
    if (!dst || !src) return hipErrorInvalidValue;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream));
}

PIML_API int piml_allgather_state_p2p(const float* own, size_t floats_per_rank, int rank, int world, float* const* peer_recv,
                                      unsigned* const* peer_flags, unsigned seq, unsigned spin_limit, int* status, void* stream) {
    if (!own || !peer_recv || !peer_flags || !status || world < 1 || world > kP2pMaxWorld || rank < 0 || rank >= world ||
        floats_per_rank == 0 || floats_per_rank % 4 != 0 || seq == 0)
        return hipErrorInvalidValue;
    P2pArgs A = {};
    A.own = own;
    for (int r = 0; r < world; ++r) {
        if (!peer_recv[r] || !peer_flags[r]) return hipErrorInvalidValue;
        A.recv[r] = peer_recv[r];
        A.flags[r] = peer_flags[r];
    }
    A.n4 = floats_per_rank / 4;
    A.rank = rank; A.world = world; A.parity = (int)(seq & 1u);
    A.seq = seq; A.spin_limit = spin_limit ? spin_limit : 125000u;      // ~0.5 s
    A.status = status;
    hipLaunchKernelGGL(p2p_allgather_kernel, dim3(world), dim3(256), 0, as_stream(stream), A);
    return hipGetLastError();
}
