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
#include <stdlib.h>
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
    if (__hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;      // a dead exchange stays dead (sticky)
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

// ---------------------------------------------------------------------------------------------------------------------
// The general exchange step (round 5): the step counter lives ON THE DEVICE, so the launch has no host-side argument that
// changes from step to step and sits inside a captured HIP graph like any other kernel of the sharded step; a message has a
// per-receiver part ("scatter": the partial d/d(state) rows of the receiver's agent block) and a common part ("broadcast":
// the rank's own records forward, its weight-gradient bucket backward); behind the wait every workgroup either copies the
// senders' parts out in rank order (all-gather) or ADDS them in rank order (reduce-scatter / all-reduce: the same sum, in the
// same order, on every rank -- bit-reproducible).  Workgroup (r, b) sends slice b of the message for receiver r; the last of
// receiver r's SPLIT workgroups to finish raises the flag.  ctr = [completed steps | finished workgroups | sent workgroups
// per receiver ...].  status is STICKY: once a wait has run out the exchange is dead -- later launches return at once (a rank
// that went on alone would otherwise run two steps ahead of a peer and overwrite a parity half the peer is still reading).
// ---------------------------------------------------------------------------------------------------------------------
struct P2pX {
    const float4* src_a; const float4* src_b;
    float4* out_a; float4* out_b;
    float* recv[kP2pMaxWorld];
    unsigned* flags[kP2pMaxWorld];
    unsigned long long na4, nb4, slot4;
    int rank, world, split, sum;
    unsigned spin_limit;
    unsigned* ctr;
    int* status;
};

__global__ __launch_bounds__(256) void p2p_exchange_kernel(P2pX A) {
    __shared__ int sh_ok;
    const int r = (int)blockIdx.x / A.split, b = (int)blockIdx.x % A.split, tid = threadIdx.x;
    if (__hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;      // a dead exchange stays dead
    const unsigned seq = __hip_atomic_load(A.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;   // (advanced by the LAST workgroup out)
    const int parity = (int)(seq & 1u);
    const unsigned long long n4 = A.na4 + A.nb4;
    {
        float4* dst = reinterpret_cast<float4*>(A.recv[r]) + ((size_t)parity * A.world + A.rank) * A.slot4;
        const float4* sa = A.src_a + (size_t)r * A.na4;
        for (unsigned long long e = (unsigned long long)b * 256 + tid; e < n4; e += (unsigned long long)A.split * 256)
            dst[e] = e < A.na4 ? sa[e] : A.src_b[e - A.na4];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores ...
    __syncthreads();                                       // ... before the one lane that signals for the workgroup
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope: the payload is visible to the peer before the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned sent = __hip_atomic_fetch_add(A.ctr + 2 + r, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (sent == (unsigned)A.split - 1u) {              // receiver r's last slice is out: its flag
            __hip_atomic_store(A.ctr + 2 + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.flags[r] + parity * A.world + A.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        sh_ok = 1;
    }
    __syncthreads();
    // every workgroup waits for this rank's own flags (one lane per sender), then takes its slice of the result
    if (tid < A.world) {
        const unsigned* f = A.flags[A.rank] + parity * A.world + tid;
        unsigned spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (++spins > A.spin_limit) { sh_ok = 0; break; }
            __builtin_amdgcn_s_sleep(127);
        }
    }
    __syncthreads();
    const bool ok = sh_ok != 0;
    if (!ok && tid == 0) __hip_atomic_store(A.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // what the senders released is what is read below
    if (ok) {
        const float4* base = reinterpret_cast<const float4*>(A.recv[A.rank]) + (size_t)parity * A.world * A.slot4;
        const unsigned long long stride = (unsigned long long)gridDim.x * 256;
        for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + tid; e < n4; e += stride) {
            const bool in_a = e < A.na4;
            float4* out = in_a ? A.out_a : A.out_b;
            if (!out) continue;
            const unsigned long long eo = in_a ? e : e - A.na4, per = in_a ? A.na4 : A.nb4;
            if (A.sum) {
                float4 acc = base[e];
                for (int s = 1; s < A.world; ++s) {        // rank order: the same sum on every rank
                    const float4 v = base[(size_t)s * A.slot4 + e];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                out[eo] = acc;
            } else {
                for (int s = 0; s < A.world; ++s) out[(size_t)s * per + eo] = base[(size_t)s * A.slot4 + e];
            }
        }
    }
    __syncthreads();
    if (tid == 0 && ok) {
        const unsigned done = __hip_atomic_fetch_add(A.ctr + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1u) {
            __hip_atomic_store(A.ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.ctr, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace piml

using namespace piml;

// Receive buffers and flag words are written by PEERS (over xGMI between the GPUs of a node) while this rank's kernels poll and
// read them: fine-grained device memory, as RCCL uses for its P2P buffers -- cross-agent visibility of coarse-grained memory
// (plain hipMalloc) is only defined at kernel boundaries, and a system-scope fence does not make this GPU's L2 coherent with a
// remote store into it.  PIML_P2P_COARSE=1 keeps hipMalloc (A/B on a single GPU, where the ranks share one L2).
PIML_API int piml_p2p_alloc(size_t bytes, void** devptr) {
    if (!devptr || bytes == 0) return hipErrorInvalidValue;
    static const bool coarse = getenv("PIML_P2P_COARSE") && atoi(getenv("PIML_P2P_COARSE")) != 0;
    hipError_t e = coarse ? hipMalloc(devptr, bytes) : hipExtMallocWithFlags(devptr, bytes, hipDeviceMallocFinegrained);
    if (e) return e;
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

PIML_API int piml_p2p_exchange(const piml_p2p_msg* msg, int rank, int world, float* const* peer_recv, unsigned* const* peer_flags,
                               size_t slot_floats, unsigned* ctr, unsigned spin_limit, int* status, void* stream) {
    if (!msg || !peer_recv || !peer_flags || !ctr || !status || world < 1 || world > kP2pMaxWorld || rank < 0 || rank >= world)
        return hipErrorInvalidValue;
    const size_t na = msg->scatter_floats, nb = msg->bcast_floats;
    if (na + nb == 0 || na % 4 || nb % 4 || slot_floats % 4 || na + nb > slot_floats || (na && !msg->scatter_src) || (nb && !msg->bcast_src))
        return hipErrorInvalidValue;
    P2pX A = {};
    A.src_a = reinterpret_cast<const float4*>(msg->scatter_src);
    A.src_b = reinterpret_cast<const float4*>(msg->bcast_src);
    A.out_a = reinterpret_cast<float4*>(msg->out_scatter);
    A.out_b = reinterpret_cast<float4*>(msg->out_bcast);
    for (int r = 0; r < world; ++r) {
        if (!peer_recv[r] || !peer_flags[r]) return hipErrorInvalidValue;
        A.recv[r] = peer_recv[r];
        A.flags[r] = peer_flags[r];
    }
    A.na4 = na / 4; A.nb4 = nb / 4; A.slot4 = slot_floats / 4;
    A.rank = rank; A.world = world; A.sum = msg->sum ? 1 : 0;
    const unsigned long long n4 = A.na4 + A.nb4;
    A.split = (int)(n4 / 1024 < 1 ? 1 : (n4 / 1024 > 16 ? 16 : n4 / 1024));
    A.spin_limit = spin_limit ? spin_limit : 125000u;      // ~0.5 s
    A.ctr = ctr;
    A.status = status;
    hipLaunchKernelGGL(p2p_exchange_kernel, dim3((unsigned)(world * A.split)), dim3(256), 0, as_stream(stream), A);
    return hipGetLastError();
}
