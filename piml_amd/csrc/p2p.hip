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
constexpr int kP2pParts = 1 + PIML_P2P_MAX_PARTS;        // the scatter part + the broadcast parts
struct P2pX {
    const float4* src[kP2pParts];          // part 0 = the scatter part (receiver r reads src[0] + r * n4[0])
    float4* out[kP2pParts];
    unsigned long long n4[kP2pParts], off4[kP2pParts];      // float4 per part, its offset inside a slot
    int nparts;
    float* recv[kP2pMaxWorld];
    unsigned* flags[kP2pMaxWorld];
    unsigned long long tot4, slot4;
    int rank, world, split, sum;
    unsigned spin_limit;
    unsigned* ctr;
    int* status;
};

// polls of the flag words sleep ~0.5 us between loads (s_sleep 16 = 1024 clocks)
__device__ __forceinline__ bool p2p_wait(const unsigned* f, unsigned want, unsigned limit, int scope_system) {
    unsigned spins = 0;
    for (;;) {
        const unsigned v = scope_system ? __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                        : __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == want) return true;
        if (++spins > limit) return false;
        __builtin_amdgcn_s_sleep(16);
    }
}

__global__ __launch_bounds__(256) void p2p_exchange_kernel(P2pX A) {
    __shared__ int sh_ok;
    const int r = (int)blockIdx.x / A.split, b = (int)blockIdx.x % A.split, tid = threadIdx.x;
    if (__hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;      // a dead exchange stays dead
    const unsigned seq = __hip_atomic_load(A.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;   // (advanced by the LAST workgroup out)
    const int parity = (int)(seq & 1u);
    // part of slot element e (parts are few: a linear walk)
    auto part_of = [&](unsigned long long e) {
        int j = 0;
        while (j + 1 < A.nparts && e >= A.off4[j + 1]) ++j;
        return j;
    };
    {
        float4* dst = reinterpret_cast<float4*>(A.recv[r]) + ((size_t)parity * A.world + A.rank) * A.slot4;
        for (unsigned long long e = (unsigned long long)b * 256 + tid; e < A.tot4; e += (unsigned long long)A.split * 256) {
            const int j = part_of(e);
            dst[e] = A.src[j][(j == 0 ? (size_t)r * A.n4[0] : 0) + (e - A.off4[j])];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores ...
    __syncthreads();                                       // ... before the one lane that signals for the workgroup
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope: the payload is visible to the peer before the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned sent = __hip_atomic_fetch_add(A.ctr + 3 + r, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (sent == (unsigned)A.split - 1u) {              // receiver r's last slice is out: its flag
            __hip_atomic_store(A.ctr + 3 + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.flags[r] + parity * A.world + A.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_fetch_add(A.ctr + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // this workgroup has read its part of the sources
        sh_ok = 1;
    }
    __syncthreads();
    // wait: (i) for this rank's own flags, one lane per sender; (ii) lane 63: until EVERY workgroup of this launch has finished
    // reading the sources -- the results may be written over them (in-place sums: out == src)
    if (tid < A.world) {
        if (!p2p_wait(A.flags[A.rank] + parity * A.world + tid, seq, A.spin_limit, 1)) sh_ok = 0;
    } else if (tid == 63) {
        if (!p2p_wait(A.ctr + 2, gridDim.x, A.spin_limit, 0)) sh_ok = 0;
    }
    __syncthreads();
    const bool ok = sh_ok != 0;
    if (!ok && tid == 0) __hip_atomic_store(A.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // what the senders released is what is read below
    if (ok) {
        const float4* base = reinterpret_cast<const float4*>(A.recv[A.rank]) + (size_t)parity * A.world * A.slot4;
        const unsigned long long stride = (unsigned long long)gridDim.x * 256;
        for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + tid; e < A.tot4; e += stride) {
            const int j = part_of(e);
            float4* out = A.out[j];
            if (!out) continue;
            const unsigned long long eo = e - A.off4[j];
            if (A.sum) {
                float4 acc = base[e];
                for (int s = 1; s < A.world; ++s) {        // rank order: the same sum on every rank
                    const float4 v = base[(size_t)s * A.slot4 + e];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                out[eo] = acc;
            } else {
                for (int s = 0; s < A.world; ++s) out[(size_t)s * A.n4[j] + eo] = base[(size_t)s * A.slot4 + e];
            }
        }
    }
    __syncthreads();
    if (tid == 0 && ok) {
        const unsigned done = __hip_atomic_fetch_add(A.ctr + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1u) {
            __hip_atomic_store(A.ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.ctr + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    if (!msg || !peer_recv || !peer_flags || !ctr || !status || world < 1 || world > kP2pMaxWorld || rank < 0 || rank >= world ||
        msg->n_bcast < 0 || msg->n_bcast > PIML_P2P_MAX_PARTS)
        return hipErrorInvalidValue;
    P2pX A = {};
    unsigned long long off = 0;
    auto add = [&](const float* src, size_t n, float* out) {
        if (n % 4 || !src) return false;
        A.src[A.nparts] = reinterpret_cast<const float4*>(src);
        A.out[A.nparts] = reinterpret_cast<float4*>(out);
        A.n4[A.nparts] = n / 4;
        A.off4[A.nparts] = off;
        off += n / 4;
        ++A.nparts;
        return true;
    };
    // part 0 is always the scatter part (possibly empty: it then never matches an element)
    if (msg->scatter_floats) {
        if (!add(msg->scatter_src, msg->scatter_floats, msg->out_scatter)) return hipErrorInvalidValue;
    } else {
        A.nparts = 1;
    }
    for (int j = 0; j < msg->n_bcast; ++j)
        if (msg->bcast_floats[j] && !add(msg->bcast_src[j], msg->bcast_floats[j], msg->out_bcast[j])) return hipErrorInvalidValue;
    if (off == 0 || slot_floats % 4 || off > slot_floats / 4) return hipErrorInvalidValue;
    for (int r = 0; r < world; ++r) {
        if (!peer_recv[r] || !peer_flags[r]) return hipErrorInvalidValue;
        A.recv[r] = peer_recv[r];
        A.flags[r] = peer_flags[r];
    }
    A.tot4 = off; A.slot4 = slot_floats / 4;
    A.rank = rank; A.world = world; A.sum = msg->sum ? 1 : 0;
    // workgroups per receiver: ~2 float4 per thread (loads of the fine-grained receive buffers are uncached round trips: many
    // threads with few loads each), all of them resident at once (the waits spin): world * split <= 512
    // PIML_P2P_MAX_SPLIT: a lower cap -- several ranks SHARING one GPU (tests/test_p2p_gpu.py: 8 processes on one device) need all
    // their launches resident together: world ranks x world x split workgroups <= 2048
    static const int max_split = getenv("PIML_P2P_MAX_SPLIT") && atoi(getenv("PIML_P2P_MAX_SPLIT")) > 0 ? atoi(getenv("PIML_P2P_MAX_SPLIT")) : 64;
    const int cap = max_split < 64 ? max_split : 64;
    A.split = (int)(off / 512 < 1 ? 1 : (off / 512 > (unsigned long long)cap ? (unsigned long long)cap : off / 512));
    A.spin_limit = spin_limit ? spin_limit : 1000000u;     // ~0.5 s
    A.ctr = ctr;
    A.status = status;
    hipLaunchKernelGGL(p2p_exchange_kernel, dim3((unsigned)(world * A.split)), dim3(256), 0, as_stream(stream), A);
    return hipGetLastError();
}
