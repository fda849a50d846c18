// The dropout keep-mask stream of libpiml_hip.so (dropout.hip: the stand-alone generator; encoder_x3.hip: the same words
// drawn inside the encoder forward).  Private to the library; restated in numpy by tests/philox_ref.py.
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11), counter = (offset lo, offset hi, row, (stream << 16) | sub),
// key = (seed lo, seed hi).  `offset` counts the draws (one per forward launch, advanced on the device), `stream` tells
// the branches of one launch apart.
//   p == 0.5  ("fair bits"): keep word w (features 32 w .. 32 w + 31) of a row = output word (w & 3) of the call
//             sub = 0xFFFF - (w >> 2): ONE call per 128 features;
//   other p:  feature c takes 16 bits: call sub = c >> 3, output word (c >> 1) & 3, half c & 1; kept iff those 16 bits
//             >= round(p * 65536).
#pragma once
#include <hip/hip_runtime.h>

namespace piml {

struct PhiloxOut { unsigned x, y, z, w; };

__device__ __forceinline__ PhiloxOut philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return PhiloxOut{c0, c1, c2, c3};
}

// the four keep words of features 0 .. 127 of `row` for p = 0.5
__device__ __forceinline__ PhiloxOut keep_words_fair(unsigned long long seed, unsigned long long offset, unsigned row, unsigned stream) {
    return philox4x32_10((unsigned)offset, (unsigned)(offset >> 32), row, (stream << 16) | 0xFFFFu, (unsigned)seed, (unsigned)(seed >> 32));
}

constexpr float kFairP = 0.5f;

// Last workgroup out advances the draw counter: state = [seed, offset, ticket, -]; every workgroup of the launch has read
// `offset` (and used it) before it takes its ticket.  Call from ONE thread per workgroup, after the workgroup's last use
// of `offset`.  No fence: a device-scope release would write back the whole L2 of the XCD (measured: + 14 us in the
// encoder forward, which has just written 100 MB); the relaxed device-scope atomic is all the counting needs, and the
// new offset only has to be visible to the NEXT launch (kernel boundary).
__device__ __forceinline__ void dropout_advance(unsigned long long* state, unsigned long long offset, unsigned nblocks) {
    unsigned* ticket = reinterpret_cast<unsigned*>(state + 2);
    if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
        state[1] = offset + 1;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace piml
