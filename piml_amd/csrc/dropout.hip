// Keep-mask bits of the PINNSF processor's train-mode dropout (reference: ResDNN.forward = Dropout_p(2 x),
// src/models/model.py:82-119 with SURVEY quirk Q3; model.train() at src/models/simulators.py:311, --dropout 0.5 at
// src/main.py:45).  The fused encoder kernels apply the mask in their epilogue / at the head of their backward chain
// (encoder_x3.hip, encoder.hip, mlpglue.hip: scale_ksum); this file only draws it.
//
// Philox4x32-10 (Salmon et al., SC'11), counter = (offset lo, offset hi, row, c >> 2), key = (seed lo, seed hi): one
// call yields the uniforms of four consecutive features.  The call counter `offset` lives on the device and is advanced
// by the launch itself, so a launch captured into a hipGraph draws a fresh mask on every replay.
#include "common.hpp"
#include "../../include/piml_hip.h"

namespace piml {

struct U4 { unsigned x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// one thread per (row, word of 32 features): eight Philox calls
__global__ __launch_bounds__(256) void dropout_keep_bits_kernel(u64* __restrict__ state, long long rows, int words, int cols,
                                                                u64 thresh, unsigned* __restrict__ bits) {
    const u64 seed = state[0], off = state[1];
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id < rows * words) {
        const unsigned row = (unsigned)(id / words), w = (unsigned)(id % words);
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const U4 r = philox4x32_10(U4{(unsigned)off, (unsigned)(off >> 32), row, w * 8 + i}, (unsigned)seed, (unsigned)(seed >> 32));
            m |= ((u64)r.x >= thresh ? 1u : 0u) << (4 * i);
            m |= ((u64)r.y >= thresh ? 2u : 0u) << (4 * i);
            m |= ((u64)r.z >= thresh ? 4u : 0u) << (4 * i);
            m |= ((u64)r.w >= thresh ? 8u : 0u) << (4 * i);
        }
        const int left = cols - 32 * (int)w;                    // features past `cols` are never kept
        if (left < 32) m &= (1u << left) - 1u;
        bits[id] = m;
    }
    // the last block out advances the call counter (every block has read `off` before it takes its ticket)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        unsigned* ticket = reinterpret_cast<unsigned*>(state + 2);
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            state[1] = off + 1;
            *ticket = 0;
            __threadfence();
        }
    }
}

}  // namespace piml

using namespace piml;

PIML_API int piml_dropout_keep_bits(unsigned long long* state, long long rows, int cols, float p, unsigned* keep_bits, void* stream) {
    if (rows < 0 || cols <= 0 || !(p >= 0.f && p <= 1.f) || rows >= (1ll << 32)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!state || !keep_bits) return hipErrorInvalidValue;
    const int words = (cols + 31) / 32;
    double t = (double)p * 4294967296.0;
    u64 thresh = (u64)(t + 0.5);
    if (thresh > 4294967296ull) thresh = 4294967296ull;
    const long long n = rows * words;
    hipLaunchKernelGGL(dropout_keep_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<u64*>(state), rows, words, cols, thresh, keep_bits);
    return hipGetLastError();
}
