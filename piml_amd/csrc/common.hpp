// Shared device helpers for the gfx950 kernels of libpiml_hip.so.
// CDNA4 only: 64-lane wavefronts are assumed throughout (no other target is built).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PIML_API extern "C" __attribute__((visibility("default")))

namespace piml {

typedef unsigned long long u64;
constexpr int kWave = 64;
constexpr u64 kEmptyKey = ~0ull;

// --- float32 arithmetic pinned to PyTorch's CPU kernels (DESIGN.md "pinned arithmetic") ---
// torch.norm(x, p=2, dim=-1) on a 2-vector evaluates sqrt(fma(y, y, x*x)).
// sqrtf (built with -fhip-fp32-correctly-rounded-divide-sqrt) is correctly rounded on gfx950;
// __fsqrt_rn is NOT (1 ulp off for ~15 % of arguments, measured: tools/probe_math.hip).
__device__ __forceinline__ float norm2(float x, float y) {
    return sqrtf(__fmaf_rn(y, y, __fmul_rn(x, x)));
}
// torch.cosine_similarity(a, b) on 2-vectors: each operand divided by max(|.|, 1e-8) first,
// products rounded separately, then one add (no fma).  (b0, b1) is already normalised.
__device__ __forceinline__ float cos_sim_prenorm(float ax, float ay, float a_norm, float b0, float b1) {
    const float n = fmaxf(a_norm, 1e-8f);
    return __fadd_rn(__fmul_rn(__fdiv_rn(ax, n), b0), __fmul_rn(__fdiv_rn(ay, n), b1));
}
__device__ __forceinline__ float sq2(float x, float y) { return __fmaf_rn(y, y, __fmul_rn(x, x)); }
__device__ __forceinline__ float nan_to_zero(float x) { return x != x ? 0.f : x; }

// --- wave-level helpers ---
__device__ __forceinline__ int lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `m` strictly below this lane
__device__ __forceinline__ unsigned mbcnt(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// `base` + the number of set bits of `m` strictly below this lane (the addend rides in the mbcnt instruction)
__device__ __forceinline__ unsigned mbcnt_add(u64 m, unsigned base) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, base));
}
// value is identical in every lane: move it to an SGPR
__device__ __forceinline__ float uniform(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ u64 readlane64(u64 x, int l) {
    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, l);
    unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), l);
    return ((u64)hi << 32) | lo;
}
// lane l receives lane l-1's value (lane 0 keeps its own)
__device__ __forceinline__ u64 shift_up1(u64 x) {
    unsigned lo = (unsigned)__shfl_up((int)(unsigned)x, 1, 64);
    unsigned hi = (unsigned)__shfl_up((int)(unsigned)(x >> 32), 1, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace piml
