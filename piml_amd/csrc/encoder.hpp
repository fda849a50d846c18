// Shared by the encoder kernels (encoder.hip: f32 matrix-core products; encoder_x3.hip: split bf16 products).
// Private to libpiml_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "pack.hpp"
#include "../../include/piml_hip.h"

namespace piml {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ENC_THREADS = 512;        // 8 waves per workgroup: 2 per SIMD
constexpr int ENC_WAVES = ENC_THREADS / 64;

struct EncArgs {
    piml_encoder_branch br[2];
    int nbr;
    int wg_split;       // workgroups [0, wg_split) serve branch 0, the rest branch 1
    float* zero;        // forward only, optional: a buffer the launch clears on the way (the decoder tails' accumulator)
    int zero_n;
    unsigned long long* gen_state;   // forward only: non-NULL = the kernel draws the p = 0.5 keep-masks itself (philox.hpp)
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// store of 4 consecutive floats of a row (non-temporal stores were measured here: forward 47.9 -> 53.7 us, reverted)
__device__ __forceinline__ void store4_stream(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// Packed image -> LDS, every 16-byte load of the thread issued before the first LDS write.  (As a plain loop this
// compiled to load / s_waitcnt vmcnt(0) / ds_write per iteration: 17 serialised L2 round trips in front of every
// workgroup's first MFMA.)
template <int NFLOATS>
__device__ __forceinline__ void stage_linear(float* lds, const float* __restrict__ src, int tid) {
    constexpr int N4 = NFLOATS / 4, ROUNDS = (N4 + ENC_THREADS - 1) / ENC_THREADS;
    static_assert(NFLOATS % 4 == 0, "float4 granularity");
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(lds);
    float4 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int e = r * ENC_THREADS + tid;
        v[r] = s4[(r + 1) * ENC_THREADS <= N4 || e < N4 ? e : 0];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int e = r * ENC_THREADS + tid;
        if ((r + 1) * ENC_THREADS <= N4 || e < N4) d4[e] = v[r];
    }
}

// features (4 consecutive) held by accumulator registers 4q .. 4q+3 of block blk in lane half h
__device__ __forceinline__ int feat0(int blk, int q, int h) { return 32 * blk + 8 * q + 4 * h; }

// encoder_x3.hip: the same stages on split bf16 products (launch only; arguments checked by the callers in encoder.hip)
int enc_x3_set_attributes();
// drop: every branch of the launch carries keep_bits (the processor's train-mode dropout)
void enc_x3_launch_fwd_pool(const EncArgs& A, int total, hipStream_t s);      // inference: layers 1-2 + the agents' sums of h2
void enc_x3_launch_fwd_sum(const EncArgs& A, int total, hipStream_t s);       // training on the agents' sums of h2 (PIML_POOL_TRAIN)
void enc_x3_launch_fwd(const EncArgs& A, int total, bool drop, hipStream_t s, bool exch = false);   // A.gen_state: draw the masks in the kernel
void enc_x3_launch_fwd_split(const EncArgs& A, int pairs0, int pairs1, bool drop, hipStream_t s);      // few rows: four waves per tile
void enc_x3_launch_bwd_dx_split(const EncArgs& A, int pairs0, int pairs1, bool drop, hipStream_t s);      // few rows: four waves per tile
void enc_x3_launch_bwd_dx(const EncArgs& A, int total, bool mask, bool drop, hipStream_t s);   // mask: the forward of these branches wrote relu_mask

// encoder_dw2.hip: the weight gradients as layer-split workgroups (half the partial bytes, h1 recomputed when absent)
int enc_dww_set_attributes();                                                          // encoder_dww.hip: the same slabs, wide staging loads
void enc_dww_launch(const EncArgs& A, int grid, bool drop, hipStream_t s);
int enc_dw2_set_attributes();
void enc_dw2_split(int w, int* n0, int* n1);          // a branch's w workgroups -> layer-0 / layer-1 workgroups
// l0_only: every workgroup takes layer 0 (dW3 / db3); the lower layers' gradients come from enc_f3_launch
void enc_dw2_launch(const EncArgs& A, int total, hipStream_t s, bool l0_only = false);

// encoder_bwd3.hip: dX chain + dW2 / dW1 / db2 / db1 in one pass (g2 / g1 never leave the CU); nA[b] workgroups for branch b,
// their slots behind slot0[b] layer-0 slots of the branch's `partials`
// encoder_bwd4.hip: the same in eight waves (two per SIMD, 16-feature blocks): the default; encoder_bwd3.hip with PIML_ENC_FUSED_V=3
int enc_f4_set_attributes();
void enc_f4_launch(const EncArgs& A, const int* nA, const int* slot0, bool with_dw3, hipStream_t s);
int enc_f3_set_attributes();
// with_dw3: the launch also does dW3 / db3 (layer-0 slots, slot = workgroup index within the branch; needs slot0[b] == nA[b])
// sums: the PIML_POOL_TRAIN form (G2 = g_pooled[agent] * [h2 > 0]: no W3^T layer, no dW3; sign words of h2 in the exchanged layout)
void enc_f3_launch(const EncArgs& A, const int* nA, const int* slot0, bool with_dw3, hipStream_t s, bool sums = false);
// encoder_bwd5.hip: the PIML_POOL_TRAIN backward as two crews of four waves (chain | weight gradients), two waves per SIMD;
// bitwise the results of enc_f3_launch(..., sums = true).  false: a shape it does not take (the caller falls back)
int enc_f5_set_attributes();
bool enc_f5_launch(const EncArgs& A, const int* nA, hipStream_t s);

}  // namespace piml
