// Device helpers of the split-product ("x3") kernels: encoder_x3.hip (forward, dX chain, weight gradients) and
// encoder_dw2.hip (weight gradients, layer-split decomposition).  Private to libpiml_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "encoder.hpp"

namespace piml {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // register arrays of HIP's u32x4 struct were left in scratch

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ReLU in one instruction (fmaxf costs a canonicalising v_max before the v_max)
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }
// keep v where bit `pos` of m is set, else 0 (one bit-field extract to an all-ones / zero word, one and)
__device__ __forceinline__ float keep_if(float v, unsigned m, int pos) {
    const int t = __builtin_amdgcn_sbfe(m, pos, 1);
    return __uint_as_float(__float_as_uint(v) & (unsigned)t);
}

// Dropout of the processor output (piml_encoder_branch.keep_bits: bit c & 31 of word c >> 5 of a row = keep feature c):
// lane (row, h) holds features 32 blk + (r & 3) + 8 (r >> 2) + 4 h of block blk in register r, so after a shift by 4 h
// the bit positions are compile-time constants.
__device__ __forceinline__ void keep_block(f32x16& a, unsigned word, int h) {
    const unsigned m = word >> (4 * h);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = keep_if(a[r], m, (r & 3) + 8 * (r >> 2));
}
__device__ __forceinline__ unsigned word_of(const uint4& k, int blk) { return blk == 0 ? k.x : (blk == 1 ? k.y : (blk == 2 ? k.z : k.w)); }

// The six products of one k-block.  The matrix core adds the 16 products of an instruction and the accumulator with the
// low bits of the aligned addends cut off, not rounded (measured: sums over many rows of x3 results drift by ~0.5 ulp of
// the accumulator per instruction, all in one direction), so the five small terms (<= 2^-8 of the product) go to a second
// accumulator, where that cut is 2^-8 smaller still, and only w_hi x_hi -- eight instructions per output, against the 128
// roundings of an f32 fmaf chain -- touches the main one.  The two are added once per output block.
__device__ __forceinline__ void kblock_x3(f32x16& acc, f32x16& small, u32x4 wh, u32x4 wm, u32x4 wl, u32x4 xh, u32x4 xm, u32x4 xl) {
    small = mfma_bf(wl, xh, small);
    small = mfma_bf(wm, xm, small);
    small = mfma_bf(wh, xl, small);
    small = mfma_bf(wm, xh, small);
    small = mfma_bf(wh, xm, small);
    acc = mfma_bf(wh, xh, acc);
}

// The same six products with the operands exchanged: the activations' pieces as the A operand, the weight fragment as B --
// the transposed output block (D'[row][feature]: lane = feature, registers = the tile's rows), every accumulator seeing the
// same products in the same order.
__device__ __forceinline__ void kblock_x3_t(f32x16& acc, f32x16& small, u32x4 wh, u32x4 wm, u32x4 wl, u32x4 xh, u32x4 xm, u32x4 xl) {
    small = mfma_bf(xh, wl, small);
    small = mfma_bf(xm, wm, small);
    small = mfma_bf(xl, wh, small);
    small = mfma_bf(xh, wm, small);
    small = mfma_bf(xm, wh, small);
    acc = mfma_bf(xh, wh, acc);
}

// batch geometry of the slab weight-gradient kernel (enc_bwd_dw_x3w_kernel, encoder_dww.hip)
constexpr int DW_X3_ROWS = 16;
constexpr int DWX_ARR = 3 * 256;                       // u32x4 of one array's three pieces
constexpr int DWX_BUF = 4 * DWX_ARR + 32;              // + x rows [16][8] floats
constexpr int DWX_LDS_BYTES = 2 * DWX_BUF * 16;
constexpr int DWX_RED = 4 * 128 * 9;                   // floats of the final cross-group exchange (reuses the buffers)
static_assert((DWX_RED + 4 * 128 * 2) * 4 <= DWX_LDS_BYTES, "exchange fits");

}  // namespace piml
