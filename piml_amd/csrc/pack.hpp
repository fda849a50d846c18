// Packed (MFMA operand fragment) images of the PINNSF weights -- layouts and the element -> source-weight maps -- and
// the slot sums of the weight-gradient partials.  Shared by the component kernels (encoder.hip, decoder.hip) and the
// one-launch pack / reduction of the whole network (network.hip).  Private to libpiml_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/piml_hip.h"

namespace piml {

constexpr int EH = 128;        // encoder hidden width
constexpr int DH = 128;        // decoder input width (= encoder width)
constexpr int DD = 64;         // decoder hidden / output width
constexpr int ENC_PART = 2 * EH * EH + EH * 8 + 3 * EH;                // floats of one encoder workgroup's dW / db partial
constexpr int DEC_PART = DD * DH + DD * DD + 2 * DD + DD + DD + 8;     // decoder: dW1 | dW2 | dW3 | db1 | db2 | db3 (+pad)

// A-fragment images of a 128 x 128 nn.Linear weight (out, in), float4 index ((blk*4 + bp)*4 + q)*64 + lane:
//   forward:       [u] = W[32 blk + i][32 bp + 8 q + 4 h + u]   (out block blk, in block bp)
//   backward (dX): [u] = W[32 bp + 8 q + 4 h + u][32 blk + i]   (A = W^T: in block blk, out block bp)
// Packed weights of one encoder (floats), written once per step by enc_pack_kernel so that every workgroup stages
// its LDS image with linear, fully coalesced copies:
//   [ W2 fragments 16384 | W3 fragments 16384 | W1 fragments 1024 | b1 b2 b3 384 ]        forward image
//   [ W3^T fragments 16384 | W2^T fragments 16384 | W1 rows padded to 8 columns 1024 ]    dX image
constexpr int PACK_FWD = 16384 * 2 + 1024 + 384;
constexpr int PACK_DX = 16384 * 2 + 1024;
constexpr int PACK_F32 = PACK_FWD + PACK_DX;

// ---- split-product ("x3") images: every f32 weight as three bf16 pieces hi + mid + lo = the f32 value exactly ----
// (hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid): 3 x 8 significand bits + round-to-nearest signs cover the 24
// of an f32.)  A-fragments of v_mfma_f32_32x32x16_bf16: fragment fb = 8 blk + kb = (out block blk, k-block kb), lane
// (i, h) holds 8 bf16 = one uint4, element t = W[32 blk + i][16 kb + 8 (t >> 2) + 4 h + (t & 3)] -- the k order in which
// a 32x32 accumulator's registers 8 (kb & 1) .. + 7 of block kb >> 1 hold the features (see encoder.hip).  One image
// (dwords): [ HM: fb 32 ][ p 2 (hi, mid) ][ lane 64 ][ 4 ]  |  [ LO: fb 32 ][ lane 64 ][ 4 ]; the transposed images hold
// W^T the same way (blk = input block of W, kb = k-block over W's outputs).
constexpr int X3_HM = 32 * 2 * 64 * 4;          // 16384 dwords
constexpr int X3_LO = 32 * 64 * 4;              //  8192 dwords
constexpr int X3_IMG = X3_HM + X3_LO;           // 24576 dwords = 96 KB
constexpr int PACK_X3 = 4 * X3_IMG;             // W2 | W3 | W3^T | W2^T
constexpr int PACK_FLOATS = PACK_F32 + PACK_X3;

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// two f32 -> one dword of two bf16 (round to nearest even; a in the low half)
__device__ __forceinline__ unsigned bf16_pair(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
// the three bf16 pieces of two f32 values, packed pairwise; the remainders are exact in f32
__device__ __forceinline__ void split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = bf16_pair(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = bf16_pair(ra, rb);
    lo = bf16_pair(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u));
}

__device__ __forceinline__ unsigned pack_bits_x3(const piml_encoder_branch& J, int e) {
    const int img = e / X3_IMG, g = e - img * X3_IMG;
    const bool tr = img >= 2;
    const float* W = (img == 0 || img == 3) ? J.w2 : J.w3;
    const int d = g & 3, lane = (g >> 2) & 63;
    int p, fb;
    if (g < X3_HM) { p = (g >> 8) & 1; fb = g >> 9; }
    else { p = 2; fb = (g - X3_HM) >> 8; }
    const int blk = fb >> 3, kb = fb & 7, i = 32 * blk + (lane & 31);
    const int c = 16 * kb + 8 * (d >> 1) + 4 * (lane >> 5) + 2 * (d & 1);       // t = 2 d, 2 d + 1
    const float a = tr ? W[(size_t)c * EH + i] : W[(size_t)i * EH + c];
    const float b = tr ? W[(size_t)(c + 1) * EH + i] : W[(size_t)i * EH + c + 1];
    unsigned hi, mid, lo;
    split3(a, b, hi, mid, lo);
    return p == 0 ? hi : (p == 1 ? mid : lo);
}

__device__ __forceinline__ float pack_value(const piml_encoder_branch& J, int e) {
    const int IN = J.in_dim;
    if (e >= PACK_F32) return __uint_as_float(pack_bits_x3(J, e - PACK_F32));
    if (e < 32768 || (e >= PACK_FWD && e < PACK_FWD + 32768)) {
        const bool tr = e >= PACK_FWD;
        const int f = tr ? e - PACK_FWD : e;
        const float* W = (f < 16384) == tr ? J.w3 : J.w2;      // fwd: W2 | W3;  dX: W3^T | W2^T
        const int g = f & 16383, u = g & 3, lane = (g >> 2) & 63, q = (g >> 8) & 3, bp = (g >> 10) & 3, blk = g >> 12;
        const int i = lane & 31, c = 32 * bp + 8 * q + 4 * (lane >> 5) + u;
        return tr ? W[(size_t)c * EH + 32 * blk + i] : W[(size_t)(32 * blk + i) * EH + c];
    }
    if (e < 32768 + 1024) {                                    // W1 fragments [blk][s][lane]
        const int g = e - 32768, l = g & 63, sidx = (g >> 6) & 3, blk = g >> 8;
        const int c = 2 * sidx + (l >> 5);
        return c < IN ? J.w1[(size_t)(32 * blk + (l & 31)) * IN + c] : 0.f;
    }
    if (e < PACK_FWD) {
        const int g = e - 32768 - 1024;
        return g < 128 ? J.b1[g] : (g < 256 ? J.b2[g - 128] : J.b3[g - 256]);
    }
    {                                                          // W1 rows padded to 8 columns: [f][8]
        const int g = e - PACK_FWD - 32768, f = g >> 3, c = g & 7;
        return c < IN ? J.w1[(size_t)f * IN + c] : 0.f;
    }
}

// packed image of one branch (floats)
constexpr int DP_A1 = 0;                         // [ob 2][bp 4][q 4][lane 64] float4   W1 (64, 128)
constexpr int DP_A2 = DP_A1 + 2 * 4 * 4 * 256;   // [ob 2][bp 2][q 4][lane 64] float4   W2 (64, 64)
constexpr int DP_A3 = DP_A2 + 2 * 2 * 4 * 256;   // [bp 2][q 4][lane 64] float4         W3 (2, 64), rows >= 2 are 0
constexpr int DP_B = DP_A3 + 2 * 4 * 256;        // b1 64 | b2 64 | b3 2 | pad 2
constexpr int DP_T3 = DP_B + 132;                // [ob 2][lane 64]                     W3^T: lane (i, h) = W3[h][32 ob + i]
constexpr int DP_T2 = DP_T3 + 128;               // [ob 2][bp 2][q 4][lane 64] float4   W2^T
constexpr int DP_T1 = DP_T2 + 2 * 2 * 4 * 256;   // [blk 4][bp 2][q 4][lane 64] float4  W1^T
// FOLDED first layer (piml_decoder_branch.fold_w3, training on the agents' sums of h2: PIML_POOL_TRAIN): the same two images of
// W1' = fold_scale * W1 * fold_w3 (64 x 128; float64 products, rounded once) and the vector c = fold_scale * W1 * fold_b3
// (the first layer's bias on the sums of an agent's k rows is b1 + k c); garbage when the branch carries no fold_w3
constexpr int DP_A1F = DP_T1 + 4 * 2 * 4 * 256;
constexpr int DP_T1F = DP_A1F + 2 * 4 * 4 * 256;
constexpr int DP_CF = DP_T1F + 4 * 2 * 4 * 256;  // c 64
constexpr int DEC_PACK = DP_CF + 64;
constexpr int DEC_PACK_PLAIN = DP_A1F;           // what dec_pack_value fills; the folded part: dec_fold_item

__device__ __forceinline__ float dec_pack_value(const piml_decoder_branch& J, int e) {
    if (e >= DEC_PACK_PLAIN) return 0.f;          // (the folded images: dec_fold_item)
    if (e < DP_B) {                 // forward fragments
        int f = e, rows_in;         // rows_in: input width of the layer
        const float* W;
        int nbp, limit_i = 64;
        if (e < DP_A2) { W = J.w1; rows_in = DH; nbp = 4; }
        else if (e < DP_A3) { f = e - DP_A2; W = J.w2; rows_in = DD; nbp = 2; }
        else { f = e - DP_A3; W = J.w3; rows_in = DD; nbp = 2; limit_i = 2; }
        const int u = f & 3, lane = (f >> 2) & 63, q = (f >> 8) & 3;
        int rest = f >> 10;
        const int bp = rest % nbp, ob = rest / nbp;           // ob = 0 for W3
        const int i = 32 * ob + (lane & 31), c = 32 * bp + 8 * q + 4 * (lane >> 5) + u;
        return i < limit_i ? W[(size_t)i * rows_in + c] : 0.f;
    }
    if (e < DP_T3) {
        const int g = e - DP_B;
        return g < 64 ? J.b1[g] : (g < 128 ? J.b2[g - 64] : (g < 130 ? J.b3[g - 128] : 0.f));
    }
    if (e < DP_T2) {                // W3^T: one k-step (k = h = output component)
        const int g = e - DP_T3, lane = g & 63, ob = g >> 6;
        return J.w3[(size_t)(lane >> 5) * DD + 32 * ob + (lane & 31)];
    }
    {                               // W2^T, W1^T: [u] = W[32 bp + 8 q + 4 h + u][32 blk + i]
        const bool t1 = e >= DP_T1;
        const int f = t1 ? e - DP_T1 : e - DP_T2;
        const float* W = t1 ? J.w1 : J.w2;
        const int cols = t1 ? DH : DD;                        // input width of the layer = columns of W
        const int u = f & 3, lane = (f >> 2) & 63, q = (f >> 8) & 3, rest = f >> 10;
        const int bp = rest & 1, blk = rest >> 1;
        const int r = 32 * bp + 8 * q + 4 * (lane >> 5) + u;  // output feature of the layer (row of W)
        return W[(size_t)r * cols + 32 * blk + (lane & 31)];
    }
}

// collision head: A1 [ob 2][bp 4][q 4][lane 64] float4 (W1 fragments for the f32 matrix instruction) | b1 64 | b2 1 + 3 pad |
// w2 64 (raw row) | W1 split into bf16 pieces for the split-product form: [ob 2][kb 8][piece 3][lane 64] u32x4 -- lane
// (i, g) holds W1[32 ob + i][16 kb + 8 g + t], t = 0 .. 7, packed pairwise
// | FOLDED (piml_collision_head.fold_w3: the head on h2 rows instead of message rows, PIML_POOL_TRAIN): the split-product image
// of W1' = fold_scale * W1 * fold_w3 and b1' = b1 + fold_scale * W1 * fold_b3 (64)
constexpr int HP_B = 2 * 4 * 4 * 256;
constexpr int HP_W2 = HP_B + 68;
constexpr int HP_X3 = HP_W2 + 64;
constexpr int HP_X3F = HP_X3 + 2 * 8 * 3 * 256;
constexpr int HP_BF = HP_X3F + 2 * 8 * 3 * 256;
constexpr int HEAD_PACK = HP_BF + 64;
constexpr int HEAD_PACK_PLAIN = HP_X3F;          // what head_pack_value fills; the folded part: head_fold_item
static_assert(HP_X3 % 4 == 0 && HP_X3F % 4 == 0, "16-byte aligned fragments");

__device__ __forceinline__ float head_pack_value(const piml_collision_head& H, int e) {
    const float* __restrict__ w1 = H.w1;
    if (e >= HEAD_PACK_PLAIN) return 0.f;         // (the folded images: head_fold_item)
    if (e < HP_B) {
        const int u = e & 3, lane = (e >> 2) & 63, q = (e >> 8) & 3, rest = e >> 10;
        const int bp = rest & 3, ob = rest >> 2;
        const int i = 32 * ob + (lane & 31), c = 32 * bp + 8 * q + 4 * (lane >> 5) + u;
        return w1[(size_t)i * DH + c];
    }
    if (e >= HP_X3) {
        const int x = e - HP_X3, d = x & 3, lane = (x >> 2) & 63, rest = x >> 8;
        const int piece = rest % 3, kb = (rest / 3) & 7, ob = rest / 24;
        const int i = 32 * ob + (lane & 31), c = 16 * kb + 8 * (lane >> 5) + 2 * d;
        unsigned hi, mid, lo;
        split3(w1[(size_t)i * DH + c], w1[(size_t)i * DH + c + 1], hi, mid, lo);
        return __uint_as_float(piece == 0 ? hi : (piece == 1 ? mid : lo));
    }
    if (e >= HP_W2) return H.w2[e - HP_W2];
    const int g = e - HP_B;
    return g < 64 ? H.b1[g] : (g == 64 ? H.b2[0] : 0.f);
}

// ---- the FOLDED images (PIML_POOL_TRAIN): W' = s W1 [W3 | b3] (64 x 129) per folded layer ----
// A 128-term dot product per element is a chain of L2 round trips for a lone thread (measured: 30 us at the tail of the relfeat
// forward, whose trailing workgroups these are, reduce.hpp), so a GROUP = (row i, 64 consecutive columns -- or the b3 column
// alone) is cut over CH waves, each taking 128 / CH terms with every load in flight at once: lanes = columns (coalesced W3
// rows), the W1 row wave-uniform (scalar loads), float64 partials meet in LDS and are added in a fixed order, the result is
// rounded once and scattered into the images.  Lean on purpose: the relfeat forward's register budget is shared.
constexpr int FOLD_GROUPS = DD * 3;                  // per folded layer: 64 rows x (columns 0 .. 63 | 64 .. 127 | the b3 column)

template <int CH>
__device__ __forceinline__ double fold_partial(const float* __restrict__ row, const float* __restrict__ col, int stride, int chunk) {
    constexpr int N = DH / CH;
    float wv[N], xv[N];
#pragma unroll
    for (int t = 0; t < N; ++t) { wv[t] = row[chunk * N + t]; xv[t] = col[(size_t)(chunk * N + t) * stride]; }
    double a = 0.0;
#pragma unroll
    for (int t = 0; t < N; ++t) a += (double)wv[t] * (double)xv[t];
    return a;
}

// the group's 64 finished values (lane = column) -> the decoder's folded images
__device__ __forceinline__ void dec_fold_store(const piml_decoder_branch& J, int i, int hc, int lane, double sum) {
    const float v = (float)((double)J.fold_scale * sum);
    if (hc == 2) {
        if (lane == 0) J.packed[DP_CF + i] = v;
        return;
    }
    const int c = 64 * hc + lane;
    // forward fragment: [ob][bp][q][lane][u], i = 32 ob + (lane & 31), c = 32 bp + 8 q + 4 h + u
    J.packed[DP_A1F + ((((i >> 5) * 4 + (c >> 5)) * 4 + ((c >> 3) & 3)) * 64 + (i & 31) + 32 * ((c >> 2) & 1)) * 4 + (c & 3)] = v;
    // transposed fragment: [blk][bp][q][lane][u], i = 32 bp + 8 q + 4 h + u, c = 32 blk + (lane & 31)
    J.packed[DP_T1F + ((((c >> 5) * 2 + (i >> 5)) * 4 + ((i >> 3) & 3)) * 64 + (c & 31) + 32 * ((i >> 2) & 1)) * 4 + (i & 3)] = v;
}
__device__ __forceinline__ void head_fold_store(const piml_collision_head& H, int i, int hc, int lane, double sum) {
    if (hc == 2) {
        if (lane == 0) H.packed[HP_BF + i] = (float)((double)H.b1[i] + (double)H.fold_scale * sum);
        return;
    }
    const float a = (float)((double)H.fold_scale * sum);
    const float b = __shfl_down(a, 1, 64);              // the pair's second column
    if (lane & 1) return;
    const int c = 64 * hc + lane;
    unsigned hi, mid, lo;
    split3(a, b, hi, mid, lo);
    // [ob][kb][piece][lane][d]: i = 32 ob + (lane & 31), c = 16 kb + 8 (lane >> 5) + 2 d
    unsigned* dst = reinterpret_cast<unsigned*>(H.packed + HP_X3F) + ((((i >> 5) * 8 + (c >> 4)) * 3) * 64 + (i & 31) + 32 * ((c >> 3) & 1)) * 4 + ((c & 7) >> 1);
    dst[0] = hi; dst[256] = mid; dst[512] = lo;
}

// ---- sums over per-workgroup partial slots (fixed order: deterministic), one block of 256 threads per column group ----
// 16 float4 columns x 16 slot groups per block (wide grid); a thread's slots are fetched eight at a time, all loads
// (clamped, unconditional) issued before the first add: the sums are latency-bound, not bandwidth-bound
// Target of float4 column j: off0 + j for j < split, off1 + (j - split) behind it (a slot whose fields land in two places of
// the gradient buffer: the layer-split slots of encoder_dw2.hip); the plain form is split = lanes, off0 = 0.
// accumulate: grads += the sum (a second backward pass through the same weights inside one optimiser step).
// (bx: the column block, blockIdx.x of a launch of its own)
__device__ __forceinline__ void sum_slots_16x16_at(int bx, const float* __restrict__ partials, float* __restrict__ grads, int B, int lanes,
                                                   int split = 0x7fffffff, int off0 = 0, int off1 = 0, bool accumulate = false) {
    __shared__ float4 sh[256];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int j = bx * 16 + col;
    const float4* parts = reinterpret_cast<const float4*>(partials);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < lanes)
        for (int q0 = grp; q0 < B; q0 += 16 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + 16 * u;
                v[u] = parts[(size_t)(q < B ? q : grp) * lanes + j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = q0 + 16 * u < B;
                s.x += ok ? v[u].x : 0.f; s.y += ok ? v[u].y : 0.f; s.z += ok ? v[u].z : 0.f; s.w += ok ? v[u].w : 0.f;
            }
        }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && j < lanes) {
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float4 v = sh[q * 16 + col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float4* dst = reinterpret_cast<float4*>(grads) + (j < split ? off0 + j : off1 + (j - split));
        if (accumulate) {
            const float4 o = *dst;
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        *dst = s;
    }
}

__device__ __forceinline__ void sum_slots_16x16(const float* __restrict__ partials, float* __restrict__ grads, int B, int lanes,
                                                int split = 0x7fffffff, int off0 = 0, int off1 = 0, bool accumulate = false) {
    sum_slots_16x16_at((int)blockIdx.x, partials, grads, B, lanes, split, off0, off1, accumulate);
}

// layer-split encoder slots (encoder_dw2.hip): float4 geometry of the two slot kinds inside a full ENC_PART gradient buffer
constexpr int DW2_L0_LANES = (EH * EH + EH) / 4, DW2_L0_SPLIT = EH * EH / 4, DW2_L0_OFF1 = (2 * EH * EH + 1024) / 4;
constexpr int DW2_L1_LANES = (EH * EH + 1024 + 2 * EH) / 4, DW2_L1_SPLIT = (EH * EH + 1024) / 4, DW2_L1_OFF0 = EH * EH / 4,
              DW2_L1_OFF1 = (2 * EH * EH + 1024 + EH) / 4;

}  // namespace piml
