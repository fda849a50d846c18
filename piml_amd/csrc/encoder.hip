// Fused PINNSF encoder on the f32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 = an fmaf chain).
//
// Reference arithmetic: src/models/model.py:40-65 (MLP = Linear/ReLU chain), :1271-1283 (ped/obs encoder ->
// processor -> sum over the k neighbours; with >= 2 processor "layers" the processor is 2 * x).
// One encoder = Linear(in<=8 -> 128) ReLU Linear(128 -> 128) ReLU Linear(128 -> 128), applied to the (agents*k)
// neighbour rows; msgs = scale * output.  These three layers are 97 % of the network's FLOPs
// (2 * 65 536 rows * 33.5 k MAC at the 4096-agent scene), run until round 1 as nine library GEMMs + glue passes.
//
// Formulation (everything is the TRANSPOSED product, features on the MFMA's M axis, data rows on its N axis):
//     D[f_out][row] = sum_f W[f_out][f] * act[row][f]          A = weights, B = activations^T
// A 32x32 accumulator then holds, in lane (row j = lane & 31, half h = lane >> 5), register r, the feature
// f(r, h) = (r & 3) + 8 (r >> 2) + 4 h of row j -- and that register IS the B operand of the next layer's MFMA
// (B[k][j]: k = h), provided the A operand carries the weights of the matching input features:
//     A-fragment (blk, bp, q, u): lane (i, h) = W[32 blk + i][32 bp + 8 q + 4 h + u].
// So activations never leave the registers between layers (no LDS round trip, no barrier); the weights are staged
// once per workgroup into LDS as ready-made A-fragments (one ds_read_b128 = the fragments of 4 k-steps) and a wave
// streams them past its 32-row tile.  Bias rides in as the accumulator's initial value, ReLU is a v_max.
// Backward: enc_bwd_dx runs the same chain with W^T (g_h = W^T g_pre, masked by the saved activations) and leaves
// the pre-activation gradients g2, g1 in HBM; enc_bwd_dw is the split-K product dW = G^T H over row slabs
// (A = G^T read straight from row-major G, B = H), per-workgroup partials, one piml_sum_leading over them.
#include <cstdlib>

#include "common.hpp"
#include "encoder.hpp"
#include "pack.hpp"
#include "reduce.hpp"
#include "philox.hpp"
#include "stages.hpp"
#include "../../include/piml_hip.h"

namespace piml {

// Dropout of the processor output (piml_encoder_branch.keep_bits, (rows, 4) dwords: bit c & 31 of word c >> 5 = keep
// feature c of the row).  Lane (row, h), block blk, register r holds feature 32 blk + (r & 3) + 8 (r >> 2) + 4 h.
__device__ __forceinline__ void keep_block_f32(f32x16& a, unsigned word, int h) {
    const unsigned m = word >> (4 * h);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = ((m >> ((r & 3) + 8 * (r >> 2))) & 1u) ? a[r] : 0.f;
}
__device__ __forceinline__ unsigned keep_word(const unsigned* __restrict__ keep, long long row, int blk, bool valid) {
    return valid ? keep[row * 4 + blk] : 0u;
}

__global__ __launch_bounds__(256) void enc_pack_kernel(EncArgs A) {
    const int b = blockIdx.y;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < PACK_FLOATS) J.packed[e] = pack_value(J, e);
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// LDS (floats): W2 fragments 16384 | W3 fragments 16384 | W1 fragments [blk 4][s 4][lane 64] 1024 | b1 b2 b3 384
constexpr int FWD_LDS_FLOATS = 16384 * 2 + 1024 + 384;      // = PACK_FWD

__global__ __launch_bounds__(ENC_THREADS) void enc_fwd_kernel(EncArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int nwg = b ? (int)gridDim.x - A.wg_split : (A.nbr > 1 ? A.wg_split : (int)gridDim.x);
    const long long R = J.rows;
    const int IN = J.in_dim;
    const long long ntiles = (R + 31) >> 5;
    const long long first = (long long)((int)blockIdx.x - wg0) * ENC_WAVES + wave;
    const long long stride = (long long)nwg * ENC_WAVES;
    if (A.zero)
        for (int e = blockIdx.x * ENC_THREADS + tid; e < A.zero_n; e += gridDim.x * ENC_THREADS) A.zero[e] = 0.f;
    if ((long long)((int)blockIdx.x - wg0) * ENC_WAVES >= ntiles) return;        // whole workgroup idle

    const float4* W2f = reinterpret_cast<const float4*>(lds);
    const float4* W3f = reinterpret_cast<const float4*>(lds + 16384);
    const float* W1f = lds + 32768;
    const float* bias = lds + 32768 + 1024;
    stage_linear<PACK_FWD>(lds, J.packed, tid);
    __syncthreads();

    const int j = lane & 31, h = lane >> 5;
    for (long long tile = first; tile < ntiles; tile += stride) {
        const long long row = tile * 32 + j;
        const bool valid = row < R;
        float xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = 2 * s + h;
            xb[s] = (valid && c < IN) ? J.x[row * IN + c] : 0.f;
        }
        f32x16 a1[4], a2[4];
        // ---- layer 1: K = in_dim (padded to 8) ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + feat0(blk, q, h));
                a1[blk][4 * q + 0] = bq.x; a1[blk][4 * q + 1] = bq.y; a1[blk][4 * q + 2] = bq.z; a1[blk][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) a1[blk] = mfma32(W1f[(blk * 4 + s) * 64 + lane], xb[s], a1[blk]);
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[blk][r] = fmaxf(a1[blk][r], 0.f);
        }
        if (J.h1 && valid) {
            float* o = J.h1 + row * EH;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4_stream(o + feat0(blk, q, h), a1[blk][4 * q], a1[blk][4 * q + 1], a1[blk][4 * q + 2], a1[blk][4 * q + 3]);
        }
        // ---- layer 2 ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + 128 + feat0(blk, q, h));
                a2[blk][4 * q + 0] = bq.x; a2[blk][4 * q + 1] = bq.y; a2[blk][4 * q + 2] = bq.z; a2[blk][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int bp = 0; bp < 4; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = W2f[((blk * 4 + bp) * 4 + q) * 64 + lane];
                    a2[blk] = mfma32(w.x, a1[bp][4 * q + 0], a2[blk]);
                    a2[blk] = mfma32(w.y, a1[bp][4 * q + 1], a2[blk]);
                    a2[blk] = mfma32(w.z, a1[bp][4 * q + 2], a2[blk]);
                    a2[blk] = mfma32(w.w, a1[bp][4 * q + 3], a2[blk]);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[blk][r] = fmaxf(a2[blk][r], 0.f);
        }
        if (J.h2 && valid) {
            float* o = J.h2 + row * EH;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4_stream(o + feat0(blk, q, h), a2[blk][4 * q], a2[blk][4 * q + 1], a2[blk][4 * q + 2], a2[blk][4 * q + 3]);
        }
        // ---- layer 3 (no activation), msgs = scale * output; a1 is dead and reused ----
        const float scale = J.scale;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + 256 + feat0(blk, q, h));
                a1[blk][4 * q + 0] = bq.x; a1[blk][4 * q + 1] = bq.y; a1[blk][4 * q + 2] = bq.z; a1[blk][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int bp = 0; bp < 4; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = W3f[((blk * 4 + bp) * 4 + q) * 64 + lane];
                    a1[blk] = mfma32(w.x, a2[bp][4 * q + 0], a1[blk]);
                    a1[blk] = mfma32(w.y, a2[bp][4 * q + 1], a1[blk]);
                    a1[blk] = mfma32(w.z, a2[bp][4 * q + 2], a1[blk]);
                    a1[blk] = mfma32(w.w, a2[bp][4 * q + 3], a1[blk]);
                }
            if (J.keep_bits) keep_block_f32(a1[blk], keep_word(J.keep_bits, row, blk, valid), h);
            if (valid) {
                float* o = J.msgs + row * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4_stream(o + feat0(blk, q, h), scale * a1[blk][4 * q], scale * a1[blk][4 * q + 1], scale * a1[blk][4 * q + 2], scale * a1[blk][4 * q + 3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward for FEW rows (rollouts of real clips: 100 .. 1000 agents = 60 .. 500 tiles for 2048 wave slots).  With one tile
// per wave most SIMDs hold one wave or none, and a lone wave issues a 32x32x2 MFMA only every ~135 cycles: the 528-MFMA
// chain takes 40 us whatever the row count.  Here FOUR waves share a tile -- wave (t, blk) computes output block blk of
// every layer of tile t, two tiles per workgroup -- and hand the activations over through LDS in accumulator layout
// (lane = row, register = feature = the B-operand layout of the next layer).  Chain per wave: 4 + 64 + 64 MFMAs.  The
// weight fragments come straight from the packed image (a wave uses each once: no staging phase); every accumulator sees
// the k-steps in the order of enc_fwd_kernel, so the outputs are bitwise identical.
// ---------------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(512) void enc_fwd_split_kernel(EncArgs A, int pairs0) {
    __shared__ float exch[2][2][4][16][64];            // [layer buffer][tile][block][register][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x >= pairs0 ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int t = wave >> 2, blk = wave & 3;
    const long long R = J.rows;
    const int IN = J.in_dim;
    const long long tile = ((long long)blockIdx.x - (b ? pairs0 : 0)) * 2 + t;
    const int j = lane & 31, h = lane >> 5;
    const long long row = tile * 32 + j;
    const bool valid = row < R;
    if (A.zero)
        for (int e = blockIdx.x * 512 + tid; e < A.zero_n; e += gridDim.x * 512) A.zero[e] = 0.f;
    const float4* W2g = reinterpret_cast<const float4*>(J.packed);
    const float4* W3g = reinterpret_cast<const float4*>(J.packed + 16384);
    const float* W1g = J.packed + 32768;
    const float* bias = J.packed + 32768 + 1024;
    float4 wf[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) wf[g] = W2g[(blk * 16 + g) * 64 + lane];           // in flight during layer 1
    float xb[4], w1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 2 * s + h;
        xb[s] = (valid && c < IN) ? J.x[(valid ? row : 0) * IN + c] : 0.f;
        w1[s] = W1g[(blk * 4 + s) * 64 + lane];
    }
    float4 bq[3][4];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[l][q] = *reinterpret_cast<const float4*>(bias + 128 * l + feat0(blk, q, h));
    f32x16 acc;
    auto init = [&](int l) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[4 * q] = bq[l][q].x; acc[4 * q + 1] = bq[l][q].y; acc[4 * q + 2] = bq[l][q].z; acc[4 * q + 3] = bq[l][q].w; }
    };
    auto store = [&](float* dst, float sc) {
        if (dst && valid) {
            float* o = dst + row * EH;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                store4_stream(o + feat0(blk, q, h), sc * acc[4 * q], sc * acc[4 * q + 1], sc * acc[4 * q + 2], sc * acc[4 * q + 3]);
        }
    };
    // ---- layer 1 ----
    init(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma32(w1[s], xb[s], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = fmaxf(acc[r], 0.f); exch[0][t][blk][r][lane] = acc[r]; }
    if (J.h1 && valid) {
        float* o = J.h1 + row * EH;
#pragma unroll
        for (int q = 0; q < 4; ++q) store4_stream(o + feat0(blk, q, h), acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    __syncthreads();
    // ---- layers 2 and 3 ----
#pragma unroll
    for (int l = 1; l < 3; ++l) {
        float in[4][16];
#pragma unroll
        for (int bp = 0; bp < 4; ++bp)
#pragma unroll
            for (int r = 0; r < 16; ++r) in[bp][r] = exch[l - 1][t][bp][r][lane];
        init(l);
        float4 wn[16];
        if (l == 1) {
#pragma unroll
            for (int g = 0; g < 16; ++g) wn[g] = W3g[(blk * 16 + g) * 64 + lane];   // next layer's, under this layer's MFMAs
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int bp = g >> 2, q = g & 3;
            acc = mfma32(wf[g].x, in[bp][4 * q + 0], acc);
            acc = mfma32(wf[g].y, in[bp][4 * q + 1], acc);
            acc = mfma32(wf[g].z, in[bp][4 * q + 2], acc);
            acc = mfma32(wf[g].w, in[bp][4 * q + 3], acc);
        }
        if (l == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = fmaxf(acc[r], 0.f); exch[1][t][blk][r][lane] = acc[r]; }
            store(J.h2, 1.f);
#pragma unroll
            for (int g = 0; g < 16; ++g) wf[g] = wn[g];
            __syncthreads();
        } else {
            if (DROP) keep_block_f32(acc, keep_word(J.keep_bits, row, blk, valid), h);
            store(J.msgs, J.scale);
        }
    }
}

// pooled[a][c] = sum over the k rows of agent a of msgs (src/models/model.py:1283 `.sum(dim=-2)`), float4 per thread
__global__ __launch_bounds__(256) void enc_ksum_kernel(const float4* __restrict__ msgs, long long agents, int k,
                                                       float4* __restrict__ pooled) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= agents * (EH / 4)) return;
    const long long a = t / (EH / 4);
    const int c = (int)(t % (EH / 4));
    const float4* p = msgs + a * k * (EH / 4) + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < k; ++i) {
        const float4 v = p[(size_t)i * (EH / 4)];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    pooled[t] = s;
}

// ---------------------------------------------------------------------------------------------------------
// backward, part 1: the dX chain.  g3[row] = scale * (g_pooled[row / k] + g_msgs[row]) (either may be absent);
// g_h2 = W3^T g3, g2 = g_h2 * [h2 > 0] (stored); g_h1 = W2^T g2, g1 = g_h1 * [h1 > 0] (stored); g_x = W1^T g1.
// ---------------------------------------------------------------------------------------------------------
// LDS (floats): W3^T fragments 16384 | W2^T fragments 16384 | W1 rows [f 128][8] = 1024
constexpr int DX_LDS_FLOATS = 16384 * 2 + 1024;               // = PACK_DX

__global__ __launch_bounds__(ENC_THREADS) void enc_bwd_dx_kernel(EncArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int nwg = b ? (int)gridDim.x - A.wg_split : (A.nbr > 1 ? A.wg_split : (int)gridDim.x);
    const long long R = J.rows;
    const int IN = J.in_dim, K = J.k;
    const long long ntiles = (R + 31) >> 5;
    const long long first = (long long)((int)blockIdx.x - wg0) * ENC_WAVES + wave;
    const long long stride = (long long)nwg * ENC_WAVES;
    if ((long long)((int)blockIdx.x - wg0) * ENC_WAVES >= ntiles) return;

    const float4* W3t = reinterpret_cast<const float4*>(lds);
    const float4* W2t = reinterpret_cast<const float4*>(lds + 16384);
    const float4* W1r = reinterpret_cast<const float4*>(lds + 32768);      // row f = float4 2 f, 2 f + 1
    const bool want_gx = J.g_x != nullptr;
    stage_linear<PACK_DX>(lds, J.packed + PACK_FWD, tid);
    __syncthreads();

    const int j = lane & 31, h = lane >> 5;
    const float scale = J.scale;
    for (long long tile = first; tile < ntiles; tile += stride) {
        const long long row = tile * 32 + j;
        const bool valid = row < R;
        f32x16 g[4], d[4];
        // ---- g3 in registers ----
        {
            const float* gp = (J.g_pooled && valid) ? J.g_pooled + (row / K) * EH : nullptr;
            const float* gm = (J.g_msgs && valid) ? J.g_msgs + row * EH : nullptr;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gp) v = *reinterpret_cast<const float4*>(gp + feat0(blk, q, h));
                    if (gm) {
                        const float4 m = *reinterpret_cast<const float4*>(gm + feat0(blk, q, h));
                        v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
                    }
                    g[blk][4 * q + 0] = scale * v.x; g[blk][4 * q + 1] = scale * v.y;
                    g[blk][4 * q + 2] = scale * v.z; g[blk][4 * q + 3] = scale * v.w;
                }
            if (J.keep_bits) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) keep_block_f32(g[blk], keep_word(J.keep_bits, row, blk, valid), h);
            }
        }
        // ---- g_h2 = W3^T g3, masked by h2 -> g2 ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            __builtin_amdgcn_sched_barrier(0);
            float4 hv[4];                                   // this block's h2 values: in flight during the MFMAs
            {
                const float* hp = J.h2 + (valid ? row : 0) * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4*>(hp + feat0(blk, q, h));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) d[blk][r] = 0.f;
#pragma unroll
            for (int bp = 0; bp < 4; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = W3t[((blk * 4 + bp) * 4 + q) * 64 + lane];
                    d[blk] = mfma32(w.x, g[bp][4 * q + 0], d[blk]);
                    d[blk] = mfma32(w.y, g[bp][4 * q + 1], d[blk]);
                    d[blk] = mfma32(w.z, g[bp][4 * q + 2], d[blk]);
                    d[blk] = mfma32(w.w, g[bp][4 * q + 3], d[blk]);
                }
            if (valid) {
                float* o = J.g2 + row * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = hv[q];
                    d[blk][4 * q + 0] = a.x > 0.f ? d[blk][4 * q + 0] : 0.f;
                    d[blk][4 * q + 1] = a.y > 0.f ? d[blk][4 * q + 1] : 0.f;
                    d[blk][4 * q + 2] = a.z > 0.f ? d[blk][4 * q + 2] : 0.f;
                    d[blk][4 * q + 3] = a.w > 0.f ? d[blk][4 * q + 3] : 0.f;
                    store4_stream(o + feat0(blk, q, h), d[blk][4 * q], d[blk][4 * q + 1], d[blk][4 * q + 2], d[blk][4 * q + 3]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) d[blk][r] = 0.f;
            }
        }
        // ---- g_h1 = W2^T g2, masked by h1 -> g1 (g is dead and reused) ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            __builtin_amdgcn_sched_barrier(0);
            float4 hv[4];
            {
                const float* hp = J.h1 + (valid ? row : 0) * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4*>(hp + feat0(blk, q, h));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) g[blk][r] = 0.f;
#pragma unroll
            for (int bp = 0; bp < 4; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = W2t[((blk * 4 + bp) * 4 + q) * 64 + lane];
                    g[blk] = mfma32(w.x, d[bp][4 * q + 0], g[blk]);
                    g[blk] = mfma32(w.y, d[bp][4 * q + 1], g[blk]);
                    g[blk] = mfma32(w.z, d[bp][4 * q + 2], g[blk]);
                    g[blk] = mfma32(w.w, d[bp][4 * q + 3], g[blk]);
                }
            if (valid) {
                float* o = J.g1 + row * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = hv[q];
                    g[blk][4 * q + 0] = a.x > 0.f ? g[blk][4 * q + 0] : 0.f;
                    g[blk][4 * q + 1] = a.y > 0.f ? g[blk][4 * q + 1] : 0.f;
                    g[blk][4 * q + 2] = a.z > 0.f ? g[blk][4 * q + 2] : 0.f;
                    g[blk][4 * q + 3] = a.w > 0.f ? g[blk][4 * q + 3] : 0.f;
                    store4_stream(o + feat0(blk, q, h), g[blk][4 * q], g[blk][4 * q + 1], g[blk][4 * q + 2], g[blk][4 * q + 3]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) g[blk][r] = 0.f;
            }
        }
        // ---- g_x = W1^T g1 on the vector pipe: the product has in_dim <= 8 output columns, an MFMA would pad them to 32
        // (64 of the tile's 576 matrix instructions for 1/5 of a block).  Lane (row, h) holds half of the row's g1
        // features: 8 partial dot products over them (explicit FMAs: the file is built with -ffp-contract=off), the other
        // half arrives with one cross-half exchange.
        if (want_gx) {
            float gx[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) gx[c] = 0.f;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_barrier(0);      // 8 LDS reads in flight per group, not all 128 (spills)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int f = feat0(blk, q, h) + u;
                        const float4 wa = W1r[2 * f], wb = W1r[2 * f + 1];
                        const float v = g[blk][4 * q + u];
                        gx[0] = __fmaf_rn(wa.x, v, gx[0]); gx[1] = __fmaf_rn(wa.y, v, gx[1]);
                        gx[2] = __fmaf_rn(wa.z, v, gx[2]); gx[3] = __fmaf_rn(wa.w, v, gx[3]);
                        gx[4] = __fmaf_rn(wb.x, v, gx[4]); gx[5] = __fmaf_rn(wb.y, v, gx[5]);
                        gx[6] = __fmaf_rn(wb.z, v, gx[6]); gx[7] = __fmaf_rn(wb.w, v, gx[7]);
                    }
                }
#pragma unroll
            for (int c = 0; c < 8; ++c) gx[c] += __shfl_xor(gx[c], 32, 64);
            if (valid) {       // half h stores input features 4 h .. 4 h + 3
                float* o = J.g_x + row * IN + 4 * h;
                const int left = IN - 4 * h;       // scalar selects (an array select goes through scratch)
                const float s0 = h ? gx[4] : gx[0], s1 = h ? gx[5] : gx[1], s2 = h ? gx[6] : gx[2], s3 = h ? gx[7] : gx[3];
                if (left > 0) o[0] = s0;
                if (left > 1) o[1] = s1;
                if (left > 2) o[2] = s2;
                if (left > 3) o[3] = s3;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// dX chain for FEW rows: four waves per tile like enc_fwd_split_kernel (same reason, same gain).  Wave (t, blk) owns
// output block blk of g2 and of g1; every wave builds the whole g3 of its tile's rows itself (loads), the g2 blocks
// travel through LDS; g_x is computed by the blk = 0 wave of the tile from all four g1 blocks in enc_bwd_dx_kernel's
// order.  Bitwise identical to enc_bwd_dx_kernel.
// ---------------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(512) void enc_bwd_dx_split_kernel(EncArgs A, int pairs0) {
    __shared__ float exch[2][2][4][16][64];            // [g2 | g1][tile][block][register][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x >= pairs0 ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int t = wave >> 2, blk = wave & 3;
    const long long R = J.rows;
    const int IN = J.in_dim, K = J.k;
    const long long tile = ((long long)blockIdx.x - (b ? pairs0 : 0)) * 2 + t;
    const int j = lane & 31, h = lane >> 5;
    const long long row = tile * 32 + j;
    const bool valid = row < R;
    const long long rr = valid ? row : 0;
    const float scale = J.scale;
    const float4* W3g = reinterpret_cast<const float4*>(J.packed + PACK_FWD);
    const float4* W2g = reinterpret_cast<const float4*>(J.packed + PACK_FWD + 16384);
    const float4* W1r = reinterpret_cast<const float4*>(J.packed + PACK_FWD + 32768);     // row f = float4 2 f, 2 f + 1
    float4 wf[16], hv[4];
#pragma unroll
    for (int g = 0; g < 16; ++g) wf[g] = W3g[(blk * 16 + g) * 64 + lane];
#pragma unroll
    for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4*>(J.h2 + rr * EH + feat0(blk, q, h));
    // ---- g3 of the tile's rows (all four blocks: the B operand of this wave's 64 MFMAs) ----
    float in[4][16];
    {
        const float* gp = J.g_pooled ? J.g_pooled + (rr / K) * EH : nullptr;
        const float* gm = J.g_msgs ? J.g_msgs + rr * EH : nullptr;
#pragma unroll
        for (int bp = 0; bp < 4; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gp) v = *reinterpret_cast<const float4*>(gp + feat0(bp, q, h));
                if (gm) {
                    const float4 m = *reinterpret_cast<const float4*>(gm + feat0(bp, q, h));
                    v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
                }
                in[bp][4 * q + 0] = valid ? scale * v.x : 0.f; in[bp][4 * q + 1] = valid ? scale * v.y : 0.f;
                in[bp][4 * q + 2] = valid ? scale * v.z : 0.f; in[bp][4 * q + 3] = valid ? scale * v.w : 0.f;
            }
        if (DROP) {
#pragma unroll
            for (int bp = 0; bp < 4; ++bp) {
                const unsigned m = keep_word(J.keep_bits, rr, bp, valid) >> (4 * h);
#pragma unroll
                for (int r = 0; r < 16; ++r) in[bp][r] = ((m >> ((r & 3) + 8 * (r >> 2))) & 1u) ? in[bp][r] : 0.f;
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {                      // l = 0: g2 = (W3^T g3) * [h2 > 0];  l = 1: g1 = (W2^T g2) * [h1 > 0]
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float4 wn[16], hn[4];
        if (l == 0) {                                  // the next layer's operands travel under this layer's MFMAs
#pragma unroll
            for (int g = 0; g < 16; ++g) wn[g] = W2g[(blk * 16 + g) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) hn[q] = *reinterpret_cast<const float4*>(J.h1 + rr * EH + feat0(blk, q, h));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int bp = g >> 2, q = g & 3;
            acc = mfma32(wf[g].x, in[bp][4 * q + 0], acc);
            acc = mfma32(wf[g].y, in[bp][4 * q + 1], acc);
            acc = mfma32(wf[g].z, in[bp][4 * q + 2], acc);
            acc = mfma32(wf[g].w, in[bp][4 * q + 3], acc);
        }
        float* dst = (l == 0 ? J.g2 : J.g1) + rr * EH;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = hv[q];
            acc[4 * q + 0] = (valid && a.x > 0.f) ? acc[4 * q + 0] : 0.f;
            acc[4 * q + 1] = (valid && a.y > 0.f) ? acc[4 * q + 1] : 0.f;
            acc[4 * q + 2] = (valid && a.z > 0.f) ? acc[4 * q + 2] : 0.f;
            acc[4 * q + 3] = (valid && a.w > 0.f) ? acc[4 * q + 3] : 0.f;
            if (valid) store4_stream(dst + feat0(blk, q, h), acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) exch[l][t][blk][r][lane] = acc[r];
        __syncthreads();
        if (l == 0) {
#pragma unroll
            for (int bp = 0; bp < 4; ++bp)
#pragma unroll
                for (int r = 0; r < 16; ++r) in[bp][r] = exch[0][t][bp][r][lane];
#pragma unroll
            for (int g = 0; g < 16; ++g) wf[g] = wn[g];
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[q] = hn[q];
        }
    }
    // ---- g_x = W1^T g1, by one wave of the tile, in enc_bwd_dx_kernel's order ----
    if (blk == 0 && J.g_x) {
        float gx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) gx[c] = 0.f;
#pragma unroll
        for (int bp = 0; bp < 4; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = feat0(bp, q, h) + u;
                    const float4 wa = W1r[2 * f], wb = W1r[2 * f + 1];
                    const float v = exch[1][t][bp][4 * q + u][lane];
                    gx[0] = __fmaf_rn(wa.x, v, gx[0]); gx[1] = __fmaf_rn(wa.y, v, gx[1]);
                    gx[2] = __fmaf_rn(wa.z, v, gx[2]); gx[3] = __fmaf_rn(wa.w, v, gx[3]);
                    gx[4] = __fmaf_rn(wb.x, v, gx[4]); gx[5] = __fmaf_rn(wb.y, v, gx[5]);
                    gx[6] = __fmaf_rn(wb.z, v, gx[6]); gx[7] = __fmaf_rn(wb.w, v, gx[7]);
                }
            }
#pragma unroll
        for (int c = 0; c < 8; ++c) gx[c] += __shfl_xor(gx[c], 32, 64);
        if (valid) {
            float* o = J.g_x + row * IN + 4 * h;
            const int left = IN - 4 * h;
            const float s0 = h ? gx[4] : gx[0], s1 = h ? gx[5] : gx[1], s2 = h ? gx[6] : gx[2], s3 = h ? gx[7] : gx[3];
            if (left > 0) o[0] = s0;
            if (left > 1) o[1] = s1;
            if (left > 2) o[2] = s2;
            if (left > 3) o[3] = s3;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward, part 2: weight gradients as split-K products over a workgroup's row slab (K = rows).
//   dW3[f3][f2] = sum_rows g3[row][f3] h2[row][f2],  dW2 = g2^T h1,  dW1 = g1^T x,  db_l = column sums of g_l.
// Wave w: M block mb = w & 3 (32 rows of dW), N half nh = w >> 2 (64 columns), every row of the slab.
// A: lane (i, h) = G[row 2 s + h][32 mb + i] (128-B coalesced, straight from row-major G);
// B: lane (n, h) = H[row 2 s + h][64 nh + 2 n + {0, 1}] (float2), so accumulator u holds dW[..][64 nh + 2 n + u].
// Partial of workgroup p (floats): dW3 16384 | dW2 16384 | dW1 128 x 8 | db3 | db2 | db1.
// ---------------------------------------------------------------------------------------------------------
// The workgroup stages batches of DW_ROWS rows of the five operand arrays (g3 computed on the fly, g2, g1, h2, h1; plus
// the x rows) into LDS with 16-byte loads -- every global byte is fetched once per workgroup, 1 KiB per wave-instruction --
// double-buffered: the loads of batch t+1 are in flight during the MFMAs of batch t, one barrier per batch.
constexpr int DW_ROWS = 16;                                  // rows per batch = 8 k-steps
constexpr int DW_BUF = 5 * DW_ROWS * EH + DW_ROWS * 8;       // floats of one LDS buffer
constexpr int DW_LDS_FLOATS = 2 * DW_BUF;

template <bool POOL, bool MSGS>
__global__ __launch_bounds__(ENC_THREADS) void enc_bwd_dw_kernel(EncArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int nwg = b ? (int)gridDim.x - A.wg_split : (A.nbr > 1 ? A.wg_split : (int)gridDim.x);
    const unsigned p = (unsigned)((int)blockIdx.x - wg0);
    const unsigned R = (unsigned)J.rows;                   // rows < 2^24 (checked on the host): 32-bit indexing
    const unsigned IN = (unsigned)J.in_dim, K = (unsigned)J.k;
    const unsigned kmagic = (unsigned)((0x100000000ull + K - 1) / K);      // row / K == umulhi(row, kmagic) for row * K < 2^32
    unsigned slab = (R + nwg - 1) / nwg;
    slab = (slab + 1) & ~1u;
    const unsigned r0 = p * slab < R ? p * slab : R;
    const unsigned r1 = r0 + slab < R ? r0 + slab : R;
    const int mb = wave & 3, nh = wave >> 2;
    const unsigned i = lane & 31, h = lane >> 5;
    const float scale = J.scale;

    f32x16 c3[2], c2[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { c3[0][r] = 0.f; c3[1][r] = 0.f; c2[0][r] = 0.f; c2[1][r] = 0.f; }
    float s3 = 0.f, s2 = 0.f, s1 = 0.f;
    // dW1 (128 x in_dim <= 8) on the vector pipe: lane (i, h) of wave (mb, nh) accumulates dW1[32 mb + i][4 nh .. 4 nh + 3]
    // over the rows of parity h from the g1 value it reads for db1 anyway (an MFMA would pad the 8 columns to 32)
    float w1a = 0.f, w1b = 0.f, w1c = 0.f, w1d = 0.f;
    const unsigned fa = 32 * mb + i;               // A column (feature of G)
    const unsigned fb = 64 * nh + 2 * i;           // first of the two B columns (features of H)
    const float* __restrict__ gpool = J.g_pooled;
    const float* __restrict__ gmsg = J.g_msgs;
    const float* __restrict__ G2 = J.g2;
    const float* __restrict__ G1 = J.g1;
    const float* __restrict__ H2 = J.h2;
    const float* __restrict__ H1 = J.h1;
    const float* __restrict__ X = J.x;
    // staging role of this thread: row srow of the batch, float4 column sc4 of every array
    const unsigned srow = tid >> 5, sc4 = (tid & 31) * 4;
    const unsigned xrow = tid >> 3, xc = tid & 7;            // threads 0..127: the x rows
    struct Stage { float4 pool, msg, g2, g1, h2, h1; float x; bool ok; unsigned keep; };
    const unsigned* __restrict__ KB = J.keep_bits;
    auto stage_load = [&](unsigned rb) -> Stage {            // issue the global loads of the batch starting at row rb
        Stage S;
        const unsigned row = rb + srow;
        S.ok = row < r1;
        const unsigned ro = S.ok ? row : r0;                 // clamped: a readable row (r0 < R whenever a batch exists)
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        S.pool = POOL ? *reinterpret_cast<const float4*>(gpool + __umulhi(ro, kmagic) * EH + sc4) : z;
        S.msg = MSGS ? *reinterpret_cast<const float4*>(gmsg + ro * EH + sc4) : z;
        S.g2 = *reinterpret_cast<const float4*>(G2 + ro * EH + sc4);
        S.g1 = *reinterpret_cast<const float4*>(G1 + ro * EH + sc4);
        S.h2 = *reinterpret_cast<const float4*>(H2 + ro * EH + sc4);
        S.h1 = *reinterpret_cast<const float4*>(H1 + ro * EH + sc4);
        S.keep = KB ? (KB[ro * 4 + (sc4 >> 5)] >> (sc4 & 31)) : 0xfu;          // the four keep bits of this float4 column
        const unsigned xr = rb + xrow;
        S.x = (tid < DW_ROWS * 8 && xr < r1 && xc < IN) ? X[xr * IN + xc] : 0.f;
        return S;
    };
    auto stage_write = [&](const Stage S, float* buf) {     // registers -> LDS (rows past the slab are zeros)
        float4* d = reinterpret_cast<float4*>(buf + srow * EH + sc4);
        const float4 g3 = make_float4((S.keep & 1u) ? (S.pool.x + S.msg.x) * scale : 0.f, (S.keep & 2u) ? (S.pool.y + S.msg.y) * scale : 0.f,
                                      (S.keep & 4u) ? (S.pool.z + S.msg.z) * scale : 0.f, (S.keep & 8u) ? (S.pool.w + S.msg.w) * scale : 0.f);
        auto sel = [&](const float4 v) {             // component-wise (a float4 ?: becomes a select through scratch)
            return make_float4(S.ok ? v.x : 0.f, S.ok ? v.y : 0.f, S.ok ? v.z : 0.f, S.ok ? v.w : 0.f);
        };
        d[0] = sel(g3);
        d[1 * DW_ROWS * EH / 4] = sel(S.g2);
        d[2 * DW_ROWS * EH / 4] = sel(S.g1);
        d[3 * DW_ROWS * EH / 4] = sel(S.h2);
        d[4 * DW_ROWS * EH / 4] = sel(S.h1);
        if (tid < DW_ROWS * 8) buf[5 * DW_ROWS * EH + tid] = S.x;
    };
    auto compute = [&](const float* buf) {
#pragma unroll
        for (int ks = 0; ks < DW_ROWS / 2; ++ks) {
            const float* rowp = buf + (2 * ks + h) * EH;
            const float a3 = rowp[fa], a2 = rowp[DW_ROWS * EH + fa], a1 = rowp[2 * DW_ROWS * EH + fa];
            const float2 b3 = *reinterpret_cast<const float2*>(rowp + 3 * DW_ROWS * EH + fb);
            const float2 b2 = *reinterpret_cast<const float2*>(rowp + 4 * DW_ROWS * EH + fb);
            c3[0] = mfma32(a3, b3.x, c3[0]);
            c3[1] = mfma32(a3, b3.y, c3[1]);
            c2[0] = mfma32(a2, b2.x, c2[0]);
            c2[1] = mfma32(a2, b2.y, c2[1]);
            const float4 xr = *reinterpret_cast<const float4*>(buf + 5 * DW_ROWS * EH + (2 * ks + h) * 8 + 4 * nh);
            w1a = __fmaf_rn(a1, xr.x, w1a); w1b = __fmaf_rn(a1, xr.y, w1b);
            w1c = __fmaf_rn(a1, xr.z, w1c); w1d = __fmaf_rn(a1, xr.w, w1d);
            s3 += a3; s2 += a2; s1 += a1;
        }
    };
    if (r0 < r1) {
        const unsigned nb = (r1 - r0 + DW_ROWS - 1) / DW_ROWS;
        Stage S = stage_load(r0);
        stage_write(S, lds);
        __syncthreads();
        for (unsigned t = 0; t < nb; ++t) {
            float* cur = lds + (t & 1) * DW_BUF;
            float* nxt = lds + ((t + 1) & 1) * DW_BUF;
            S = stage_load(r0 + (t + 1) * DW_ROWS);         // past the slab: clamped + zeroed, written but never read
            // pinned: without the two fences the scheduler lifts stage_write's arithmetic on the freshly loaded registers
            // in between the first MFMAs, i.e. waits for the NEXT batch's loads (s_waitcnt vmcnt(0) after four MFMAs)
            // before this batch is computed -- one exposed memory round trip per batch instead of none
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            stage_write(S, nxt);
            __syncthreads();
        }
    }
    float* P = J.partials + (size_t)p * ENC_PART;
    // accumulator u, register r, lane (n, h): dW[32 mb + (r & 3) + 8 (r >> 2) + 4 h][64 nh + 2 n + u]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int orow = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * h;
        *reinterpret_cast<float2*>(P + (size_t)orow * EH + fb) = make_float2(c3[0][r], c3[1][r]);
        *reinterpret_cast<float2*>(P + 16384 + (size_t)orow * EH + fb) = make_float2(c2[0][r], c2[1][r]);
    }
    {   // dW1 row-major (128, in_dim) at the head of its 1024 floats
        w1a += __shfl_xor(w1a, 32, 64); w1b += __shfl_xor(w1b, 32, 64);
        w1c += __shfl_xor(w1c, 32, 64); w1d += __shfl_xor(w1d, 32, 64);
        const unsigned c0 = 4 * nh;
        float* o = P + 32768 + fa * IN + c0;
        if (h == 0) {
            if (c0 + 0 < IN) o[0] = w1a;
            if (c0 + 1 < IN) o[1] = w1b;
            if (c0 + 2 < IN) o[2] = w1c;
            if (c0 + 3 < IN) o[3] = w1d;
        }
    }
    if (nh == 0) {
        s3 += __shfl_xor(s3, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (h == 0) {
            P[32768 + 1024 + fa] = s3;
            P[32768 + 1024 + 128 + fa] = s2;
            P[32768 + 1024 + 256 + fa] = s1;
        }
    }
}

// grads = sum over the branch's partial slots (sum_slots_16x16, pack.hpp); blockIdx.y = branch
__global__ __launch_bounds__(256) void enc_reduce_kernel(EncArgs A, int lanes, int accumulate) {
    const int b = blockIdx.y;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int B = b ? 256 - A.wg_split : (A.nbr > 1 ? A.wg_split : 256);
    sum_slots_16x16(J.partials, J.grads, B, lanes, 0x7fffffff, 0, 0, accumulate != 0);
}

// the same for layer-split slots (encoder_dw2.hip): blockIdx.y = 2 * branch + layer
struct Reduce2Args { piml_encoder_branch br[2]; int n0[2], n1[2]; int accumulate; };
__global__ __launch_bounds__(256) void enc_reduce2_kernel(Reduce2Args A) {
    const int b = blockIdx.y >> 1, L = blockIdx.y & 1;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    if (L == 0) {
        if ((int)blockIdx.x * 16 < DW2_L0_LANES) sum_slots_16x16(J.partials, J.grads, A.n0[b], DW2_L0_LANES, DW2_L0_SPLIT, 0, DW2_L0_OFF1, A.accumulate != 0);
    } else {
        if ((int)blockIdx.x * 16 < DW2_L1_LANES)
            sum_slots_16x16(J.partials + (size_t)A.n0[b] * (DW2_L0_LANES * 4), J.grads, A.n1[b], DW2_L1_LANES, DW2_L1_SPLIT, DW2_L1_OFF0, DW2_L1_OFF1,
                            A.accumulate != 0);
    }
}

static int split_workgroups(const piml_encoder_branch* br, int nbr, int total, long long unit) {
    // workgroups for branch 0, proportional to the rows (each branch gets at least one)
    if (nbr < 2) return total;
    const double r0 = (double)br[0].rows, r1 = (double)br[1].rows;
    int w = (int)(total * r0 / (r0 + r1) + 0.5);
    if (w < 1) w = 1;
    if (w > total - 1) w = total - 1;
    (void)unit;
    return w;
}

static bool branch_ok(const piml_encoder_branch& b) {
    return b.rows > 0 && b.rows < (1ll << 24) && b.in_dim >= 1 && b.in_dim <= 8 && b.x && b.w1 && b.b1 && b.w2 && b.b2 && b.w3 && b.b3 &&
           b.packed;
}

static int fill_args(EncArgs& A, const piml_encoder_branch* br, int nbr) {
    A.nbr = nbr;
    A.zero = nullptr;
    A.zero_n = 0;
    A.gen_state = nullptr;
    for (int i = 0; i < nbr; ++i) A.br[i] = br[i];
    if (nbr == 1) A.br[1] = br[0];
    A.wg_split = split_workgroups(br, nbr, 256, 1);
    return 256;
}

}  // namespace piml

using namespace piml;

PIML_API int piml_encoder_partial_floats(void) { return ENC_PART; }

// workgroups the forward / backward launches use for these branches (partials must hold that many slots per
// branch: see piml_encoder_bwd)
PIML_API int piml_encoder_workgroups(const piml_encoder_branch* br, int nbr, int* wg_branch0) {
    if (!br || nbr < 1 || nbr > 2) return 0;
    const int total = 256;
    const int w0 = split_workgroups(br, nbr, total, 1);
    if (wg_branch0) *wg_branch0 = w0;
    return total;
}

PIML_API int piml_encoder_pack_floats(void) { return PACK_FLOATS; }

static int enc_check(const piml_encoder_branch* br, int nbr) {
    if (!br || nbr < 1 || nbr > 2) return hipErrorInvalidValue;
    for (int i = 0; i < nbr; ++i)
        if (!branch_ok(br[i])) return hipErrorInvalidValue;
    // dropout is a property of the launch: both branches carry keep_bits or neither does
    if (nbr == 2 && (br[0].keep_bits != nullptr) != (br[1].keep_bits != nullptr)) return hipErrorInvalidValue;
    return hipSuccess;
}

static int enc_bwd_check(const piml_encoder_branch* br, int nbr);
static bool enc_bwd_is_fused(const piml_encoder_branch* br, int nbr);


// dynamic LDS above 64 KB has to be enabled per kernel once per process
static int enc_set_lds(const void* f, int bytes) {
    return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int piml::enc_stage_pack(const piml_encoder_branch* br, int nbr, hipStream_t s) {
    if (!br || nbr < 1 || nbr > 2) return hipErrorInvalidValue;
    for (int i = 0; i < nbr; ++i) {          // the pack reads the weights only (no rows yet)
        const piml_encoder_branch& b = br[i];
        if (b.in_dim < 1 || b.in_dim > 8 || !b.w1 || !b.b1 || !b.w2 || !b.b2 || !b.w3 || !b.b3 || !b.packed) return hipErrorInvalidValue;
    }
    EncArgs A;
    A.nbr = nbr;
    A.br[0] = br[0];
    A.br[1] = br[nbr - 1];
    A.wg_split = 0;
    A.zero = nullptr;
    A.zero_n = 0;
    A.gen_state = nullptr;
    hipLaunchKernelGGL(enc_pack_kernel, dim3((PACK_FLOATS + 255) / 256, nbr), dim3(256), 0, s, A);
    return hipGetLastError();
}

// row tiles (of 32) up to which the forward uses enc_fwd_split_kernel (PIML_ENC_SPLIT_TILES, piml_encoder_split_tiles)
// (lone forward, graph-replayed, tools/sweep_lone_forward.py, with the wave-major tile order of round 5: four waves per tile 24.7 us
// against 26.8 at 488 tiles, 33.2 against 29.0 at 751 -- the bound was 1024 while the one-wave kernels filled a workgroup's eight waves
// before they used the next CU)
static long long split_tiles_default() { return getenv("PIML_ENC_SPLIT_TILES") ? atoll(getenv("PIML_ENC_SPLIT_TILES")) : 640; }
// ... and the bound of a TRAINING pass (every branch carries relu_mask: a backward follows).  Round 5: with the one-pass backward
// and the layer-1 slots the many-rows kernels win from the real clips' sizes on -- forward + backward of `pinnsf_m`, few-rows /
// many-rows kernels: 84 / 84 us at 122 agents (62 tiles), 96 / 94 at 1024, 132 / 111 at 2048 (one rank of the 8-way sharded
// 16384-agent scene), and 84 / 74, 96 / 83, 132 / 95 where the sums path applies (tools/sweep_split_tiles.py) -- while a lone
// forward (validation, rollouts below the pooled path's bound) still wants four waves per tile
static long long split_tiles_train_default() {
    if (getenv("PIML_ENC_SPLIT_TILES_TRAIN")) return atoll(getenv("PIML_ENC_SPLIT_TILES_TRAIN"));
    return getenv("PIML_ENC_SPLIT_TILES") ? atoll(getenv("PIML_ENC_SPLIT_TILES")) : 48;
}
static long long g_split_tiles = split_tiles_default();
static long long g_split_tiles_train = split_tiles_train_default();

// tiles >= 0: BOTH bounds (A/B, tests); -1: query the forward-only bound; -2: both back to their defaults
PIML_API long long piml_encoder_split_tiles(long long tiles) {
    const long long old = g_split_tiles;
    if (tiles >= 0) g_split_tiles = g_split_tiles_train = tiles;
    else if (tiles == -2) { g_split_tiles = split_tiles_default(); g_split_tiles_train = split_tiles_train_default(); }
    return old;
}
PIML_API long long piml_encoder_split_tiles_train(long long tiles) {
    const long long old = g_split_tiles_train;
    if (tiles >= 0) g_split_tiles_train = tiles;
    return old;
}
// the bound that applies to these branches: a training pass (all of them carry relu_mask: the one-pass backward follows), or a lone
// forward.  Until the wave-major tile order (round 5, encoder_x3.hip) steps with a dropout mask kept the lone forward's bound: the
// four-waves-per-tile kernels were ahead there up to ~512 tiles (fine-tuning step of 4 x 5 x 122 agents at p = 0.5 0.45 ms against 0.50).
// With it (tools/train_mode_steps.py, p = 0.5, PIML_ENC_SPLIT_TILES = 1024 / 48): pinnsf_m fine-tuning step 0.436 / 0.433 ms, pointwise
// step of 1024 rows 0.174 / 0.165, pinnsf_bm fine-tuning step 0.900 / 0.871 -- one bound for every training pass.
static long long split_bound(const piml_encoder_branch* br, int nbr) {
    for (int i = 0; i < nbr; ++i)
        if (!br[i].relu_mask) return g_split_tiles;
    return g_split_tiles_train;
}

// products of the two 128 x 128 layers: 1 = split bf16 products (encoder_x3.hip, f32-exact to one rounding per product),
// 0 = the f32 matrix-core instruction (PIML_ENC_PRODUCTS=f32, piml_encoder_products)
static int g_x3 = !(getenv("PIML_ENC_PRODUCTS") && getenv("PIML_ENC_PRODUCTS")[0] == 'f');

PIML_API int piml_encoder_products(int x3) {
    const int old = g_x3;
    if (x3 >= 0) g_x3 = x3 ? 1 : 0;
    return old;
}

// One-pass backward (encoder_bwd3.hip): where the layer-split weight gradients run AND the forward left the sign bits AND the
// branches carry the same kinds of upstream gradients, the dX chain and dW2 / dW1 / db2 / db1 are one launch that keeps g2 / g1
// on the CU; dW3 / db3 stay with the layer-0 workgroups of encoder_dw2.hip, now all of them.  PIML_ENC_FUSED_BWD=0 keeps the
// two-kernel form (A/B).
static int g_f3 = getenv("PIML_ENC_FUSED_BWD") ? (atoi(getenv("PIML_ENC_FUSED_BWD")) == 2 ? 2 : atoi(getenv("PIML_ENC_FUSED_BWD")) != 0) : 1;

PIML_API int piml_encoder_fused_bwd(int on) {
    const int old = g_f3;
    if (on >= 0) g_f3 = on == 2 ? 2 : (on ? 1 : 0);
    return old;
}

// Backward of the sums path (PIML_POOL_TRAIN): 2 (default) = two crews of four waves, two waves per SIMD (encoder_bwd5.hip);
// 1 = the one-wave-per-SIMD kernel of round 5 (encoder_bwd3.hip, SUMS).  Bitwise the same results (A/B).  PIML_ENC_SUMS_BWD=1 / 2
static int g_sums_bwd = getenv("PIML_ENC_SUMS_BWD") && atoi(getenv("PIML_ENC_SUMS_BWD")) == 1 ? 1 : 2;

PIML_API int piml_encoder_sums_bwd(int form) {
    const int old = g_sums_bwd;
    if (form == 1 || form == 2) g_sums_bwd = form;
    return old;
}

// PIML_ENC_FUSED_DW3=0: dW3 / db3 stay a launch of their own (the layer-0 workgroups of encoder_dw2.hip) behind the one-pass kernel
static int g_f3_dw3 = !(getenv("PIML_ENC_FUSED_DW3") && atoi(getenv("PIML_ENC_FUSED_DW3")) == 0);

// PIML_ENC_DX_SPLIT=f32: the few-rows dX chain on the f32 matrix instruction even with split products elsewhere (A/B)
static const bool g_dx_split_f32 = getenv("PIML_ENC_DX_SPLIT") && getenv("PIML_ENC_DX_SPLIT")[0] == 'f';

bool piml::enc_f32_images_needed() { return !g_x3 || g_dx_split_f32; }

// h1 may be absent (all branches) exactly when the backward runs without it: the dX chain on sign bits (relu_mask, more than
// piml_encoder_split_tiles() tiles, split products) and the weight gradients on the layer-split kernel, which recomputes it
static int enc_bwd_check(const piml_encoder_branch* br, int nbr) {
    if (int e = enc_check(br, nbr)) return e;
    bool no_h1 = false;
    for (int i = 0; i < nbr; ++i) {
        const piml_encoder_branch& b = br[i];
        if (!b.h2 || !b.partials || !b.grads || b.k < 1 || (!b.g_pooled && !b.g_msgs)) return hipErrorInvalidValue;
        no_h1 = no_h1 || !b.h1;
    }
    if (!enc_bwd_is_fused(br, nbr))           // g2 / g1 scratch: only the two-kernel form passes them through memory
        for (int i = 0; i < nbr; ++i)
            if (!br[i].g2 || !br[i].g1) return hipErrorInvalidValue;
    if (no_h1) {
        for (int i = 0; i < nbr; ++i)
            if (br[i].h1 || !br[i].relu_mask) return hipErrorInvalidValue;
        if (!enc_dw2_used(br, nbr, nullptr, nullptr)) return hipErrorInvalidValue;
    }
    return hipSuccess;
}

static int x3_ready() {
    static int state = -1;
    if (state < 0) state = enc_x3_set_attributes();
    return state;
}

int piml::enc_stage_fwd(const piml_encoder_branch* br, int nbr, hipStream_t s, float* zero, long long zero_n, bool msum) {
    if (int e = enc_check(br, nbr)) return e;
    for (int i = 0; i < nbr; ++i)
        if (!br[i].msgs && !msum) return hipErrorInvalidValue;          // (PIML_POOL_MSGS: message rows only where somebody reads them)
    EncArgs A;
    const int total = fill_args(A, br, nbr);
    static bool attr_set = false;
    if (!attr_set) {
        if (int e = enc_set_lds(reinterpret_cast<const void*>(enc_fwd_kernel), FWD_LDS_FLOATS * 4)) return e;
        attr_set = true;
    }
    if (zero && zero_n > 0) {
        if (zero_n >= (1ll << 31)) return hipErrorInvalidValue;       // a clear that cannot be honoured is an error, not a skip
        A.zero = zero;
        A.zero_n = (int)zero_n;
    }
    // Train-mode dropout drawn by this call (piml_encoder_branch.drop_state): p = 0.5 on the split-product kernels inside the
    // forward kernel, everything else by one generator launch for all branches in front of it
    if (br[0].drop_state) {
        for (int i = 0; i < nbr; ++i)
            if (br[i].drop_state != br[0].drop_state || br[i].drop_p != br[0].drop_p || !br[i].keep_bits) return hipErrorInvalidValue;
        if (g_x3 && br[0].drop_p == kFairP) {
            A.gen_state = br[0].drop_state;
        } else {
            long long rows[2];
            unsigned* bits[2];
            unsigned streams[2] = {0u, 1u};
            for (int i = 0; i < nbr; ++i) { rows[i] = br[i].rows; bits[i] = br[i].keep_bits; }
            if (int e = dropout_stage(br[0].drop_state, rows, bits, streams, nbr, EH, br[0].drop_p, s)) return e;
        }
    } else if (nbr > 1 && br[1].drop_state) {
        return hipErrorInvalidValue;
    }
    long long tiles[2] = {(br[0].rows + 31) / 32, nbr > 1 ? (br[1].rows + 31) / 32 : 0};
    if (msum) {          // PIML_POOL_MSGS: the agents' sums of the messages from the one-wave kernel's registers
        if (!enc_pool_msgs_ok(br, nbr)) return hipErrorInvalidValue;
        for (int i = 0; i < nbr; ++i)
            if (!br[i].sum_a || !br[i].sum_b) return hipErrorInvalidValue;
        if (int e = x3_ready()) return e;
        enc_x3_launch_fwd(A, total, br[0].keep_bits != nullptr, s, true);
        return hipGetLastError();
    }
    if (tiles[0] + tiles[1] <= split_bound(br, nbr)) {       // few rows: four waves per tile (see enc_fwd_split_kernel)
        const int pairs0 = (int)((tiles[0] + 1) / 2), pairs1 = (int)((tiles[1] + 1) / 2);
        if (g_x3) {
            if (int e = x3_ready()) return e;
            enc_x3_launch_fwd_split(A, pairs0, pairs1, br[0].keep_bits != nullptr, s);
            return hipGetLastError();
        }
        if (br[0].keep_bits) hipLaunchKernelGGL(enc_fwd_split_kernel<true>, dim3((unsigned)(pairs0 + pairs1)), dim3(512), 0, s, A, pairs0);
        else hipLaunchKernelGGL(enc_fwd_split_kernel<false>, dim3((unsigned)(pairs0 + pairs1)), dim3(512), 0, s, A, pairs0);
        return hipGetLastError();
    }
    if (g_x3) {
        if (int e = x3_ready()) return e;
        enc_x3_launch_fwd(A, total, br[0].keep_bits != nullptr, s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(enc_fwd_kernel, dim3(total), dim3(ENC_THREADS), FWD_LDS_FLOATS * 4, s, A);
    return hipGetLastError();
}

// The inference forward on pooled h2 serves: split products, no dropout, k = 6 or 10 neighbours per agent, whole agents, and
// more than 32 tiles (PIML_POOL_H2_MIN_TILES).  It also replaces the few-rows forward (four waves per tile): half the matrix
// work and no message rows weigh more than the shorter chains -- rollout frame 44 -> 40 us at 512 agents, 48 -> 42 at 1024,
// 65 -> 50 at 2048, 77 -> 58 at 4096; level at 122 (42 / 42, 46 / 38).
bool piml::enc_pool_h2_ok(const piml_encoder_branch* br, int nbr) {
    static const bool off = getenv("PIML_POOL_H2") && atoi(getenv("PIML_POOL_H2")) == 0;
    static const long long min_tiles = getenv("PIML_POOL_H2_MIN_TILES") ? atoll(getenv("PIML_POOL_H2_MIN_TILES")) : -1;
    if (off || !g_x3 || !br || nbr < 1 || nbr > 2) return false;
    long long tiles = 0;
    for (int i = 0; i < nbr; ++i) {
        const piml_encoder_branch& b = br[i];
        if ((b.k != 6 && b.k != 10) || b.rows <= 0 || b.rows % b.k || b.keep_bits || b.drop_state || b.in_dim > 8) return false;
        tiles += (b.rows + 31) / 32;
    }
    return tiles > (min_tiles >= 0 ? min_tiles : 32);
}

int piml::enc_stage_fwd_pool(const piml_encoder_branch* br, int nbr, hipStream_t s, float* zero, long long zero_n) {
    if (int e = enc_check(br, nbr)) return e;
    if (!enc_pool_h2_ok(br, nbr)) return hipErrorInvalidValue;
    for (int i = 0; i < nbr; ++i)
        if (!br[i].msgs || !br[i].h2) return hipErrorInvalidValue;
    EncArgs A;
    const int total = fill_args(A, br, nbr);
    if (zero && zero_n > 0) {
        if (zero_n >= (1ll << 31)) return hipErrorInvalidValue;
        A.zero = zero;
        A.zero_n = (int)zero_n;
    }
    if (int e = x3_ready()) return e;
    enc_x3_launch_fwd_pool(A, total, s);
    return hipGetLastError();
}

int piml::enc_stage_bwd_dx(const piml_encoder_branch* br, int nbr, hipStream_t s) {
    if (int e = enc_bwd_check(br, nbr)) return e;
    EncArgs A;
    const int total = fill_args(A, br, nbr);
    if (enc_bwd_is_fused(br, nbr)) {          // dX chain + dW2 / dW1 / db2 / db1, one workgroup (four waves, one per SIMD) per CU
        if (int e = x3_ready()) return e;
        static int ready = -1;
        if (ready < 0) {
            ready = enc_f3_set_attributes();
            if (!ready) ready = enc_f4_set_attributes();
        }
        if (ready) return ready;
        const int nA[2] = {nbr > 1 ? A.wg_split : total, nbr > 1 ? total - A.wg_split : 0};
        if (g_f3 == 2) enc_f4_launch(A, nA, nA, g_f3_dw3 != 0, s);      // eight waves of 16-feature blocks (encoder_bwd4.hip)
        else enc_f3_launch(A, nA, nA, g_f3_dw3 != 0, s);                // four waves of 32-feature blocks (encoder_bwd3.hip): the default
        return hipGetLastError();
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (int e = enc_set_lds(reinterpret_cast<const void*>(enc_bwd_dx_kernel), DX_LDS_FLOATS * 4)) return e;
        attr_set = true;
    }
    long long tiles[2] = {(br[0].rows + 31) / 32, nbr > 1 ? (br[1].rows + 31) / 32 : 0};
    // few rows: four waves per tile.  (Measured: 23 vs 43 us at 256 tiles, 25 vs 44 at 512, 44 vs 45 at 1024, 62 vs 47 at 1536:
    // the backward form breaks even earlier than the forward, every wave rebuilding the whole g3.)
    if ((tiles[0] + tiles[1]) * 4 <= split_bound(br, nbr) * 3) {
        const int pairs0 = (int)((tiles[0] + 1) / 2), pairs1 = (int)((tiles[1] + 1) / 2);
        if (g_x3 && !g_dx_split_f32) {
            if (int e = x3_ready()) return e;
            enc_x3_launch_bwd_dx_split(A, pairs0, pairs1, br[0].keep_bits != nullptr, s);
            return hipGetLastError();
        }
        if (br[0].keep_bits) hipLaunchKernelGGL(enc_bwd_dx_split_kernel<true>, dim3((unsigned)(pairs0 + pairs1)), dim3(512), 0, s, A, pairs0);
        else hipLaunchKernelGGL(enc_bwd_dx_split_kernel<false>, dim3((unsigned)(pairs0 + pairs1)), dim3(512), 0, s, A, pairs0);
        return hipGetLastError();
    }
    if (g_x3) {
        if (int e = x3_ready()) return e;
        // the sign bits exist iff the forward ran on enc_fwd_x3_kernel (same rule as enc_stage_fwd) and was given the buffer
        bool mask = tiles[0] + tiles[1] > split_bound(br, nbr);
        for (int i = 0; i < nbr; ++i) mask = mask && br[i].relu_mask != nullptr;
        enc_x3_launch_bwd_dx(A, total, mask, br[0].keep_bits != nullptr, s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(enc_bwd_dx_kernel, dim3(total), dim3(ENC_THREADS), DX_LDS_FLOATS * 4, s, A);
    return hipGetLastError();
}

// Layer-split weight gradients (encoder_dw2.hip): split products, the one-wave path (more than piml_encoder_split_tiles()
// tiles), both branches with the same kinds of upstream gradients / keep bits / h1, at least two workgroups per branch.
// PIML_ENC_DW2=0 keeps the slab kernel of encoder_dww.hip at every size (A/B).
static int g_dw2 = !(getenv("PIML_ENC_DW2") && atoi(getenv("PIML_ENC_DW2")) == 0);

PIML_API int piml_encoder_dw2(int on) {
    const int old = g_dw2;
    if (on >= 0) g_dw2 = on ? 1 : 0;
    return old;
}

static bool enc_f3_used(const piml_encoder_branch* br, int nbr) {
    if (!g_f3) return false;
    for (int i = 0; i < nbr; ++i)
        if (!br[i].relu_mask || (br[i].g_pooled != nullptr) != (br[0].g_pooled != nullptr) ||
            (br[i].g_msgs != nullptr) != (br[0].g_msgs != nullptr) || (!br[i].g_pooled && !br[i].g_msgs) ||
            (br[i].g_x != nullptr) != (br[0].g_x != nullptr) || br[i].rows >= (1ll << 22))      // (32-bit byte offsets into (rows, 128) arrays)
            return false;
    return true;
}

bool piml::enc_dw2_used(const piml_encoder_branch* br, int nbr, int* n0, int* n1) {
    if (!g_dw2 || !g_x3 || !br || nbr < 1 || nbr > 2) return false;
    long long tiles = 0;
    for (int i = 0; i < nbr; ++i) {
        tiles += (br[i].rows + 31) / 32;
        if ((br[i].keep_bits != nullptr) != (br[0].keep_bits != nullptr) || (br[i].h1 != nullptr) != (br[0].h1 != nullptr)) return false;
    }
    if (tiles <= split_bound(br, nbr)) return false;
    const int total = 256, w0 = split_workgroups(br, nbr, total, 1);
    const int w[2] = {w0, total - w0};
    const bool f3 = enc_f3_used(br, nbr);
    for (int i = 0; i < nbr; ++i) {
        if (w[i] < 2) return false;
        int a, c;
        enc_dw2_split(w[i], &a, &c);
        if (f3) a = c = w[i];        // every workgroup of the branch writes a layer-0 slot in one launch and a layer-1 slot in the other
        if (n0) n0[i] = a;
        if (n1) n1[i] = c;
    }
    return true;
}

// true: the backward of these branches is enc_f3_launch + the layer-0 half of enc_dw2_launch
static bool enc_bwd_is_fused(const piml_encoder_branch* br, int nbr) {
    return enc_dw2_used(br, nbr, nullptr, nullptr) && enc_f3_used(br, nbr);
}

int piml::enc_stage_bwd_dw(const piml_encoder_branch* br, int nbr, hipStream_t s) {
    if (int e = enc_bwd_check(br, nbr)) return e;
    EncArgs A;
    const int total = fill_args(A, br, nbr);
    if (enc_dw2_used(br, nbr, nullptr, nullptr)) {
        static int ready = -1;
        if (ready < 0) ready = enc_dw2_set_attributes();
        if (ready) return ready;
        // the kernel variant (which upstream gradients exist) is per launch: branches that disagree are launched separately,
        // each on its own workgroups and slots
        if (enc_f3_used(br, nbr)) {                 // the lower layers' gradients came with the dX chain (enc_stage_bwd_dx)
            if (!g_f3_dw3) enc_dw2_launch(A, total, s, true);        // (and dW3 / db3 too, unless PIML_ENC_FUSED_DW3=0)
        } else if (nbr == 1 || ((br[0].g_pooled != nullptr) == (br[1].g_pooled != nullptr) && (br[0].g_msgs != nullptr) == (br[1].g_msgs != nullptr))) {
            enc_dw2_launch(A, total, s);
        } else {
            for (int i = 0; i < 2; ++i) {
                EncArgs B = A;
                B.nbr = 1;
                B.br[0] = B.br[1] = A.br[i];
                enc_dw2_launch(B, i == 0 ? A.wg_split : total - A.wg_split, s);
            }
        }
        return hipGetLastError();
    }
    static bool attr_set = false;
    if (!attr_set) {
        const void* dw[3] = {reinterpret_cast<const void*>(enc_bwd_dw_kernel<true, true>),
                             reinterpret_cast<const void*>(enc_bwd_dw_kernel<true, false>),
                             reinterpret_cast<const void*>(enc_bwd_dw_kernel<false, true>)};
        for (const void* f : dw)
            if (int e = enc_set_lds(f, DW_LDS_FLOATS * 4)) return e;
        attr_set = true;
    }
    if (g_x3)
        if (int e = x3_ready()) return e;
    if (g_x3) {
        static int wide_ready = -1;
        if (wide_ready < 0) wide_ready = enc_dww_set_attributes();
        if (wide_ready) return wide_ready;
    }
    auto launch_dw = [&](const EncArgs& B, int grid) {
        if (g_x3) return enc_dww_launch(B, grid, B.br[0].keep_bits != nullptr, s);
        const bool pool = B.br[0].g_pooled != nullptr, msgs = B.br[0].g_msgs != nullptr;
        if (pool && msgs) hipLaunchKernelGGL((enc_bwd_dw_kernel<true, true>), dim3(grid), dim3(ENC_THREADS), DW_LDS_FLOATS * 4, s, B);
        else if (pool) hipLaunchKernelGGL((enc_bwd_dw_kernel<true, false>), dim3(grid), dim3(ENC_THREADS), DW_LDS_FLOATS * 4, s, B);
        else hipLaunchKernelGGL((enc_bwd_dw_kernel<false, true>), dim3(grid), dim3(ENC_THREADS), DW_LDS_FLOATS * 4, s, B);
    };
    // the kernel variant (which upstream gradients exist) is per launch: branches that disagree are launched
    // separately, each on its own share of the partial slots
    const bool same = nbr == 1 || ((br[0].g_pooled != nullptr) == (br[1].g_pooled != nullptr) &&
                                   (br[0].g_msgs != nullptr) == (br[1].g_msgs != nullptr));
    if (same) {
        launch_dw(A, total);
    } else {
        for (int i = 0; i < 2; ++i) {
            EncArgs B = A;
            B.nbr = 1;
            B.br[0] = B.br[1] = A.br[i];
            launch_dw(B, i == 0 ? A.wg_split : total - A.wg_split);
        }
    }
    return hipGetLastError();
}

int piml::enc_stage_reduce(const piml_encoder_branch* br, int nbr, hipStream_t s, bool accumulate, bool defer) {
    if (int e = enc_bwd_check(br, nbr)) return e;
    EncArgs A;
    fill_args(A, br, nbr);
    Reduce2Args R2 = {};
    R2.accumulate = accumulate ? 1 : 0;
    if (defer) {          // PIML_DEFER_SLOT_SUMS: the description goes to the relfeat backward's launch (network.hip: reduce_all's sets)
        ReduceAll R = {};
        R.accumulate = accumulate ? 1 : 0;
        int n = 0, maxl = 0;
        auto add = [&](const float* parts, float* grads, int slots, int lanes, int split, int off0, int off1) {
            R.set[n++] = ReduceSet{parts, grads, slots, lanes, split, off0, off1};
            if (lanes > maxl) maxl = lanes;
        };
        int n0[2] = {0, 0}, n1[2] = {0, 0}, w0 = 0;
        const int total = piml_encoder_workgroups(br, nbr, &w0);
        const bool dw2 = enc_dw2_used(br, nbr, n0, n1);
        for (int i = 0; i < nbr; ++i) {
            if (dw2) {
                add(br[i].partials, br[i].grads, n0[i], DW2_L0_LANES, DW2_L0_SPLIT, 0, DW2_L0_OFF1);
                add(br[i].partials + (size_t)n0[i] * (DW2_L0_LANES * 4), br[i].grads, n1[i], DW2_L1_LANES, DW2_L1_SPLIT, DW2_L1_OFF0, DW2_L1_OFF1);
            } else {
                add(br[i].partials, br[i].grads, nbr == 1 ? total : (i == 0 ? w0 : total - w0), ENC_PART / 4, 0x7fffffff, 0, 0);
            }
        }
        R.nsets = n;
        R.gx = (maxl + 15) / 16;
        return pending_slot_sums_leave(R, s);
    }
    if (enc_dw2_used(br, nbr, R2.n0, R2.n1)) {
        for (int i = 0; i < nbr; ++i) R2.br[i] = br[i];
        hipLaunchKernelGGL(enc_reduce2_kernel, dim3((DW2_L1_LANES + 15) / 16, 2 * nbr), dim3(256), 0, s, R2);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(enc_reduce_kernel, dim3((ENC_PART / 4 + 15) / 16, nbr), dim3(256), 0, s, A, ENC_PART / 4, accumulate ? 1 : 0);
    return hipGetLastError();
}

// Training on the agents' sums of h2 (PIML_POOL_TRAIN) serves: split products, the one-wave forward and the four-wave one-pass
// backward (more than piml_encoder_split_tiles() tiles, layer-split slots, at least two workgroups per branch), k = 2, 6 or 10
// neighbours per agent, whole agents, no dropout mask.  PIML_POOL_TRAIN=0 in the environment turns it off (A/B).
bool piml::enc_pool_train_ok(const piml_encoder_branch* br, int nbr) {
    static const bool off = getenv("PIML_POOL_TRAIN") && atoi(getenv("PIML_POOL_TRAIN")) == 0;
    if (off || !g_x3 || !g_dw2 || g_f3 != 1 || !br || nbr < 1 || nbr > 2) return false;
    long long tiles = 0;
    for (int i = 0; i < nbr; ++i) {
        const piml_encoder_branch& b = br[i];
        if ((b.k != 2 && b.k != 6 && b.k != 10) || b.rows <= 0 || b.rows % b.k || b.keep_bits || b.drop_state || b.in_dim < 1 || b.in_dim > 8 ||
            b.rows >= (1ll << 22))
            return false;
        tiles += (b.rows + 31) / 32;
    }
    if (tiles <= g_split_tiles_train) return false;
    const int total = 256, w0 = split_workgroups(br, nbr, total, 1);
    return nbr == 1 || (w0 >= 2 && total - w0 >= 2);
}

// The agents' sums of the MESSAGES from the forward's registers (PIML_POOL_MSGS: the last layer with exchanged operands,
// enc_fwd_x3_kernel<DROP, true>) serves: split products, a training pass above the one-wave bound (the one-pass backward follows:
// every branch carries relu_mask), k = 2, 6 or 10 neighbours per agent, whole agents; with or without a dropout mask.
// PIML_POOL_MSGS=0 in the environment turns it off (A/B).
bool piml::enc_pool_msgs_ok(const piml_encoder_branch* br, int nbr) {
    static const bool off = getenv("PIML_POOL_MSGS") && atoi(getenv("PIML_POOL_MSGS")) == 0;
    if (off || !g_x3 || !br || nbr < 1 || nbr > 2) return false;
    long long tiles = 0;
    for (int i = 0; i < nbr; ++i) {
        const piml_encoder_branch& b = br[i];
        if ((b.k != 2 && b.k != 6 && b.k != 10) || b.rows <= 0 || b.rows % b.k || !b.relu_mask || b.in_dim < 1 || b.in_dim > 8 ||
            b.rows >= (1ll << 22))
            return false;
        tiles += (b.rows + 31) / 32;
    }
    return tiles > g_split_tiles_train;
}

int piml::enc_stage_fwd_sum(const piml_encoder_branch* br, int nbr, hipStream_t s, float* zero, long long zero_n) {
    if (int e = enc_check(br, nbr)) return e;
    if (!enc_pool_train_ok(br, nbr)) return hipErrorInvalidValue;
    for (int i = 0; i < nbr; ++i)
        if (!br[i].sum_a || !br[i].sum_b || !br[i].relu_mask) return hipErrorInvalidValue;
    EncArgs A;
    const int total = fill_args(A, br, nbr);
    if (zero && zero_n > 0) {
        if (zero_n >= (1ll << 31)) return hipErrorInvalidValue;
        A.zero = zero;
        A.zero_n = (int)zero_n;
    }
    if (int e = x3_ready()) return e;
    enc_x3_launch_fwd_sum(A, total, s);
    return hipGetLastError();
}

// backward of enc_stage_fwd_sum: the one-pass kernel without its W3^T layer and without dW3 (encoder_bwd3.hip, SUMS = true)
int piml::enc_stage_bwd_sum(const piml_encoder_branch* br, int nbr, hipStream_t s) {
    if (int e = enc_check(br, nbr)) return e;
    if (!enc_pool_train_ok(br, nbr)) return hipErrorInvalidValue;
    for (int i = 0; i < nbr; ++i) {
        const piml_encoder_branch& b = br[i];
        if (!b.g_pooled || b.g_msgs || !b.relu_mask || !b.partials || !b.grads || (b.g_x != nullptr) != (br[0].g_x != nullptr)) return hipErrorInvalidValue;
    }
    EncArgs A;
    const int total = fill_args(A, br, nbr);
    if (int e = x3_ready()) return e;
    static int ready = -1;
    if (ready < 0) { ready = enc_f3_set_attributes(); if (!ready) ready = enc_f5_set_attributes(); }
    if (ready) return ready;
    const int nA[2] = {nbr > 1 ? A.wg_split : total, nbr > 1 ? total - A.wg_split : 0};
    const int zero[2] = {0, 0};
    if (g_sums_bwd == 1 || !enc_f5_launch(A, nA, s)) enc_f3_launch(A, nA, zero, false, s, true);
    return hipGetLastError();
}

PIML_API int piml_encoder_pack(const piml_encoder_branch* br, int nbr, void* stream) {
    return enc_stage_pack(br, nbr, as_stream(stream));
}

PIML_API int piml_encoder_fwd(const piml_encoder_branch* br, int nbr, void* stream) {
    if (int e = enc_stage_pack(br, nbr, as_stream(stream))) return e;
    return enc_stage_fwd(br, nbr, as_stream(stream));
}

// `packed` already holds the operand images of these weights (piml_encoder_pack / piml_pinnsf_pack)
PIML_API int piml_encoder_fwd_packed(const piml_encoder_branch* br, int nbr, void* stream) {
    if (int e = pending_pack_flush(as_stream(stream))) return e;          // a deferred pack (PIML_DEFER_PACK) nobody took: now
    return enc_stage_fwd(br, nbr, as_stream(stream));
}

PIML_API int piml_encoder_ksum(const float* msgs, long long agents, int k, float* pooled, void* stream) {
    if (agents == 0) return hipSuccess;
    if (!msgs || !pooled || agents < 0 || k < 0) return hipErrorInvalidValue;
    const long long n = agents * (EH / 4);
    hipLaunchKernelGGL(enc_ksum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(msgs), agents, k, reinterpret_cast<float4*>(pooled));
    return hipGetLastError();
}

PIML_API int piml_encoder_bwd_acc(const piml_encoder_branch* br, int nbr, int accumulate, void* stream) {
    // `accumulate`: 0 / 1, or flags -- PIML_ACCUMULATE and / or PIML_DEFER_SLOT_SUMS (the header)
    const bool acc = (accumulate & 1) || (accumulate & PIML_ACCUMULATE), defer = (accumulate & PIML_DEFER_SLOT_SUMS) != 0;
    if (int e = enc_stage_bwd_dx(br, nbr, as_stream(stream))) return e;
    if (int e = enc_stage_bwd_dw(br, nbr, as_stream(stream))) return e;
    return enc_stage_reduce(br, nbr, as_stream(stream), acc, defer);
}

PIML_API int piml_encoder_bwd(const piml_encoder_branch* br, int nbr, void* stream) { return piml_encoder_bwd_acc(br, nbr, 0, stream); }
