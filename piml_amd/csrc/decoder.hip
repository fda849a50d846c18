// Fused PINNSF decoder tail on the f32 matrix cores: neighbour-axis sum -> decoder MLP(128 -> 64 ReLU -> 64) ->
// predictor Linear(64 -> 2), both branches, plus the desired-force term -- and its backward.
//
// Reference arithmetic: src/models/model.py:1283-1294 (`pinnsf_m`; same lines in `pinnsf`):
//     ped_embeddings = sum_k ped_msgs;  acc = ped_predictor(ped_decoder(ped_embeddings))  [+ the obstacle branch]
//     predictions = acc + (v0 * dest / |dest| - v) / tau
// Per step these were ~45 launch-bound library GEMM / glue kernels (6 Linear layers on 4096 rows, forward and
// backward).  Same formulation as encoder.hip: features on the MFMA's M axis, agents on its N axis (lane = agent),
// accumulators chained as the next layer's B operand, weights pre-packed as A-fragments (here read straight from
// the packed global image: a wave owns one 32-agent tile and uses every fragment once, so LDS staging buys nothing).
// One workgroup = one 32-agent tile, wave 0 = pedestrian branch, wave 1 = obstacle branch, combined through LDS.
#include "common.hpp"
#include "../../include/piml_hip.h"

namespace piml {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DH = 128;        // decoder input width (= encoder width)
constexpr int DD = 64;         // decoder hidden / output width
// packed image of one branch (floats)
constexpr int DP_A1 = 0;                         // [ob 2][bp 4][q 4][lane 64] float4   W1 (64, 128)
constexpr int DP_A2 = DP_A1 + 2 * 4 * 4 * 256;   // [ob 2][bp 2][q 4][lane 64] float4   W2 (64, 64)
constexpr int DP_A3 = DP_A2 + 2 * 2 * 4 * 256;   // [bp 2][q 4][lane 64] float4         W3 (2, 64), rows >= 2 are 0
constexpr int DP_B = DP_A3 + 2 * 4 * 256;        // b1 64 | b2 64 | b3 2 | pad 2
constexpr int DP_T3 = DP_B + 132;                // [ob 2][lane 64]                     W3^T: lane (i, h) = W3[h][32 ob + i]
constexpr int DP_T2 = DP_T3 + 128;               // [ob 2][bp 2][q 4][lane 64] float4   W2^T
constexpr int DP_T1 = DP_T2 + 2 * 2 * 4 * 256;   // [blk 4][bp 2][q 4][lane 64] float4  W1^T
constexpr int DEC_PACK = DP_T1 + 4 * 2 * 4 * 256;
constexpr int DEC_PART = DD * DH + DD * DD + 2 * DD + DD + DD + 8;     // dW1 | dW2 | dW3 | db1 | db2 | db3 (+pad)
constexpr int DEC_SLAB = 32;                                           // agents per workgroup of the dW kernel

struct DecArgs {
    piml_decoder_branch br[2];
    int nbr;
    const float* self_features;   // (agents, 7) or NULL
    float tau;
    float* acc;                   // fwd out (agents, 2)
    const float* g_pred;          // bwd in (agents, 2)
    float* g_self;                // bwd out (agents, 7) or NULL
    int wg_split;                 // dW kernel: workgroups [0, wg_split) serve branch 0
};

__device__ __forceinline__ f32x16 dmfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int dfeat0(int blk, int q, int h) { return 32 * blk + 8 * q + 4 * h; }

__device__ __forceinline__ float dec_pack_value(const piml_decoder_branch& J, int e) {
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

__global__ __launch_bounds__(256) void dec_pack_kernel(DecArgs A) {
    const int b = blockIdx.y;
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < DEC_PACK) J.packed[e] = dec_pack_value(J, e);
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// pooled[a][:] = sum over the k rows of agent a of msgs (model.py:1283), both branches in one launch (blockIdx.y);
// one float4 column per thread.  (Summing inside dec_fwd serialises k dependent load rounds in one wave: 23 us.)
__global__ __launch_bounds__(256) void dec_pool_kernel(DecArgs A) {
    const piml_decoder_branch J = blockIdx.y ? A.br[1] : A.br[0];
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= J.agents * (DH / 4)) return;
    const long long a = t / (DH / 4);
    const int c = (int)(t % (DH / 4));
    const float4* p = reinterpret_cast<const float4*>(J.msgs) + a * J.k * (DH / 4) + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < J.k; ++i) {
        const float4 v = p[(size_t)i * (DH / 4)];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(J.pooled)[t] = s;
}

__global__ __launch_bounds__(128) void dec_fwd_kernel(DecArgs A) {
    __shared__ float comb[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const bool active = wave < A.nbr;
    const piml_decoder_branch J = wave ? A.br[1] : A.br[0];
    const long long agent = (long long)blockIdx.x * 32 + j;
    const bool valid = active && agent < J.agents;
    float ax = 0.f, ay = 0.f;
    if (active) {
        const float4* PK = reinterpret_cast<const float4*>(J.packed);
        const float* bias = J.packed + DP_B;
        f32x16 P[4];
        {
            const float* base = J.pooled + (valid ? agent : 0) * DH;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(base + dfeat0(blk, q, h));
                    P[blk][4 * q] = valid ? v.x : 0.f; P[blk][4 * q + 1] = valid ? v.y : 0.f;
                    P[blk][4 * q + 2] = valid ? v.z : 0.f; P[blk][4 * q + 3] = valid ? v.w : 0.f;
                }
        }
        // every weight fragment of the three layers is requested before the first MFMA (72 x 1 KiB per wave, L2-resident):
        // the chain below then runs at the MFMA rate instead of one L2 round trip per group of four
        float4 w1f[2][16], w2f[2][8], w3f[8];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int t = 0; t < 16; ++t) w1f[ob][t] = PK[DP_A1 / 4 + (ob * 16 + t) * 64 + lane];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int t = 0; t < 8; ++t) w2f[ob][t] = PK[DP_A2 / 4 + (ob * 8 + t) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 8; ++t) w3f[t] = PK[DP_A3 / 4 + t * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);       // the loads stay up here
        // ---- decoder layer 1: 128 -> 64, ReLU ----
        f32x16 a1[2], a2[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + dfeat0(ob, q, h));
                a1[ob][4 * q] = bq.x; a1[ob][4 * q + 1] = bq.y; a1[ob][4 * q + 2] = bq.z; a1[ob][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int bp = 0; bp < 4; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = w1f[ob][bp * 4 + q];
                    a1[ob] = dmfma(w.x, P[bp][4 * q + 0], a1[ob]);
                    a1[ob] = dmfma(w.y, P[bp][4 * q + 1], a1[ob]);
                    a1[ob] = dmfma(w.z, P[bp][4 * q + 2], a1[ob]);
                    a1[ob] = dmfma(w.w, P[bp][4 * q + 3], a1[ob]);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[ob][r] = fmaxf(a1[ob][r], 0.f);
        }
        if (J.h1 && valid) {
            float* o = J.h1 + agent * DD;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) =
                        make_float4(a1[ob][4 * q], a1[ob][4 * q + 1], a1[ob][4 * q + 2], a1[ob][4 * q + 3]);
        }
        // ---- decoder layer 2: 64 -> 64, no activation ----
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + 64 + dfeat0(ob, q, h));
                a2[ob][4 * q] = bq.x; a2[ob][4 * q + 1] = bq.y; a2[ob][4 * q + 2] = bq.z; a2[ob][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = w2f[ob][bp * 4 + q];
                    a2[ob] = dmfma(w.x, a1[bp][4 * q + 0], a2[ob]);
                    a2[ob] = dmfma(w.y, a1[bp][4 * q + 1], a2[ob]);
                    a2[ob] = dmfma(w.z, a1[bp][4 * q + 2], a2[ob]);
                    a2[ob] = dmfma(w.w, a1[bp][4 * q + 3], a2[ob]);
                }
        }
        if (J.d2 && valid) {
            float* o = J.d2 + agent * DD;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) =
                        make_float4(a2[ob][4 * q], a2[ob][4 * q + 1], a2[ob][4 * q + 2], a2[ob][4 * q + 3]);
        }
        // ---- predictor: 64 -> 2 (M padded to 32; output component c sits in register c of the h = 0 lanes) ----
        f32x16 a3;
#pragma unroll
        for (int r = 0; r < 16; ++r) a3[r] = 0.f;
        if (h == 0) { a3[0] = bias[128]; a3[1] = bias[129]; }
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = w3f[bp * 4 + q];
                a3 = dmfma(w.x, a2[bp][4 * q + 0], a3);
                a3 = dmfma(w.y, a2[bp][4 * q + 1], a3);
                a3 = dmfma(w.z, a2[bp][4 * q + 2], a3);
                a3 = dmfma(w.w, a2[bp][4 * q + 3], a3);
            }
        ax = a3[0];
        ay = a3[1];
    }
    if (wave == 1 && h == 0) { comb[2 * j] = ax; comb[2 * j + 1] = ay; }
    __syncthreads();
    if (wave == 0 && h == 0 && agent < A.br[0].agents) {
        if (A.nbr > 1) { ax += comb[2 * j]; ay += comb[2 * j + 1]; }
        if (A.self_features) {        // + (v0 * d / t - v) / tau,  t = |d| (+0.1 where |d| == 0)   (model.py:1289-1294)
            const float* s = A.self_features + agent * 7;
            const float dx = s[0], dy = s[1], vx = s[2], vy = s[3], v0 = s[6];
            float t = norm2(dx, dy);
            t = (t == 0.f) ? t + 0.1f : t;
            ax += (v0 * (dx / t) - vx) / A.tau;
            ay += (v0 * (dy / t) - vy) / A.tau;
        }
        reinterpret_cast<float2*>(A.acc)[agent] = make_float2(ax, ay);
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward, dX chain: g_pred (agents, 2) -> g_pre2 = W3^T g_pred -> g_pre1 = (W2^T g_pre2) * [h1 > 0] ->
// g_pooled = W1^T g_pre1; wave 0 also writes the desired-force gradient g_self when asked.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void dec_bwd_dx_kernel(DecArgs A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    if (wave >= A.nbr) return;
    const piml_decoder_branch J = wave ? A.br[1] : A.br[0];
    const long long agent = (long long)blockIdx.x * 32 + j;
    const bool valid = agent < J.agents;
    const float4* PK = reinterpret_cast<const float4*>(J.packed);
    float2 gp = make_float2(0.f, 0.f);
    if (valid) gp = reinterpret_cast<const float2*>(A.g_pred)[agent];
    const float bg = h ? gp.y : gp.x;
    f32x16 g2[2], g1[2];
    float4 hv[2][4];
    {
        const float* hp = J.h1 + (valid ? agent : 0) * DD;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[ob][q] = *reinterpret_cast<const float4*>(hp + dfeat0(ob, q, h));
    }
    float t3f[2];
    float4 t2f[2][8], t1f[4][8];          // all W^T fragments requested before the first MFMA
    t3f[0] = J.packed[DP_T3 + lane];
    t3f[1] = J.packed[DP_T3 + 64 + lane];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int t = 0; t < 8; ++t) t2f[ob][t] = PK[DP_T2 / 4 + (ob * 8 + t) * 64 + lane];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int t = 0; t < 8; ++t) t1f[blk][t] = PK[DP_T1 / 4 + (blk * 8 + t) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);           // the loads stay up here
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g2[ob][r] = 0.f;
        g2[ob] = dmfma(t3f[ob], bg, g2[ob]);
        if (valid) {
            float* o = J.g_pre2 + agent * DD;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) =
                    make_float4(g2[ob][4 * q], g2[ob][4 * q + 1], g2[ob][4 * q + 2], g2[ob][4 * q + 3]);
        }
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[ob][r] = 0.f;
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = t2f[ob][bp * 4 + q];
                g1[ob] = dmfma(w.x, g2[bp][4 * q + 0], g1[ob]);
                g1[ob] = dmfma(w.y, g2[bp][4 * q + 1], g1[ob]);
                g1[ob] = dmfma(w.z, g2[bp][4 * q + 2], g1[ob]);
                g1[ob] = dmfma(w.w, g2[bp][4 * q + 3], g1[ob]);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = hv[ob][q];
            g1[ob][4 * q + 0] = (valid && a.x > 0.f) ? g1[ob][4 * q + 0] : 0.f;
            g1[ob][4 * q + 1] = (valid && a.y > 0.f) ? g1[ob][4 * q + 1] : 0.f;
            g1[ob][4 * q + 2] = (valid && a.z > 0.f) ? g1[ob][4 * q + 2] : 0.f;
            g1[ob][4 * q + 3] = (valid && a.w > 0.f) ? g1[ob][4 * q + 3] : 0.f;
        }
        if (valid) {
            float* o = J.g_pre1 + agent * DD;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) =
                    make_float4(g1[ob][4 * q], g1[ob][4 * q + 1], g1[ob][4 * q + 2], g1[ob][4 * q + 3]);
        }
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        f32x16 gpool;
#pragma unroll
        for (int r = 0; r < 16; ++r) gpool[r] = 0.f;
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = t1f[blk][bp * 4 + q];
                gpool = dmfma(w.x, g1[bp][4 * q + 0], gpool);
                gpool = dmfma(w.y, g1[bp][4 * q + 1], gpool);
                gpool = dmfma(w.z, g1[bp][4 * q + 2], gpool);
                gpool = dmfma(w.w, g1[bp][4 * q + 3], gpool);
            }
        if (valid) {
            float* o = J.g_pooled + agent * DH;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + dfeat0(blk, q, h)) =
                    make_float4(gpool[4 * q], gpool[4 * q + 1], gpool[4 * q + 2], gpool[4 * q + 3]);
        }
    }
    // desired-force backward (pinnsf_epilogue_bwd_kernel's arithmetic), one lane per agent
    if (wave == 0 && h == 0 && valid && A.g_self && A.self_features) {
        const float* s = A.self_features + agent * 7;
        const float dx = s[0], dy = s[1], v0 = s[6], tau = A.tau;
        const float n = norm2(dx, dy);
        const float t = (n == 0.f) ? n + 0.1f : n;
        const float ex = dx / t, ey = dy / t;
        const float gex = gp.x * v0 / tau, gey = gp.y * v0 / tau;
        const float gt = -(gex * dx + gey * dy) / (t * t);
        float gdx = gex / t, gdy = gey / t;
        if (n != 0.f) { gdx += gt * (dx / n); gdy += gt * (dy / n); }
        float* o = A.g_self + agent * 7;
        o[0] = gdx; o[1] = gdy; o[2] = -gp.x / tau; o[3] = -gp.y / tau; o[4] = 0.f; o[5] = 0.f;
        o[6] = (gp.x * ex + gp.y * ey) / tau;
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward, weight gradients (K = agents of the workgroup's slab): dW1 = g_pre1^T pooled (64 x 128),
// dW2 = g_pre2^T h1 (64 x 64), dW3 = g_pred^T d2 (2 x 64), db = column sums.  8 waves: wave w owns block
// (w >> 2, w & 3) of dW1; waves 0-3 also block (w >> 1, w & 1) of dW2; waves 4, 5 also column block w & 1 of dW3.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void dec_bwd_dw_kernel(DecArgs A) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int p = (int)blockIdx.x - wg0;
    const long long R = J.agents;
    const long long slab = DEC_SLAB;                       // nwg = ceil(agents / DEC_SLAB), see dec_dw_workgroups
    const long long r0 = (long long)p * slab < R ? (long long)p * slab : R;
    const long long r1 = r0 + slab < R ? r0 + slab : R;
    const int i = lane & 31, h = lane >> 5;
    const int mb1 = w >> 2, nb1 = w & 3, mb2 = (w >> 1) & 1, nb2 = w & 1;
    const bool do2 = w < 4, do3 = w == 4 || w == 5;
    f32x16 c1, c2, c3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    // the slab is at most DEC_SLAB agents = DEC_SLAB / 2 k-steps: every load is issued before the first MFMA
    constexpr int U = DEC_SLAB / 2;
    {
        float a1[U], b1[U], a2[U], b2[U], a3[U], b3[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = r0 + 2 * u + h;
            const bool ok = row < r1;
            const long long ro = ok ? row : (r0 < R ? r0 : 0);
            a1[u] = J.g_pre1[ro * DD + 32 * mb1 + i];
            b1[u] = J.pooled[ro * DH + 32 * nb1 + i];
            a2[u] = do2 ? J.g_pre2[ro * DD + 32 * mb2 + i] : 0.f;
            b2[u] = do2 ? J.h1[ro * DD + 32 * nb2 + i] : 0.f;
            a3[u] = (do3 && i < 2) ? A.g_pred[ro * 2 + i] : 0.f;
            b3[u] = do3 ? J.d2[ro * DD + 32 * nb2 + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = r0 + 2 * u + h < r1;
            const float x1 = ok ? a1[u] : 0.f, x2 = ok ? a2[u] : 0.f, x3 = ok ? a3[u] : 0.f;
            c1 = dmfma(x1, b1[u], c1);
            if (do2) c2 = dmfma(x2, b2[u], c2);
            if (do3) c3 = dmfma(x3, b3[u], c3);
            s1 += x1; s2 += x2; s3 += x3;
        }
    }
    float* P = J.partials + (size_t)p * DEC_PART;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * h;
        P[(size_t)(32 * mb1 + ri) * DH + 32 * nb1 + i] = c1[r];
        if (do2) P[DD * DH + (32 * mb2 + ri) * DD + 32 * nb2 + i] = c2[r];
        if (do3 && ri < 2) P[DD * DH + DD * DD + ri * DD + 32 * nb2 + i] = c3[r];
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    s3 += __shfl_xor(s3, 32, 64);
    float* Pb = P + DD * DH + DD * DD + 2 * DD;
    if (h == 0) {
        if (nb1 == 0) Pb[32 * mb1 + i] = s1;                       // waves 0 and 4
        if (do2 && nb2 == 0) Pb[DD + 32 * mb2 + i] = s2;           // waves 0 and 2
        if (w == 4 && i < 8) Pb[2 * DD + i] = i < 2 ? s3 : 0.f;    // db3 + padding
    }
}

__global__ __launch_bounds__(256) void dec_reduce_kernel(DecArgs A, int B, int lanes) {
    // 16 float4 columns x 16 slot groups per block (the partials are few MB spread over many slots: wide grid)
    __shared__ float4 sh[256];
    const piml_decoder_branch J = blockIdx.y ? A.br[1] : A.br[0];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + col;
    const float4* parts = reinterpret_cast<const float4*>(J.partials);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < lanes)
        for (int q = grp; q < B; q += 16) {
            const float4 v = parts[(size_t)q * lanes + j];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && j < lanes) {
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float4 v = sh[q * 16 + col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(J.grads)[j] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// collision head of `pinnsf_m` (src/models/model.py:1246 `ped_collision_predictor = MLP(128, [64, 1])`, :1296-1300):
// out[row] = sigmoid(w2 . relu(W1 msgs[row] + b1) + b2) for every neighbour row, forward only (the reference trains the
// head for `pinnsf_bm` only; a backward through it falls back to torch ops, ops.collision_head).
// packed: A1 [ob 2][bp 4][q 4][lane 64] float4 | A2 [bp 2][q 4][lane 64] float4 (rows >= 1 are 0) | b1 64 | b2 1 + 3 pad
// ---------------------------------------------------------------------------------------------------------
constexpr int HP_A2 = 2 * 4 * 4 * 256;
constexpr int HP_B = HP_A2 + 2 * 4 * 256;
constexpr int HEAD_PACK = HP_B + 68;

__global__ __launch_bounds__(256) void head_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ packed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= HEAD_PACK) return;
    float v;
    if (e < HP_B) {
        const bool second = e >= HP_A2;
        const int f = second ? e - HP_A2 : e;
        const int u = f & 3, lane = (f >> 2) & 63, q = (f >> 8) & 3, rest = f >> 10;
        const int bp = second ? rest : (rest & 3), ob = second ? 0 : (rest >> 2);
        const int i = 32 * ob + (lane & 31), c = 32 * bp + 8 * q + 4 * (lane >> 5) + u;
        v = second ? (i < 1 ? w2[c] : 0.f) : w1[(size_t)i * DH + c];
    } else {
        const int g = e - HP_B;
        v = g < 64 ? b1[g] : (g == 64 ? b2[0] : 0.f);
    }
    packed[e] = v;
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ msgs, long long rows,
                                                       const float* __restrict__ packed, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long long row = ((long long)blockIdx.x * 4 + wave) * 32 + j;
    if (((long long)blockIdx.x * 4 + wave) * 32 >= rows) return;
    const bool valid = row < rows;
    const float4* PK = reinterpret_cast<const float4*>(packed);
    const float* bias = packed + HP_B;
    f32x16 X[4];
    {
        const float* base = msgs + (valid ? row : 0) * DH;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(base + dfeat0(blk, q, h));
                X[blk][4 * q] = v.x; X[blk][4 * q + 1] = v.y; X[blk][4 * q + 2] = v.z; X[blk][4 * q + 3] = v.w;
            }
    }
    float4 w1f[2][16], w2f[8];           // all weight fragments requested before the first MFMA
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int t = 0; t < 16; ++t) w1f[ob][t] = PK[(ob * 16 + t) * 64 + lane];
#pragma unroll
    for (int t = 0; t < 8; ++t) w2f[t] = PK[HP_A2 / 4 + t * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);           // the loads stay up here
    f32x16 a1[2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = *reinterpret_cast<const float4*>(bias + dfeat0(ob, q, h));
            a1[ob][4 * q] = bq.x; a1[ob][4 * q + 1] = bq.y; a1[ob][4 * q + 2] = bq.z; a1[ob][4 * q + 3] = bq.w;
        }
#pragma unroll
        for (int bp = 0; bp < 4; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = w1f[ob][bp * 4 + q];
                a1[ob] = dmfma(w.x, X[bp][4 * q + 0], a1[ob]);
                a1[ob] = dmfma(w.y, X[bp][4 * q + 1], a1[ob]);
                a1[ob] = dmfma(w.z, X[bp][4 * q + 2], a1[ob]);
                a1[ob] = dmfma(w.w, X[bp][4 * q + 3], a1[ob]);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[ob][r] = fmaxf(a1[ob][r], 0.f);
    }
    f32x16 a2;
#pragma unroll
    for (int r = 0; r < 16; ++r) a2[r] = 0.f;
    if (h == 0) a2[0] = bias[64];
#pragma unroll
    for (int bp = 0; bp < 2; ++bp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = w2f[bp * 4 + q];
            a2 = dmfma(w.x, a1[bp][4 * q + 0], a2);
            a2 = dmfma(w.y, a1[bp][4 * q + 1], a2);
            a2 = dmfma(w.z, a1[bp][4 * q + 2], a2);
            a2 = dmfma(w.w, a1[bp][4 * q + 3], a2);
        }
    if (h == 0 && valid) out[row] = 1.f / (1.f + expf(-a2[0]));
}

static bool dec_branch_ok(const piml_decoder_branch& b) {
    return b.agents > 0 && b.agents < (1ll << 21) && b.k >= 1 && b.msgs && b.w1 && b.b1 && b.w2 && b.b2 && b.w3 && b.b3 && b.packed;
}

static int dec_dw_workgroups(long long agents) { return (int)((agents + DEC_SLAB - 1) / DEC_SLAB); }

}  // namespace piml

using namespace piml;

PIML_API int piml_decoder_pack_floats(void) { return DEC_PACK; }
PIML_API int piml_decoder_partial_floats(void) { return DEC_PART; }
PIML_API int piml_decoder_workgroups(long long agents) { return dec_dw_workgroups(agents); }

PIML_API int piml_decoder_fwd(const piml_decoder_branch* br, int nbr, const float* self_features, float tau,
                              float* acc, void* stream) {
    if (!br || nbr < 1 || nbr > 2 || !acc) return hipErrorInvalidValue;
    DecArgs A = {};
    A.nbr = nbr;
    for (int i = 0; i < nbr; ++i) {
        if (!dec_branch_ok(br[i]) || br[i].agents != br[0].agents || !br[i].pooled) return hipErrorInvalidValue;
        A.br[i] = br[i];
    }
    if (nbr == 1) A.br[1] = br[0];
    A.self_features = self_features;
    A.tau = tau;
    A.acc = acc;
    hipLaunchKernelGGL(dec_pack_kernel, dim3((DEC_PACK + 255) / 256, nbr), dim3(256), 0, as_stream(stream), A);
    hipLaunchKernelGGL(dec_pool_kernel, dim3((unsigned)((br[0].agents * (DH / 4) + 255) / 256), nbr), dim3(256), 0,
                       as_stream(stream), A);
    const unsigned tiles = (unsigned)((br[0].agents + 31) / 32);
    hipLaunchKernelGGL(dec_fwd_kernel, dim3(tiles), dim3(128), 0, as_stream(stream), A);
    return hipGetLastError();
}

PIML_API int piml_decoder_bwd(const piml_decoder_branch* br, int nbr, const float* g_pred, const float* self_features,
                              float tau, float* g_self, void* stream) {
    if (!br || nbr < 1 || nbr > 2 || !g_pred) return hipErrorInvalidValue;
    DecArgs A = {};
    A.nbr = nbr;
    for (int i = 0; i < nbr; ++i) {
        const piml_decoder_branch& b = br[i];
        if (!dec_branch_ok(b) || b.agents != br[0].agents || !b.pooled || !b.h1 || !b.d2 || !b.g_pre2 || !b.g_pre1 ||
            !b.g_pooled || !b.partials || !b.grads)
            return hipErrorInvalidValue;
        A.br[i] = b;
    }
    if (nbr == 1) A.br[1] = br[0];
    A.self_features = self_features;
    A.tau = tau;
    A.g_pred = g_pred;
    A.g_self = g_self;
    const unsigned tiles = (unsigned)((br[0].agents + 31) / 32);
    hipLaunchKernelGGL(dec_bwd_dx_kernel, dim3(tiles), dim3(128), 0, as_stream(stream), A);
    const int per = dec_dw_workgroups(br[0].agents);
    A.wg_split = per;
    hipLaunchKernelGGL(dec_bwd_dw_kernel, dim3(per * nbr), dim3(512), 0, as_stream(stream), A);
    hipLaunchKernelGGL(dec_reduce_kernel, dim3((DEC_PART / 4 + 15) / 16, nbr), dim3(256), 0, as_stream(stream), A, per,
                       DEC_PART / 4);
    return hipGetLastError();
}

PIML_API int piml_collision_head_pack_floats(void) { return HEAD_PACK; }

PIML_API int piml_collision_head_fwd(const float* msgs, long long rows, const float* w1, const float* b1, const float* w2,
                                     const float* b2, float* packed, float* out, void* stream) {
    if (rows < 0) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!msgs || !w1 || !b1 || !w2 || !b2 || !packed || !out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_pack_kernel, dim3((HEAD_PACK + 255) / 256), dim3(256), 0, as_stream(stream), w1, b1, w2, b2,
                       packed);
    const long long tiles = (rows + 31) / 32;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, as_stream(stream), msgs, rows,
                       packed, out);
    return hipGetLastError();
}
