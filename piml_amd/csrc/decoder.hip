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
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "pack.hpp"
#include "reduce.hpp"
#include "stages.hpp"
#include "x3.hpp"
#include "../../include/piml_hip.h"

namespace piml {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
    int pool_h2;                  // forward, PIML_POOL_H2: `pooled` holds the agents' first parts, `msgs` (agents, 128) the second
                                  // parts of the agents whose k rows straddle two 32-row tiles (enc_fwd_pool_x3_kernel)
                                  // 2 = PIML_POOL_TRAIN (forward and backward): the same, with the FOLDED first-layer images of
                                  // `packed` (DP_A1F / DP_T1F / DP_CF, pack.hpp), bias b1 + k c, and the completed sums written
                                  // back to `pooled` for the backward's dW1
                                  // 3 = PIML_POOL_MSGS: `pooled` / `msgs` hold the two parts of the agents' sums of the MESSAGES
                                  // (enc_fwd_x3_kernel<DROP, true>): plain images, completed sums written back like 2
};

#ifdef PIML_DEC_STAMPS
// diagnostic build only (tools/dec_stamps.py): shader-clock stamps of thread 0 of every workgroup of the LAST decoder launch,
// `wait` = drain the memory counters first, so that the stamp marks the arrival of everything requested so far
__device__ unsigned long long g_dec_stamps[1024 * 16];
#define DEC_STAMP(i, wait)                                                                         \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (wait) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                      \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_dec_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define DEC_STAMP(i, wait)
#endif

__device__ __forceinline__ f32x16 dmfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int dfeat0(int blk, int q, int h) { return 32 * blk + 8 * q + 4 * h; }

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
__device__ __forceinline__ void dec_pool_body(const piml_decoder_branch& J, long long bx) {
    const long long t = bx * 256 + threadIdx.x;
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

__global__ __launch_bounds__(256) void dec_pool_kernel(DecArgs A) {
    dec_pool_body(blockIdx.y ? A.br[1] : A.br[0], blockIdx.x);
}

// One 32-agent tile per workgroup, both branches.  The three layers are a dependent chain of 224 MFMAs per branch
// (7.5 us when one wave runs it): the chain is cut across 4 waves per branch -- wave (ob, kh) owns output block ob and
// half kh of the contraction of layers 1 and 2 -- with the partial sums exchanged through LDS in accumulator layout
// (lane = agent, register = feature: exactly the B-operand layout of the next layer, so an exchange is 16 conflict-free
// ds_write_b32 + 32 ds_read_b32 per lane).  Chain per wave: 32 + 16 + 16 MFMAs and three barriers.
// POOL: the neighbour-axis sum (model.py:1283) is done here instead of by dec_pool_kernel: wave (branch, w) sums feature
// block w of the tile's agents over their k message rows (k x 4 16-byte loads per lane, POOL_ROWS rows = 160 registers in
// flight at a time -- the reference's k = 6 and k = 10 in ONE round trip --, added in row order like dec_pool_kernel), writes `pooled` for the backward pass and hands the block to the other
// waves through LDS.
constexpr int POOL_ROWS = 10;

// SPLIT: one 4-wave workgroup per (tile, branch) instead of 8 waves per tile -- twice the CUs pull the message rows
// (the inline pooling is bound by the L1 miss rate of a CU) --; each branch then ADDS its acceleration to `acc`, which
// the caller has zeroed: two float atomic adds onto zero commute exactly (0 + x = x, x + y = y + x), so the result is
// still deterministic and equal to the unsplit kernel's (p + d) + o.
// ROWS (with SPLIT): the bottleneck variants' per-neighbour-row use -- the input rows are the (rows, 128) embeddings
// themselves (J.msgs), `agents` = rows of THIS branch, and the predictor output of every row goes to J.pred.
template <bool POOL, bool SPLIT = false, bool ROWS = false>
__device__ __forceinline__ void dec_fwd_body(const DecArgs& A, long long tile, int split_branch = 0) {
    constexpr int NB = SPLIT ? 1 : 2;
    __shared__ float part1[NB][2][2][16][64];      // [branch][ob][kh][register][lane]
    __shared__ float part2[NB][2][2][16][64];
    __shared__ float part3[NB][2][2][32];          // [branch][kh][component][agent]
    __shared__ float poolx[POOL ? NB : 1][POOL ? 4 : 1][POOL ? 16 : 1][64];     // [branch][feature block][register][lane]
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));   // in an SGPR: per-wave selects stay scalar
    const int j = lane & 31, h = lane >> 5;
    const int b = SPLIT ? split_branch : wave >> 2, ob = (wave >> 1) & 1, kh = wave & 1;
    const int bi = SPLIT ? 0 : b;                  // index of the branch in the LDS arrays
    const bool active = b < A.nbr;
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const long long agent = tile * 32 + j;
    const bool valid = active && agent < J.agents;
    const float4* PK = reinterpret_cast<const float4*>(J.packed);
    const float* bias = J.packed + DP_B;
    float4 w2f[4], w3f[4], w1f[2][4], b2v[4];
    float b3x = 0.f, b3y = 0.f;
    if (active) {
#pragma unroll
        for (int bl = 0; bl < 2; ++bl)
#pragma unroll
            for (int q = 0; q < 4; ++q) w1f[bl][q] = PK[(A.pool_h2 == 2 ? DP_A1F : DP_A1) / 4 + ((ob * 4 + 2 * kh + bl) * 4 + q) * 64 + lane];
    }
    float4 b1v[4];
    float sfv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    // everything the later phases read from memory, requested in one go BEFORE the first barrier (left where they are used,
    // these loads sit behind the barriers: one exposed L2 round trip per layer)
    auto late_loads = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b1v[q] = *reinterpret_cast<const float4*>(bias + dfeat0(ob, q, h));
            if (!POOL && !ROWS && A.pool_h2 == 2) {          // folded first layer: b1 + k c (pack.hpp: DP_CF)
                const float4 cv = *reinterpret_cast<const float4*>(J.packed + DP_CF + dfeat0(ob, q, h));
                const float kf = (float)J.k;
                b1v[q].x += kf * cv.x; b1v[q].y += kf * cv.y; b1v[q].z += kf * cv.z; b1v[q].w += kf * cv.w;
            }
            w2f[q] = PK[DP_A2 / 4 + ((ob * 2 + kh) * 4 + q) * 64 + lane];
            w3f[q] = PK[DP_A3 / 4 + (kh * 4 + q) * 64 + lane];
            b2v[q] = *reinterpret_cast<const float4*>(bias + 64 + dfeat0(ob, q, h));
        }
        b3x = bias[128];
        b3y = bias[129];
        if (A.self_features && b == 0 && wave == 0) {          // (wave-uniform condition) the epilogue's desired-force inputs
            const float* sp = A.self_features + (agent < A.br[0].agents ? agent : 0) * 7;
            sfv[0] = sp[0]; sfv[1] = sp[1]; sfv[2] = sp[2]; sfv[3] = sp[3]; sfv[4] = sp[6];
        }
    };
    if (POOL) {
        if (active) {
            const int blk = wave & 3, k = J.k;
            const float* mp = J.msgs + (valid ? agent : 0) * (long long)k * DH;
            float4 sum[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sum[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r0 = 0; r0 < k; r0 += POOL_ROWS) {
                float4 v[POOL_ROWS][4];
#pragma unroll
                for (int u = 0; u < POOL_ROWS; ++u) {
                    const int r = r0 + u < k ? r0 + u : 0;             // clamped, unconditional: all loads in flight
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[u][q] = *reinterpret_cast<const float4*>(mp + (long long)r * DH + dfeat0(blk, q, h));
                }
#pragma unroll
                for (int u = 0; u < POOL_ROWS; ++u) {
                    const bool ok = r0 + u < k;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sum[q].x += ok ? v[u][q].x : 0.f; sum[q].y += ok ? v[u][q].y : 0.f;
                        sum[q].z += ok ? v[u][q].z : 0.f; sum[q].w += ok ? v[u][q].w : 0.f;
                    }
                }
            }
            late_loads();
            __builtin_amdgcn_sched_barrier(0);       // ... in flight under the LDS exchange and the barrier
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (valid) *reinterpret_cast<float4*>(J.pooled + agent * DH + dfeat0(blk, q, h)) = sum[q];
                poolx[bi][blk][4 * q + 0][lane] = valid ? sum[q].x : 0.f; poolx[bi][blk][4 * q + 1][lane] = valid ? sum[q].y : 0.f;
                poolx[bi][blk][4 * q + 2][lane] = valid ? sum[q].z : 0.f; poolx[bi][blk][4 * q + 3][lane] = valid ? sum[q].w : 0.f;
            }
        }
        __syncthreads();
    }
    float4 pv[2][4];
    if (SPLIT && !POOL && !ROWS) DEC_STAMP(0, false);
    if (active) {
        // ---- layer 1 partial: features of block ob, contraction over input blocks 2 kh, 2 kh + 1 ----
        if (POOL) {
#pragma unroll
            for (int bl = 0; bl < 2; ++bl)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    pv[bl][q] = make_float4(poolx[bi][2 * kh + bl][4 * q][lane], poolx[bi][2 * kh + bl][4 * q + 1][lane],
                                            poolx[bi][2 * kh + bl][4 * q + 2][lane], poolx[bi][2 * kh + bl][4 * q + 3][lane]);
        } else {
            const float* base = (ROWS ? J.msgs : J.pooled) + (valid ? agent : 0) * DH;
#pragma unroll
            for (int bl = 0; bl < 2; ++bl)
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[bl][q] = *reinterpret_cast<const float4*>(base + dfeat0(2 * kh + bl, q, h));
            if (!ROWS && A.pool_h2) {
                // (round 5, measured and dropped: the second parts requested for every lane together with the first parts, instead of a
                // branch on the straddlers -- one round trip less on paper, 12.8 -> 13.8 us in fact: this launch is bound by the bytes its
                // waves pull through the CUs' L1s at its start, and that form pulls a quarter more)
                const long long g0 = (valid ? agent : 0) * J.k;
                if (valid && (g0 >> 5) != ((g0 + J.k - 1) >> 5)) {          // the agent's rows straddle two tiles: + its second part
                    const float* more = J.msgs + agent * DH;
#pragma unroll
                    for (int bl = 0; bl < 2; ++bl)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 c = *reinterpret_cast<const float4*>(more + dfeat0(2 * kh + bl, q, h));
                            pv[bl][q].x += c.x; pv[bl][q].y += c.y; pv[bl][q].z += c.z; pv[bl][q].w += c.w;
                        }
                }
            }
        }
        if (!POOL) late_loads();
        f32x16 a1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = b1v[q];
            a1[4 * q] = kh ? 0.f : bq.x; a1[4 * q + 1] = kh ? 0.f : bq.y;
            a1[4 * q + 2] = kh ? 0.f : bq.z; a1[4 * q + 3] = kh ? 0.f : bq.w;
        }
        __builtin_amdgcn_sched_barrier(0);       // every load above is in flight before the first MFMA
        if (SPLIT && !POOL && !ROWS) { DEC_STAMP(1, false); DEC_STAMP(2, true); }
#pragma unroll
        for (int bl = 0; bl < 2; ++bl)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = w1f[bl][q], x = pv[bl][q];
                a1 = dmfma(w.x, valid ? x.x : 0.f, a1);
                a1 = dmfma(w.y, valid ? x.y : 0.f, a1);
                a1 = dmfma(w.z, valid ? x.z : 0.f, a1);
                a1 = dmfma(w.w, valid ? x.w : 0.f, a1);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) part1[bi][ob][kh][r][lane] = a1[r];
        if (SPLIT && !POOL && !ROWS) DEC_STAMP(3, true);
    }
    __syncthreads();
    if (SPLIT && !POOL && !ROWS) DEC_STAMP(4, false);
    if (active) {
        // training on the sums (pool_h2 == 2): the completed sum is the backward's layer-1 input (dW1 = g_pre1^T pooled); written
        // behind the barrier, when every wave of the workgroup has finished reading the first parts
        if (!POOL && !ROWS && A.pool_h2 >= 2 && ob == 0) {
            const long long g0 = (valid ? agent : 0) * J.k;
            if (valid && (g0 >> 5) != ((g0 + J.k - 1) >> 5)) {
#pragma unroll
                for (int bl = 0; bl < 2; ++bl)
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(J.pooled + agent * DH + dfeat0(2 * kh + bl, q, h)) = pv[bl][q];
            }
        }
        // ---- layer 2 partial: output block ob, contraction over hidden block kh (= relu of the summed layer-1 block kh) ----
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = fmaxf(part1[bi][kh][0][r][lane] + part1[bi][kh][1][r][lane], 0.f);
        if (ob == 0 && J.h1 && valid) {
            float* o = J.h1 + agent * DD;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + dfeat0(kh, q, h)) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
        }
        f32x16 a2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = b2v[q];
            a2[4 * q] = kh ? 0.f : bq.x; a2[4 * q + 1] = kh ? 0.f : bq.y;
            a2[4 * q + 2] = kh ? 0.f : bq.z; a2[4 * q + 3] = kh ? 0.f : bq.w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = w2f[q];
            a2 = dmfma(w.x, x[4 * q + 0], a2);
            a2 = dmfma(w.y, x[4 * q + 1], a2);
            a2 = dmfma(w.z, x[4 * q + 2], a2);
            a2 = dmfma(w.w, x[4 * q + 3], a2);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part2[bi][ob][kh][r][lane] = a2[r];
        if (SPLIT && !POOL && !ROWS) DEC_STAMP(5, true);
    }
    __syncthreads();
    if (SPLIT && !POOL && !ROWS) DEC_STAMP(6, false);
    if (active && ob == 0) {
        // ---- predictor partial over decoder-output block kh (M padded to 32: component c = register c of the h = 0 lanes) ----
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = part2[bi][kh][0][r][lane] + part2[bi][kh][1][r][lane];
        if (J.d2 && valid) {
            float* o = J.d2 + agent * DD;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + dfeat0(kh, q, h)) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        }
        f32x16 a3;
#pragma unroll
        for (int r = 0; r < 16; ++r) a3[r] = 0.f;
        if (h == 0 && kh == 0) { a3[0] = b3x; a3[1] = b3y; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = w3f[q];
            a3 = dmfma(w.x, y[4 * q + 0], a3);
            a3 = dmfma(w.y, y[4 * q + 1], a3);
            a3 = dmfma(w.z, y[4 * q + 2], a3);
            a3 = dmfma(w.w, y[4 * q + 3], a3);
        }
        if (h == 0) { part3[bi][kh][0][j] = a3[0]; part3[bi][kh][1][j] = a3[1]; }
        if (SPLIT && !POOL && !ROWS) DEC_STAMP(7, true);
    }
    __syncthreads();
    if (SPLIT && !POOL && !ROWS) DEC_STAMP(8, false);
    if (ROWS) {
        if ((wave & 3) == 0 && h == 0 && agent < J.agents)
            reinterpret_cast<float2*>(J.pred)[agent] = make_float2(part3[0][0][0][j] + part3[0][1][0][j],
                                                                   part3[0][0][1][j] + part3[0][1][1][j]);
        return;
    }
    if (wave == 0 && h == 0 && agent < A.br[0].agents) {
        // sum order (also of the split form): (pedestrian branch + desired force) + obstacle branch
        float ax = part3[0][0][0][j] + part3[0][1][0][j], ay = part3[0][0][1][j] + part3[0][1][1][j];
        if (A.self_features && b == 0) {   // + (v0 * d / t - v) / tau,  t = |d| (+0.1 where |d| == 0)   (model.py:1289-1294)
            const float dx = sfv[0], dy = sfv[1], vx = sfv[2], vy = sfv[3], v0 = sfv[4];
            float t = norm2(dx, dy);
            t = (t == 0.f) ? t + 0.1f : t;
            ax += (v0 * (dx / t) - vx) / A.tau;
            ay += (v0 * (dy / t) - vy) / A.tau;
        }
        if (SPLIT) {
            if (A.nbr > 1) {
                atomicAdd(A.acc + 2 * agent, ax);
                atomicAdd(A.acc + 2 * agent + 1, ay);
            } else {
                reinterpret_cast<float2*>(A.acc)[agent] = make_float2(ax, ay);
            }
        } else {
            if (A.nbr > 1) { ax += part3[1][0][0][j] + part3[1][1][0][j]; ay += part3[1][0][1][j] + part3[1][1][1][j]; }
            reinterpret_cast<float2*>(A.acc)[agent] = make_float2(ax, ay);
        }
    }
    if (SPLIT && !POOL && !ROWS) DEC_STAMP(9, true);
}

__global__ __launch_bounds__(512) void dec_fwd_kernel(DecArgs A) { dec_fwd_body<false>(A, blockIdx.x); }

// ---------------------------------------------------------------------------------------------------------
// backward, dX chain: g_pred (agents, 2) -> g_pre2 = W3^T g_pred -> g_pre1 = (W2^T g_pre2) * [h1 > 0] ->
// g_pooled = W1^T g_pre1; wave 0 also writes the desired-force gradient g_self when asked.
// ---------------------------------------------------------------------------------------------------------
// 4 waves per branch (the chain of 194 MFMAs is cut like dec_fwd_kernel's): wave (ob, kh) computes g_pre2 block kh itself
// (one MFMA), the partial of g_pre1 block ob over it (16 MFMAs), and after one LDS exchange g_pooled block 2 ob + kh over
// the complete, masked g_pre1 (32 MFMAs).
// ROWS: per-neighbour-row use (see dec_fwd_body): one 4-wave workgroup per (tile, branch), g_pred per row from
// J.g_pred_rows, optional extra gradient J.g_d2 on the decoder output, no desired-force part.
// SPLIT (round 3): the agent-level chain with one 4-wave workgroup per (tile, branch) like the forward's SPLIT form --
// the 8-wave form runs 128 workgroups at the 4096-agent scene, half the CUs.
template <bool ROWS = false, bool SPLIT = false>
__device__ __forceinline__ void dec_bwd_dx_body(const DecArgs& A, long long tile, int rows_branch = 0, float* keep_g2 = nullptr,
                                                float* keep_g1 = nullptr) {
    constexpr int NB = (ROWS || SPLIT) ? 1 : 2;
    __shared__ float part[NB][2][2][16][64];       // [branch][ob][kh][register][lane]
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));   // in an SGPR: per-wave selects stay scalar
    const int j = lane & 31, h = lane >> 5;
    const int b = (ROWS || SPLIT) ? rows_branch : wave >> 2, ob = (wave >> 1) & 1, kh = wave & 1, blk = wave & 3;
    const int bi = (ROWS || SPLIT) ? 0 : b;
    const bool active = b < A.nbr;
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const long long agent = tile * 32 + j;
    const bool valid = active && agent < J.agents;
    const float4* PK = reinterpret_cast<const float4*>(J.packed);
    float2 gp = make_float2(0.f, 0.f);
    float4 hv[2][4], t1f[8];
    float sfv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        // unconditional, clamped loads (a branch around a load makes the compiler wait for it at the join: three serial
        // round trips at the head of this kernel before the change)
        gp = reinterpret_cast<const float2*>(ROWS ? J.g_pred_rows : A.g_pred)[valid ? agent : 0];
        if (!valid) gp = make_float2(0.f, 0.f);
        float4 gd2[4];
        if (ROWS && J.g_d2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) gd2[q] = *reinterpret_cast<const float4*>(J.g_d2 + (valid ? agent : 0) * DD + dfeat0(kh, q, h));
        }
        if (!ROWS && wave == 0 && b == 0 && A.g_self && A.self_features) {        // wave-uniform: the desired-force gradient's inputs
            const float* sp = A.self_features + (valid ? agent : 0) * 7;
            sfv[0] = sp[0]; sfv[1] = sp[1]; sfv[4] = sp[6];
        }
        const float bg = h ? gp.y : gp.x;
        const float* hp = J.h1 + (valid ? agent : 0) * DD;
#pragma unroll
        for (int o2 = 0; o2 < 2; ++o2)
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[o2][q] = *reinterpret_cast<const float4*>(hp + dfeat0(o2, q, h));
        const float t3f = J.packed[DP_T3 + 64 * kh + lane];
        float4 t2f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t2f[q] = PK[DP_T2 / 4 + (ob * 8 + kh * 4 + q) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 8; ++t) t1f[t] = PK[((!ROWS && A.pool_h2 == 2) ? DP_T1F : DP_T1) / 4 + (blk * 8 + t) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);           // the loads stay up here
        if (SPLIT) { DEC_STAMP(1, false); DEC_STAMP(2, true); }
        f32x16 g2, g1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { g2[r] = 0.f; g1[r] = 0.f; }
        g2 = dmfma(t3f, bg, g2);
        if (ROWS && J.g_d2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                g2[4 * q] += valid ? gd2[q].x : 0.f; g2[4 * q + 1] += valid ? gd2[q].y : 0.f;
                g2[4 * q + 2] += valid ? gd2[q].z : 0.f; g2[4 * q + 3] += valid ? gd2[q].w : 0.f;
            }
        }
        if (SPLIT && keep_g2) {            // the tile's g_pre2 stays in the workgroup (LDS [agent][64]): its only reader is the dW body
            if (ob == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(keep_g2 + j * DD + dfeat0(kh, q, h)) =
                        valid ? make_float4(g2[4 * q], g2[4 * q + 1], g2[4 * q + 2], g2[4 * q + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (ob == 0 && valid) {
            float* o = J.g_pre2 + agent * DD;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + dfeat0(kh, q, h)) = make_float4(g2[4 * q], g2[4 * q + 1], g2[4 * q + 2], g2[4 * q + 3]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = t2f[q];
            g1 = dmfma(w.x, g2[4 * q + 0], g1);
            g1 = dmfma(w.y, g2[4 * q + 1], g1);
            g1 = dmfma(w.z, g2[4 * q + 2], g1);
            g1 = dmfma(w.w, g2[4 * q + 3], g1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[bi][ob][kh][r][lane] = g1[r];
        if (SPLIT) DEC_STAMP(3, true);
    }
    __syncthreads();
    if (SPLIT) DEC_STAMP(4, false);
    if (active) {
    float g1c[2][16];
#pragma unroll
    for (int o2 = 0; o2 < 2; ++o2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = hv[o2][q];
            const float m[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = part[bi][o2][0][4 * q + u][lane] + part[bi][o2][1][4 * q + u][lane];
                g1c[o2][4 * q + u] = (valid && m[u] > 0.f) ? v : 0.f;
            }
        }
    if (SPLIT && keep_g1) {          // (g1c is already zero for agents past the end)
        if (blk < 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v0 = make_float4(g1c[0][4 * q], g1c[0][4 * q + 1], g1c[0][4 * q + 2], g1c[0][4 * q + 3]);
                const float4 v1 = make_float4(g1c[1][4 * q], g1c[1][4 * q + 1], g1c[1][4 * q + 2], g1c[1][4 * q + 3]);
                if (blk == 0) *reinterpret_cast<float4*>(keep_g1 + j * DD + dfeat0(0, q, h)) = v0;
                else *reinterpret_cast<float4*>(keep_g1 + j * DD + dfeat0(1, q, h)) = v1;
            }
        }
    } else if (blk < 2 && valid) {          // (compile-time register indices: a runtime g1c[blk] becomes a select chain)
        float* o = J.g_pre1 + agent * DD;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v0 = make_float4(g1c[0][4 * q], g1c[0][4 * q + 1], g1c[0][4 * q + 2], g1c[0][4 * q + 3]);
            const float4 v1 = make_float4(g1c[1][4 * q], g1c[1][4 * q + 1], g1c[1][4 * q + 2], g1c[1][4 * q + 3]);
            if (blk == 0) *reinterpret_cast<float4*>(o + dfeat0(0, q, h)) = v0;
            else *reinterpret_cast<float4*>(o + dfeat0(1, q, h)) = v1;
        }
    }
    {
        f32x16 gpool;
#pragma unroll
        for (int r = 0; r < 16; ++r) gpool[r] = 0.f;
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = t1f[bp * 4 + q];
                gpool = dmfma(w.x, g1c[bp][4 * q + 0], gpool);
                gpool = dmfma(w.y, g1c[bp][4 * q + 1], gpool);
                gpool = dmfma(w.z, g1c[bp][4 * q + 2], gpool);
                gpool = dmfma(w.w, g1c[bp][4 * q + 3], gpool);
            }
        if (valid) {
            float* o = J.g_pooled + agent * DH;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + dfeat0(blk, q, h)) =
                    make_float4(gpool[4 * q], gpool[4 * q + 1], gpool[4 * q + 2], gpool[4 * q + 3]);
        }
    }
    }
    if (SPLIT) DEC_STAMP(5, true);
    // desired-force backward (pinnsf_epilogue_bwd_kernel's arithmetic), one lane per agent
    if (!ROWS && wave == 0 && b == 0 && h == 0 && valid && A.g_self && A.self_features) {
        const float dx = sfv[0], dy = sfv[1], v0 = sfv[4], tau = A.tau;
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

__global__ __launch_bounds__(512) void dec_bwd_dx_kernel(DecArgs A) { dec_bwd_dx_body<false>(A, blockIdx.x); }

// ---------------------------------------------------------------------------------------------------------
// backward, weight gradients (K = agents of the workgroup's slab): dW1 = g_pre1^T pooled (64 x 128),
// dW2 = g_pre2^T h1 (64 x 64), dW3 = g_pred^T d2 (2 x 64), db = column sums.  8 waves: wave w owns block
// (w >> 2, w & 3) of dW1; waves 0-3 also block (w >> 1, w & 1) of dW2; waves 4, 5 also column block w & 1 of dW3.
// ---------------------------------------------------------------------------------------------------------
// (branch b, slab p) for the 8 waves of a workgroup; w = wave index.  ROWS: the per-neighbour-row use -- the slab is
// `slab` rows (a multiple of DEC_SLAB), walked in chunks of DEC_SLAB; the layer-1 input rows are J.msgs, g_pred is per row.
template <bool ROWS = false>
__device__ __forceinline__ void dec_bwd_dw_body(const DecArgs& A, int b, int p, int w, int lane, long long slab = DEC_SLAB) {
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const long long R = J.agents;
    const long long s0 = (long long)p * slab < R ? (long long)p * slab : R;
    const long long s1e = s0 + slab < R ? s0 + slab : R;
    const float* __restrict__ in1 = ROWS ? J.msgs : J.pooled;
    const float* __restrict__ gpr = ROWS ? J.g_pred_rows : A.g_pred;
    const int i = lane & 31, h = lane >> 5;
    const int mb1 = w >> 2, nb1 = w & 3, mb2 = (w >> 1) & 1, nb2 = w & 1;
    const bool do2 = w < 4, do3 = w == 4 || w == 5;
    f32x16 c1, c2, c3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    // a chunk is DEC_SLAB agents = DEC_SLAB / 2 k-steps: every load of the chunk is issued before its first MFMA
    constexpr int U = DEC_SLAB / 2;
    for (long long r0 = s0; r0 < s1e || r0 == s0; r0 += DEC_SLAB) {
        const long long r1 = r0 + DEC_SLAB < s1e ? r0 + DEC_SLAB : s1e;
        float a1[U], b1[U], a2[U], b2[U], a3[U], b3[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = r0 + 2 * u + h;
            const bool ok = row < r1;
            const long long ro = ok ? row : (r0 < R ? r0 : 0);
            a1[u] = J.g_pre1[ro * DD + 32 * mb1 + i];
            b1[u] = in1[ro * DH + 32 * nb1 + i];
            a2[u] = do2 ? J.g_pre2[ro * DD + 32 * mb2 + i] : 0.f;
            b2[u] = do2 ? J.h1[ro * DD + 32 * nb2 + i] : 0.f;
            a3[u] = (do3 && i < 2) ? gpr[ro * 2 + i] : 0.f;
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

// The same partials by FOUR waves (the (tile, branch) workgroups of dec_bwd_split_kernel): wave w owns blocks (0, w) and
// (1, w) of dW1 (they share the pooled-column operand), block (w >> 1, w & 1) of dW2, waves 0 / 1 column block w of dW3.
__device__ __forceinline__ void dec_bwd_dw_body4(const DecArgs& A, int b, int p, int w, int lane, const float* keep_g2, const float* keep_g1) {
    // keep_g2 / keep_g1: the tile's gradients from the dX chain of the same workgroup (LDS [agent][64], zero past the end).
    // Round 5 (in-kernel stamps, tools/dec_stamps.py): as one loop with 64-bit clamped row indices, per-lane predicated loads and a
    // run-time choice of the gradients' source, REQUESTING the operands took 4.6 k of the workgroup's 26 k clocks and the slot stores
    // 2.3 k -- 40 clocks per memory instruction, all address arithmetic and branches.  Now: one base pointer per operand, a full slab
    // (wave-uniform) addressed by compile-time offsets, wave-uniform branches around the waves' optional third product.
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const long long R = J.agents;
    const long long s0 = (long long)p * DEC_SLAB < R ? (long long)p * DEC_SLAB : R;
    const int n = (int)(R - s0 < DEC_SLAB ? R - s0 : DEC_SLAB);          // agents of the slab (wave-uniform)
    const long long sb = n > 0 ? s0 : 0;
    const int i = lane & 31, h = lane >> 5;
    const int mb2 = w >> 1, nb2 = w & 1;
    const bool do3 = w < 2;
    constexpr int U = DEC_SLAB / 2;
    const float* __restrict__ pooled0 = J.pooled + sb * DH + 32 * w + i;
    const float* __restrict__ h10 = J.h1 + sb * DD + 32 * nb2 + i;
    const float* __restrict__ d20 = J.d2 + sb * DD + 32 * nb2 + i;
    const float* __restrict__ gp0 = A.g_pred + sb * 2 + (i & 1);
    const float* kg1 = keep_g1 + h * DD + i;
    const float* kg2 = keep_g2 + h * DD + 32 * mb2 + i;
    float a1a[U], a1b[U], b1[U], a2[U], b2[U], a3[U], b3[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                 // every load of the slab is issued before its first MFMA
        a1a[u] = kg1[2 * u * DD];
        a1b[u] = kg1[2 * u * DD + 32];
        a2[u] = kg2[2 * u * DD];
        a3[u] = 0.f; b3[u] = 0.f;
    }
    if (n == DEC_SLAB) {
        const float* pb = pooled0 + h * DH;
        const float* hb = h10 + h * DD;
#pragma unroll
        for (int u = 0; u < U; ++u) { b1[u] = pb[2 * u * DH]; b2[u] = hb[2 * u * DD]; }
        if (do3) {
            const float* db = d20 + h * DD;
            const float* gb = gp0 + h * 2;
#pragma unroll
            for (int u = 0; u < U; ++u) { b3[u] = db[2 * u * DD]; a3[u] = gb[4 * u]; }
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {             // rows past the end read the slab's first row; their products are masked
            const int rc = 2 * u + h < n ? 2 * u + h : 0;
            b1[u] = pooled0[rc * DH]; b2[u] = h10[rc * DD];
        }
        if (do3) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rc = 2 * u + h < n ? 2 * u + h : 0;
                b3[u] = d20[rc * DD]; a3[u] = gp0[rc * 2];
            }
        }
    }
    DEC_STAMP(8, false);
    DEC_STAMP(9, true);
    f32x16 c1a, c1b, c2, c3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1a[r] = 0.f; c1b[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    float s1a = 0.f, s1b = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        c1a = dmfma(a1a[u], b1[u], c1a);
        c1b = dmfma(a1b[u], b1[u], c1b);
        c2 = dmfma(a2[u], b2[u], c2);
        s1a += a1a[u]; s1b += a1b[u]; s2 += a2[u];
    }
    if (do3) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float x3 = (i < 2 && 2 * u + h < n) ? a3[u] : 0.f;        // g_pred has two columns
            c3 = dmfma(x3, b3[u], c3);
            s3 += x3;
        }
    }
    float* P = J.partials + (size_t)p * DEC_PART;
    DEC_STAMP(10, false);
    {   // (the stamp must see the products done: read one accumulator)
#ifdef PIML_DEC_STAMPS
        asm volatile("v_mov_b32 %0, %0" : "+v"(c1a[0]));
        DEC_STAMP(11, false);
#endif
    }
    {
        float* P1 = P + (size_t)(4 * h) * DH + 32 * w + i;                       // row rho(r) + 4 h of dW1's halves
        float* P2 = P + DD * DH + (32 * mb2 + 4 * h) * DD + 32 * nb2 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2);
            P1[rr * DH] = c1a[r];
            P1[(32 + rr) * DH] = c1b[r];
            P2[rr * DD] = c2[r];
        }
        if (do3 && h == 0) {                                                       // dW3's two rows are registers 0 and 1 of half 0
            float* P3 = P + DD * DH + DD * DD + 32 * nb2 + i;
            P3[0] = c3[0]; P3[DD] = c3[1];
        }
    }
    s1a += __shfl_xor(s1a, 32, 64);
    s1b += __shfl_xor(s1b, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    s3 += __shfl_xor(s3, 32, 64);
    float* Pb = P + DD * DH + DD * DD + 2 * DD;
    if (h == 0) {
        if (w == 0) { Pb[i] = s1a; Pb[32 + i] = s1b; }
        if (nb2 == 0) Pb[DD + 32 * mb2 + i] = s2;                  // waves 0 and 2
        if (w == 0 && i < 8) Pb[2 * DD + i] = i < 2 ? s3 : 0.f;    // db3 + padding
    }
    DEC_STAMP(12, false);
    DEC_STAMP(13, true);
}

// dX chain + weight-gradient partials per (32-agent tile, branch): 2 x tiles workgroups of four waves
__global__ __launch_bounds__(256) void dec_bwd_split_kernel(DecArgs A) {
    const int bx = blockIdx.x;
    const int b = A.nbr > 1 ? (bx & 1) : 0;
    const long long tile = A.nbr > 1 ? (bx >> 1) : bx;
    // g_pre2 / g_pre1 of the tile never leave the workgroup: the dX chain leaves them in LDS ([agent][64]) and the weight-gradient
    // products read them there -- no stores to drain in front of the barrier, no round trip through L2 behind it (round 4)
    __shared__ __align__(16) float keep[2][32 * DD];
    DEC_STAMP(0, false);
    dec_bwd_dx_body<false, true>(A, tile, b, keep[0], keep[1]);
    DEC_STAMP(6, false);
    __syncthreads();
    DEC_STAMP(7, false);
    dec_bwd_dw_body4(A, b, (int)tile, uniform((int)(threadIdx.x >> 6)), threadIdx.x & 63, keep[0], keep[1]);
}

__global__ __launch_bounds__(512) void dec_bwd_dw_kernel(DecArgs A) {
    const int lane = threadIdx.x & 63, w = uniform((int)(threadIdx.x >> 6));
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    dec_bwd_dw_body<false>(A, b, (int)blockIdx.x - (b ? A.wg_split : 0), w, lane);
}

// dX chain and weight-gradient partials of a 32-agent tile in ONE launch: the dW slab of a workgroup is exactly the tile
// whose g_pre2 / g_pre1 it has just written (visible to the whole workgroup after the barrier: one CU, one L1).
__global__ __launch_bounds__(512) void dec_bwd_kernel(DecArgs A) {
    dec_bwd_dx_body<false>(A, blockIdx.x);
    __threadfence_block();
    __syncthreads();
    const int lane = threadIdx.x & 63, w = uniform((int)(threadIdx.x >> 6));
    for (int b = 0; b < A.nbr; ++b) dec_bwd_dw_body<false>(A, b, (int)blockIdx.x, w, lane);
}

// ---- the same network per neighbour ROW (bottleneck variants): (tile, branch) workgroups, branches of different sizes ----
__global__ __launch_bounds__(256) void rowdec_fwd_kernel(DecArgs A, int tiles0) {
    const int bx = blockIdx.x;
    if (bx < tiles0) dec_fwd_body<false, true, true>(A, bx, 0);
    else dec_fwd_body<false, true, true>(A, bx - tiles0, 1);
}

__global__ __launch_bounds__(256) void rowdec_bwd_dx_kernel(DecArgs A, int tiles0) {
    const int bx = blockIdx.x;
    if (bx < tiles0) dec_bwd_dx_body<true>(A, bx, 0);
    else dec_bwd_dx_body<true>(A, bx - tiles0, 1);
}

// ---- many rows (the reference's 65 536 neighbour rows per step): the encoder kernels' structure -- one wave per 32-row
// tile, the branch's weight fragments staged once per workgroup in LDS, the three layers chained in registers (no LDS
// exchange, no barrier), tiles strided over 256 workgroups.  The (tile, branch) kernels above cut the chain across four
// waves for LATENCY (a few hundred tiles); here there are 2048 tiles for 2048 wave slots and the matrix pipe is the
// limit: 29 -> 21 us forward, 38 -> 24 us dX at the reference's row counts.
template <int NFLOATS>
__device__ __forceinline__ void rowdec_stage(float* lds, const float* __restrict__ src, int tid) {
    constexpr int N4 = NFLOATS / 4, ROUNDS = (N4 + 511) / 512;
    static_assert(NFLOATS % 4 == 0, "float4 granularity");
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(lds);
    float4 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int e = r * 512 + tid;
        v[r] = s4[e < N4 ? e : 0];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int e = r * 512 + tid;
        if (e < N4) d4[e] = v[r];
    }
}

__global__ __launch_bounds__(512) void rowdec_fwd_big_kernel(DecArgs A, int wg_split) {
    __shared__ __align__(16) float lds[DP_T3];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));
    const int b = (A.nbr > 1 && (int)blockIdx.x >= wg_split) ? 1 : 0;
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? wg_split : 0;
    const int nwg = b ? (int)gridDim.x - wg_split : (A.nbr > 1 ? wg_split : (int)gridDim.x);
    const long long R = J.agents, ntiles = (R + 31) >> 5;
    if ((long long)((int)blockIdx.x - wg0) * 8 >= ntiles) return;
    rowdec_stage<DP_T3>(lds, J.packed, tid);
    __syncthreads();
    const float4* F = reinterpret_cast<const float4*>(lds);
    const float* bias = lds + DP_B;
    const int j = lane & 31, h = lane >> 5;
    for (long long tile = (long long)((int)blockIdx.x - wg0) * 8 + wave; tile < ntiles; tile += (long long)nwg * 8) {
        const long long row = tile * 32 + j;
        const bool valid = row < R;
        f32x16 X[4];
        {
            const float* base = J.msgs + (valid ? row : 0) * DH;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(base + dfeat0(blk, q, h));
                    X[blk][4 * q] = valid ? v.x : 0.f; X[blk][4 * q + 1] = valid ? v.y : 0.f;
                    X[blk][4 * q + 2] = valid ? v.z : 0.f; X[blk][4 * q + 3] = valid ? v.w : 0.f;
                }
        }
        f32x16 a1[2], a2[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + dfeat0(ob, q, h));
                a1[ob][4 * q] = bq.x; a1[ob][4 * q + 1] = bq.y; a1[ob][4 * q + 2] = bq.z; a1[ob][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int bp = 0; bp < 4; ++bp) {
                __builtin_amdgcn_sched_barrier(0);       // a block's fragment reads at a time (hoisting them all spills)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = F[DP_A1 / 4 + ((ob * 4 + bp) * 4 + q) * 64 + lane];
                    a1[ob] = dmfma(w.x, X[bp][4 * q + 0], a1[ob]);
                    a1[ob] = dmfma(w.y, X[bp][4 * q + 1], a1[ob]);
                    a1[ob] = dmfma(w.z, X[bp][4 * q + 2], a1[ob]);
                    a1[ob] = dmfma(w.w, X[bp][4 * q + 3], a1[ob]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[ob][r] = fmaxf(a1[ob][r], 0.f);
        }
        if (valid) {
            float* o = J.h1 + row * DD;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) = make_float4(a1[ob][4 * q], a1[ob][4 * q + 1], a1[ob][4 * q + 2], a1[ob][4 * q + 3]);
        }
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + 64 + dfeat0(ob, q, h));
                a2[ob][4 * q] = bq.x; a2[ob][4 * q + 1] = bq.y; a2[ob][4 * q + 2] = bq.z; a2[ob][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int bp = 0; bp < 2; ++bp) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = F[DP_A2 / 4 + ((ob * 2 + bp) * 4 + q) * 64 + lane];
                    a2[ob] = dmfma(w.x, a1[bp][4 * q + 0], a2[ob]);
                    a2[ob] = dmfma(w.y, a1[bp][4 * q + 1], a2[ob]);
                    a2[ob] = dmfma(w.z, a1[bp][4 * q + 2], a2[ob]);
                    a2[ob] = dmfma(w.w, a1[bp][4 * q + 3], a2[ob]);
                }
            }
        }
        if (valid) {
            float* o = J.d2 + row * DD;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) = make_float4(a2[ob][4 * q], a2[ob][4 * q + 1], a2[ob][4 * q + 2], a2[ob][4 * q + 3]);
        }
        f32x16 a3;
#pragma unroll
        for (int r = 0; r < 16; ++r) a3[r] = 0.f;
        if (h == 0) { a3[0] = bias[128]; a3[1] = bias[129]; }
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = F[DP_A3 / 4 + (bp * 4 + q) * 64 + lane];
                a3 = dmfma(w.x, a2[bp][4 * q + 0], a3);
                a3 = dmfma(w.y, a2[bp][4 * q + 1], a3);
                a3 = dmfma(w.z, a2[bp][4 * q + 2], a3);
                a3 = dmfma(w.w, a2[bp][4 * q + 3], a3);
            }
        }
        if (h == 0 && valid) reinterpret_cast<float2*>(J.pred)[row] = make_float2(a3[0], a3[1]);
    }
}

constexpr int RDX_LDS = DEC_PACK - DP_T3;        // W3^T | W2^T | W1^T fragments

__global__ __launch_bounds__(512) void rowdec_bwd_dx_big_kernel(DecArgs A, int wg_split) {
    __shared__ __align__(16) float lds[RDX_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));
    const int b = (A.nbr > 1 && (int)blockIdx.x >= wg_split) ? 1 : 0;
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? wg_split : 0;
    const int nwg = b ? (int)gridDim.x - wg_split : (A.nbr > 1 ? wg_split : (int)gridDim.x);
    const long long R = J.agents, ntiles = (R + 31) >> 5;
    if ((long long)((int)blockIdx.x - wg0) * 8 >= ntiles) return;
    rowdec_stage<RDX_LDS>(lds, J.packed + DP_T3, tid);
    __syncthreads();
    const float* T3 = lds;
    const float4* T2 = reinterpret_cast<const float4*>(lds + (DP_T2 - DP_T3));
    const float4* T1 = reinterpret_cast<const float4*>(lds + (DP_T1 - DP_T3));
    const int j = lane & 31, h = lane >> 5;
    for (long long tile = (long long)((int)blockIdx.x - wg0) * 8 + wave; tile < ntiles; tile += (long long)nwg * 8) {
        const long long row = tile * 32 + j;
        const bool valid = row < R;
        const long long rr = valid ? row : 0;
        float2 gp = reinterpret_cast<const float2*>(J.g_pred_rows)[rr];
        if (!valid) gp = make_float2(0.f, 0.f);
        const float bg = h ? gp.y : gp.x;
        float4 hv[2][4], gd[2][4];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hv[ob][q] = *reinterpret_cast<const float4*>(J.h1 + rr * DD + dfeat0(ob, q, h));
                gd[ob][q] = J.g_d2 ? *reinterpret_cast<const float4*>(J.g_d2 + rr * DD + dfeat0(ob, q, h)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        f32x16 g2[2], g1[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g2[ob][r] = 0.f;
            g2[ob] = dmfma(T3[64 * ob + lane], bg, g2[ob]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                g2[ob][4 * q] += valid ? gd[ob][q].x : 0.f; g2[ob][4 * q + 1] += valid ? gd[ob][q].y : 0.f;
                g2[ob][4 * q + 2] += valid ? gd[ob][q].z : 0.f; g2[ob][4 * q + 3] += valid ? gd[ob][q].w : 0.f;
            }
            if (valid) {
                float* o = J.g_pre2 + row * DD;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) = make_float4(g2[ob][4 * q], g2[ob][4 * q + 1], g2[ob][4 * q + 2], g2[ob][4 * q + 3]);
            }
        }
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[ob][r] = 0.f;
#pragma unroll
            for (int bp = 0; bp < 2; ++bp) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = T2[((ob * 2 + bp) * 4 + q) * 64 + lane];
                    g1[ob] = dmfma(w.x, g2[bp][4 * q + 0], g1[ob]);
                    g1[ob] = dmfma(w.y, g2[bp][4 * q + 1], g1[ob]);
                    g1[ob] = dmfma(w.z, g2[bp][4 * q + 2], g1[ob]);
                    g1[ob] = dmfma(w.w, g2[bp][4 * q + 3], g1[ob]);
                }
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
                float* o = J.g_pre1 + row * DD;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) = make_float4(g1[ob][4 * q], g1[ob][4 * q + 1], g1[ob][4 * q + 2], g1[ob][4 * q + 3]);
            }
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x16 ge;
#pragma unroll
            for (int r = 0; r < 16; ++r) ge[r] = 0.f;
#pragma unroll
            for (int bp = 0; bp < 2; ++bp) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = T1[((blk * 2 + bp) * 4 + q) * 64 + lane];
                    ge = dmfma(w.x, g1[bp][4 * q + 0], ge);
                    ge = dmfma(w.y, g1[bp][4 * q + 1], ge);
                    ge = dmfma(w.z, g1[bp][4 * q + 2], ge);
                    ge = dmfma(w.w, g1[bp][4 * q + 3], ge);
                }
            }
            if (valid) {
                float* o = J.g_pooled + row * DH;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(blk, q, h)) = make_float4(ge[4 * q], ge[4 * q + 1], ge[4 * q + 2], ge[4 * q + 3]);
            }
        }
    }
}

// ---- the same two kernels on SPLIT bf16 products (round 6; x3.hpp / encoder_x3.hip: every f32 operand as three bf16 pieces, a
// k-block of 16 as six v_mfma_f32_32x32x16_bf16 = 192 cycles against 8 x 64 of the f32 instruction).  No new packed image: a split
// fragment (fb = (output block, k-block kb), lane (i, h), element t = W[32 ob + i][16 kb + 8 (t >> 2) + 4 h + (t & 3)]) is the two
// float4 (q = 2 (kb & 1), 2 (kb & 1) + 1; bp = kb >> 1) of the SAME lane of the f32 fragment image, so the workgroup splits the
// weights while it stages them (three items per thread, once per launch).  The 64 -> 2 predictor is 64 FMAs per lane and one
// cross-half add instead of 32 padded matrix instructions; W3^T (k = 2) stays on the f32 instruction (two per tile).
// LDS (u32x4): [fb][piece 3][lane 64]. ----
// dX kernel: 1 = the first tile's row loads between the staging's loads and its split, 0 (default, measured 0.5 us ahead) = behind
// the staging's LDS stores
#ifndef PIML_ROWDEC_HOIST
#define PIML_ROWDEC_HOIST 0
#endif
struct RowPieces { u32x4 hi[8], mid[8], lo[8]; };
template <int NB>          // NB accumulator blocks -> 2 NB k-blocks
__device__ __forceinline__ void rowdec_split(const f32x16 (&in)[NB], RowPieces& P) {
#pragma unroll
    for (int kb = 0; kb < 2 * NB; ++kb) {
        const f32x16& a = in[kb >> 1];
        const int r = 8 * (kb & 1);
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) split3(a[r + 2 * d], a[r + 2 * d + 1], hi[d], mid[d], lo[d]);
        P.hi[kb] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
        P.mid[kb] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
        P.lo[kb] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
    }
}
// the f32 fragment image `src` ([ob][bp NBP][q 4][lane 64] float4) of NOB x NBP blocks -> split fragments [fb = ob * 2 NBP + kb][3][64];
// in two halves: the loads (issued in FRONT of the first tile's row loads: the memory counter retires in order, and the weights
// come from L2) and the split + LDS stores (behind them, while the rows are on their way)
template <int NOB, int NBP>
struct RowdecStage {
    static constexpr int ITEMS = NOB * 2 * NBP * 64, ROUNDS = (ITEMS + 511) / 512;
    float4 a[ROUNDS], b[ROUNDS];
    __device__ __forceinline__ void load(const float* __restrict__ src, int tid) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int e = (r * 512 + tid) < ITEMS ? r * 512 + tid : 0;
            const int lane = e & 63, fb = e >> 6, kb = fb % (2 * NBP), ob = fb / (2 * NBP);
            a[r] = s4[((ob * NBP + (kb >> 1)) * 4 + 2 * (kb & 1)) * 64 + lane];
            b[r] = s4[((ob * NBP + (kb >> 1)) * 4 + 2 * (kb & 1) + 1) * 64 + lane];
        }
    }
    __device__ __forceinline__ void land(u32x4* dst, int tid) const {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int e = r * 512 + tid;
            if (ITEMS % 512 == 0 || e < ITEMS) {
                const int lane = e & 63, fb = e >> 6;
                unsigned hi[4], mid[4], lo[4];
                split3(a[r].x, a[r].y, hi[0], mid[0], lo[0]);
                split3(a[r].z, a[r].w, hi[1], mid[1], lo[1]);
                split3(b[r].x, b[r].y, hi[2], mid[2], lo[2]);
                split3(b[r].z, b[r].w, hi[3], mid[3], lo[3]);
                dst[(fb * 3 + 0) * 64 + lane] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
                dst[(fb * 3 + 1) * 64 + lane] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
                dst[(fb * 3 + 2) * 64 + lane] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
            }
        }
    }
};

constexpr int RFX_W1 = 0, RFX_W2 = RFX_W1 + 16 * 3 * 64, RFX_F32 = RFX_W2 + 8 * 3 * 64;      // u32x4 units; then floats: bias 132 | W3 rows 128
constexpr int RFX_LDS_BYTES = RFX_F32 * 16 + (132 + 128) * 4;

__global__ __launch_bounds__(512) void rowdec_fwd_x3_kernel(DecArgs A, int wg_split) {
    extern __shared__ __align__(16) unsigned char rx_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));
    const int b = (A.nbr > 1 && (int)blockIdx.x >= wg_split) ? 1 : 0;
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? wg_split : 0;
    const int nwg = b ? (int)gridDim.x - wg_split : (A.nbr > 1 ? wg_split : (int)gridDim.x);
    const long long R = J.agents, ntiles = (R + 31) >> 5;
    if ((long long)((int)blockIdx.x - wg0) * 8 >= ntiles) return;
    u32x4* img = reinterpret_cast<u32x4*>(rx_smem);
    float* fl = reinterpret_cast<float*>(rx_smem + RFX_F32 * 16);
    const int j = lane & 31, h = lane >> 5;
    const long long first = (long long)((int)blockIdx.x - wg0) * 8 + wave, stride = (long long)nwg * 8;
    {
        RowdecStage<2, 4> S1;
        RowdecStage<2, 2> S2;
        S1.load(J.packed + DP_A1, tid);
        S2.load(J.packed + DP_A2, tid);
        const float fv = tid < 132 ? J.packed[DP_B + tid] : ((tid >= 256 && tid < 384) ? J.w3[tid - 256] : 0.f);
        S1.land(img + RFX_W1, tid);
        S2.land(img + RFX_W2, tid);
        if (tid < 132) fl[tid] = fv;
        else if (tid >= 256 && tid < 384) fl[132 + tid - 256] = fv;
    }
    __syncthreads();
    const float* bias = fl;
    const float* w3r = fl + 132;
    for (long long tile = first; tile < ntiles; tile += stride) {
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));                      // (opaque per tile: keeps the fragment reads of a tile inside the loop)
        const u32x4* W1 = img + RFX_W1 + lane_t;
        const u32x4* W2 = img + RFX_W2 + lane_t;
        const long long row = tile * 32 + j;
        const bool valid = row < R;
        RowPieces P;
        {
            // (the rows are requested HERE: in front of the weight staging, or a tile ahead, measured 2 us slower at the reference's
            // row counts, where a wave has one tile -- 214 registers instead of 162 and every wave's loads in one burst)
            f32x16 X[4];
            const float* base = J.msgs + (valid ? row : 0) * DH;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(base + dfeat0(blk, q, h));
                    X[blk][4 * q] = valid ? v.x : 0.f; X[blk][4 * q + 1] = valid ? v.y : 0.f;
                    X[blk][4 * q + 2] = valid ? v.z : 0.f; X[blk][4 * q + 3] = valid ? v.w : 0.f;
                }
            rowdec_split<4>(X, P);
        }
        f32x16 a1[2], a2[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            f32x16 acc, sm;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + dfeat0(ob, q, h));
                acc[4 * q] = bq.x; acc[4 * q + 1] = bq.y; acc[4 * q + 2] = bq.z; acc[4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = ob * 8 + kb;
                kblock_x3(acc, sm, W1[(fb * 3) * 64], W1[(fb * 3 + 1) * 64], W1[(fb * 3 + 2) * 64], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[ob][r] = relu1(acc[r] + sm[r]);
        }
        if (valid) {
            float* o = J.h1 + row * DD;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) = make_float4(a1[ob][4 * q], a1[ob][4 * q + 1], a1[ob][4 * q + 2], a1[ob][4 * q + 3]);
        }
        rowdec_split<2>(a1, P);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            f32x16 acc, sm;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + 64 + dfeat0(ob, q, h));
                acc[4 * q] = bq.x; acc[4 * q + 1] = bq.y; acc[4 * q + 2] = bq.z; acc[4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int fb = ob * 4 + kb;
                kblock_x3(acc, sm, W2[(fb * 3) * 64], W2[(fb * 3 + 1) * 64], W2[(fb * 3 + 2) * 64], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[ob][r] = acc[r] + sm[r];
        }
        if (valid) {
            float* o = J.d2 + row * DD;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) = make_float4(a2[ob][4 * q], a2[ob][4 * q + 1], a2[ob][4 * q + 2], a2[ob][4 * q + 3]);
        }
        // the predictor: this lane's 32 features of the row, then the other lane half's
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 wa = *reinterpret_cast<const float4*>(w3r + dfeat0(ob, q, h));
                const float4 wb = *reinterpret_cast<const float4*>(w3r + 64 + dfeat0(ob, q, h));
                p0 = __fmaf_rn(a2[ob][4 * q], wa.x, p0); p0 = __fmaf_rn(a2[ob][4 * q + 1], wa.y, p0);
                p0 = __fmaf_rn(a2[ob][4 * q + 2], wa.z, p0); p0 = __fmaf_rn(a2[ob][4 * q + 3], wa.w, p0);
                p1 = __fmaf_rn(a2[ob][4 * q], wb.x, p1); p1 = __fmaf_rn(a2[ob][4 * q + 1], wb.y, p1);
                p1 = __fmaf_rn(a2[ob][4 * q + 2], wb.z, p1); p1 = __fmaf_rn(a2[ob][4 * q + 3], wb.w, p1);
            }
        p0 += __shfl_xor(p0, 32, 64);
        p1 += __shfl_xor(p1, 32, 64);
        if (h == 0 && valid) reinterpret_cast<float2*>(J.pred)[row] = make_float2(p0 + bias[128], p1 + bias[129]);
    }
}

constexpr int RDXX_T2 = 0, RDXX_T1 = RDXX_T2 + 8 * 3 * 64, RDXX_F32 = RDXX_T1 + 16 * 3 * 64;      // u32x4 units; then W3^T 128 floats
constexpr int RDXX_LDS_BYTES = RDXX_F32 * 16 + 128 * 4;

__global__ __launch_bounds__(512) void rowdec_bwd_dx_x3_kernel(DecArgs A, int wg_split) {
    extern __shared__ __align__(16) unsigned char rx_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));
    const int b = (A.nbr > 1 && (int)blockIdx.x >= wg_split) ? 1 : 0;
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? wg_split : 0;
    const int nwg = b ? (int)gridDim.x - wg_split : (A.nbr > 1 ? wg_split : (int)gridDim.x);
    const long long R = J.agents, ntiles = (R + 31) >> 5;
    if ((long long)((int)blockIdx.x - wg0) * 8 >= ntiles) return;
    u32x4* img = reinterpret_cast<u32x4*>(rx_smem);
    float* T3 = reinterpret_cast<float*>(rx_smem + RDXX_F32 * 16);
    const int j = lane & 31, h = lane >> 5;
    const long long first = (long long)((int)blockIdx.x - wg0) * 8 + wave, stride = (long long)nwg * 8;
    // the tile's rows: requested in front of the weight staging (a further tile's: behind the first products of this one)
    float2 gpv;
    float4 hvv[2][4], gdv[2][4];
    auto load_rows = [&](long long tile_) {
        const long long row_ = tile_ * 32 + j, rr_ = row_ < R ? row_ : 0;
        gpv = reinterpret_cast<const float2*>(J.g_pred_rows)[rr_];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hvv[ob][q] = *reinterpret_cast<const float4*>(J.h1 + rr_ * DD + dfeat0(ob, q, h));
                gdv[ob][q] = J.g_d2 ? *reinterpret_cast<const float4*>(J.g_d2 + rr_ * DD + dfeat0(ob, q, h)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    };
    {
        RowdecStage<2, 2> S2;
        RowdecStage<4, 2> S1;
        S2.load(J.packed + DP_T2, tid);
        S1.load(J.packed + DP_T1, tid);
        const float fv = tid < 128 ? J.packed[DP_T3 + tid] : 0.f;
        if (PIML_ROWDEC_HOIST && first < ntiles) load_rows(first);
        S2.land(img + RDXX_T2, tid);
        S1.land(img + RDXX_T1, tid);
        if (tid < 128) T3[tid] = fv;
    }
    if (!PIML_ROWDEC_HOIST && first < ntiles) load_rows(first);
    __syncthreads();
    for (long long tile = first; tile < ntiles; tile += stride) {
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const u32x4* T2 = img + RDXX_T2 + lane_t;
        const u32x4* T1 = img + RDXX_T1 + lane_t;
        const long long row = tile * 32 + j;
        const bool valid = row < R;
        float2 gp = gpv;
        if (!valid) gp = make_float2(0.f, 0.f);
        const float bg = h ? gp.y : gp.x;
        float4 hv[2][4], gd[2][4];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int q = 0; q < 4; ++q) { hv[ob][q] = hvv[ob][q]; gd[ob][q] = gdv[ob][q]; }
        f32x16 g2[2], g1[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g2[ob][r] = 0.f;
            g2[ob] = dmfma(T3[64 * ob + lane_t], bg, g2[ob]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                g2[ob][4 * q] += valid ? gd[ob][q].x : 0.f; g2[ob][4 * q + 1] += valid ? gd[ob][q].y : 0.f;
                g2[ob][4 * q + 2] += valid ? gd[ob][q].z : 0.f; g2[ob][4 * q + 3] += valid ? gd[ob][q].w : 0.f;
            }
            if (valid) {
                float* o = J.g_pre2 + row * DD;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) = make_float4(g2[ob][4 * q], g2[ob][4 * q + 1], g2[ob][4 * q + 2], g2[ob][4 * q + 3]);
            }
        }
        RowPieces P;
        rowdec_split<2>(g2, P);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            f32x16 acc, sm;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; sm[r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int fb = ob * 4 + kb;
                kblock_x3(acc, sm, T2[(fb * 3) * 64], T2[(fb * 3 + 1) * 64], T2[(fb * 3 + 2) * 64], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = hv[ob][q];
                g1[ob][4 * q + 0] = (valid && a.x > 0.f) ? acc[4 * q + 0] + sm[4 * q + 0] : 0.f;
                g1[ob][4 * q + 1] = (valid && a.y > 0.f) ? acc[4 * q + 1] + sm[4 * q + 1] : 0.f;
                g1[ob][4 * q + 2] = (valid && a.z > 0.f) ? acc[4 * q + 2] + sm[4 * q + 2] : 0.f;
                g1[ob][4 * q + 3] = (valid && a.w > 0.f) ? acc[4 * q + 3] + sm[4 * q + 3] : 0.f;
            }
            if (valid) {
                float* o = J.g_pre1 + row * DD;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(ob, q, h)) = make_float4(g1[ob][4 * q], g1[ob][4 * q + 1], g1[ob][4 * q + 2], g1[ob][4 * q + 3]);
            }
        }
        rowdec_split<2>(g1, P);
        if (tile + stride < ntiles) load_rows(tile + stride);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x16 acc, sm;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; sm[r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int fb = blk * 4 + kb;
                kblock_x3(acc, sm, T1[(fb * 3) * 64], T1[(fb * 3 + 1) * 64], T1[(fb * 3 + 2) * 64], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
            if (valid) {
                float* o = J.g_pooled + row * DH;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + dfeat0(blk, q, h)) =
                        make_float4(acc[4 * q] + sm[4 * q], acc[4 * q + 1] + sm[4 * q + 1], acc[4 * q + 2] + sm[4 * q + 2], acc[4 * q + 3] + sm[4 * q + 3]);
            }
        }
    }
}

// The row-wise weight gradients with LDS staging (the encoder dW kernel's structure): a slab holds hundreds of rows, and
// the agent-level body above would fetch every chunk with 96 dword loads per lane, several waves fetching the same
// columns, and wait for each chunk in turn.  Here the workgroup stages 32-row chunks of the five operand arrays with
// 16-byte loads (every byte once), double-buffered: the loads of chunk t+1 are in flight during the MFMAs of chunk t.
constexpr int RD_CHUNK = 32;
constexpr int RD_G1 = 0, RD_G2 = RD_CHUNK * DD, RD_H1 = 2 * RD_CHUNK * DD, RD_D2 = 3 * RD_CHUNK * DD, RD_E = 4 * RD_CHUNK * DD,
              RD_GP = RD_E + RD_CHUNK * DH, RD_BUF = RD_GP + RD_CHUNK * 2;

// the products of one 32-row chunk for a wave's role (1: c1 alone, 2: c1 + c2, 3: c1 + c3; cx / sx = the second accumulator and
// its column sum): a wave's role is fixed, so one branch per chunk and straight-line code behind it, the operands of four k-steps
// read ahead of their products -- with the role tested inside the k loop every product sat behind its own two LDS reads and
// their full latency.  (Eight k-steps ahead, or the sums captured by a nested lambda: scratch traffic inside the products, whose
// reloads wait for every global load in flight.)
template <int ROLE>
__device__ __forceinline__ float2 rd_products(const float* buf, int h, int i, int mb1, int nb1, int mb2, int nb2, f32x16& c1, f32x16& cx,
                                              float s1, float sx) {
    constexpr int AHEAD = 4;                // (s1, sx: the running column sums of the two A operands, in and out BY VALUE -- as
                                            // reference parameters they ended up in scratch; one running sum: the order of the
                                            // additions is the k order, as before)
#pragma unroll
    for (int k0 = 0; k0 < RD_CHUNK / 2; k0 += AHEAD) {
        float a1[AHEAD], b1[AHEAD], a2[AHEAD], b2[AHEAD];
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            const int row = 2 * (k0 + u) + h;
            a1[u] = buf[RD_G1 + row * DD + 32 * mb1 + i]; b1[u] = buf[RD_E + row * DH + 32 * nb1 + i];
            if (ROLE == 2) { a2[u] = buf[RD_G2 + row * DD + 32 * mb2 + i]; b2[u] = buf[RD_H1 + row * DD + 32 * nb2 + i]; }
            if (ROLE == 3) { a2[u] = i < 2 ? buf[RD_GP + row * 2 + i] : 0.f; b2[u] = buf[RD_D2 + row * DD + 32 * nb2 + i]; }
        }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            c1 = dmfma(a1[u], b1[u], c1);
            s1 += a1[u];
            if (ROLE != 1) { cx = dmfma(a2[u], b2[u], cx); sx += a2[u]; }
        }
    }
    return make_float2(s1, sx);
}

__global__ __launch_bounds__(512) void rowdec_bwd_dw_lds_kernel(DecArgs A, int slots0, long long slab0, long long slab1) {
    extern __shared__ __align__(16) float rd_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = uniform((int)(tid >> 6));
    const int b = (int)blockIdx.x >= slots0 ? 1 : 0;
    const int p = (int)blockIdx.x - (b ? slots0 : 0);
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const long long R = J.agents, slab = b ? slab1 : slab0;
    const long long s0 = (long long)p * slab < R ? (long long)p * slab : R;
    const long long s1e = s0 + slab < R ? s0 + slab : R;
    const int i = lane & 31, h = lane >> 5;
    const int mb1 = w >> 2, nb1 = w & 3, mb2 = (w >> 1) & 1, nb2 = w & 1;
    const bool do2 = w < 4, do3 = w == 4 || w == 5;
    f32x16 c1, cx;           // cx: the wave's second product (do2: dW2 block, do3: the predictor's dW), sx its column sum
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1[r] = 0.f; cx[r] = 0.f; }
    float s1 = 0.f, sx = 0.f;
    // staging role: row srow of the chunk, float4 column sc4 of the 64-wide arrays (and sc4, sc4 + 64 of the embeddings)
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;
    struct Stage { float4 g1, g2, h1, d2, e0, e1; float gp; bool ok; };
    auto stage_load = [&](long long rb) -> Stage {
        Stage S;
        const long long row = rb + srow;
        S.ok = row < s1e;
        const long long ro = S.ok ? row : (s0 < R ? s0 : 0);       // clamped: a readable row
        S.g1 = *reinterpret_cast<const float4*>(J.g_pre1 + ro * DD + sc4);
        S.g2 = *reinterpret_cast<const float4*>(J.g_pre2 + ro * DD + sc4);
        S.h1 = *reinterpret_cast<const float4*>(J.h1 + ro * DD + sc4);
        S.d2 = *reinterpret_cast<const float4*>(J.d2 + ro * DD + sc4);
        S.e0 = *reinterpret_cast<const float4*>(J.msgs + ro * DH + sc4);
        S.e1 = *reinterpret_cast<const float4*>(J.msgs + ro * DH + 64 + sc4);
        // (every thread loads -- clamped -- and the value is looked at only when it is written to LDS: as a conditional load
        // with its select right behind it the compiler put s_waitcnt vmcnt(0) in FRONT of the chunk's products in wave 0, and
        // the whole workgroup waited for that wave at the barrier: loads and products took turns)
        const long long gr = rb + ((tid & 63) >> 1);               // threads 0..63 matter: the chunk's (row, component) of g_pred
        S.gp = J.g_pred_rows[(gr < s1e ? gr : (s0 < R ? s0 : 0)) * 2 + (tid & 1)];
        return S;
    };
    auto stage_write = [&](const Stage S, float* buf, long long rb) {
        auto sel = [&](const float4 v) { return make_float4(S.ok ? v.x : 0.f, S.ok ? v.y : 0.f, S.ok ? v.z : 0.f, S.ok ? v.w : 0.f); };
        *reinterpret_cast<float4*>(buf + RD_G1 + srow * DD + sc4) = sel(S.g1);
        *reinterpret_cast<float4*>(buf + RD_G2 + srow * DD + sc4) = sel(S.g2);
        *reinterpret_cast<float4*>(buf + RD_H1 + srow * DD + sc4) = sel(S.h1);
        *reinterpret_cast<float4*>(buf + RD_D2 + srow * DD + sc4) = sel(S.d2);
        *reinterpret_cast<float4*>(buf + RD_E + srow * DH + sc4) = sel(S.e0);
        *reinterpret_cast<float4*>(buf + RD_E + srow * DH + 64 + sc4) = sel(S.e1);
        if (tid < 2 * RD_CHUNK) buf[RD_GP + tid] = rb + (tid >> 1) < s1e ? S.gp : 0.f;
    };
    auto compute = [&](const float* buf) {
        float2 ds;
        if (do2) ds = rd_products<2>(buf, h, i, mb1, nb1, mb2, nb2, c1, cx, s1, sx);
        else if (do3) ds = rd_products<3>(buf, h, i, mb1, nb1, mb2, nb2, c1, cx, s1, sx);
        else ds = rd_products<1>(buf, h, i, mb1, nb1, mb2, nb2, c1, cx, s1, sx);
        s1 = ds.x; sx = ds.y;
    };
    if (s0 < s1e) {
        const int nb = (int)((s1e - s0 + RD_CHUNK - 1) / RD_CHUNK);
        Stage S = stage_load(s0);
        stage_write(S, rd_lds, s0);
        __syncthreads();
        for (int t = 0; t < nb; ++t) {
            float* cur = rd_lds + (t & 1) * RD_BUF;
            float* nxt = rd_lds + ((t + 1) & 1) * RD_BUF;
            S = stage_load(s0 + (long long)(t + 1) * RD_CHUNK);    // past the slab: clamped + zeroed, written but never read
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            stage_write(S, nxt, s0 + (long long)(t + 1) * RD_CHUNK);
            __syncthreads();
        }
    }
    float* P = J.partials + (size_t)p * DEC_PART;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * h;
        P[(size_t)(32 * mb1 + ri) * DH + 32 * nb1 + i] = c1[r];
        if (do2) P[DD * DH + (32 * mb2 + ri) * DD + 32 * nb2 + i] = cx[r];
        if (do3 && ri < 2) P[DD * DH + DD * DD + ri * DD + 32 * nb2 + i] = cx[r];
    }
    s1 += __shfl_xor(s1, 32, 64);
    sx += __shfl_xor(sx, 32, 64);
    float* Pb = P + DD * DH + DD * DD + 2 * DD;
    if (h == 0) {
        if (nb1 == 0) Pb[32 * mb1 + i] = s1;                       // waves 0 and 4
        if (do2 && nb2 == 0) Pb[DD + 32 * mb2 + i] = sx;           // waves 0 and 2
        if (w == 4 && i < 8) Pb[2 * DD + i] = i < 2 ? sx : 0.f;    // db3 + padding
    }
}

// ---- the row-wise weight gradients on split bf16 products (round 6).  All three products contract over the ROWS, so both operands
// of each are activations: a 32-row chunk of the five arrays is split into bf16 pieces while it is staged and laid into three
// [piece 3][row 32][feature 128] images (8-byte chunks XOR-swizzled by the row, the layout of encoder_bwd3.hip's phase 2):
//     E = the embeddings (128)      G = g_pre1 | g_pre2 (64 | 64)      H = h1 | d2 (64 | 64)
// ds_read_b64_tr_b16 hands a lane its FEATURE's rows: fragments with the feature on the lane and the rows as the k index, for the A
// operand (g_pre1 / g_pre2 blocks) and the B operand (embedding / h1 / d2 blocks) alike.  Wave w owns block (w >> 2, w & 3) of dW1
// (64 x 128); waves 0 - 3 also block ((w >> 1) & 1, w & 1) of dW2 (64 x 64); waves 4, 5 the predictor's dW (2 x 64: the A fragment
// is g_pred's two columns, built from the chunk's f32 copy, 30 zero rows).  A chunk is 12 (+ 12) products per wave instead of 16
// (+ 16) f32 instructions of twice the cycles each.  The column sums (db1, db2, db3) are taken by the staging threads from the f32
// values on their way into the images: per thread over its chunks, then over the 32 staging rows in a fixed order.  Two image sets:
// one barrier per chunk. ----
constexpr int RWX_IMG = 3 * 32 * 256;                          // bytes of one image
constexpr int RWX_E = 0, RWX_G = RWX_IMG, RWX_H = 2 * RWX_IMG, RWX_GP = 3 * RWX_IMG, RWX_BUF = RWX_GP + 256;      // GP: [row 32][2] floats
constexpr int RWX_LDS_BYTES = 2 * RWX_BUF;
static_assert(RWX_LDS_BYTES <= 160 * 1024 && 32 * 132 * 4 <= RWX_LDS_BYTES, "fits the CU; the column sums' exchange fits the buffers");

__global__ __launch_bounds__(512) void rowdec_bwd_dw_x3_kernel(DecArgs A, int slots0, long long slab0, long long slab1) {
    extern __shared__ __align__(16) unsigned char rx_smem[];
    typedef short rw_s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) rw_s16x4 rw_lds_s16x4;
    const int tid = threadIdx.x, lane = tid & 63, w = uniform((int)(tid >> 6));
    const int b = (int)blockIdx.x >= slots0 ? 1 : 0;
    const int p = (int)blockIdx.x - (b ? slots0 : 0);
    const piml_decoder_branch J = b ? A.br[1] : A.br[0];
    const long long R = J.agents, slab = b ? slab1 : slab0;
    const long long s0 = (long long)p * slab < R ? (long long)p * slab : R;
    const long long s1e = s0 + slab < R ? s0 + slab : R;
    const int i = lane & 31, h = lane >> 5;
    const int mb1 = w >> 2, nb1 = w & 3, mb2 = (w >> 1) & 1, nb2 = w & 1;
    const bool do2 = w < 4, do3 = w == 4 || w == 5;
    f32x16 c1, s1, cx, sx;          // main and small-term accumulators (x3.hpp) of the wave's two products
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1[r] = 0.f; s1[r] = 0.f; cx[r] = 0.f; sx[r] = 0.f; }
    // staging role: row srow of the chunk, float4 column sc4 of the 64-wide arrays (and sc4, sc4 + 64 of the embeddings)
    const int srow = tid >> 4, c16 = tid & 15, sc4 = c16 * 4;
    float4 sum1 = make_float4(0.f, 0.f, 0.f, 0.f), sum2 = sum1;      // column sums of g_pre1 / g_pre2 over this thread's rows
    float sumgp = 0.f;                                               // of g_pred (threads 0 .. 63: row tid >> 1, component tid & 1)
    struct Stage { float4 g1, g2, h1, d2, e0, e1; float gp; bool ok; };
    auto stage_load = [&](long long rb) -> Stage {
        Stage S;
        const long long row = rb + srow;
        S.ok = row < s1e;
        const long long ro = S.ok ? row : (s0 < R ? s0 : 0);       // clamped: a readable row
        S.g1 = *reinterpret_cast<const float4*>(J.g_pre1 + ro * DD + sc4);
        S.g2 = *reinterpret_cast<const float4*>(J.g_pre2 + ro * DD + sc4);
        S.h1 = *reinterpret_cast<const float4*>(J.h1 + ro * DD + sc4);
        S.d2 = *reinterpret_cast<const float4*>(J.d2 + ro * DD + sc4);
        S.e0 = *reinterpret_cast<const float4*>(J.msgs + ro * DH + sc4);
        S.e1 = *reinterpret_cast<const float4*>(J.msgs + ro * DH + 64 + sc4);
        const long long gr = rb + ((tid & 63) >> 1);
        S.gp = J.g_pred_rows[(gr < s1e ? gr : (s0 < R ? s0 : 0)) * 2 + (tid & 1)];
        return S;
    };
    // chunk `chunk` (4 features) of row srow of an image: the three pieces of v
    auto lay = [&](unsigned char* im, int chunk, const float4 v) {
        unsigned h0, m0, l0, h1_, m1, l1;
        split3(v.x, v.y, h0, m0, l0);
        split3(v.z, v.w, h1_, m1, l1);
        unsigned char* d = im + srow * 256 + ((chunk ^ (8 * (srow & 3))) * 8);
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1_);
        *reinterpret_cast<uint2*>(d + 8192) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(d + 16384) = make_uint2(l0, l1);
    };
    auto stage_write = [&](const Stage S, unsigned char* buf, long long rb) {
        auto sel = [&](const float4 v) { return make_float4(S.ok ? v.x : 0.f, S.ok ? v.y : 0.f, S.ok ? v.z : 0.f, S.ok ? v.w : 0.f); };
        const float4 g1 = sel(S.g1), g2 = sel(S.g2);
        sum1.x += g1.x; sum1.y += g1.y; sum1.z += g1.z; sum1.w += g1.w;
        sum2.x += g2.x; sum2.y += g2.y; sum2.z += g2.z; sum2.w += g2.w;
        lay(buf + RWX_G, c16, g1);
        lay(buf + RWX_G, 16 + c16, g2);
        lay(buf + RWX_H, c16, sel(S.h1));
        lay(buf + RWX_H, 16 + c16, sel(S.d2));
        lay(buf + RWX_E, c16, sel(S.e0));
        lay(buf + RWX_E, 16 + c16, sel(S.e1));
        if (tid < 64) {
            const float g = rb + (tid >> 1) < s1e ? S.gp : 0.f;
            sumgp += g;
            reinterpret_cast<float*>(buf + RWX_GP)[tid] = g;
        }
    };
    // reader (encoder_bwd3.hip, phase 2): lane 4 q + pp of 16-lane group g16 supplies row r0 + q, chunk 8 blk + 4 (g16 & 1) + pp
    int tr_off;
    {
        const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        tr_off = (4 * (g16 >> 1) + q) * 256 + (4 * (g16 & 1) + pp) * 8;
    }
    const int trq = (lane >> 2) & 3;
    auto frag = [&](const unsigned char* im, int blk, int s_, int piece) -> u32x4 {
        const int o0 = tr_off + (16 * s_) * 256 + ((8 * blk) ^ (8 * trq)) * 8 + piece * 8192;
        const rw_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rw_lds_s16x4*)(im + o0));
        const rw_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rw_lds_s16x4*)(im + o0 + 8 * 256));
        const uint2 x = __builtin_bit_cast(uint2, lo4), y = __builtin_bit_cast(uint2, hi4);
        return (u32x4){x.x, x.y, y.x, y.y};
    };
    auto compute = [&](const unsigned char* buf) {
        u32x4 a[2][3], bq[2][3];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) { a[s_][pc] = frag(buf + RWX_G, mb1, s_, pc); bq[s_][pc] = frag(buf + RWX_E, nb1, s_, pc); }
        u32x4 a2[2][3], b2[2][3];
        if (do2) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) { a2[s_][pc] = frag(buf + RWX_G, 2 + mb2, s_, pc); b2[s_][pc] = frag(buf + RWX_H, nb2, s_, pc); }
        } else if (do3) {
            const float* gpl = reinterpret_cast<const float*>(buf + RWX_GP);
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = i < 2 ? gpl[(16 * s_ + 8 * (t >> 2) + 4 * h + (t & 3)) * 2 + i] : 0.f;
                unsigned hh[4], mm[4], ll[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) split3(v[2 * d], v[2 * d + 1], hh[d], mm[d], ll[d]);
                a2[s_][0] = (u32x4){hh[0], hh[1], hh[2], hh[3]};
                a2[s_][1] = (u32x4){mm[0], mm[1], mm[2], mm[3]};
                a2[s_][2] = (u32x4){ll[0], ll[1], ll[2], ll[3]};
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) b2[s_][pc] = frag(buf + RWX_H, 2 + nb2, s_, pc);
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            kblock_x3(c1, s1, a[s_][0], a[s_][1], a[s_][2], bq[s_][0], bq[s_][1], bq[s_][2]);
            if (do2 || do3) kblock_x3(cx, sx, a2[s_][0], a2[s_][1], a2[s_][2], b2[s_][0], b2[s_][1], b2[s_][2]);
        }
    };
    if (s0 < s1e) {
        const int nb = (int)((s1e - s0 + RD_CHUNK - 1) / RD_CHUNK);
        // the loads of chunk t + 2 are requested as soon as the registers of chunk t + 1 are free -- in front of the barrier -- and
        // looked at behind the products of chunk t + 1 (requested in front of the products of chunk t they had the products' time
        // only: 25.8 us for the launch at the reference's row counts)
        // (the barrier: this wave's LDS operations done, nothing said about its loads in flight -- __syncthreads() waits for them)
#ifndef PIML_RWX_EARLY
#define PIML_RWX_EARLY 1
#endif
        Stage S = stage_load(s0);
        stage_write(S, rx_smem, s0);
        if (PIML_RWX_EARLY) S = stage_load(s0 + RD_CHUNK);          // past the slab: clamped + zeroed, written but never read
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int t = 0; t < nb; ++t) {
            unsigned char* cur = rx_smem + (t & 1) * RWX_BUF;
            unsigned char* nxt = rx_smem + ((t + 1) & 1) * RWX_BUF;
            if (!PIML_RWX_EARLY) S = stage_load(s0 + (long long)(t + 1) * RD_CHUNK);
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            stage_write(S, nxt, s0 + (long long)(t + 1) * RD_CHUNK);
            if (PIML_RWX_EARLY) S = stage_load(s0 + (long long)(t + 2) * RD_CHUNK);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float* P = J.partials + (size_t)p * DEC_PART;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * h;
        P[(size_t)(32 * mb1 + ri) * DH + 32 * nb1 + i] = c1[r] + s1[r];
        if (do2) P[DD * DH + (32 * mb2 + ri) * DD + 32 * nb2 + i] = cx[r] + sx[r];
        if (do3 && ri < 2) P[DD * DH + DD * DD + ri * DD + 32 * nb2 + i] = cx[r] + sx[r];
    }
    // column sums: [staging row 32][g_pre1 64 | g_pre2 64 | g_pred 2 | pad 2] floats over the (dead) images, summed over the rows in order
    float* red = reinterpret_cast<float*>(rx_smem);
    *reinterpret_cast<float4*>(red + srow * 132 + sc4) = sum1;
    *reinterpret_cast<float4*>(red + srow * 132 + 64 + sc4) = sum2;
    if (tid < 64) red[(tid >> 1) * 132 + 128 + (tid & 1)] = sumgp;
    __syncthreads();
    float* Pb = P + DD * DH + DD * DD + 2 * DD;
    if (tid < 136) {
        float v = 0.f;
        if (tid < 130)
#pragma unroll 8
            for (int r = 0; r < 32; ++r) v += red[r * 132 + tid];
        Pb[tid] = v;                                                  // db1 64 | db2 64 | db3 2 | padding 6
    }
}

__global__ __launch_bounds__(256) void rowdec_reduce_kernel(DecArgs A, int B0, int B1, int lanes, int accumulate) {
    const piml_decoder_branch J = blockIdx.y ? A.br[1] : A.br[0];
    sum_slots_16x16(J.partials, J.grads, blockIdx.y ? B1 : B0, lanes, 0x7fffffff, 0, 0, accumulate != 0);
}

__global__ __launch_bounds__(256) void dec_reduce_kernel(DecArgs A, int B, int lanes) {
    const piml_decoder_branch J = blockIdx.y ? A.br[1] : A.br[0];
    sum_slots_16x16(J.partials, J.grads, B, lanes);
}

// ---------------------------------------------------------------------------------------------------------
// collision head of `pinnsf_m` (src/models/model.py:1246 `ped_collision_predictor = MLP(128, [64, 1])`, :1296-1300):
// out[row] = sigmoid(w2 . relu(W1 msgs[row] + b1) + b2) for every neighbour row, forward only (the reference trains the
// head for `pinnsf_bm` only; a backward through it falls back to torch ops, ops.collision_head).
// packed: pack.hpp (W1 fragments | b1 | b2 | raw w2 row)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_pack_kernel(piml_collision_head H) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < HEAD_PACK) H.packed[e] = head_pack_value(H, e);
}

// One wave per 32-row tile.  Layer 1 (128 -> 64) on the matrix pipe; its fragments stream from the packed image (L2)
// four at a time, one group ahead, so the kernel keeps < 128 VGPRs (the pooling blocks that share its launch want
// occupancy).  Layer 2 (64 -> 1) is a dot product per row: 32 FMAs per lane + one cross-half add, not 32 padded MFMAs.
template <int WAVES = 4>
__device__ __forceinline__ void head_fwd_body(const float* __restrict__ msgs, long long rows,
                                              const float* __restrict__ packed, float* __restrict__ out, long long bx) {
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));   // in an SGPR: per-wave selects stay scalar
    const int j = lane & 31, h = lane >> 5;
    const long long row = (bx * WAVES + wave) * 32 + j;
    if ((bx * WAVES + wave) * 32 >= rows) return;
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
    float4 wb[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wb[0][q] = PK[q * 64 + lane];
    float dot = 0.f;
    f32x16 a1;
#pragma unroll
    for (int g = 0; g < 8; ++g) {                 // g = ob * 4 + bp: the fragment groups in image order
        const int ob = g >> 2, bp = g & 3;
        if (g + 1 < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wb[(g + 1) & 1][q] = PK[((g + 1) * 4 + q) * 64 + lane];
        }
        if (bp == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + dfeat0(ob, q, h));
                a1[4 * q] = bq.x; a1[4 * q + 1] = bq.y; a1[4 * q + 2] = bq.z; a1[4 * q + 3] = bq.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // the next group's loads are issued before this group's MFMAs
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = wb[g & 1][q];
            a1 = dmfma(w.x, X[bp][4 * q + 0], a1);
            a1 = dmfma(w.y, X[bp][4 * q + 1], a1);
            a1 = dmfma(w.z, X[bp][4 * q + 2], a1);
            a1 = dmfma(w.w, X[bp][4 * q + 3], a1);
        }
        if (bp == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w2 = *reinterpret_cast<const float4*>(packed + HP_W2 + dfeat0(ob, q, h));
                dot += w2.x * fmaxf(a1[4 * q], 0.f) + w2.y * fmaxf(a1[4 * q + 1], 0.f) + w2.z * fmaxf(a1[4 * q + 2], 0.f) +
                       w2.w * fmaxf(a1[4 * q + 3], 0.f);
            }
        }
    }
    dot += __shfl_xor(dot, 32, 64);
    if (h == 0 && valid) out[row] = 1.f / (1.f + expf(-(dot + bias[64])));
}

// Layer 1 (128 -> 64) as SPLIT bf16 PRODUCTS (x3.hpp: f32 = hi + mid + lo, six products per k-block, f32 accumulation with
// the small terms apart -- the encoder layers' arithmetic): 96 matrix instructions of 32 cycles per wave instead of a chain
// of 128 f32 ones of 64.  The head's waves share their SIMDs with the decoder chains of the same launch, and those chains
// wait for the matrix pipe: the launch took 3.1 us less without the head's matrix instructions (variant build).  The row's
// 128 values are split once (lane (row, g): features 16 kb + 8 g + t of k-block kb, the order of the packed W1 pieces);
// W1's pieces stream from the packed image (L2), one k-block ahead.  PIML_HEAD_PRODUCTS=f32 keeps the f32 instruction.
// FOLD (PIML_POOL_TRAIN): the rows are h2 rows and W1 / b1 the images with the encoder's last layer folded in (pack.hpp: HP_X3F / HP_BF)
// LDSW (round 5; the launches that pair the head with the decoder tails): the four waves of a workgroup read the SAME 48 KB of
// weight pieces; streamed from L2 by every wave they were three quarters of what the launch pulled through the CUs' L1s at its
// start (in-kernel stamps, tools/dec_stamps_fwd.py: issuing a head wave's first 25 loads took 9 k clocks, its sixteen k-blocks
// 13 k for 3 k of products).  The workgroup now stages k-block groups 0 .. HEAD_LDS_GROUPS - 1 in LDS once (`wlds`, dynamic shared
// memory) and reads the last group from L2 as before: fifteen groups = 45 KB is what fits beside the decoder body's static 33 KB
// with two workgroups per CU.
// (dec_fwd_head_kernel, the message path: its decoder body also holds the pooled tile, 49.5 KB static -- ten groups there.)
constexpr int HEAD_LDS_GROUPS = 15, HEAD_LDS_GROUPS_POOL = 10;
constexpr int head_lds_bytes(int groups) { return groups * 3 * 64 * 16; }
template <int WAVES = 4, bool FOLD = false, int HEAD_LDS_G = 0>
__device__ __forceinline__ void head_fwd_body_x3(const float* __restrict__ msgs, long long rows,
                                                 const float* __restrict__ packed, float* __restrict__ out, long long bx,
                                                 u32x4* wlds = nullptr) {
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const long long row = (bx * WAVES + wave) * 32 + j;
    constexpr bool LDSW = HEAD_LDS_G > 0;
    const bool idle = (bx * WAVES + wave) * 32 >= rows;
    if (!LDSW && idle) return;
    const bool valid = row < rows;
    const float* bias = packed + (FOLD ? HP_BF : HP_B);
    const u32x4* Wg = reinterpret_cast<const u32x4*>(packed + (FOLD ? HP_X3F : HP_X3));       // [ob][kb][piece] 64 apart
    const u32x4* W = Wg + lane;
    constexpr int HEAD_PF = 2;
    u32x4 wf[HEAD_PF][3];
    if (FOLD) DEC_STAMP(0, false);
    // the row's 128 values first (their split overlaps the wait for the weights)
    float4 v[8][2];
    {
        const float* base = msgs + (valid ? row : 0) * DH + 8 * h;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            v[kb][0] = *reinterpret_cast<const float4*>(base + 16 * kb);
            v[kb][1] = *reinterpret_cast<const float4*>(base + 16 * kb + 4);
        }
    }
    // biases and second-layer weights of both output blocks too: left at their uses they were four more exposed round trips
    float4 bq[2][4], w2q[2][4];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bq[ob][q] = *reinterpret_cast<const float4*>(bias + dfeat0(ob, q, h));
            w2q[ob][q] = *reinterpret_cast<const float4*>(packed + HP_W2 + dfeat0(ob, q, h));
        }
    const float b2s = packed[HP_B + 64];
    u32x4 wtail[3];
    if (LDSW) {
        constexpr int N4 = HEAD_LDS_G * 3 * 64, ROUNDS = LDSW ? (N4 + WAVES * 64 - 1) / (WAVES * 64) : 1;
        u32x4 st[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int e = r * WAVES * 64 + (int)threadIdx.x;
            st[r] = Wg[e < N4 ? e : 0];
        }
        if (HEAD_LDS_G == 15) {
#pragma unroll
            for (int p = 0; p < 3; ++p) wtail[p] = W[(15 * 3 + p) * 64];
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int e = r * WAVES * 64 + (int)threadIdx.x;
            if (e < N4) wlds[e] = st[r];
        }
    } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) wf[0][p] = W[p * 64];
    }
    if (FOLD) { DEC_STAMP(1, false); DEC_STAMP(2, true); }
    u32x4 xh[8], xm[8], xl[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        unsigned hi[4], mid[4], lo[4];
        split3(v[kb][0].x, v[kb][0].y, hi[0], mid[0], lo[0]);
        split3(v[kb][0].z, v[kb][0].w, hi[1], mid[1], lo[1]);
        split3(v[kb][1].x, v[kb][1].y, hi[2], mid[2], lo[2]);
        split3(v[kb][1].z, v[kb][1].w, hi[3], mid[3], lo[3]);
        xh[kb] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
        xm[kb] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
        xl[kb] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
    }
    const u32x4* WL = wlds + lane;
    if (LDSW) {
        __syncthreads();
        if (idle) return;
#pragma unroll
        for (int p = 0; p < 3; ++p) wf[0][p] = WL[p * 64];
    }
    float dot = 0.f;
    if (FOLD) DEC_STAMP(3, false);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        f32x16 acc, sm;
#pragma unroll
        for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[4 * q] = bq[ob][q].x; acc[4 * q + 1] = bq[ob][q].y; acc[4 * q + 2] = bq[ob][q].z; acc[4 * q + 3] = bq[ob][q].w;
        }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const int g = ob * 8 + kb;
            if (g + 1 < 16) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    wf[(g + 1) & 1][p] = g + 1 < HEAD_LDS_G ? WL[((g + 1) * 3 + p) * 64]
                                                          : ((HEAD_LDS_G == 15 && g + 1 == 15) ? wtail[p] : W[((g + 1) * 3 + p) * 64]);
            }
            kblock_x3(acc, sm, wf[g & 1][0], wf[g & 1][1], wf[g & 1][2], xh[kb], xm[kb], xl[kb]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w2 = w2q[ob][q];
            dot += w2.x * fmaxf(acc[4 * q] + sm[4 * q], 0.f) + w2.y * fmaxf(acc[4 * q + 1] + sm[4 * q + 1], 0.f) +
                   w2.z * fmaxf(acc[4 * q + 2] + sm[4 * q + 2], 0.f) + w2.w * fmaxf(acc[4 * q + 3] + sm[4 * q + 3], 0.f);
        }
    }
    dot += __shfl_xor(dot, 32, 64);
    if (FOLD) DEC_STAMP(4, false);
    if (h == 0 && valid) out[row] = 1.f / (1.f + expf(-(dot + b2s)));
    if (FOLD) DEC_STAMP(5, true);
}

// PIML_HEAD_PRODUCTS=f32: the collision head's 128 -> 64 layer on the f32 matrix instruction (A/B); default: split bf16 products
static const bool g_head_x3 = !(getenv("PIML_HEAD_PRODUCTS") && getenv("PIML_HEAD_PRODUCTS")[0] == 'f');

template <bool X3>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ msgs, long long rows,
                                                       const float* __restrict__ packed, float* __restrict__ out) {
    if (X3) head_fwd_body_x3(msgs, rows, packed, out, blockIdx.x);
    else head_fwd_body(msgs, rows, packed, out, blockIdx.x);
}


static bool dec_branch_ok(const piml_decoder_branch& b) {
    return b.agents > 0 && b.agents < (1ll << 21) && b.k >= 1 && b.msgs && b.w1 && b.b1 && b.w2 && b.b2 && b.w3 && b.b3 && b.packed;
}

static int dec_dw_workgroups(long long agents) { return (int)((agents + DEC_SLAB - 1) / DEC_SLAB); }

}  // namespace piml

#ifdef PIML_DEC_STAMPS
extern "C" __attribute__((visibility("default"))) int piml_dec_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(piml::g_dec_stamps), sizeof(unsigned long long) * 1024 * 16);
}
#endif

using namespace piml;

PIML_API int piml_decoder_pack_floats(void) { return DEC_PACK; }
PIML_API int piml_decoder_partial_floats(void) { return DEC_PART; }
PIML_API int piml_decoder_workgroups(long long agents) { return dec_dw_workgroups(agents); }

static int head_check(const piml_collision_head* h);

static int dec_fill(DecArgs& A, const piml_decoder_branch* br, int nbr) {
    if (!br || nbr < 1 || nbr > 2) return hipErrorInvalidValue;
    A = DecArgs{};
    A.nbr = nbr;
    for (int i = 0; i < nbr; ++i) {
        if (!dec_branch_ok(br[i]) || br[i].agents != br[0].agents || !br[i].pooled) return hipErrorInvalidValue;
        A.br[i] = br[i];
    }
    if (nbr == 1) A.br[1] = br[0];
    return hipSuccess;
}

static int dec_fill_bwd(DecArgs& A, const piml_decoder_branch* br, int nbr, const float* g_pred) {
    if (!g_pred) return hipErrorInvalidValue;
    if (int e = dec_fill(A, br, nbr)) return e;
    for (int i = 0; i < nbr; ++i) {
        const piml_decoder_branch& b = br[i];
        if (!b.h1 || !b.d2 || !b.g_pre2 || !b.g_pre1 || !b.g_pooled || !b.partials || !b.grads) return hipErrorInvalidValue;
    }
    A.g_pred = g_pred;
    return hipSuccess;
}

int piml::dec_stage_pack(const piml_decoder_branch* br, int nbr, hipStream_t s) {
    if (!br || nbr < 1 || nbr > 2) return hipErrorInvalidValue;
    DecArgs A = {};
    A.nbr = nbr;
    for (int i = 0; i < nbr; ++i) {          // the pack reads the weights only (no agents yet)
        const piml_decoder_branch& b = br[i];
        if (!b.w1 || !b.b1 || !b.w2 || !b.b2 || !b.w3 || !b.b3 || !b.packed) return hipErrorInvalidValue;
        A.br[i] = b;
    }
    if (nbr == 1) A.br[1] = br[0];
    hipLaunchKernelGGL(dec_pack_kernel, dim3((DEC_PACK + 255) / 256, nbr), dim3(256), 0, s, A);
    return hipGetLastError();
}

int piml::dec_stage_pool(const piml_decoder_branch* br, int nbr, hipStream_t s) {
    DecArgs A;
    if (int e = dec_fill(A, br, nbr)) return e;
    hipLaunchKernelGGL(dec_pool_kernel, dim3((unsigned)((br[0].agents * (DH / 4) + 255) / 256), nbr), dim3(256), 0, s, A);
    return hipGetLastError();
}

// The decoder tails (incl. their neighbour-axis sums) and the collision head in ONE launch: both consume the encoders'
// messages and nothing of each other.  Blocks [0, dec_blocks) are (32-agent tile, branch) units, the rest head blocks of
// 4 x 32 message rows.  `acc` must be zero on entry when there are two branches (enc_stage_fwd clears it on the way).
// (two waves per SIMD: decoder and head workgroups of one launch are all resident at once -- 448 x 4 waves at the 4096-agent
// scene; left alone the compiler gives the split-product head 64 accumulation registers on top of the decoder body's 224)
template <bool X3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void dec_fwd_head_kernel(DecArgs A, piml_collision_head Hd, int dec_blocks) {
    const int bx = blockIdx.x;
    if (bx < dec_blocks) {
        if (A.nbr > 1) dec_fwd_body<true, true>(A, bx >> 1, bx & 1);
        else dec_fwd_body<true, true>(A, bx, 0);
    } else {
        extern __shared__ __align__(16) float head_lds[];
        if (X3) head_fwd_body_x3<4, false, HEAD_LDS_GROUPS_POOL>(Hd.msgs, Hd.rows, Hd.packed, Hd.out, (long long)bx - dec_blocks, reinterpret_cast<u32x4*>(head_lds));
        else head_fwd_body<4>(Hd.msgs, Hd.rows, Hd.packed, Hd.out, (long long)bx - dec_blocks);
    }
}

// PIML_POOL_TRAIN: the decoder tails on the agents' sums of h2 (no neighbour-axis sum here: dec_fwd_ph2_kernel's body with the
// folded first layer, A.pool_h2 = 2) and the collision head on the h2 rows with the folded W1, in one launch as above
// FOLD = false (PIML_POOL_MSGS): the sums are sums of the messages themselves -- plain first layers, the head on the message rows
template <bool FOLD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void dec_fwd_head_sum_kernel(DecArgs A, piml_collision_head Hd, int dec_blocks) {
    const int bx = blockIdx.x;
    if (bx < dec_blocks) {
        if (A.nbr > 1) dec_fwd_body<false, true>(A, bx >> 1, bx & 1);
        else dec_fwd_body<false, true>(A, bx, 0);
    } else {
        extern __shared__ __align__(16) float head_lds[];
        head_fwd_body_x3<4, FOLD, HEAD_LDS_GROUPS>(Hd.msgs, Hd.rows, Hd.packed, Hd.out, (long long)bx - dec_blocks, reinterpret_cast<u32x4*>(head_lds));
    }
}

int piml::dec_stage_fwd_sum(const piml_decoder_branch* br, int nbr, const piml_collision_head* h, const float* self_features,
                            float tau, float* acc, hipStream_t s, bool fold) {
    DecArgs A;
    if (!acc) return hipErrorInvalidValue;
    if (int e = dec_fill(A, br, nbr)) return e;
    for (int i = 0; i < nbr; ++i)
        if (!br[i].msgs || (fold && (!br[i].fold_w3 || !br[i].fold_b3))) return hipErrorInvalidValue;
    piml_collision_head Hd = {};
    int head_blocks = 0;
    if (h && h->rows > 0) {
        if (int e = head_check(h)) return e;
        if (fold && (!h->fold_w3 || !h->fold_b3)) return hipErrorInvalidValue;
        if (!g_head_x3) return hipErrorInvalidValue;          // (the paired launch has the split-product head only)
        Hd = *h;
        head_blocks = (int)(((h->rows + 31) / 32 + 3) / 4);
    }
    A.self_features = self_features;
    A.tau = tau;
    A.acc = acc;
    A.pool_h2 = fold ? 2 : 3;
    const int tiles = (int)((br[0].agents + 31) / 32) * nbr;
    const dim3 g((unsigned)(tiles + head_blocks));
    const int lds = head_blocks ? head_lds_bytes(HEAD_LDS_GROUPS) : 0;
    if (fold) hipLaunchKernelGGL(dec_fwd_head_sum_kernel<true>, g, dim3(256), lds, s, A, Hd, tiles);
    else hipLaunchKernelGGL(dec_fwd_head_sum_kernel<false>, g, dim3(256), lds, s, A, Hd, tiles);
    return hipGetLastError();
}

int piml::dec_stage_fwd_fused(const piml_decoder_branch* br, int nbr, const piml_collision_head* h, const float* self_features,
                              float tau, float* acc, hipStream_t s) {
    DecArgs A;
    if (!acc) return hipErrorInvalidValue;
    if (int e = dec_fill(A, br, nbr)) return e;
    piml_collision_head Hd = {};
    int head_blocks = 0;
    if (h && h->rows > 0) {
        if (int e = head_check(h)) return e;
        Hd = *h;
        head_blocks = (int)(((h->rows + 31) / 32 + 3) / 4);
    }
    A.self_features = self_features;
    A.tau = tau;
    A.acc = acc;
    const int tiles = (int)((br[0].agents + 31) / 32) * nbr;
    if (g_head_x3) hipLaunchKernelGGL(dec_fwd_head_kernel<true>, dim3((unsigned)(tiles + head_blocks)), dim3(256), head_blocks ? head_lds_bytes(HEAD_LDS_GROUPS_POOL) : 0, s, A, Hd, tiles);
    else hipLaunchKernelGGL(dec_fwd_head_kernel<false>, dim3((unsigned)(tiles + head_blocks)), dim3(256), 0, s, A, Hd, tiles);
    return hipGetLastError();
}

// PIML_POOL_H2: the decoder tails on the sums the inference forward left (no neighbour-axis sum, no head); (tile, branch)
// workgroups like dec_fwd_head_kernel, `acc` cleared by the encoder launch when there are two branches
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void dec_fwd_ph2_kernel(DecArgs A) {
    const int bx = blockIdx.x;
    if (A.nbr > 1) dec_fwd_body<false, true>(A, bx >> 1, bx & 1);
    else dec_fwd_body<false, true>(A, bx, 0);
}

int piml::dec_stage_fwd_ph2(const piml_decoder_branch* br, int nbr, const float* self_features, float tau, float* acc,
                            hipStream_t s) {
    DecArgs A;
    if (!acc) return hipErrorInvalidValue;
    if (int e = dec_fill(A, br, nbr)) return e;
    for (int i = 0; i < nbr; ++i)
        if (!br[i].msgs) return hipErrorInvalidValue;
    A.self_features = self_features;
    A.tau = tau;
    A.acc = acc;
    A.pool_h2 = 1;
    const int tiles = (int)((br[0].agents + 31) / 32) * nbr;
    hipLaunchKernelGGL(dec_fwd_ph2_kernel, dim3((unsigned)tiles), dim3(256), 0, s, A);
    return hipGetLastError();
}

int piml::dec_stage_fwd(const piml_decoder_branch* br, int nbr, const float* self_features, float tau, float* acc,
                        hipStream_t s) {
    DecArgs A;
    if (!acc) return hipErrorInvalidValue;
    if (int e = dec_fill(A, br, nbr)) return e;
    A.self_features = self_features;
    A.tau = tau;
    A.acc = acc;
    const unsigned tiles = (unsigned)((br[0].agents + 31) / 32);
    hipLaunchKernelGGL(dec_fwd_kernel, dim3(tiles), dim3(512), 0, s, A);
    return hipGetLastError();
}

int piml::dec_stage_bwd_dx(const piml_decoder_branch* br, int nbr, const float* g_pred, const float* self_features,
                           float tau, float* g_self, hipStream_t s) {
    DecArgs A;
    if (int e = dec_fill_bwd(A, br, nbr, g_pred)) return e;
    A.self_features = self_features;
    A.tau = tau;
    A.g_self = g_self;
    const unsigned tiles = (unsigned)((br[0].agents + 31) / 32);
    hipLaunchKernelGGL(dec_bwd_dx_kernel, dim3(tiles), dim3(512), 0, s, A);
    return hipGetLastError();
}

int piml::dec_stage_bwd_fused(const piml_decoder_branch* br, int nbr, const float* g_pred, const float* self_features,
                              float tau, float* g_self, hipStream_t s, bool sums) {
    DecArgs A;
    if (int e = dec_fill_bwd(A, br, nbr, g_pred)) return e;
    A.self_features = self_features;
    A.tau = tau;
    A.g_self = g_self;
    if (sums) {          // PIML_POOL_TRAIN: W1'^T fragments (DP_T1F) in the dX chain; `pooled` = the completed sums of h2
        for (int i = 0; i < nbr; ++i)
            if (!br[i].fold_w3) return hipErrorInvalidValue;
        A.pool_h2 = 2;
    }
    static_assert(DEC_SLAB == 32, "the dW slab of a workgroup is its dX tile");
    const unsigned tiles = (unsigned)((br[0].agents + 31) / 32);
    static const bool whole = getenv("PIML_DEC_BWD_SPLIT") && atoi(getenv("PIML_DEC_BWD_SPLIT")) == 0;     // A/B: the 8-wave form
    if (whole && !sums) hipLaunchKernelGGL(dec_bwd_kernel, dim3(tiles), dim3(512), 0, s, A);
    else hipLaunchKernelGGL(dec_bwd_split_kernel, dim3(tiles * (unsigned)nbr), dim3(256), 0, s, A);
    return hipGetLastError();
}

int piml::dec_stage_bwd_dw(const piml_decoder_branch* br, int nbr, const float* g_pred, bool reduce, hipStream_t s) {
    DecArgs A;
    if (int e = dec_fill_bwd(A, br, nbr, g_pred)) return e;
    const int per = dec_dw_workgroups(br[0].agents);
    A.wg_split = per;
    hipLaunchKernelGGL(dec_bwd_dw_kernel, dim3(per * nbr), dim3(512), 0, s, A);
    if (reduce)
        hipLaunchKernelGGL(dec_reduce_kernel, dim3((DEC_PART / 4 + 15) / 16, nbr), dim3(256), 0, s, A, per, DEC_PART / 4);
    return hipGetLastError();
}

static int head_check(const piml_collision_head* h) {
    if (!h || h->rows < 0) return hipErrorInvalidValue;
    if (h->rows > 0 && (!h->msgs || !h->w1 || !h->b1 || !h->w2 || !h->b2 || !h->packed || !h->out)) return hipErrorInvalidValue;
    return hipSuccess;
}

int piml::head_stage_pack(const piml_collision_head* h, hipStream_t s) {
    if (!h || !h->w1 || !h->b1 || !h->w2 || !h->b2 || !h->packed) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_pack_kernel, dim3((HEAD_PACK + 255) / 256), dim3(256), 0, s, *h);
    return hipGetLastError();
}

int piml::head_stage_fwd(const piml_collision_head* h, hipStream_t s) {
    if (int e = head_check(h)) return e;
    if (h->rows == 0) return hipSuccess;
    const long long tiles = (h->rows + 31) / 32;
    if (g_head_x3) hipLaunchKernelGGL(head_fwd_kernel<true>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, h->msgs, h->rows, h->packed, h->out);
    else hipLaunchKernelGGL(head_fwd_kernel<false>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, h->msgs, h->rows, h->packed, h->out);
    return hipGetLastError();
}

PIML_API int piml_decoder_fwd(const piml_decoder_branch* br, int nbr, const float* self_features, float tau,
                              float* acc, void* stream) {
    hipStream_t s = as_stream(stream);
    if (int e = dec_stage_pack(br, nbr, s)) return e;
    if (int e = dec_stage_pool(br, nbr, s)) return e;
    return dec_stage_fwd(br, nbr, self_features, tau, acc, s);
}

PIML_API int piml_decoder_bwd(const piml_decoder_branch* br, int nbr, const float* g_pred, const float* self_features,
                              float tau, float* g_self, void* stream) {
    hipStream_t s = as_stream(stream);
    if (int e = dec_stage_bwd_dx(br, nbr, g_pred, self_features, tau, g_self, s)) return e;
    return dec_stage_bwd_dw(br, nbr, g_pred, true, s);
}

// slab of the row-wise dW kernel: a multiple of DEC_SLAB rows, at least 256 (the workgroup holds ~100 KB of LDS, one per
// CU: with the reference's 24 576 + 40 960 rows the two branches make 96 + 160 = 256 workgroups, each pipelining 8 chunks),
// and at most 256 slots per branch
// (few rows -- the training rollout on real clips, 3 000 - 5 000 rows per branch: slabs of 64, or a dozen workgroups walk
// eight chunks each while 240 CUs idle: pinnsf_bm fine-tuning step 1.45 -> 1.38 ms)
static long long rowdec_slab(long long rows) {
    long long slab = (rows + 255) / 256;
    slab = (slab + DEC_SLAB - 1) / DEC_SLAB * DEC_SLAB;
    const long long least = rows < 16384 ? 64 : 256;
    return slab < least ? least : slab;
}

// 32-row tiles (both branches) above which the row decoder runs one wave per tile with LDS-staged fragments
constexpr int kRowdecBigTiles = 1024;

// workgroups of branch 0 in a launch of `total`, proportional to the rows (each branch at least one)
static int rowdec_wg_split(const piml_decoder_branch* br, int nbr, int total) {
    if (nbr < 2) return total;
    int w = (int)(total * (double)br[0].agents / ((double)br[0].agents + (double)br[1].agents) + 0.5);
    return w < 1 ? 1 : (w > total - 1 ? total - 1 : w);
}

PIML_API int piml_rowdecoder_slots(long long rows) {
    if (rows <= 0) return 0;
    const long long slab = rowdec_slab(rows);
    return (int)((rows + slab - 1) / slab);
}

static int rowdec_fill(DecArgs& A, const piml_decoder_branch* br, int nbr, bool bwd) {
    if (!br || nbr < 1 || nbr > 2) return hipErrorInvalidValue;
    A = DecArgs{};
    A.nbr = nbr;
    for (int i = 0; i < nbr; ++i) {
        const piml_decoder_branch& b = br[i];
        if (!dec_branch_ok(b) || !b.h1 || !b.d2) return hipErrorInvalidValue;
        if (!bwd && !b.pred) return hipErrorInvalidValue;
        if (bwd && (!b.g_pred_rows || !b.g_pre2 || !b.g_pre1 || !b.g_pooled || !b.partials || !b.grads)) return hipErrorInvalidValue;
        A.br[i] = b;
    }
    if (nbr == 1) A.br[1] = br[0];
    return hipSuccess;
}

// products of the row decoder's many-rows kernels: 1 = split bf16 products (rowdec_*_x3_kernel), 0 = the f32 matrix-core
// instruction (PIML_ROWDEC_PRODUCTS=f32, piml_rowdecoder_products)
static int g_rowdec_x3 = !(getenv("PIML_ROWDEC_PRODUCTS") && getenv("PIML_ROWDEC_PRODUCTS")[0] == 'f');

PIML_API int piml_rowdecoder_products(int x3) {
    const int old = g_rowdec_x3;
    if (x3 >= 0) g_rowdec_x3 = x3 ? 1 : 0;
    return old;
}

static int rowdec_x3_attributes() {
    static bool done = false;
    if (done) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowdec_fwd_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RFX_LDS_BYTES))
        return e;
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowdec_bwd_dx_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RDXX_LDS_BYTES))
        return e;
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowdec_bwd_dw_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RWX_LDS_BYTES))
        return e;
    done = true;
    return hipSuccess;
}

static int rowdecoder_fwd(const piml_decoder_branch* br, int nbr, bool pack, void* stream);
PIML_API int piml_rowdecoder_fwd(const piml_decoder_branch* br, int nbr, void* stream) { return rowdecoder_fwd(br, nbr, true, stream); }
// `packed` already holds the operand images of these weights (piml_pinnsf_pack / an earlier piml_rowdecoder_fwd)
PIML_API int piml_rowdecoder_fwd_packed(const piml_decoder_branch* br, int nbr, void* stream) {
    if (int e = pending_pack_flush(as_stream(stream))) return e;          // a deferred pack (PIML_DEFER_PACK) nobody took: now
    return rowdecoder_fwd(br, nbr, false, stream);
}

static int rowdecoder_fwd(const piml_decoder_branch* br, int nbr, bool pack, void* stream) {
    hipStream_t s = as_stream(stream);
    DecArgs A;
    if (int e = rowdec_fill(A, br, nbr, false)) return e;
    if (pack)
        if (int e = dec_stage_pack(br, nbr, s)) return e;
    const int tiles0 = (int)((br[0].agents + 31) / 32), tiles1 = nbr > 1 ? (int)((br[1].agents + 31) / 32) : 0;
    if (tiles0 + tiles1 > kRowdecBigTiles) {       // many rows: one wave per tile, fragments in LDS (rowdec_fwd_big_kernel)
        const int split = rowdec_wg_split(br, nbr, 256);
        if (g_rowdec_x3) {
            if (int e = rowdec_x3_attributes()) return e;
            hipLaunchKernelGGL(rowdec_fwd_x3_kernel, dim3(256), dim3(512), RFX_LDS_BYTES, s, A, split);
        } else {
            hipLaunchKernelGGL(rowdec_fwd_big_kernel, dim3(256), dim3(512), 0, s, A, split);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(rowdec_fwd_kernel, dim3((unsigned)(tiles0 + tiles1)), dim3(256), 0, s, A, tiles0);
    return hipGetLastError();
}

PIML_API int piml_rowdecoder_bwd(const piml_decoder_branch* br, int nbr, void* stream) { return piml_rowdecoder_bwd_acc(br, nbr, 0, stream); }

PIML_API int piml_rowdecoder_bwd_acc(const piml_decoder_branch* br, int nbr, int accumulate, void* stream) {
    hipStream_t s = as_stream(stream);
    DecArgs A;
    if (int e = rowdec_fill(A, br, nbr, true)) return e;
    const int tiles0 = (int)((br[0].agents + 31) / 32), tiles1 = nbr > 1 ? (int)((br[1].agents + 31) / 32) : 0;
    if (tiles0 + tiles1 > kRowdecBigTiles) {
        if (g_rowdec_x3) {
            if (int e = rowdec_x3_attributes()) return e;
            hipLaunchKernelGGL(rowdec_bwd_dx_x3_kernel, dim3(256), dim3(512), RDXX_LDS_BYTES, s, A, rowdec_wg_split(br, nbr, 256));
        } else {
            hipLaunchKernelGGL(rowdec_bwd_dx_big_kernel, dim3(256), dim3(512), 0, s, A, rowdec_wg_split(br, nbr, 256));
        }
    } else
        hipLaunchKernelGGL(rowdec_bwd_dx_kernel, dim3((unsigned)(tiles0 + tiles1)), dim3(256), 0, s, A, tiles0);
    const int slots0 = piml_rowdecoder_slots(br[0].agents), slots1 = nbr > 1 ? piml_rowdecoder_slots(br[1].agents) : 0;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowdec_bwd_dw_lds_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RD_BUF * 4))
            return e;
        attr_set = true;
    }
    if (g_rowdec_x3 && tiles0 + tiles1 > kRowdecBigTiles) {
        if (int e = rowdec_x3_attributes()) return e;
        hipLaunchKernelGGL(rowdec_bwd_dw_x3_kernel, dim3((unsigned)(slots0 + slots1)), dim3(512), RWX_LDS_BYTES, s, A, slots0,
                           rowdec_slab(br[0].agents), nbr > 1 ? rowdec_slab(br[1].agents) : (long long)DEC_SLAB);
    } else {
        hipLaunchKernelGGL(rowdec_bwd_dw_lds_kernel, dim3((unsigned)(slots0 + slots1)), dim3(512), 2 * RD_BUF * 4, s, A, slots0,
                           rowdec_slab(br[0].agents), nbr > 1 ? rowdec_slab(br[1].agents) : (long long)DEC_SLAB);
    }
    // `accumulate`: 0 / 1, or flags -- PIML_ACCUMULATE and / or PIML_DEFER_SLOT_SUMS (the header)
    const bool acc = (accumulate & 1) || (accumulate & PIML_ACCUMULATE);
    if (accumulate & PIML_DEFER_SLOT_SUMS) {          // the slot sums ride in the relfeat backward's launch (network.hip)
        ReduceAll R = {};
        R.accumulate = acc ? 1 : 0;
        R.set[0] = ReduceSet{br[0].partials, br[0].grads, slots0, DEC_PART / 4, 0x7fffffff, 0, 0};
        if (nbr > 1) R.set[1] = ReduceSet{br[1].partials, br[1].grads, slots1, DEC_PART / 4, 0x7fffffff, 0, 0};
        R.nsets = nbr;
        R.gx = (DEC_PART / 4 + 15) / 16;
        if (hipError_t e = hipGetLastError()) return e;
        return pending_slot_sums_leave(R, s);
    }
    hipLaunchKernelGGL(rowdec_reduce_kernel, dim3((DEC_PART / 4 + 15) / 16, nbr), dim3(256), 0, s, A, slots0, slots1,
                       DEC_PART / 4, acc ? 1 : 0);
    return hipGetLastError();
}

PIML_API int piml_collision_head_pack_floats(void) { return HEAD_PACK; }

PIML_API int piml_collision_head_fwd(const float* msgs, long long rows, const float* w1, const float* b1, const float* w2,
                                     const float* b2, float* packed, float* out, void* stream) {
    const piml_collision_head h = {msgs, rows, w1, b1, w2, b2, packed, out, nullptr, nullptr, 0.f};
    if (int e = head_stage_pack(&h, as_stream(stream))) return e;
    return head_stage_fwd(&h, as_stream(stream));
}
