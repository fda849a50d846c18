// The corrector of `pinnsf_res` on hand-written kernels, forward AND backward (round 4; until then library GEMMs + glue,
// ~0.5 ms of a 0.86 ms step).  Reference: src/models/model.py:1016-1020 (the three modules), :1050-1052 (their use),
// :950-970 (attn_pooling), :82-119 (ResDNN, whose forward keeps only its LAST block -- an empty MLP plus the skip
// connection -- so that with >= 2 "layers" it computes dropout(2 x)):
//     r      = keep * scale * enc                                  enc (agents * k, 128): the pedestrian encoder's raw output
//     hid    = relu(Wa r + ba),  s = wb . hid + bb                 attn_pooling.get_weights = MLP(128, [128, 1])
//     attn   = softmax_k(exp(s)),  pooled = sum_k attn r           per agent, over its k neighbour rows
//     out    = Wd relu(Wc pooled + bc) + bd                        corrector[2] = MLP(128, [64, 2])
//
// Four kernels besides the slot sum, all exact f32 (v_mfma_f32_32x32x2_f32 is an fmaf chain; the layouts are head64.hip's:
// features on the instruction's M axis, the 32 rows of a tile on its N axis, lane (j, h) holds features 8 q + 4 h + u of
// row j in accumulator registers 4 q + u):
//   corr_rows_fwd    one wave per 32-row tile: hid (256 matrix instructions), s
//   corr_agents_fwd  one wave per agent: the two exponentials, the weighted sum over k, the 128 -> 64 -> 2 tail
//                    (Wc transposed in LDS once per workgroup)
//   corr_agents_bwd  one wave per agent: tail backward (dWc as 128 accumulators per lane over the workgroup's agents),
//                    g_pooled, the softmax / exp backward -> g_s
//   corr_rows_bwd    a workgroup of four waves = four tiles: g_hid, g_r = Wa^T g_hid + attn g_pooled, d/d(enc); the tiles'
//                    g_hid and r meet in LDS and wave w accumulates rows 32 w .. 32 w + 31 of dWa = g_hid^T r over all
//                    128 rows (256 matrix instructions), dba / dwb / dbb as column sums
// Weight gradients: one slot per workgroup, summed in a fixed order (no atomics: bit-reproducible).
#include "common.hpp"
#include "pack.hpp"
#include "reduce.hpp"
#include "trace.hpp"
#include "../../include/piml_hip.h"

namespace piml {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CH = 128, CD = 64;
constexpr int CORRA_PART = CH * CH + CH + CH + 4;             // dWa | dba | dwb | dbb + pad
constexpr int CORRB_PART = CD * CH + CD + 2 * CD + 4;         // dWc | dbc | dWd | dbd + pad
constexpr int CORR_TSTRIDE = CH + 4;                          // row stride of the LDS tiles (floats): the two lane halves on different banks
constexpr int CORR_ROWS_BWD_LDS = 2 * 128 * CORR_TSTRIDE * 4 + 64;

__device__ __forceinline__ f32x16 cmfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int cfeat0(int blk, int q, int h) { return 32 * blk + 8 * q + 4 * h; }

// r = keep * scale * enc for row `row` in the accumulator layout (4 blocks of 16 registers); rows past the end give zeros
__device__ __forceinline__ void corr_load_r(const piml_corrector& A, long long row, bool valid, int h, f32x16 (&X)[4]) {
    const float* xr = A.enc + (valid ? row : 0) * CH;
    uint4 kb = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (A.keep_bits && valid) kb = *reinterpret_cast<const uint4*>(A.keep_bits + row * 4);
    const unsigned kw[4] = {kb.x, kb.y, kb.z, kb.w};
    const float sc = valid ? A.scale : 0.f;
#pragma unroll
    for (int bp = 0; bp < 4; ++bp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(xr + cfeat0(bp, q, h));
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) X[bp][4 * q + u] = (kw[bp] >> (8 * q + 4 * h + u)) & 1u ? sc * e[u] : 0.f;
        }
}

__global__ __launch_bounds__(256) void corr_rows_fwd_kernel(piml_corrector A) {
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const long long rows = A.agents * A.k;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile * 32 >= rows) return;
    const long long row = tile * 32 + j;
    const bool valid = row < rows;
    f32x16 X[4];
    corr_load_r(A, row, valid, h, X);
    float dot = 0.f;
#pragma unroll 1
    for (int ob = 0; ob < 4; ++ob) {
        f32x16 a;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = *reinterpret_cast<const float4*>(A.ba + cfeat0(ob, q, h));
            a[4 * q] = bq.x; a[4 * q + 1] = bq.y; a[4 * q + 2] = bq.z; a[4 * q + 3] = bq.w;
        }
        const float* wrow = A.wa + (size_t)(32 * ob + j) * CH;             // lane (i = j, h): Wa[32 ob + i][...]
#pragma unroll
        for (int bp = 0; bp < 4; ++bp) {
            float4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const float4*>(wrow + cfeat0(bp, q, h));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a = cmfma(w[q].x, X[bp][4 * q + 0], a);
                a = cmfma(w[q].y, X[bp][4 * q + 1], a);
                a = cmfma(w[q].z, X[bp][4 * q + 2], a);
                a = cmfma(w[q].w, X[bp][4 * q + 3], a);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
        if (A.hid && valid) {
            float* o = A.hid + row * CH;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + cfeat0(ob, q, h)) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w2 = *reinterpret_cast<const float4*>(A.wb + cfeat0(ob, q, h));
            dot += w2.x * a[4 * q] + w2.y * a[4 * q + 1] + w2.z * a[4 * q + 2] + w2.w * a[4 * q + 3];
        }
    }
    dot += __shfl_xor(dot, 32, 64);
    if (h == 0 && valid) A.score[row] = dot + A.bb[0];
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

// the two features 2 lane, 2 lane + 1 of r for one row
__device__ __forceinline__ float2 corr_r2(const piml_corrector& A, long long row, int lane) {
    const float2 v = *reinterpret_cast<const float2*>(A.enc + row * CH + 2 * lane);
    unsigned w = ~0u;
    if (A.keep_bits) w = A.keep_bits[row * 4 + (lane >> 4)];
    const int b = (2 * lane) & 31;
    return make_float2((w >> b) & 1u ? A.scale * v.x : 0.f, (w >> (b + 1)) & 1u ? A.scale * v.y : 0.f);
}

__global__ __launch_bounds__(256) void corr_agents_fwd_kernel(piml_corrector A) {
    __shared__ float wct[CH * CD];          // Wc transposed: [input 128][output 64]
    __shared__ float pl[4][CH];
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    for (int e = threadIdx.x; e < CH * CD; e += 256) {
        const int c = e / CH, i = e - c * CH;           // coalesced read of Wc (64, 128)
        wct[i * CD + c] = A.wc[e];
    }
    __syncthreads();
    const int k = A.k;
    const float bc = A.bc[lane], wd0 = A.wd[lane], wd1 = A.wd[CD + lane], bd0 = A.bd[0], bd1 = A.bd[1];
    for (long long agent = (long long)blockIdx.x * 4 + wave; agent < A.agents; agent += (long long)gridDim.x * 4) {
        const long long row0 = agent * k;
        const float s = lane < k ? A.score[row0 + lane] : 0.f;
        const float e = lane < k ? expf(s) : -INFINITY;                     // attn = softmax(exp(s)) over the k rows (model.py:966-967)
        const float m = wave_max(e);
        const float p = lane < k ? expf(e - m) : 0.f;
        const float a = p / wave_sum(p);
        if (lane < k) A.attn[row0 + lane] = a;
        float2 acc = make_float2(0.f, 0.f);
        for (int i = 0; i < k; ++i) {
            const float ai = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), i));
            const float2 r = corr_r2(A, row0 + i, lane);
            acc.x += ai * r.x; acc.y += ai * r.y;
        }
        *reinterpret_cast<float2*>(A.pooled + agent * CH + 2 * lane) = acc;
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<float2*>(&pl[wave][2 * lane]) = acc;
        __builtin_amdgcn_wave_barrier();
        float hsum = bc;
#pragma unroll 8
        for (int i = 0; i < CH; ++i) hsum += wct[i * CD + lane] * pl[wave][i];
        hsum = fmaxf(hsum, 0.f);
        A.chid[agent * CD + lane] = hsum;
        const float o0 = wave_sum(hsum * wd0), o1 = wave_sum(hsum * wd1);
        if (lane == 0) *reinterpret_cast<float2*>(A.out + agent * 2) = make_float2(o0 + bd0, o1 + bd1);
    }
}

__global__ __launch_bounds__(256) void corr_agents_bwd_kernel(piml_corrector A) {
    __shared__ __attribute__((aligned(16))) float wcl[CD * CH];       // Wc row-major [output 64][input 128]; later the workgroup's slot
    __shared__ float gl[4][CD];
    __shared__ float small[4][4 * CD + 4];                             // per wave: dbc | dWd (2 x 64) | pad, dbd
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    for (int e = threadIdx.x; e < CD * CH; e += 256) wcl[e] = A.wc[e];
    __syncthreads();
    const int k = A.k;
    const float wd0 = A.wd[lane], wd1 = A.wd[CD + lane];
    float dwc[CD][2];
#pragma unroll
    for (int c = 0; c < CD; ++c) { dwc[c][0] = 0.f; dwc[c][1] = 0.f; }
    float dbc = 0.f, dwd0 = 0.f, dwd1 = 0.f, dbd0 = 0.f, dbd1 = 0.f;
    for (long long agent = (long long)blockIdx.x * 4 + wave; agent < A.agents; agent += (long long)gridDim.x * 4) {
        const long long row0 = agent * k;
        const float2 g = *reinterpret_cast<const float2*>(A.g_out + agent * 2);
        const float ch = A.chid[agent * CD + lane];
        const float gch = ch > 0.f ? wd0 * g.x + wd1 * g.y : 0.f;
        dwd0 += g.x * ch; dwd1 += g.y * ch; dbc += gch; dbd0 += g.x; dbd1 += g.y;
        __builtin_amdgcn_wave_barrier();
        gl[wave][lane] = gch;
        __builtin_amdgcn_wave_barrier();
        const float2 pool = *reinterpret_cast<const float2*>(A.pooled + agent * CH + 2 * lane);
        float2 gp = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < CD; ++c) {
            const float gc = gl[wave][c];
            const float2 w = *reinterpret_cast<const float2*>(&wcl[c * CH + 2 * lane]);
            gp.x += w.x * gc; gp.y += w.y * gc;
            dwc[c][0] += gc * pool.x; dwc[c][1] += gc * pool.y;
        }
        *reinterpret_cast<float2*>(A.g_pooled + agent * CH + 2 * lane) = gp;
        // pooled = sum_i a_i r_i, a = softmax(e), e = exp(s):  g_a_i = g_pooled . r_i,  g_e_i = a_i (g_a_i - sum_j a_j g_a_j),
        // g_s_i = g_e_i e_i
        float ga_mine = 0.f, t = 0.f;
        for (int i = 0; i < k; ++i) {
            const float2 r = corr_r2(A, row0 + i, lane);
            const float ga = wave_sum(gp.x * r.x + gp.y * r.y);
            t += A.attn[row0 + i] * ga;
            if (lane == i) ga_mine = ga;
        }
        if (lane < k) {
            const float a = A.attn[row0 + lane];
            A.g_score[row0 + lane] = a * (ga_mine - t) * expf(A.score[row0 + lane]);
        }
    }
    // ---- the four waves add up in LDS, wave 0 first (fixed order): one slot per workgroup ----
    __syncthreads();                                                   // Wc in LDS is dead
    float* P = wcl;                                                     // dWc [64][128]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < CD; ++c) {
                float2* p = reinterpret_cast<float2*>(&P[c * CH + 2 * lane]);
                const float2 o = w == 0 ? make_float2(0.f, 0.f) : *p;
                *p = make_float2(o.x + dwc[c][0], o.y + dwc[c][1]);
            }
            small[w][lane] = dbc; small[w][CD + lane] = dwd0; small[w][2 * CD + lane] = dwd1;
            if (lane == 0) { small[w][3 * CD] = dbd0; small[w][3 * CD + 1] = dbd1; }
        }
        __syncthreads();
    }
    float* out = A.partials_b + (size_t)blockIdx.x * CORRB_PART;
    for (int e = threadIdx.x; e < CD * CH; e += 256) out[e] = P[e];
    for (int e = threadIdx.x; e < 3 * CD + 4; e += 256) {
        float v = 0.f;
        if (e < 3 * CD + 2)
            for (int w = 0; w < 4; ++w) v += small[w][e];
        out[CD * CH + e] = v;
    }
}

__global__ __launch_bounds__(256) void corr_rows_bwd_kernel(piml_corrector A) {
    extern __shared__ __attribute__((aligned(16))) float cl[];
    float* const G = cl;                                   // [128 rows][CORR_TSTRIDE]: g_hid, later g_s * hid
    float* const R = cl + 128 * CORR_TSTRIDE;              // [128 rows][CORR_TSTRIDE]: r
    float* const sm = R + 128 * CORR_TSTRIDE;              // 4 floats: the waves' sums of g_s
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const long long rows = A.agents * A.k;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const long long row = tile * 32 + j;
    const bool valid = row < rows;
    const long long rr = valid ? row : 0;
    const float gs = valid ? A.g_score[row] : 0.f;
    // ---- phase 1: this wave's tile ----
    f32x16 hid[4], gh[4];
    {
        f32x16 X[4];
        corr_load_r(A, row, valid, h, X);
#pragma unroll
        for (int bp = 0; bp < 4; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 hv = *reinterpret_cast<const float4*>(A.hid + rr * CH + cfeat0(bp, q, h));
                const float4 w2 = *reinterpret_cast<const float4*>(A.wb + cfeat0(bp, q, h));
                const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, ww[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    hid[bp][4 * q + u] = valid ? hh[u] : 0.f;
                    gh[bp][4 * q + u] = (valid && hh[u] > 0.f) ? gs * ww[u] : 0.f;
                }
                float* gt = G + (32 * wave + j) * CORR_TSTRIDE + cfeat0(bp, q, h);
                float* rt = R + (32 * wave + j) * CORR_TSTRIDE + cfeat0(bp, q, h);
                *reinterpret_cast<float4*>(gt) = make_float4(gh[bp][4 * q], gh[bp][4 * q + 1], gh[bp][4 * q + 2], gh[bp][4 * q + 3]);
                *reinterpret_cast<float4*>(rt) = make_float4(X[bp][4 * q], X[bp][4 * q + 1], X[bp][4 * q + 2], X[bp][4 * q + 3]);
            }
    }
    if (A.g_enc) {
        // g_r = Wa^T g_hid + attn g_pooled[agent];  d/d(enc) = keep * scale * g_r
        const long long agent = rr / A.k;
        const float at = valid ? A.attn[rr] : 0.f;
        uint4 kb = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (A.keep_bits && valid) kb = *reinterpret_cast<const uint4*>(A.keep_bits + row * 4);
#pragma unroll 1
        for (int blk = 0; blk < 4; ++blk) {
            const unsigned kwb = blk == 0 ? kb.x : (blk == 1 ? kb.y : (blk == 2 ? kb.z : kb.w));
            f32x16 gx;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 gp = *reinterpret_cast<const float4*>(A.g_pooled + agent * CH + cfeat0(blk, q, h));
                gx[4 * q] = at * gp.x; gx[4 * q + 1] = at * gp.y; gx[4 * q + 2] = at * gp.z; gx[4 * q + 3] = at * gp.w;
            }
#pragma unroll
            for (int bp = 0; bp < 4; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float w[4];                                           // lane (i = j, h): Wa[32 bp + 8 q + 4 h + u][32 blk + i]
#pragma unroll
                    for (int u = 0; u < 4; ++u) w[u] = A.wa[(size_t)(cfeat0(bp, q, h) + u) * CH + 32 * blk + j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) gx = cmfma(w[u], gh[bp][4 * q + u], gx);
                }
            if (valid) {
                float* o = A.g_enc + row * CH;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (kwb >> (8 * q + 4 * h + u)) & 1u ? A.scale * gx[4 * q + u] : 0.f;
                    *reinterpret_cast<float4*>(o + cfeat0(blk, q, h)) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    const float sgs = wave_sum(h == 0 ? gs : 0.f);
    if (lane == 0) sm[wave] = sgs;
    __syncthreads();
    // ---- phase 2: wave w accumulates rows 32 w .. 32 w + 31 of dWa = g_hid^T r over the workgroup's 128 rows ----
    f32x16 dw[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[b][r] = 0.f;
    float s_dba = 0.f;
#pragma unroll 4
    for (int s = 0; s < 64; ++s) {                                        // k-step s: rows 2 s + h
        const float* gr = G + (2 * s + h) * CORR_TSTRIDE;
        const float* rw = R + (2 * s + h) * CORR_TSTRIDE;
        const float a = gr[32 * wave + j];                                // A: lane (f = j, h) = g_hid[row][32 w + f]
        dw[0] = cmfma(a, rw[j], dw[0]); dw[1] = cmfma(a, rw[32 + j], dw[1]);
        dw[2] = cmfma(a, rw[64 + j], dw[2]); dw[3] = cmfma(a, rw[96 + j], dw[3]);
        s_dba += a;
    }
    s_dba += __shfl_xor(s_dba, 32, 64);
    __syncthreads();
    // ---- phase 3: g_s * hid through the same tile, column sums = dwb ----
#pragma unroll
    for (int bp = 0; bp < 4; ++bp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* gt = G + (32 * wave + j) * CORR_TSTRIDE + cfeat0(bp, q, h);
            *reinterpret_cast<float4*>(gt) = make_float4(gs * hid[bp][4 * q], gs * hid[bp][4 * q + 1], gs * hid[bp][4 * q + 2],
                                                         gs * hid[bp][4 * q + 3]);
        }
    __syncthreads();
    float s_dwb = 0.f;
#pragma unroll 8
    for (int s = 0; s < 64; ++s) s_dwb += G[(2 * s + h) * CORR_TSTRIDE + 32 * wave + j];
    s_dwb += __shfl_xor(s_dwb, 32, 64);
    // ---- the slot: every wave owns its rows of dWa and its 32 entries of dba / dwb ----
    float* out = A.partials_a + (size_t)blockIdx.x * CORRA_PART;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)      // register r, lane (c = j, h): dWa[32 w + (r & 3) + 8 (r >> 2) + 4 h][32 b + c]
            out[(size_t)(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h) * CH + 32 * b + j] = dw[b][r];
    if (h == 0) {
        out[CH * CH + 32 * wave + j] = s_dba;
        out[CH * CH + CH + 32 * wave + j] = s_dwb;
    }
    if (threadIdx.x < 4) out[CH * CH + 2 * CH + threadIdx.x] = threadIdx.x == 0 ? (sm[0] + sm[1]) + (sm[2] + sm[3]) : 0.f;
}

}  // namespace piml

using namespace piml;

PIML_API int piml_corrector_partial_floats(int which) { return which == 0 ? CORRA_PART : CORRB_PART; }
// slots of the two partial sets: 0 = the row tiles' (dWa | dba | dwb | dbb), 1 = the agents' (dWc | dbc | dWd | dbd)
PIML_API int piml_corrector_slots(int which, long long agents, int k) {
    if (agents <= 0 || k <= 0) return 0;
    if (which == 0) return (int)((agents * k + 127) / 128);
    const long long wg = (agents + 3) / 4;
    return (int)(wg < 256 ? wg : 256);
}

static int corrector_check(const piml_corrector* A, bool bwd) {
    if (!A || A->agents < 0 || A->k < 1 || A->k > 64 || A->agents * A->k >= (1ll << 31)) return hipErrorInvalidValue;
    if (A->agents == 0) return hipSuccess;
    if (!A->enc || !A->wa || !A->ba || !A->wb || !A->bb || !A->wc || !A->bc || !A->wd || !A->bd || !A->score || !A->attn ||
        !A->pooled || !A->chid || !A->out)
        return hipErrorInvalidValue;
    if (bwd && (!A->hid || !A->g_out || !A->g_pooled || !A->g_score || !A->partials_a || !A->partials_b || !A->grads))
        return hipErrorInvalidValue;
    return hipSuccess;
}

PIML_API int piml_corrector_fwd(const piml_corrector* A, void* stream) {
    if (int e = corrector_check(A, false)) return e;
    if (A->agents == 0) return hipSuccess;
    const long long rows = A->agents * A->k;
    hipLaunchKernelGGL(corr_rows_fwd_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, as_stream(stream), *A);
    hipLaunchKernelGGL(corr_agents_fwd_kernel, dim3((unsigned)piml_corrector_slots(1, A->agents, A->k)), dim3(256), 0, as_stream(stream), *A);
    trace_mark("corrector_fwd", as_stream(stream));
    return hipGetLastError();
}

PIML_API int piml_corrector_bwd(const piml_corrector* A, int accumulate, void* stream) {
    if (int e = corrector_check(A, true)) return e;
    if (A->agents == 0) return hipSuccess;
    static int attr = -1;      // dynamic LDS above 64 KB has to be enabled per kernel once per process
    if (attr < 0)
        attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(corr_rows_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        CORR_ROWS_BWD_LDS);
    if (attr) return attr;
    const int sa = piml_corrector_slots(0, A->agents, A->k), sb = piml_corrector_slots(1, A->agents, A->k);
    hipLaunchKernelGGL(corr_agents_bwd_kernel, dim3((unsigned)sb), dim3(256), 0, as_stream(stream), *A);
    hipLaunchKernelGGL(corr_rows_bwd_kernel, dim3((unsigned)sa), dim3(256), CORR_ROWS_BWD_LDS, as_stream(stream), *A);
    ReduceAll R = {};
    R.accumulate = accumulate ? 1 : 0;
    R.set[0] = ReduceSet{A->partials_a, A->grads, sa, CORRA_PART / 4, 0x7fffffff, 0, 0};
    R.set[1] = ReduceSet{A->partials_b, A->grads + CORRA_PART, sb, CORRB_PART / 4, 0x7fffffff, 0, 0};
    R.nsets = 2;
    R.gx = (CORRA_PART / 4 + 15) / 16;
    const int e = launch_slot_sums(R, as_stream(stream));
    trace_mark("corrector_bwd", as_stream(stream));
    return e;
}
