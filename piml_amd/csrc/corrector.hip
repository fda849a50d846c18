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
//   corr_rows_fwd    one workgroup per 32-row tile, one wave per 32-feature block of hid (64 matrix instructions), s
//   corr_agents_fwd  one wave per agent: the two exponentials, the weighted sum over k, the 128 -> 64 -> 2 tail
//                    (Wc transposed in LDS once per workgroup)
//   corr_agents_bwd  one wave per agent: g_chid, g_pooled, the softmax / exp backward -> g_s
//   corr_rows_bwd    sixteen waves = four tiles x four blocks: g_hid, g_r = Wa^T g_hid + attn g_pooled, d/d(enc); the tiles'
//                    g_hid and r meet in LDS and every wave accumulates one 32 x 32 block of dWa = g_hid^T r over all 128
//                    rows, dba / dwb / dbb as column sums; further workgroups of the same launch take 128 agents each for
//                    the tail's dWc = g_chid^T pooled (matrix instructions) and dbc / dWd / dbd
// (first form, one wave per tile and four agents per wave with dWc in 128 accumulators per lane: 111 us for the four
// kernels at 4096 agents x 6 rows; this form: see profiles/)
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

// One workgroup per 32-row tile, wave `ob` computes output block ob of the hidden layer (64 matrix instructions) and its
// share of the 128 -> 1 dot product; the four shares meet in LDS and are added in a fixed order.
__global__ __launch_bounds__(256) void corr_rows_fwd_kernel(piml_corrector A) {
    __shared__ float sd[4][32];
    const int lane = threadIdx.x & 63, ob = uniform((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const long long rows = A.agents * A.k;
    const long long row = (long long)blockIdx.x * 32 + j;
    const bool valid = row < rows;
    const float* wrow = A.wa + (size_t)(32 * ob + j) * CH;                 // lane (i = j, h): Wa[32 ob + i][...]
    float4 w[4][4];
#pragma unroll
    for (int bp = 0; bp < 4; ++bp)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[bp][q] = *reinterpret_cast<const float4*>(wrow + cfeat0(bp, q, h));
    f32x16 X[4];
    corr_load_r(A, row, valid, h, X);
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bq = *reinterpret_cast<const float4*>(A.ba + cfeat0(ob, q, h));
        a[4 * q] = bq.x; a[4 * q + 1] = bq.y; a[4 * q + 2] = bq.z; a[4 * q + 3] = bq.w;
    }
#pragma unroll
    for (int bp = 0; bp < 4; ++bp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a = cmfma(w[bp][q].x, X[bp][4 * q + 0], a);
            a = cmfma(w[bp][q].y, X[bp][4 * q + 1], a);
            a = cmfma(w[bp][q].z, X[bp][4 * q + 2], a);
            a = cmfma(w[bp][q].w, X[bp][4 * q + 3], a);
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
    if (A.hid && valid) {
        float* o = A.hid + row * CH;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(o + cfeat0(ob, q, h)) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    }
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w2 = *reinterpret_cast<const float4*>(A.wb + cfeat0(ob, q, h));
        dot += w2.x * a[4 * q] + w2.y * a[4 * q + 1] + w2.z * a[4 * q + 2] + w2.w * a[4 * q + 3];
    }
    dot += __shfl_xor(dot, 32, 64);
    if (h == 0) sd[ob][j] = dot;
    __syncthreads();
    if (ob == 0 && h == 0 && valid) A.score[row] = ((sd[0][j] + sd[1][j]) + (sd[2][j] + sd[3][j])) + A.bb[0];
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

constexpr int CORR_WCT = CD + 1;            // row stride of the transposed Wc image: staging writes and matvec reads both conflict-free

__global__ __launch_bounds__(256) void corr_agents_fwd_kernel(piml_corrector A) {
    __shared__ float wct[CH * CORR_WCT];    // Wc transposed: [input 128][output 64 (+ 1)]
    __shared__ float pl[4][CH];
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    for (int e = threadIdx.x; e < CH * CD; e += 256) {
        const int c = e / CH, i = e - c * CH;           // coalesced read of Wc (64, 128)
        wct[i * CORR_WCT + c] = A.wc[e];
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
        float hs[4] = {bc, 0.f, 0.f, 0.f};                                   // four chains: the sum's latency, not its length
#pragma unroll 8
        for (int i = 0; i < CH; i += 4) {
            const float4 pv = *reinterpret_cast<const float4*>(&pl[wave][i]);
            hs[0] += wct[i * CORR_WCT + lane] * pv.x; hs[1] += wct[(i + 1) * CORR_WCT + lane] * pv.y;
            hs[2] += wct[(i + 2) * CORR_WCT + lane] * pv.z; hs[3] += wct[(i + 3) * CORR_WCT + lane] * pv.w;
        }
        const float hsum = fmaxf((hs[0] + hs[1]) + (hs[2] + hs[3]), 0.f);
        A.chid[agent * CD + lane] = hsum;
        const float o0 = wave_sum(hsum * wd0), o1 = wave_sum(hsum * wd1);
        if (lane == 0) *reinterpret_cast<float2*>(A.out + agent * 2) = make_float2(o0 + bd0, o1 + bd1);
    }
}

// Per agent: g_chid (stored for the weight-gradient tiles of corr_rows_bwd), g_pooled, the softmax / exp backward -> g_s.
__global__ __launch_bounds__(256) void corr_agents_bwd_kernel(piml_corrector A) {
    __shared__ __attribute__((aligned(16))) float wcl[CD * CH];       // Wc row-major [output 64][input 128]
    __shared__ __attribute__((aligned(16))) float gl[4][CD];
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    for (int e = threadIdx.x; e < CD * CH / 4; e += 256) reinterpret_cast<float4*>(wcl)[e] = reinterpret_cast<const float4*>(A.wc)[e];
    __syncthreads();
    const int k = A.k;
    const float wd0 = A.wd[lane], wd1 = A.wd[CD + lane];
    for (long long agent = (long long)blockIdx.x * 4 + wave; agent < A.agents; agent += (long long)gridDim.x * 4) {
        const long long row0 = agent * k;
        const float2 g = *reinterpret_cast<const float2*>(A.g_out + agent * 2);
        const float ch = A.chid[agent * CD + lane];
        const float gch = ch > 0.f ? wd0 * g.x + wd1 * g.y : 0.f;
        A.g_chid[agent * CD + lane] = gch;
        __builtin_amdgcn_wave_barrier();
        gl[wave][lane] = gch;
        __builtin_amdgcn_wave_barrier();
        float2 gp0 = make_float2(0.f, 0.f), gp1 = make_float2(0.f, 0.f);
#pragma unroll 4
        for (int c = 0; c < CD; c += 4) {
            const float4 gc = *reinterpret_cast<const float4*>(&gl[wave][c]);
            const float2 w0 = *reinterpret_cast<const float2*>(&wcl[c * CH + 2 * lane]);
            const float2 w1 = *reinterpret_cast<const float2*>(&wcl[(c + 1) * CH + 2 * lane]);
            const float2 w2 = *reinterpret_cast<const float2*>(&wcl[(c + 2) * CH + 2 * lane]);
            const float2 w3 = *reinterpret_cast<const float2*>(&wcl[(c + 3) * CH + 2 * lane]);
            gp0.x += w0.x * gc.x; gp0.y += w0.y * gc.x; gp1.x += w1.x * gc.y; gp1.y += w1.y * gc.y;
            gp0.x += w2.x * gc.z; gp0.y += w2.y * gc.z; gp1.x += w3.x * gc.w; gp1.y += w3.y * gc.w;
        }
        const float2 gp = make_float2(gp0.x + gp1.x, gp0.y + gp1.y);
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
}

// Workgroups [0, sa): four 32-row tiles each, sixteen waves -- wave 4 t + b owns tile t's block b.
//   phase 1  block b of g_r = Wa^T g_hid + attn g_pooled (64 matrix instructions), d/d(enc); block b of the tile's g_hid and
//            r into the LDS tiles
//   phase 2  wave (rb, cb) = (w & 3, w >> 2): block (rb, cb) of dWa = g_hid^T r over the 128 rows (64 matrix instructions);
//            the waves of cb = 0 also the column sums dba
//   phase 3  g_s * hid through the same tile, column sums = dwb (waves of cb = 0)
// Workgroups [sa, sa + sb): 128 AGENTS each -- the tail's weight gradients: dWc = g_chid^T pooled (waves 0..7, one 32 x 32
// block each over 64 k-steps), dbc / dWd / dbd as column sums (waves 8..12).
__global__ __launch_bounds__(1024) void corr_rows_bwd_kernel(piml_corrector A, int sa) {
    extern __shared__ __attribute__((aligned(16))) float cl[];
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    if ((int)blockIdx.x >= sa) {
        // ---- the tail's weight gradients over 128 agents ----
        float* const GC = cl;                                  // [128 agents][CD + 4]: g_chid
        float* const PL = cl + 128 * (CD + 4);                 // [128 agents][CORR_TSTRIDE]: pooled
        const long long a0 = (long long)((int)blockIdx.x - sa) * 128;
        for (int e = threadIdx.x; e < 128 * (CD / 4); e += 1024) {
            const int a = e / (CD / 4), c4 = e - a * (CD / 4);
            const float4 v = a0 + a < A.agents ? *reinterpret_cast<const float4*>(A.g_chid + (a0 + a) * CD + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(GC + a * (CD + 4) + 4 * c4) = v;
        }
        for (int e = threadIdx.x; e < 128 * (CH / 4); e += 1024) {
            const int a = e / (CH / 4), c4 = e - a * (CH / 4);
            const float4 v = a0 + a < A.agents ? *reinterpret_cast<const float4*>(A.pooled + (a0 + a) * CH + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(PL + a * CORR_TSTRIDE + 4 * c4) = v;
        }
        __syncthreads();
        float* out = A.partials_b + (size_t)((int)blockIdx.x - sa) * CORRB_PART;
        if (wave < 8) {
            const int rb = wave & 1, cb = wave >> 1;           // dWc[32 rb + ..][32 cb + ..]
            f32x16 dw;
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[r] = 0.f;
#pragma unroll 8
            for (int s = 0; s < 64; ++s) dw = cmfma(GC[(2 * s + h) * (CD + 4) + 32 * rb + j], PL[(2 * s + h) * CORR_TSTRIDE + 32 * cb + j], dw);
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h) * CH + 32 * cb + j] = dw[r];
        } else if (wave < 10) {                                 // dbc
            const int c = 32 * (wave - 8) + j;
            float sum = 0.f;
            for (int s = 0; s < 64; ++s) sum += GC[(2 * s + h) * (CD + 4) + c];
            sum += __shfl_xor(sum, 32, 64);
            if (h == 0) out[CD * CH + c] = sum;
        } else if (wave < 12) {                                 // dWd[o][c] = sum_a g_out[a][o] chid[a][c]
            const int c = 32 * (wave - 10) + j;
            float s0 = 0.f, s1 = 0.f;
            for (int s = 0; s < 64; ++s) {
                const long long a = a0 + 2 * s + h;
                if (a < A.agents) {
                    const float2 g = *reinterpret_cast<const float2*>(A.g_out + a * 2);
                    const float ch = A.chid[a * CD + c];
                    s0 += g.x * ch; s1 += g.y * ch;
                }
            }
            s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
            if (h == 0) { out[CD * CH + CD + c] = s0; out[CD * CH + 2 * CD + c] = s1; }
        } else if (wave == 12) {                                // dbd
            float s0 = 0.f, s1 = 0.f;
            for (int a = lane; a < 128; a += 64)
                if (a0 + a < A.agents) {
                    const float2 g = *reinterpret_cast<const float2*>(A.g_out + (a0 + a) * 2);
                    s0 += g.x; s1 += g.y;
                }
            s0 = wave_sum(s0); s1 = wave_sum(s1);
            if (lane == 0) { out[CD * CH + 3 * CD] = s0; out[CD * CH + 3 * CD + 1] = s1; out[CD * CH + 3 * CD + 2] = 0.f; out[CD * CH + 3 * CD + 3] = 0.f; }
        }
        return;
    }
    float* const G = cl;                                   // [128 rows][CORR_TSTRIDE]: g_hid, later g_s * hid
    float* const R = cl + 128 * CORR_TSTRIDE;              // [128 rows][CORR_TSTRIDE]: r
    float* const sm = R + 128 * CORR_TSTRIDE;              // 4 floats: the tiles' sums of g_s
    const int t = wave >> 2, b = wave & 3;                 // tile of the workgroup, block of the tile
    const long long rows = A.agents * A.k;
    const long long row = ((long long)blockIdx.x * 4 + t) * 32 + j;
    const bool valid = row < rows;
    const long long rr = valid ? row : 0;
    const float gs = valid ? A.g_score[row] : 0.f;
    // ---- phase 1 ----
    f32x16 hidb;                                            // this wave's block of hid
    {
        // this wave's block of g_hid and of r into the tiles (row 32 t + j, features of block b)
        uint4 kb = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (A.keep_bits && valid) kb = *reinterpret_cast<const uint4*>(A.keep_bits + row * 4);
        const unsigned kwb = b == 0 ? kb.x : (b == 1 ? kb.y : (b == 2 ? kb.z : kb.w));
        const float sc = valid ? A.scale : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 hv = *reinterpret_cast<const float4*>(A.hid + rr * CH + cfeat0(b, q, h));
            const float4 w2 = *reinterpret_cast<const float4*>(A.wb + cfeat0(b, q, h));
            const float4 ev = *reinterpret_cast<const float4*>(A.enc + rr * CH + cfeat0(b, q, h));
            const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, ww[4] = {w2.x, w2.y, w2.z, w2.w}, ee[4] = {ev.x, ev.y, ev.z, ev.w};
            float gv[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                hidb[4 * q + u] = valid ? hh[u] : 0.f;
                gv[u] = (valid && hh[u] > 0.f) ? gs * ww[u] : 0.f;
                rv[u] = (kwb >> (8 * q + 4 * h + u)) & 1u ? sc * ee[u] : 0.f;
            }
            *reinterpret_cast<float4*>(G + (32 * t + j) * CORR_TSTRIDE + cfeat0(b, q, h)) = make_float4(gv[0], gv[1], gv[2], gv[3]);
            *reinterpret_cast<float4*>(R + (32 * t + j) * CORR_TSTRIDE + cfeat0(b, q, h)) = make_float4(rv[0], rv[1], rv[2], rv[3]);
        }
        if (A.g_enc) {
            // block b of g_r = Wa^T g_hid + attn g_pooled[agent];  d/d(enc) = keep * scale * g_r
            const long long agent = rr / A.k;
            const float at = valid ? A.attn[rr] : 0.f;
            f32x16 gx;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 gp = *reinterpret_cast<const float4*>(A.g_pooled + agent * CH + cfeat0(b, q, h));
                gx[4 * q] = at * gp.x; gx[4 * q + 1] = at * gp.y; gx[4 * q + 2] = at * gp.z; gx[4 * q + 3] = at * gp.w;
            }
#pragma unroll
            for (int bp = 0; bp < 4; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // g_hid of the row's features 32 bp + 8 q + 4 h + u (every wave of the tile re-derives all four blocks:
                    // 64 registers of them held across the loop did not fit beside sixteen waves per CU)
                    const float4 hv = *reinterpret_cast<const float4*>(A.hid + rr * CH + cfeat0(bp, q, h));
                    const float4 w2 = *reinterpret_cast<const float4*>(A.wb + cfeat0(bp, q, h));
                    const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, ww[4] = {w2.x, w2.y, w2.z, w2.w};
                    float w[4];                                           // lane (i = j, h): Wa[32 bp + 8 q + 4 h + u][32 b + i]
#pragma unroll
                    for (int u = 0; u < 4; ++u) w[u] = A.wa[(size_t)(cfeat0(bp, q, h) + u) * CH + 32 * b + j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) gx = cmfma(w[u], (valid && hh[u] > 0.f) ? gs * ww[u] : 0.f, gx);
                }
            if (valid) {
                float* o = A.g_enc + row * CH;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (kwb >> (8 * q + 4 * h + u)) & 1u ? A.scale * gx[4 * q + u] : 0.f;
                    *reinterpret_cast<float4*>(o + cfeat0(b, q, h)) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    if (b == 0) {
        const float sgs = wave_sum(h == 0 ? gs : 0.f);
        if (lane == 0) sm[t] = sgs;
    }
    __syncthreads();
    // ---- phase 2: wave (rb, cb): block (rb, cb) of dWa = g_hid^T r over the workgroup's 128 rows ----
    const int rb = wave & 3, cb = wave >> 2;
    f32x16 dw;
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[r] = 0.f;
    float s_dba = 0.f;
#pragma unroll 8
    for (int s = 0; s < 64; ++s) {                                        // k-step s: rows 2 s + h
        const float a = G[(2 * s + h) * CORR_TSTRIDE + 32 * rb + j];      // A: lane (f = j, h) = g_hid[row][32 rb + f]
        dw = cmfma(a, R[(2 * s + h) * CORR_TSTRIDE + 32 * cb + j], dw);
        s_dba += a;
    }
    s_dba += __shfl_xor(s_dba, 32, 64);
    __syncthreads();
    // ---- phase 3: g_s * hid through the same tile, column sums = dwb ----
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(G + (32 * t + j) * CORR_TSTRIDE + cfeat0(b, q, h)) =
            make_float4(gs * hidb[4 * q], gs * hidb[4 * q + 1], gs * hidb[4 * q + 2], gs * hidb[4 * q + 3]);
    __syncthreads();
    float* out = A.partials_a + (size_t)blockIdx.x * CORRA_PART;
#pragma unroll
    for (int r = 0; r < 16; ++r)          // register r, lane (c = j, h): dWa[32 rb + (r & 3) + 8 (r >> 2) + 4 h][32 cb + c]
        out[(size_t)(32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h) * CH + 32 * cb + j] = dw[r];
    if (cb == 0) {
        float s_dwb = 0.f;
#pragma unroll 8
        for (int s = 0; s < 64; ++s) s_dwb += G[(2 * s + h) * CORR_TSTRIDE + 32 * rb + j];
        s_dwb += __shfl_xor(s_dwb, 32, 64);
        if (h == 0) {
            out[CH * CH + 32 * rb + j] = s_dba;
            out[CH * CH + CH + 32 * rb + j] = s_dwb;
        }
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
    return (int)((agents + 127) / 128);
}

// workgroups of the per-agent kernels: four agents at a time, at most 1024 (each stages Wc once)
static int corr_agent_groups(long long agents) {
    const long long wg = (agents + 3) / 4;
    return (int)(wg < 1024 ? wg : 1024);
}

static int corrector_check(const piml_corrector* A, bool bwd) {
    if (!A || A->agents < 0 || A->k < 1 || A->k > 64 || A->agents * A->k >= (1ll << 31)) return hipErrorInvalidValue;
    if (A->agents == 0) return hipSuccess;
    if (!A->enc || !A->wa || !A->ba || !A->wb || !A->bb || !A->wc || !A->bc || !A->wd || !A->bd || !A->score || !A->attn ||
        !A->pooled || !A->chid || (!bwd && !A->out))
        return hipErrorInvalidValue;
    if (bwd && (!A->hid || !A->g_out || !A->g_pooled || !A->g_score || !A->g_chid || !A->partials_a || !A->partials_b || !A->grads))
        return hipErrorInvalidValue;
    return hipSuccess;
}

PIML_API int piml_corrector_fwd(const piml_corrector* A, void* stream) {
    if (int e = corrector_check(A, false)) return e;
    if (A->agents == 0) return hipSuccess;
    const long long rows = A->agents * A->k;
    hipLaunchKernelGGL(corr_rows_fwd_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, as_stream(stream), *A);
    hipLaunchKernelGGL(corr_agents_fwd_kernel, dim3((unsigned)corr_agent_groups(A->agents)), dim3(256), 0, as_stream(stream), *A);
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
    hipLaunchKernelGGL(corr_agents_bwd_kernel, dim3((unsigned)corr_agent_groups(A->agents)), dim3(256), 0, as_stream(stream), *A);
    hipLaunchKernelGGL(corr_rows_bwd_kernel, dim3((unsigned)(sa + sb)), dim3(1024), CORR_ROWS_BWD_LDS, as_stream(stream), *A, sa);
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
