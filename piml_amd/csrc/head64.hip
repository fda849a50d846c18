// Collision head of `pinnsf_bm` on the f32 matrix cores, forward AND backward (the reference trains this head:
// src/models/model.py:1183 `ped_collision_predictor = MLP(64, [64, 1])`, :1214-1215 `sigmoid(...)` on the decoder output of
// every pedestrian neighbour row; BCE against the 1-s collision labels at src/models/simulators.py:348-355):
//     hid = relu(W1 x + b1),  out = sigmoid(w2 . hid + b2)           x (rows, 64) = the row decoder's output
// Until round 3 this was the last piece of the shipped experiments' model on library GEMMs (~40 us of a 0.29 ms step).
//
// Same formulation as decoder.hip (features on the MFMA's M axis, rows on its N axis; lane (j, h) of a 32-row tile holds
// features 8 q + 4 h + u of row j in accumulator registers 4 q + u).  W1 is 16 KB: the A fragments are read straight from
// the row-major weight (forward: one float4 per fragment; transposed products: four dwords), no packed image.
//   forward   one wave per tile: 64 MFMAs for the hidden layer, the 64 -> 1 layer as 32 FMAs per lane + a cross-half add
//   backward  one wave per tile: gz = g_out * out * (1 - out); g_hid = gz * w2 * [hid > 0]; g_x = W1^T g_hid (64 MFMAs);
//             dW1 = g_hid^T x over the tile's 32 rows (64 MFMAs; g_hid transposed through LDS, x read in operand layout),
//             db1 / dw2 / db2 as column sums; the four waves of a workgroup add their partials in LDS in a fixed order,
//             one slot per workgroup, a slot sum afterwards (no atomics: bit-reproducible).
#include "common.hpp"
#include "pack.hpp"
#include "reduce.hpp"
#include "trace.hpp"
#include "../../include/piml_hip.h"

namespace piml {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HD = 64;
constexpr int H64_PART = HD * HD + HD + HD + 4;        // dW1 | db1 | dw2 | db2 + pad

__device__ __forceinline__ f32x16 hmfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int hfeat0(int blk, int q, int h) { return 32 * blk + 8 * q + 4 * h; }

__global__ __launch_bounds__(256) void head64_fwd_kernel(piml_head64 A) {
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile * 32 >= A.rows) return;
    const long long row = tile * 32 + j;
    const bool valid = row < A.rows;
    const float* xr = A.x + (valid ? row : 0) * HD;
    f32x16 X[2];
#pragma unroll
    for (int bp = 0; bp < 2; ++bp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(xr + hfeat0(bp, q, h));
            X[bp][4 * q] = v.x; X[bp][4 * q + 1] = v.y; X[bp][4 * q + 2] = v.z; X[bp][4 * q + 3] = v.w;
        }
    float dot = 0.f;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        f32x16 a;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = *reinterpret_cast<const float4*>(A.b1 + hfeat0(ob, q, h));
            a[4 * q] = bq.x; a[4 * q + 1] = bq.y; a[4 * q + 2] = bq.z; a[4 * q + 3] = bq.w;
        }
        const float* wrow = A.w1 + (size_t)(32 * ob + j) * HD;           // lane (i = j, h): W1[32 ob + i][...]
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) {
            float4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const float4*>(wrow + hfeat0(bp, q, h));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a = hmfma(w[q].x, X[bp][4 * q + 0], a);
                a = hmfma(w[q].y, X[bp][4 * q + 1], a);
                a = hmfma(w[q].z, X[bp][4 * q + 2], a);
                a = hmfma(w[q].w, X[bp][4 * q + 3], a);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
        if (A.hidden && valid) {
            float* o = A.hidden + row * HD;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(o + hfeat0(ob, q, h)) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w2 = *reinterpret_cast<const float4*>(A.w2 + hfeat0(ob, q, h));
            dot += w2.x * a[4 * q] + w2.y * a[4 * q + 1] + w2.z * a[4 * q + 2] + w2.w * a[4 * q + 3];
        }
    }
    dot += __shfl_xor(dot, 32, 64);
    if (h == 0 && valid) A.out[row] = 1.f / (1.f + expf(-(dot + A.b2[0])));
}

__global__ __launch_bounds__(256) void head64_bwd_kernel(piml_head64 A) {
    __shared__ __attribute__((aligned(16))) float lds[4][32 * HD];       // per wave: a 32 x 64 tile, rows x features
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const long long row = tile * 32 + j;
    const bool valid = row < A.rows;
    const long long rr = valid ? row : 0;
    float* T = lds[wave];
    // ---- gz, g_hid (accumulator layout), gz * hid ----
    float gz = 0.f;
    if (valid) {
        const float o = A.out[row];
        gz = A.g_out[row] * o * (1.f - o);
    }
    f32x16 gh[2];
    float gzh[2][16];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 hv = *reinterpret_cast<const float4*>(A.hidden + rr * HD + hfeat0(ob, q, h));
            const float4 w2 = *reinterpret_cast<const float4*>(A.w2 + hfeat0(ob, q, h));
            const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, ww[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gh[ob][4 * q + u] = hh[u] > 0.f ? gz * ww[u] : 0.f;
                gzh[ob][4 * q + u] = gz * hh[u];
            }
        }
    // ---- g_x = W1^T g_hid ----
    if (A.g_x) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 gx;
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[r] = 0.f;
#pragma unroll
            for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float w[4];                                           // lane (i = j, h): W1[32 bp + 8 q + 4 h + u][32 blk + i]
#pragma unroll
                    for (int u = 0; u < 4; ++u) w[u] = A.w1[(size_t)(hfeat0(bp, q, h) + u) * HD + 32 * blk + j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) gx = hmfma(w[u], gh[bp][4 * q + u], gx);
                }
            if (valid) {
                float* o = A.g_x + row * HD;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + hfeat0(blk, q, h)) = make_float4(gx[4 * q], gx[4 * q + 1], gx[4 * q + 2], gx[4 * q + 3]);
            }
        }
    }
    // ---- weight gradients of this tile: g_hid transposed through LDS (lane = feature, k = row) ----
    auto put = [&](const float (&v)[16], int ob) {                       // T[row j][feature]
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(T + j * HD + hfeat0(ob, q, h)) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    };
    {
        float t0[16], t1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { t0[r] = gh[0][r]; t1[r] = gh[1][r]; }
        put(t0, 0); put(t1, 1);
    }
    __builtin_amdgcn_wave_barrier();
    f32x16 dw[2][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[u >> 1][u & 1][r] = 0.f;
    float s_db1[2] = {0.f, 0.f}, s_dw2[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {                                        // k-step s: rows 2 s + h of the tile
        const long long xrow = tile * 32 + 2 * s + h;
        const float* xp = A.x + (xrow < A.rows ? xrow : 0) * HD;
        const float b0 = xp[j], b1 = xp[32 + j];                          // B: lane (c = j, h) = x[row 2 s + h][32 blk + c]
        const float a0 = T[(2 * s + h) * HD + j], a1 = T[(2 * s + h) * HD + 32 + j];   // A: lane (f = j, h) = g_hid[row 2 s + h][32 ob + f]
        dw[0][0] = hmfma(a0, b0, dw[0][0]); dw[0][1] = hmfma(a0, b1, dw[0][1]);
        dw[1][0] = hmfma(a1, b0, dw[1][0]); dw[1][1] = hmfma(a1, b1, dw[1][1]);
        s_db1[0] += a0; s_db1[1] += a1;
    }
    __builtin_amdgcn_wave_barrier();
    put(gzh[0], 0); put(gzh[1], 1);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        s_dw2[0] += T[(2 * s + h) * HD + j];
        s_dw2[1] += T[(2 * s + h) * HD + 32 + j];
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        s_db1[ob] += __shfl_xor(s_db1[ob], 32, 64);
        s_dw2[ob] += __shfl_xor(s_dw2[ob], 32, 64);
    }
    float s_db2 = h == 0 ? gz : 0.f;
    s_db2 = wave_sum(s_db2);
    // ---- the workgroup's four waves add up in LDS, wave 0 first (fixed order), one slot per workgroup ----
    __syncthreads();
    float* P = &lds[0][0];                                               // H64_PART floats, the tiles are dead
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            // accumulator (ob, blk), register r, lane (c = j, h): dW1[32 ob + (r & 3) + 8 (r >> 2) + 4 h][32 blk + c]
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* p = P + (32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h) * HD + 32 * blk + j;
                        *p = (w == 0 ? 0.f : *p) + dw[ob][blk][r];
                    }
            if (h == 0) {
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    float* p1 = P + HD * HD + 32 * ob + j;
                    float* p2 = P + HD * HD + HD + 32 * ob + j;
                    *p1 = (w == 0 ? 0.f : *p1) + s_db1[ob];
                    *p2 = (w == 0 ? 0.f : *p2) + s_dw2[ob];
                }
                if (lane == 0) {
                    float* p3 = P + HD * HD + 2 * HD;
                    *p3 = (w == 0 ? 0.f : *p3) + s_db2;
                }
            }
        }
        __syncthreads();
    }
    float* out = A.partials + (size_t)blockIdx.x * H64_PART;
    for (int e = threadIdx.x; e < H64_PART; e += 256) out[e] = e < HD * HD + 2 * HD + 1 ? P[e] : 0.f;
}

// The same backward with the FOUR waves of a workgroup on ONE tile (few rows: the training loops' 3 000 neighbour rows are 92 tiles =
// 23 workgroups of the kernel above, each wave a chain of 128 dependent-issue matrix instructions and the workgroup a four-step
// serial sum; here 92 workgroups, wave w = (K half w >> 1, output block w & 1) of g_x -- the halves meet in LDS, fixed order -- and
// block (w >> 1, w & 1) of dW1: 16 + 16 instructions per wave, disjoint blocks, no serial sum; round 6).  One slot per TILE.
__global__ __launch_bounds__(256) void head64_bwd_coop_kernel(piml_head64 A) {
    __shared__ __attribute__((aligned(16))) float lds[4][32 * HD];       // per wave: its transposed 32 x 64 tile; later the slot image
    __shared__ __attribute__((aligned(16))) float gxp[2][16][64];        // g_x partials of the waves with K half 1: [block][register][lane]
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const int wb = wave & 1, wk = wave >> 1;                             // g_x: output block / K half;  dW1: block (ob = wk, blk = wb)
    const long long tile = blockIdx.x;
    const long long row = tile * 32 + j;
    const bool valid = row < A.rows;
    const long long rr = valid ? row : 0;
    float* T = lds[wave];
    float gz = 0.f;
    if (valid) {
        const float o = A.out[row];
        gz = A.g_out[row] * o * (1.f - o);
    }
    // this wave's half of the hidden features (block wk): g_hid (accumulator layout) and gz * hid
    float ghv[16], gzh[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 hv = *reinterpret_cast<const float4*>(A.hidden + rr * HD + hfeat0(wk, q, h));
        const float4 w2 = *reinterpret_cast<const float4*>(A.w2 + hfeat0(wk, q, h));
        const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, ww[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ghv[4 * q + u] = hh[u] > 0.f ? gz * ww[u] : 0.f;
            gzh[4 * q + u] = gz * hh[u];
        }
    }
    // ---- g_x block wb over the K half wk ----
    f32x16 gx;
#pragma unroll
    for (int r = 0; r < 16; ++r) gx[r] = 0.f;
    if (A.g_x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float w[4];                                                   // lane (i = j, h): W1[32 wk + 8 q + 4 h + u][32 wb + i]
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = A.w1[(size_t)(hfeat0(wk, q, h) + u) * HD + 32 * wb + j];
#pragma unroll
            for (int u = 0; u < 4; ++u) gx = hmfma(w[u], ghv[4 * q + u], gx);
        }
        if (wk == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gxp[wb][r][lane] = gx[r];
        }
    }
    // ---- this wave's transposed half tile: T[row j][feature f of block wk] ----
    auto put = [&](const float (&v)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(T + j * HD + hfeat0(0, q, h)) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    };
    put(ghv);
    __syncthreads();
    if (A.g_x && wk == 0 && valid) {
        float* o = A.g_x + row * HD;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(o + hfeat0(wb, q, h)) =
                make_float4(gx[4 * q] + gxp[wb][4 * q][lane], gx[4 * q + 1] + gxp[wb][4 * q + 1][lane], gx[4 * q + 2] + gxp[wb][4 * q + 2][lane],
                            gx[4 * q + 3] + gxp[wb][4 * q + 3][lane]);
    }
    // ---- dW1 block (wk, wb) = g_hid[:, block wk]^T x[:, block wb] over the tile's 32 rows ----
    f32x16 dw;
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[r] = 0.f;
    float s_db1 = 0.f, s_dw2 = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {                                        // k-step s: rows 2 s + h of the tile
        const long long xrow = tile * 32 + 2 * s + h;
        const float b0 = A.x[(xrow < A.rows ? xrow : 0) * HD + 32 * wb + j];
        const float a0 = T[(2 * s + h) * HD + j];
        dw = hmfma(a0, b0, dw);
        s_db1 += a0;
    }
    __builtin_amdgcn_wave_barrier();
    put(gzh);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 16; ++s) s_dw2 += T[(2 * s + h) * HD + j];
    s_db1 += __shfl_xor(s_db1, 32, 64);
    s_dw2 += __shfl_xor(s_dw2, 32, 64);
    float s_db2 = h == 0 ? gz : 0.f;
    s_db2 = wave_sum(s_db2);
    // ---- the slot: disjoint blocks, every wave writes its own ----
    float* out = A.partials + (size_t)blockIdx.x * H64_PART;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(32 * wk + (r & 3) + 8 * (r >> 2) + 4 * h) * HD + 32 * wb + j] = dw[r];
    if (wb == 0 && h == 0) {
        out[HD * HD + 32 * wk + j] = s_db1;
        out[HD * HD + HD + 32 * wk + j] = s_dw2;
    }
    if (wave == 0 && lane < 4) out[HD * HD + 2 * HD + lane] = lane == 0 ? s_db2 : 0.f;
}

__global__ __launch_bounds__(256) void head64_reduce_kernel(const float* __restrict__ partials, float* __restrict__ grads, int slots,
                                                            int accumulate) {
    sum_slots_16x16(partials, grads, slots, H64_PART / 4, 0x7fffffff, 0, 0, accumulate != 0);
}

}  // namespace piml

using namespace piml;

PIML_API int piml_head64_partial_floats(void) { return H64_PART; }
// one slot per workgroup: four tiles per workgroup above kHead64CoopTiles tiles, one tile per workgroup (head64_bwd_coop_kernel) below
constexpr long long kHead64CoopTiles = 2048;
static bool head64_coop(long long rows) { return (rows + 31) / 32 <= kHead64CoopTiles; }
PIML_API int piml_head64_slots(long long rows) {
    return rows <= 0 ? 0 : (int)(head64_coop(rows) ? (rows + 31) / 32 : (rows + 127) / 128);
}

static int head64_check(const piml_head64* A, bool bwd) {
    if (!A || A->rows < 0 || A->rows >= (1ll << 25)) return hipErrorInvalidValue;
    if (A->rows == 0) return hipSuccess;
    if (!A->x || !A->w1 || !A->b1 || !A->w2 || !A->b2 || !A->out) return hipErrorInvalidValue;
    if (bwd && (!A->hidden || !A->g_out || !A->partials || !A->grads)) return hipErrorInvalidValue;
    return hipSuccess;
}

PIML_API int piml_head64_fwd(const piml_head64* A, void* stream) {
    if (int e = head64_check(A, false)) return e;
    if (A->rows == 0) return hipSuccess;
    hipLaunchKernelGGL(head64_fwd_kernel, dim3((unsigned)((A->rows + 127) / 128)), dim3(256), 0, as_stream(stream), *A);
    trace_mark("head64_fwd", as_stream(stream));
    return hipGetLastError();
}

PIML_API int piml_head64_bwd(const piml_head64* A, void* stream) { return piml_head64_bwd_acc(A, 0, stream); }

PIML_API int piml_head64_bwd_acc(const piml_head64* A, int accumulate, void* stream) {
    if (int e = head64_check(A, true)) return e;
    if (A->rows == 0) return hipSuccess;
    const int slots = piml_head64_slots(A->rows);
    if (head64_coop(A->rows)) hipLaunchKernelGGL(head64_bwd_coop_kernel, dim3((unsigned)slots), dim3(256), 0, as_stream(stream), *A);
    else hipLaunchKernelGGL(head64_bwd_kernel, dim3((unsigned)slots), dim3(256), 0, as_stream(stream), *A);
    // `accumulate`: 0 / 1, or flags -- PIML_ACCUMULATE and / or PIML_DEFER_SLOT_SUMS (the header)
    const bool acc = (accumulate & 1) || (accumulate & PIML_ACCUMULATE);
    if (accumulate & PIML_DEFER_SLOT_SUMS) {          // the slot sums ride in the relfeat backward's launch (network.hip)
        ReduceAll R = {};
        R.accumulate = acc ? 1 : 0;
        R.set[0] = ReduceSet{A->partials, A->grads, slots, H64_PART / 4, 0x7fffffff, 0, 0};
        R.nsets = 1;
        R.gx = (H64_PART / 4 + 15) / 16;
        if (hipError_t e = hipGetLastError()) return e;
        trace_mark("head64_bwd", as_stream(stream));
        return pending_slot_sums_leave(R, as_stream(stream));
    }
    hipLaunchKernelGGL(head64_reduce_kernel, dim3((H64_PART / 4 + 15) / 16), dim3(256), 0, as_stream(stream), A->partials, A->grads, slots,
                       acc ? 1 : 0);
    trace_mark("head64_bwd", as_stream(stream));
    return hipGetLastError();
}
