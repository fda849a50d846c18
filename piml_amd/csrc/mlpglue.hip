// Glue kernels around the PINNSF network's GEMMs (which stay rocBLAS / hipBLASLt calls made by
// PyTorch-ROCm): the desired-force epilogue, the self-feature rows, the neighbour-axis sum and the
// backward of Linear(+ReLU) minus its two GEMMs.  Reference: src/models/model.py:40-65 (MLP),
// :1279-1294 (k-sum, desired force); SURVEY.md row a8.  All of them are single-pass, HBM/L2-bound
// element streams; what they buy is the removal of ~60 launch-bound torch kernels per step.
#include "common.hpp"
#include "../../include/piml_hip.h"

namespace piml {

// ---------------------------------------------------------------------------------------------
// predictions = acc_ped + acc_obs + (v0 * d / t - v) / tau,  t = |d| (+0.1 where |d| == 0)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pinnsf_epilogue_fwd_kernel(const float2* __restrict__ acc_ped,
                                                                   const float2* __restrict__ acc_obs,
                                                                   const float* __restrict__ sf, size_t rows, float tau,
                                                                   float2* __restrict__ out) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* s = sf + r * 7;
    const float dx = s[0], dy = s[1], vx = s[2], vy = s[3], v0 = s[6];
    float t = norm2(dx, dy);
    t = (t == 0.f) ? t + 0.1f : t;
    float2 a = acc_ped[r];
    if (acc_obs) {
        const float2 o = acc_obs[r];
        a.x += o.x;
        a.y += o.y;
    }
    out[r] = make_float2(a.x + (v0 * (dx / t) - vx) / tau, a.y + (v0 * (dy / t) - vy) / tau);
}

// g_self from g_out.  e = d / t; g_e = g v0 / tau; g_d = g_e / t + g_t * d / n  (0 where n == 0),
// g_t = -(g_e . d) / t^2; g_v = -g / tau; g_a = 0; g_v0 = (g . e) / tau.
__device__ __forceinline__ void epilogue_bwd_row(const float2* __restrict__ g_out, const float* __restrict__ sf, size_t r, float tau,
                                                 float* __restrict__ g_self) {
    const float* s = sf + r * 7;
    const float dx = s[0], dy = s[1], v0 = s[6];
    const float2 g = g_out[r];
    const float n = norm2(dx, dy);
    const float t = (n == 0.f) ? n + 0.1f : n;
    const float ex = dx / t, ey = dy / t;
    const float gex = g.x * v0 / tau, gey = g.y * v0 / tau;
    const float gt = -(gex * dx + gey * dy) / (t * t);
    float gdx = gex / t, gdy = gey / t;
    if (n != 0.f) {
        gdx += gt * (dx / n);
        gdy += gt * (dy / n);
    }
    float* o = g_self + r * 7;
    o[0] = gdx;
    o[1] = gdy;
    o[2] = -g.x / tau;
    o[3] = -g.y / tau;
    o[4] = 0.f;
    o[5] = 0.f;
    o[6] = (g.x * ex + g.y * ey) / tau;
}

__global__ __launch_bounds__(256) void pinnsf_epilogue_bwd_kernel(const float2* __restrict__ g_out,
                                                                   const float* __restrict__ sf, size_t rows, float tau,
                                                                   float* __restrict__ g_self) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    epilogue_bwd_row(g_out, sf, r, tau, g_self);
}

// Bottleneck variants: predictions = sum_k pred_ped[a, k] + sum_k pred_obs[a, k] + desired force (one thread per agent;
// the per-neighbour predictor outputs are summed here instead of by two torch reductions), and its backward: the same
// upstream gradient broadcast to every neighbour row of the agent + g_self.
__global__ __launch_bounds__(256) void pinnsf_epilogue_ksum_fwd_kernel(const float2* __restrict__ pred_ped, int kp,
                                                                        const float2* __restrict__ pred_obs, int ko,
                                                                        const float* __restrict__ sf, size_t rows, float tau,
                                                                        float2* __restrict__ out) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* s = sf + r * 7;
    const float dx = s[0], dy = s[1], vx = s[2], vy = s[3], v0 = s[6];
    float t = norm2(dx, dy);
    t = (t == 0.f) ? t + 0.1f : t;
    float2 a = make_float2(0.f, 0.f), o = make_float2(0.f, 0.f);
    for (int i = 0; i < kp; ++i) { const float2 v = pred_ped[r * kp + i]; a.x += v.x; a.y += v.y; }
    if (pred_obs) {
        for (int i = 0; i < ko; ++i) { const float2 v = pred_obs[r * ko + i]; o.x += v.x; o.y += v.y; }
        a.x += o.x;
        a.y += o.y;
    }
    out[r] = make_float2(a.x + (v0 * (dx / t) - vx) / tau, a.y + (v0 * (dy / t) - vy) / tau);
}

__global__ __launch_bounds__(256) void pinnsf_epilogue_ksum_bwd_kernel(const float2* __restrict__ g_out, size_t rows, int kp,
                                                                        int ko, float2* __restrict__ g_ped,
                                                                        float2* __restrict__ g_obs) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float2 g = g_out[r];
    if (g_ped) for (int i = 0; i < kp; ++i) g_ped[r * kp + i] = g;
    if (g_obs) for (int i = 0; i < ko; ++i) g_obs[r * ko + i] = g;
}

// both halves of piml_pinnsf_epilogue_ksum_bwd in ONE launch (round 5: they were two launches of ~5 us over the same rows): the
// first `blocks_self` workgroups write the desired-force gradient, the rest the broadcasts
__global__ __launch_bounds__(256) void pinnsf_epilogue_ksum_bwd_both_kernel(const float2* __restrict__ g_out, const float* __restrict__ sf,
                                                                             size_t rows, float tau, int kp, int ko, float* __restrict__ g_self,
                                                                             float2* __restrict__ g_ped, float2* __restrict__ g_obs,
                                                                             unsigned blocks_self) {
    if (blockIdx.x >= blocks_self) {
        const size_t r = (size_t)(blockIdx.x - blocks_self) * blockDim.x + threadIdx.x;
        if (r >= rows) return;
        const float2 g = g_out[r];
        if (g_ped) for (int i = 0; i < kp; ++i) g_ped[r * kp + i] = g;
        if (g_obs) for (int i = 0; i < ko; ++i) g_obs[r * ko + i] = g;
        return;
    }
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    epilogue_bwd_row(g_out, sf, r, tau, g_self);
}

// The same tail for channelled (C, N, 7) input with the reference's dim=1 norm (quirk Q2,
// src/models/model.py:1290): t[c, comp] = || d[c, :, comp] ||_2 over the AGENTS of slice c, per component.
// One workgroup per slice: block reductions for the two norms (and, backward, for sum_n g_e d).
__device__ __forceinline__ float2 block_sum2(float2 v, float2* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v.x = wave_sum(v.x);
    v.y = wave_sum(v.y);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float2 s = make_float2(0.f, 0.f);
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        s.x += sh[w].x;
        s.y += sh[w].y;
    }
    return s;
}

// kp / ko >= 1: acc_ped / acc_obs hold kp / ko rows of 2 per agent (the bottleneck variants' per-neighbour predictions) and are
// summed over them here (round 4: the training rollout's channelled frames took two torch reductions in front of this launch)
__global__ __launch_bounds__(256) void pinnsf_epilogue_agentnorm_fwd_kernel(const float2* __restrict__ acc_ped,
                                                                             const float2* __restrict__ acc_obs,
                                                                             const float* __restrict__ sf, int N,
                                                                             float tau, float2* __restrict__ out, int kp = 1,
                                                                             int ko = 1) {
    __shared__ float2 sh[4];
    const size_t base = (size_t)blockIdx.x * N;
    float2 sq = make_float2(0.f, 0.f);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* s = sf + (base + n) * 7;
        sq.x += s[0] * s[0];
        sq.y += s[1] * s[1];
    }
    sq = block_sum2(sq, sh);
    float tx = sqrtf(sq.x), ty = sqrtf(sq.y);
    tx = (tx == 0.f) ? tx + 0.1f : tx;
    ty = (ty == 0.f) ? ty + 0.1f : ty;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* s = sf + (base + n) * 7;
        float2 a = make_float2(0.f, 0.f);
        for (int i = 0; i < kp; ++i) {
            const float2 q = acc_ped[(base + n) * kp + i];
            a.x += q.x; a.y += q.y;
        }
        if (acc_obs) {
            float2 o = make_float2(0.f, 0.f);
            for (int i = 0; i < ko; ++i) {
                const float2 q = acc_obs[(base + n) * ko + i];
                o.x += q.x; o.y += q.y;
            }
            a.x += o.x;
            a.y += o.y;
        }
        out[base + n] = make_float2(a.x + (s[6] * (s[0] / tx) - s[2]) / tau, a.y + (s[6] * (s[1] / ty) - s[3]) / tau);
    }
}

__global__ __launch_bounds__(256) void pinnsf_epilogue_agentnorm_bwd_kernel(const float2* __restrict__ g_out,
                                                                             const float* __restrict__ sf, int N,
                                                                             float tau, float* __restrict__ g_self) {
    __shared__ float2 sh[4];
    const size_t base = (size_t)blockIdx.x * N;
    float2 sq = make_float2(0.f, 0.f), dot = make_float2(0.f, 0.f);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* s = sf + (base + n) * 7;
        const float2 g = g_out[base + n];
        sq.x += s[0] * s[0];
        sq.y += s[1] * s[1];
        dot.x += (g.x * s[6] / tau) * s[0];
        dot.y += (g.y * s[6] / tau) * s[1];
    }
    sq = block_sum2(sq, sh);
    dot = block_sum2(dot, sh);
    const float nx = sqrtf(sq.x), ny = sqrtf(sq.y);
    const float tx = (nx == 0.f) ? nx + 0.1f : nx, ty = (ny == 0.f) ? ny + 0.1f : ny;
    const float gtx = -dot.x / (tx * tx), gty = -dot.y / (ty * ty);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* s = sf + (base + n) * 7;
        const float2 g = g_out[base + n];
        const float gex = g.x * s[6] / tau, gey = g.y * s[6] / tau;
        float* o = g_self + (base + n) * 7;
        o[0] = gex / tx + (nx != 0.f ? gtx * (s[0] / nx) : 0.f);
        o[1] = gey / ty + (ny != 0.f ? gty * (s[1] / ny) : 0.f);
        o[2] = -g.x / tau;
        o[3] = -g.y / tau;
        o[4] = 0.f;
        o[5] = 0.f;
        o[6] = (g.x * (s[0] / tx) + g.y * (s[1] / ty)) / tau;
    }
}

// ---------------------------------------------------------------------------------------------
// self_features rows = [dest_feat, v, a, v0]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void self_features_fwd_kernel(const float* __restrict__ dest_feat, int dest_ld,
                                                                 const float* __restrict__ state,
                                                                 const float* __restrict__ speed, size_t rows,
                                                                 float* __restrict__ out) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* s = state + r * 6;
    float* o = out + r * 7;
    if (dest_feat) {      // NULL: columns 0-1 were already written in place (piml_relfeat_fwd, dest_feat_ld = 7)
        const float* d = dest_feat + r * (size_t)dest_ld;
        o[0] = d[0];
        o[1] = d[1];
    }
    o[2] = s[2];
    o[3] = s[3];
    o[4] = s[4];
    o[5] = s[5];
    o[6] = speed[r];
}

__global__ __launch_bounds__(256) void self_features_bwd_kernel(const float* __restrict__ g_self, size_t rows,
                                                                 float2* __restrict__ g_dest,
                                                                 float* __restrict__ g_state,
                                                                 float* __restrict__ g_speed) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* g = g_self + r * 7;
    if (g_dest) g_dest[r] = make_float2(g[0], g[1]);
    if (g_state) {
        float* o = g_state + r * 6;
        o[0] = 0.f;
        o[1] = 0.f;
        o[2] = g[2];
        o[3] = g[3];
        o[4] = g[4];
        o[5] = g[5];
    }
    if (g_speed) g_speed[r] = g[6];
}

// ---------------------------------------------------------------------------------------------
// g_pre = g * [y > 0], db = column sums of g_pre
// ---------------------------------------------------------------------------------------------
constexpr int kColsumThreads = 256;
constexpr int kColsumMaxBlocks = 2048;
constexpr int kColsumSlabBytes = 16384;     // one slab = 4 float4 (or 16 float) loads per thread

template <int V>
struct Vec;
template <>
struct Vec<4> {
    typedef float4 type;
    static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ void add(float4& a, const float4& b) {
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    static __device__ __forceinline__ float4 mask(const float4& g, const float4& y) {
        return make_float4(y.x <= 0.f ? 0.f : g.x, y.y <= 0.f ? 0.f : g.y, y.z <= 0.f ? 0.f : g.z,
                           y.w <= 0.f ? 0.f : g.w);
    }
};
template <>
struct Vec<1> {
    typedef float type;
    static __device__ __forceinline__ float zero() { return 0.f; }
    static __device__ __forceinline__ void add(float& a, const float& b) { a += b; }
    static __device__ __forceinline__ float mask(const float& g, const float& y) { return y <= 0.f ? 0.f : g; }
};

// Column sums of rows [r0, r1) of a (., lanes * V) matrix; MASK: apply the ReLU mask and store g_pre.
// Returns this thread's partial (row group `grp`, column lane `lane`).
template <int V, bool MASK>
__device__ __forceinline__ typename Vec<V>::type colsum_rows(const float* __restrict__ g, const float* __restrict__ y,
                                                             float* __restrict__ g_pre, size_t r0, size_t r1,
                                                             int lanes, int groups, int lane, int grp) {
    typedef typename Vec<V>::type T;
    T acc = Vec<V>::zero();
    const T* gp = reinterpret_cast<const T*>(g);
    const T* yp = reinterpret_cast<const T*>(y);
    T* op = reinterpret_cast<T*>(g_pre);
    size_t r = r0 + grp;
    const size_t step = (size_t)groups;
    // four independent rows in flight per thread
    for (; r + 3 * step < r1; r += 4 * step) {
        T v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = gp[(r + u * step) * lanes + lane];
        if (MASK) {
            T w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = yp[(r + u * step) * lanes + lane];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = Vec<V>::mask(v[u], w[u]);
                op[(r + u * step) * lanes + lane] = v[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) Vec<V>::add(acc, v[u]);
    }
    for (; r < r1; r += step) {
        T v = gp[r * lanes + lane];
        if (MASK) {
            v = Vec<V>::mask(v, yp[r * lanes + lane]);
            op[r * lanes + lane] = v;
        }
        Vec<V>::add(acc, v);
    }
    return acc;
}

// sum the row groups of one block; valid in threads with grp == 0
template <int V>
__device__ __forceinline__ typename Vec<V>::type block_groups_sum(typename Vec<V>::type acc,
                                                                  typename Vec<V>::type* sh, int lanes, int groups,
                                                                  int lane, int grp, bool active) {
    typedef typename Vec<V>::type T;
    __syncthreads();
    if (active) sh[grp * lanes + lane] = acc;
    __syncthreads();
    T s = Vec<V>::zero();
    if (active && grp == 0)
        for (int q = 0; q < groups; ++q) Vec<V>::add(s, sh[q * lanes + lane]);
    return s;
}

// stage 1: one slab of rows per block -> partials[block] (or db itself when the grid is one block)
template <int V, bool MASK>
__global__ __launch_bounds__(kColsumThreads) void act_bwd_colsum_kernel(const float* __restrict__ g,
                                                                        const float* __restrict__ y, size_t rows,
                                                                        int cols, size_t rows_per_block,
                                                                        float* __restrict__ g_pre,
                                                                        float* __restrict__ partials,
                                                                        float* __restrict__ db) {
    typedef typename Vec<V>::type T;
    __shared__ T sh[kColsumThreads];
    const int lanes = cols / V;
    const int groups = kColsumThreads / lanes;
    const int lane = threadIdx.x % lanes, grp = threadIdx.x / lanes;
    const bool active = grp < groups;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block;
    const size_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    T acc = Vec<V>::zero();
    if (active) acc = colsum_rows<V, MASK>(g, y, g_pre, r0, r1, lanes, groups, lane, grp);
    const T s = block_groups_sum<V>(acc, sh, lanes, groups, lane, grp, active);
    if (active && grp == 0)
        reinterpret_cast<T*>(gridDim.x == 1 ? db : partials + (size_t)blockIdx.x * cols)[lane] = s;
}

// stage 2: block `lane` sums column lane `lane` of the (nb, lanes) partials in a fixed order
template <int V>
__device__ __forceinline__ void colsum_stage2_body(const float* __restrict__ partials, int nb, int lanes, int lane,
                                                   float* __restrict__ db, typename Vec<V>::type* sh) {
    typedef typename Vec<V>::type T;
    const T* p = reinterpret_cast<const T*>(partials);
    T acc = Vec<V>::zero();
    for (int r = threadIdx.x; r < nb; r += kColsumThreads) Vec<V>::add(acc, p[(size_t)r * lanes + lane]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int w = kColsumThreads / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) Vec<V>::add(sh[threadIdx.x], sh[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) reinterpret_cast<T*>(db)[lane] = sh[0];
}

template <int V>
__global__ __launch_bounds__(kColsumThreads) void colsum_stage2_kernel(const float* __restrict__ partials, int nb,
                                                                       int lanes, float* __restrict__ db) {
    __shared__ typename Vec<V>::type sh[kColsumThreads];
    colsum_stage2_body<V>(partials, nb, lanes, blockIdx.x, db, sh);
}

// ---------------------------------------------------------------------------------------------
// msgs = scale * e ; pooled[agent] = sum_k msgs[agent, k]
// ---------------------------------------------------------------------------------------------
// the four keep bits of float4 lane `lane` of row `row` (piml_dropout_keep_bits layout), applied to v
__device__ __forceinline__ void keep4(float4& v, const unsigned* __restrict__ keep, size_t row, int words, int lane) {
    const unsigned m = keep[row * words + (lane >> 3)] >> (4 * (lane & 7));
    v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
}

__global__ __launch_bounds__(256) void scale_ksum_fwd_kernel(const float4* __restrict__ e,
                                                              const float4* __restrict__ bias, size_t agents, int k,
                                                              int lanes, float scale, const unsigned* __restrict__ keep,
                                                              float4* __restrict__ msgs, float4* __restrict__ pooled) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t agent = id / lanes;
    const int lane = (int)(id % lanes);
    if (agent >= agents) return;
    const size_t base = agent * k * lanes + lane;
    const int words = (lanes + 7) / 8;
    const float4 b = bias ? bias[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < k; ++j) {
        float4 v = e[base + (size_t)j * lanes];
        if (bias) { v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (keep) keep4(v, keep, agent * k + j, words, lane);
        msgs[base + (size_t)j * lanes] = v;
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    pooled[agent * lanes + lane] = s;
}

__global__ __launch_bounds__(256) void scale_ksum_bwd_kernel(const float4* __restrict__ g_pooled,
                                                              const float4* __restrict__ g_msgs, size_t agents, int k,
                                                              int lanes, float scale, const unsigned* __restrict__ keep,
                                                              float4* __restrict__ g_e, float4* __restrict__ col_partials) {
    __shared__ float4 sh[256];
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t agent = id / lanes;
    const int lane = (int)(id % lanes);
    const int words = (lanes + 7) / 8;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (agent < agents) {
        const size_t base = agent * k * lanes + lane;
        const float4 gp = g_pooled ? g_pooled[agent * lanes + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; ++j) {
            float4 v = gp;
            if (g_msgs) {
                const float4 m = g_msgs[base + (size_t)j * lanes];
                v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
            }
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            if (keep) keep4(v, keep, agent * k + j, words, lane);
            g_e[base + (size_t)j * lanes] = v;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (!col_partials) return;
    // column sums of this block's rows of g_e (lanes divides 256: thread t holds column lane t % lanes) ->
    // one partial row per block; the bias gradient of the Linear that produced e is their sum
    sh[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < lanes) {
        float4 s = sh[threadIdx.x];
        for (int q = (int)threadIdx.x + lanes; q < 256; q += lanes) {
            const float4 v = sh[q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        col_partials[(size_t)blockIdx.x * lanes + threadIdx.x] = s;
    }
}

// out[j] = sum_b parts[b, j]: the reduction over row chunks of the chunked weight-gradient GEMM
// (parts (B, n) row-major, n % 4 == 0).  64 float4 column lanes x 4 chunk groups per block (a fixed
// summation order: group g takes chunks g, g+4, ...; the groups are added 0..3).
__device__ __forceinline__ void sum_leading_body(const float4* __restrict__ parts, int B, size_t lanes, unsigned block,
                                                 float4* __restrict__ out, float4* sh) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const size_t j = (size_t)block * 64 + lane;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < lanes) {
        int b = grp;
        for (; b + 12 < B; b += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = parts[(size_t)(b + 4 * u) * lanes + j];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; b < B; b += 4) {
            const float4 v = parts[(size_t)b * lanes + j];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && j < lanes) {
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 v = sh[q * 64 + lane];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        out[j] = s;
    }
}

// One launch for the two small reductions that close a layer's backward: blocks [0, blocks_a) reduce the
// weight-gradient chunks, the remaining blocks the bias-gradient partials of act_bwd_colsum's first stage.
template <int V>
__global__ __launch_bounds__(256) void layer_reduce_kernel(const float4* __restrict__ parts, int B, size_t lanes,
                                                            float4* __restrict__ out, unsigned blocks_a,
                                                            const float* __restrict__ col_partials, int nb,
                                                            int col_lanes, float* __restrict__ db) {
    __shared__ float4 sh[256];
    if (blockIdx.x < blocks_a)
        sum_leading_body(parts, B, lanes, blockIdx.x, out, sh);
    else
        colsum_stage2_body<V>(col_partials, nb, col_lanes, (int)(blockIdx.x - blocks_a), db,
                              reinterpret_cast<typename Vec<V>::type*>(sh));
}

inline unsigned blocks_for(size_t n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace piml

using namespace piml;

PIML_API int piml_pinnsf_epilogue_fwd(const float* acc_ped, const float* acc_obs, const float* self_features,
                                      size_t rows, float tau, float* out, void* stream) {
    if (rows == 0) return hipSuccess;
    if (!acc_ped || !self_features || !out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pinnsf_epilogue_fwd_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(acc_ped), reinterpret_cast<const float2*>(acc_obs),
                       self_features, rows, tau, reinterpret_cast<float2*>(out));
    return hipGetLastError();
}

PIML_API int piml_pinnsf_epilogue_bwd(const float* g_out, const float* self_features, size_t rows, float tau,
                                      float* g_self, void* stream) {
    if (rows == 0) return hipSuccess;
    if (!g_out || !self_features || !g_self) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pinnsf_epilogue_bwd_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(g_out), self_features, rows, tau, g_self);
    return hipGetLastError();
}

PIML_API int piml_pinnsf_epilogue_ksum_fwd(const float* pred_ped, int kp, const float* pred_obs, int ko,
                                           const float* self_features, size_t rows, float tau, float* out, void* stream) {
    if (rows == 0) return hipSuccess;
    if (!pred_ped || !self_features || !out || kp < 1 || (pred_obs && ko < 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pinnsf_epilogue_ksum_fwd_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(pred_ped), kp, reinterpret_cast<const float2*>(pred_obs), ko,
                       self_features, rows, tau, reinterpret_cast<float2*>(out));
    return hipGetLastError();
}

// g_out (rows, 2) -> g_self (rows, 7) (may be NULL) and the broadcasts g_pred_ped (rows, kp, 2), g_pred_obs (rows, ko, 2)
PIML_API int piml_pinnsf_epilogue_ksum_bwd(const float* g_out, const float* self_features, size_t rows, float tau, int kp,
                                           int ko, float* g_self, float* g_pred_ped, float* g_pred_obs, void* stream) {
    if (rows == 0) return hipSuccess;
    if (!g_out || !self_features) return hipErrorInvalidValue;
    if (g_self && (g_pred_ped || g_pred_obs)) {
        const unsigned nb = blocks_for(rows, 256);
        hipLaunchKernelGGL(pinnsf_epilogue_ksum_bwd_both_kernel, dim3(2 * nb), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const float2*>(g_out), self_features, rows, tau, kp, ko, g_self,
                           reinterpret_cast<float2*>(g_pred_ped), reinterpret_cast<float2*>(g_pred_obs), nb);
        return hipGetLastError();
    }
    if (g_self)
        hipLaunchKernelGGL(pinnsf_epilogue_bwd_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const float2*>(g_out), self_features, rows, tau, g_self);
    if (g_pred_ped || g_pred_obs)
        hipLaunchKernelGGL(pinnsf_epilogue_ksum_bwd_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const float2*>(g_out), rows, kp, ko, reinterpret_cast<float2*>(g_pred_ped),
                           reinterpret_cast<float2*>(g_pred_obs));
    return hipGetLastError();
}

PIML_API int piml_pinnsf_epilogue_agentnorm_fwd(const float* acc_ped, const float* acc_obs,
                                                const float* self_features, int C, int N, float tau, float* out,
                                                void* stream) {
    if (C < 0 || N < 0) return hipErrorInvalidValue;
    if (C == 0 || N == 0) return hipSuccess;
    if (!acc_ped || !self_features || !out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pinnsf_epilogue_agentnorm_fwd_kernel, dim3(C), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(acc_ped), reinterpret_cast<const float2*>(acc_obs),
                       self_features, N, tau, reinterpret_cast<float2*>(out));
    return hipGetLastError();
}

PIML_API int piml_pinnsf_epilogue_ksum_agentnorm_fwd(const float* pred_ped, int kp, const float* pred_obs, int ko,
                                                     const float* self_features, int C, int N, float tau, float* out,
                                                     void* stream) {
    if (C < 0 || N < 0 || kp < 1 || (pred_obs && ko < 1)) return hipErrorInvalidValue;
    if (C == 0 || N == 0) return hipSuccess;
    if (!pred_ped || !self_features || !out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pinnsf_epilogue_agentnorm_fwd_kernel, dim3(C), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(pred_ped), reinterpret_cast<const float2*>(pred_obs), self_features, N, tau,
                       reinterpret_cast<float2*>(out), kp, pred_obs ? ko : 1);
    return hipGetLastError();
}

PIML_API int piml_pinnsf_epilogue_agentnorm_bwd(const float* g_out, const float* self_features, int C, int N,
                                                float tau, float* g_self, void* stream) {
    if (C < 0 || N < 0) return hipErrorInvalidValue;
    if (C == 0 || N == 0) return hipSuccess;
    if (!g_out || !self_features || !g_self) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pinnsf_epilogue_agentnorm_bwd_kernel, dim3(C), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(g_out), self_features, N, tau, g_self);
    return hipGetLastError();
}

PIML_API int piml_self_features_fwd(const float* dest_feat, int dest_ld, const float* state,
                                    const float* desired_speed, size_t rows, float* out, void* stream) {
    if (dest_feat && dest_ld < 2) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!state || !desired_speed || !out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(self_features_fwd_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, as_stream(stream),
                       dest_feat, dest_ld, state, desired_speed, rows, out);
    return hipGetLastError();
}

PIML_API int piml_self_features_bwd(const float* g_self, size_t rows, float* g_dest, float* g_state, float* g_speed,
                                    void* stream) {
    if (rows == 0) return hipSuccess;
    if (!g_self) return hipErrorInvalidValue;
    hipLaunchKernelGGL(self_features_bwd_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, as_stream(stream), g_self,
                       rows, reinterpret_cast<float2*>(g_dest), g_state, g_speed);
    return hipGetLastError();
}

PIML_API int piml_colsum_blocks(size_t rows, int cols) {
    if (cols <= 0) return 0;
    size_t slab = kColsumSlabBytes / (sizeof(float) * (size_t)cols);     // rows per block
    if (slab < 1) slab = 1;
    size_t b = (rows + slab - 1) / slab;
    if (b < 1) b = 1;
    if (b > (size_t)kColsumMaxBlocks) b = kColsumMaxBlocks;
    return (int)b;
}

static int act_bwd_colsum_launch(const float* g, const float* y, size_t rows, int cols, float* g_pre,
                                 float* partials, float* db, void* stream, bool second_stage) {
    if (cols <= 0) return hipErrorInvalidValue;
    const bool vec = (cols % 4 == 0);
    if (vec ? cols > 1024 : cols > 256) return hipErrorInvalidValue;
    if (!db) return hipErrorInvalidValue;
    if (rows == 0) return hipMemsetAsync(db, 0, sizeof(float) * cols, as_stream(stream));
    if (!g || (y && !g_pre)) return hipErrorInvalidValue;
    const int nb = piml_colsum_blocks(rows, cols);
    if (nb > 1 && !partials) return hipErrorInvalidValue;
    const size_t rpb = (rows + nb - 1) / nb;
#define PIML_COLSUM(V, MASK)                                                                                        \
    hipLaunchKernelGGL((act_bwd_colsum_kernel<V, MASK>), dim3(nb), dim3(kColsumThreads), 0, as_stream(stream), g, y, \
                       rows, cols, rpb, g_pre, partials, db)
    if (vec) {
        if (y) PIML_COLSUM(4, true); else PIML_COLSUM(4, false);
        if (nb > 1 && second_stage)
            hipLaunchKernelGGL(colsum_stage2_kernel<4>, dim3(cols / 4), dim3(kColsumThreads), 0, as_stream(stream),
                               partials, nb, cols / 4, db);
    } else {
        if (y) PIML_COLSUM(1, true); else PIML_COLSUM(1, false);
        if (nb > 1 && second_stage)
            hipLaunchKernelGGL(colsum_stage2_kernel<1>, dim3(cols), dim3(kColsumThreads), 0, as_stream(stream),
                               partials, nb, cols, db);
    }
#undef PIML_COLSUM
    return hipGetLastError();
}

PIML_API int piml_act_bwd_colsum(const float* g, const float* y, size_t rows, int cols, float* g_pre,
                                 float* partials, float* db, void* stream) {
    return act_bwd_colsum_launch(g, y, rows, cols, g_pre, partials, db, stream, true);
}

PIML_API int piml_act_bwd_colsum_stage1(const float* g, const float* y, size_t rows, int cols, float* g_pre,
                                        float* partials, float* db, void* stream) {
    return act_bwd_colsum_launch(g, y, rows, cols, g_pre, partials, db, stream, false);
}

PIML_API int piml_scale_ksum_fwd(const float* e, const float* bias, size_t agents, int k, int cols, float scale,
                                 const unsigned* keep_bits, float* msgs, float* pooled, void* stream) {
    if (k <= 0 || cols <= 0 || cols % 4) return hipErrorInvalidValue;
    if (agents == 0) return hipSuccess;
    if (!e || !msgs || !pooled) return hipErrorInvalidValue;
    const int lanes = cols / 4;
    hipLaunchKernelGGL(scale_ksum_fwd_kernel, dim3(blocks_for(agents * lanes, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(e), reinterpret_cast<const float4*>(bias), agents, k, lanes, scale,
                       keep_bits, reinterpret_cast<float4*>(msgs), reinterpret_cast<float4*>(pooled));
    return hipGetLastError();
}

PIML_API int piml_ksum_blocks(size_t agents, int cols) {
    if (cols <= 0 || cols % 4) return 0;
    return (int)blocks_for(agents * (size_t)(cols / 4), 256);
}

PIML_API int piml_scale_ksum_bwd(const float* g_pooled, const float* g_msgs, size_t agents, int k, int cols,
                                 float scale, const unsigned* keep_bits, float* g_e, float* col_partials, void* stream) {
    if (k <= 0 || cols <= 0 || cols % 4) return hipErrorInvalidValue;
    if (agents == 0) return hipSuccess;
    if (!g_e) return hipErrorInvalidValue;
    const int lanes = cols / 4;
    if (col_partials && 256 % lanes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scale_ksum_bwd_kernel, dim3(blocks_for(agents * lanes, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(g_pooled), reinterpret_cast<const float4*>(g_msgs), agents, k,
                       lanes, scale, keep_bits, reinterpret_cast<float4*>(g_e), reinterpret_cast<float4*>(col_partials));
    return hipGetLastError();
}

PIML_API int piml_layer_reduce(const float* parts, int B, size_t n, float* out, const float* col_partials, int nb,
                               int cols, float* db, void* stream) {
    const bool have_a = parts != nullptr && B > 0 && n > 0;
    const bool have_b = col_partials != nullptr && nb > 1;
    if (have_a && (n % 4 || !out)) return hipErrorInvalidValue;
    if (have_b && (cols <= 0 || !db || (cols % 4 == 0 ? cols > 1024 : cols > 256))) return hipErrorInvalidValue;
    if (!have_a && !have_b) return hipSuccess;
    const unsigned blocks_a = have_a ? blocks_for(n / 4, 64) : 0u;
    const bool vec = have_b ? (cols % 4 == 0) : true;
    const int col_lanes = have_b ? (vec ? cols / 4 : cols) : 0;
    const dim3 grid(blocks_a + (unsigned)col_lanes);
    if (vec)
        hipLaunchKernelGGL(layer_reduce_kernel<4>, grid, dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const float4*>(parts), B, n / 4, reinterpret_cast<float4*>(out), blocks_a,
                           col_partials, nb, col_lanes, db);
    else
        hipLaunchKernelGGL(layer_reduce_kernel<1>, grid, dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const float4*>(parts), B, n / 4, reinterpret_cast<float4*>(out), blocks_a,
                           col_partials, nb, col_lanes, db);
    return hipGetLastError();
}

PIML_API int piml_sum_leading(const float* parts, int B, size_t n, float* out, void* stream) {
    if (B <= 0 || n % 4) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    if (!parts || !out) return hipErrorInvalidValue;
    return piml_layer_reduce(parts, B, n, out, nullptr, 0, 0, nullptr, stream);
}
