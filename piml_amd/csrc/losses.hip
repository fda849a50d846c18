// Rollout losses of the fine-tuning step (HOT LOOP C) in one launch + one for the gradient.
//
// Reference: src/models/simulators.py:172-249 (multiple_rollout_mse_loss, multiple_rollout_collision_avoidance_loss,
// multiple_rollout_collision_loss) as test_multiple_rollouts_for_training assembles them (:790-819):
//     p_res  = where(keep & gate_t, p, 0)          keep = mask_p_pred != 0,  gate_t = any agent of frame t is predicted
//     lab    = where(keep, labels[..., :2], 0)
//     mse    = sum (p_res - lab)^2 * decay^(T - 1 - t)
//     n_i    = (lab[:, T-1] - lab[:, 0]) / (|.| + 1e-6)                       per (window, agent)
//     avoid  = ((p_res - (p_res . n) n) - (lab - (lab . n) n))^2 * decay^(T - 1 - t)
//     focus  = sum [sum_t collisions > 0] * avoid * abnormal_mask             once for `collisions`, once for `hard_collisions`
// On torch operators this is ~45 forward and ~60 backward launches of a few microseconds each on 4 x 5 x 122 x 2 numbers
// (rocprofv3: 258 kernels per fine-tuning step, 200 of them such glue).  Here one thread owns a (window, agent) pair and
// walks its T frames; the three sums are reduced in a fixed order (deterministic) -- by the last workgroup out when the
// launch has more than one -- and the three gradient fields d(sum)/d(p) are written on the way, so the backward is one
// scaled sum of them.
#include "common.hpp"
#include "../../include/piml_hip.h"

#include <cmath>

namespace piml {

constexpr int LOSS_THREADS = 256;

struct LossArgs {
    const float* p;              // (C, T, N, 2)
    const float* lab;            // (C, T, N, ld): columns 0, 1 = label position
    long long ld;
    const long long* keep;       // (C, T, N) mask_p_pred (!= 0: predicted)
    const unsigned char* gate;   // (T)
    const float* coll;           // (C, T, N) or NULL
    const float* hard;           // (C, T, N) or NULL
    const float* abn;            // (N) or NULL
    int C, T, N;
    float time_decay;
    float* out;                  // 3 sums
    float* g_mse;                // (C, T, N, 2) each
    float* g_coll;
    float* g_hard;
    float* partial;              // (blocks, 6)
    unsigned* ticket;            // zero on entry, zero again on exit
    // round 4 (piml_rollout_losses_frames): the collision count records of the frames as they were produced -- frames[t] =
    // (2, C, N) floats [collisions | hard collisions] of frame t, NULL = zeros -- gated here (x gate[t]) instead of stacked and
    // multiplied by torch operators; focus = 0: the counts only feed the statistics; the weighted total in out[6..8]
    const float* frames[32];
    int use_frames, focus, stats;
    float w_coll, w_hard;
};

constexpr int LOSS_SUMS = 6;       // mse, collision focus, hard focus, sum of collisions, of hard collisions, predicted entries

__device__ __forceinline__ void block_sum6(float (&v)[LOSS_SUMS], float* red) {
    // fixed-order tree over the workgroup's threads
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < LOSS_SUMS; ++q) red[q * LOSS_THREADS + tid] = v[q];
    __syncthreads();
    for (int s = LOSS_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int q = 0; q < LOSS_SUMS; ++q) red[q * LOSS_THREADS + tid] += red[q * LOSS_THREADS + tid + s];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < LOSS_SUMS; ++q) v[q] = red[q * LOSS_THREADS];
    __syncthreads();
}

__device__ __forceinline__ void loss_write_out(const LossArgs& A, const float (&s)[LOSS_SUMS]) {
    const int t = threadIdx.x;
    if (t < 3) A.out[t] = s[t];
    if (A.stats) {
        if (t >= 3 && t < LOSS_SUMS) A.out[t] = s[t];
        if (t == 6) A.out[6] = s[0] + A.w_coll * s[1] + A.w_hard * s[2];
        if (t == 7) A.out[7] = A.w_coll * s[1];
        if (t == 8) A.out[8] = A.w_hard * s[2];
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void rollout_losses_kernel(LossArgs A) {
    __shared__ float red[LOSS_SUMS * LOSS_THREADS];
    __shared__ unsigned last;
    const int T = A.T, N = A.N;
    const long long pairs = (long long)A.C * N;
    float s[LOSS_SUMS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long e = (long long)blockIdx.x * LOSS_THREADS + threadIdx.x; e < pairs; e += (long long)gridDim.x * LOSS_THREADS) {
        const long long c = e / N, n = e - c * N;
        auto at = [&](int t) { return (c * T + t) * N + n; };
        auto label = [&](int t, float& x, float& y) {
            const long long i = at(t);
            const bool k = A.keep[i] != 0;
            x = k ? A.lab[i * A.ld] : 0.f;
            y = k ? A.lab[i * A.ld + 1] : 0.f;
        };
        auto counts = [&](int t, float& co, float& ha) {      // the frame's collision counts of this (window, agent), gated
            co = 0.f; ha = 0.f;
            if (A.use_frames) {
                const float* f = A.frames[t];
                if (f && A.gate[t] != 0) { co = f[e]; ha = f[pairs + e]; }
            } else {
                if (A.coll) co = A.coll[at(t)];
                if (A.hard) ha = A.hard[at(t)];
            }
        };
        float l0x, l0y, l1x, l1y;
        label(0, l0x, l0y);
        label(T - 1, l1x, l1y);
        float nx = l1x - l0x, ny = l1y - l0y;
        const float nn = sqrtf(nx * nx + ny * ny) + 1e-6f;         // torch.norm(ni, p=2, dim=-1) + 1e-6
        nx = nx / nn; ny = ny / nn;
        float cs = 0.f, hs = 0.f;
        for (int t = 0; t < T; ++t) {
            float co, ha;
            counts(t, co, ha);
            cs += co; hs += ha;
            if (A.keep[at(t)] == 1) s[5] += 1.f;
        }
        s[3] += cs; s[4] += hs;
        const float ab = A.abn ? A.abn[n] : 1.f;
        const bool foc = A.use_frames ? A.focus != 0 : true;
        const float wc = (foc && (A.use_frames || A.coll) && cs > 0.f) ? ab : 0.f;
        const float wh = (foc && (A.use_frames || A.hard) && hs > 0.f) ? ab : 0.f;
        for (int t = 0; t < T; ++t) {
            const long long i = at(t);
            const bool k = A.keep[i] != 0, live = k && A.gate[t] != 0;
            float lx, ly;
            label(t, lx, ly);
            const float px = live ? A.p[i * 2] : 0.f, py = live ? A.p[i * 2 + 1] : 0.f;
            const float decay = powf(A.time_decay, (float)(T - 1 - t));
            const float dx = px - lx, dy = py - ly;
            s[0] += dx * dx * decay + dy * dy * decay;
            const float dp = px * nx + py * ny, dl = lx * nx + ly * ny;
            const float ex = (px - dp * nx) - (lx - dl * nx), ey = (py - dp * ny) - (ly - dl * ny);
            const float av = ex * ex * decay + ey * ey * decay;
            s[1] += wc * av;
            s[2] += wh * av;
            // d/dp: the mask passes the gradient only where p_res is p; d|e|^2 / dp = 2 (e - (e . n) n)
            const float m = live ? 2.f * decay : 0.f;
            const float en = ex * nx + ey * ny;
            const float ax = m * (ex - en * nx), ay = m * (ey - en * ny);
            A.g_mse[i * 2] = m * dx; A.g_mse[i * 2 + 1] = m * dy;
            A.g_coll[i * 2] = wc * ax; A.g_coll[i * 2 + 1] = wc * ay;
            A.g_hard[i * 2] = wh * ax; A.g_hard[i * 2 + 1] = wh * ay;
        }
    }
    block_sum6(s, red);
    if (gridDim.x == 1) {
        loss_write_out(A, s);
        return;
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < LOSS_SUMS; ++q) A.partial[blockIdx.x * LOSS_SUMS + q] = s[q];
        __threadfence();
        last = atomicAdd(A.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float r[LOSS_SUMS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (unsigned b = threadIdx.x; b < gridDim.x; b += LOSS_THREADS)
#pragma unroll
        for (int q = 0; q < LOSS_SUMS; ++q) r[q] += A.partial[b * LOSS_SUMS + q];
    __syncthreads();
    block_sum6(r, red);
    loss_write_out(A, r);
    if (threadIdx.x == 0) *A.ticket = 0u;
}

__global__ __launch_bounds__(LOSS_THREADS) void rollout_losses_bwd_kernel(const float* __restrict__ g0, const float* __restrict__ g1,
                                                                         const float* __restrict__ g2, const float* __restrict__ g_mse,
                                                                         const float* __restrict__ g_coll, const float* __restrict__ g_hard,
                                                                         long long n, float* __restrict__ g_p, const float* __restrict__ g3,
                                                                         float w_coll, float w_hard) {
    // g0..g2: upstream of the three raw sums (frames form: of mse, of w_coll * coll, of w_hard * hard); g3: of the weighted total
    const float gt = g3 ? *g3 : 0.f;
    const float a = (g0 ? *g0 : 0.f) + gt, b = (g1 ? *g1 : 0.f) * (g3 ? w_coll : 1.f) + gt * w_coll,
                c = (g2 ? *g2 : 0.f) * (g3 ? w_hard : 1.f) + gt * w_hard;
    for (long long e = (long long)blockIdx.x * LOSS_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * LOSS_THREADS)
        g_p[e] = a * g_mse[e] + b * g_coll[e] + c * g_hard[e];
}

// ---- the collision-prediction loss of `pinnsf_bm` in the fine-tuning step (src/models/simulators.py:731-733, 826-830) ----
//     pred_collisions[:, t] = model(...)[-1] * gate_t      true_collision[:, t] = calculate_collision_label(ped_features) * gate_t
//     loss = F.binary_cross_entropy(pred_collisions, true_collision, reduction='sum') * collision_pred_weight
//     acc  = sum(round(pred_collisions) == true_collision) / numel          (frames before t_start: zeros on both sides)
// On torch operators: a label launch per frame, two stacks, two products, BCE, sum, round, ==, sum, / and their backward incl. one
// strided copy per frame -- 25 launches of a few microseconds on 4 x 5 x 122 x 6 numbers.  Here: ONE launch reads the frames'
// predictions and pedestrian features where the model / the feature kernel left them, evaluates the label (data.py:514-535, the
// arithmetic of collision_label_kernel), the clamped logarithms of binary_cross_entropy (log(p), log1p(-p), both >= -100) and
// round-half-even, reduces both sums in a fixed order (the last workgroup out when there are several) and writes
// d(loss)/d(prediction) = w gate (p - y) / max((1 - p) p, 1e-12) per frame, contiguous: the backward is one scaled copy.
struct CplArgs {
    const float* pred[32];       // frame f: (rows * k) predictions in (0, 1)
    const float* feat[32];       // frame f: (rows * k, ld) pedestrian features: relative position, relative velocity
    int nframes, k, ld, t_start, T_total;
    long long n;                 // rows * k
    const float* gates;          // (T_total) 0 / 1 floats
    float weight;
    float* out;                  // weighted loss | accuracy
    float* grad;                 // (nframes, n)
    float* partial;              // (blocks, 2)
    unsigned* ticket;
};

__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    const int tid = threadIdx.x;
    red[tid] = a; red[LOSS_THREADS + tid] = b;
    __syncthreads();
    for (int s = LOSS_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) { red[tid] += red[tid + s]; red[LOSS_THREADS + tid] += red[LOSS_THREADS + tid + s]; }
        __syncthreads();
    }
    a = red[0]; b = red[LOSS_THREADS];
    __syncthreads();
}

__global__ __launch_bounds__(LOSS_THREADS) void collision_pred_loss_kernel(CplArgs A) {
    __shared__ float red[2 * LOSS_THREADS];
    __shared__ unsigned last;
    const int f = blockIdx.y;
    const float* pp = A.pred[0];
    const float* ff = A.feat[0];
#pragma unroll 1
    for (int q = 1; q < 32; ++q) {                              // (a by-value pointer table indexed at run time lands in scratch)
        pp = (q == f) ? A.pred[q] : pp;
        ff = (q == f) ? A.feat[q] : ff;
    }
    const float g = A.gates[A.t_start + f];
    float loss = 0.f, equal = 0.f;
    for (long long e = (long long)blockIdx.x * LOSS_THREADS + threadIdx.x; e < A.n; e += (long long)gridDim.x * LOSS_THREADS) {
        const float* fr = ff + e * A.ld;
        const float px = fr[0], py = fr[1], vx = fr[2], vy = fr[3];
        float hit = 0.f;
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const float tau = __fmul_rn((float)t, 0.1f);        // torch.arange(10) * 0.1 in float32
            const float d = norm2(__fadd_rn(px, __fmul_rn(vx, tau)), __fadd_rn(py, __fmul_rn(vy, tau)));
            if (d < 0.5f && d != 0.f) hit = 1.f;
        }
        const float p = pp[e] * g, y = hit * g;
        loss += (y - 1.f) * fmaxf(log1pf(-p), -100.f) - y * fmaxf(logf(p), -100.f);
        equal += rintf(p) == y ? 1.f : 0.f;
        A.grad[(long long)f * A.n + e] = A.weight * g * ((p - y) / fmaxf((1.f - p) * p, 1e-12f));
    }
    block_sum2(loss, equal, red);
    const unsigned nblocks = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    auto finish = [&](float l, float q) {
        if (threadIdx.x == 0) {
            const float numel = (float)A.T_total * (float)A.n;
            A.out[0] = l * A.weight;
            A.out[1] = (q + (float)(A.T_total - A.nframes) * (float)A.n) / numel;      // the frames outside the rollout: 0 == 0
        }
    };
    if (nblocks == 1) { finish(loss, equal); return; }
    if (threadIdx.x == 0) {
        A.partial[bid * 2] = loss; A.partial[bid * 2 + 1] = equal;
        __threadfence();
        last = atomicAdd(A.ticket, 1u) == nblocks - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float l = 0.f, q = 0.f;
    for (unsigned b = threadIdx.x; b < nblocks; b += LOSS_THREADS) { l += A.partial[b * 2]; q += A.partial[b * 2 + 1]; }
    __syncthreads();
    block_sum2(l, q, red);
    finish(l, q);
    if (threadIdx.x == 0) *A.ticket = 0u;
}

// ---- the losses of a pointwise pre-training batch (HOT LOOP A, src/models/simulators.py:333-352, pinnsf_interaction 'sim') ----
//     mse = F.mse_loss(pred, labels[:, 4:6], reduction='sum')
//     reg = sum(reg_weight * |p_msg|)                                                  (reg_weight > 0)
//     cp  = F.binary_cross_entropy(predictions[-1], labels[:, 6:], reduction='sum')    (`pinnsf_bm` under collision_pred_weight > 0)
//     loss = mse + reg + cp
// and their gradients -- 2 (pred - label), reg_weight sign(p_msg), (p - y) / max((1 - p) p, 1e-12) -- in one launch: on torch
// operators 8 - 14 launches of a few microseconds on 128 rows.  One buffer for the three gradient fields: the backward is one
// scaled copy (none at all when the upstream gradient is the step's constant one).
struct PwlArgs {
    const float* pred;           // (rows, 2)
    const float* lab;            // (rows, ld): columns 4, 5 = the acceleration label, 6 .. 6 + k - 1 = the collision labels
    long long ld, rows;
    const float* msgs;           // (nmsg) or NULL
    long long nmsg;
    float reg_weight;
    const float* coll;           // (rows, k) or NULL
    int k;
    float* out;                  // loss | mse | reg | cp
    float* grad;                 // [2 rows | nmsg | rows k]
    float* partial;              // (blocks, 3)
    unsigned* ticket;
};

__global__ __launch_bounds__(LOSS_THREADS) void pointwise_losses_kernel(PwlArgs A) {
    __shared__ float red[3 * LOSS_THREADS];
    __shared__ unsigned last;
    const long long n0 = 2 * A.rows, n1 = A.msgs ? A.nmsg : 0, n2 = A.coll ? A.rows * A.k : 0;
    float s[3] = {0.f, 0.f, 0.f};
    for (long long e = (long long)blockIdx.x * LOSS_THREADS + threadIdx.x; e < n0 + n1 + n2; e += (long long)gridDim.x * LOSS_THREADS) {
        if (e < n0) {
            const float d = A.pred[e] - A.lab[(e >> 1) * A.ld + 4 + (e & 1)];
            s[0] += d * d;
            A.grad[e] = 2.f * d;
        } else if (e < n0 + n1) {
            const float x = A.msgs[e - n0];
            s[1] += A.reg_weight * fabsf(x);
            A.grad[e] = x > 0.f ? A.reg_weight : (x < 0.f ? -A.reg_weight : 0.f);
        } else {
            const long long q = e - n0 - n1, r = q / A.k;
            const float p = A.coll[q], y = A.lab[r * A.ld + 6 + (q - r * A.k)];
            s[2] += (y - 1.f) * fmaxf(log1pf(-p), -100.f) - y * fmaxf(logf(p), -100.f);
            A.grad[e] = (p - y) / fmaxf((1.f - p) * p, 1e-12f);
        }
    }
    auto block_sum3 = [&](float (&v)[3]) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int q = 0; q < 3; ++q) red[q * LOSS_THREADS + tid] = v[q];
        __syncthreads();
        for (int st = LOSS_THREADS / 2; st > 0; st >>= 1) {
            if (tid < st) {
#pragma unroll
                for (int q = 0; q < 3; ++q) red[q * LOSS_THREADS + tid] += red[q * LOSS_THREADS + tid + st];
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = red[q * LOSS_THREADS];
        __syncthreads();
    };
    auto finish = [&](const float (&v)[3]) {
        if (threadIdx.x == 0) {
            // the order of the reference's additions: loss = mse; loss = loss + reg; loss = loss + cp
            float t = v[0];
            if (A.msgs) t = t + v[1];
            if (A.coll) t = t + v[2];
            A.out[0] = t; A.out[1] = v[0]; A.out[2] = v[1]; A.out[3] = v[2];
        }
    };
    block_sum3(s);
    if (gridDim.x == 1) { finish(s); return; }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) A.partial[blockIdx.x * 3 + q] = s[q];
        __threadfence();
        last = atomicAdd(A.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float r[3] = {0.f, 0.f, 0.f};
    for (unsigned b = threadIdx.x; b < gridDim.x; b += LOSS_THREADS)
#pragma unroll
        for (int q = 0; q < 3; ++q) r[q] += A.partial[b * 3 + q];
    __syncthreads();
    block_sum3(r);
    finish(r);
    if (threadIdx.x == 0) *A.ticket = 0u;
}

__global__ __launch_bounds__(LOSS_THREADS) void scale_by_scalar_kernel(const float* __restrict__ g, const float* __restrict__ x, long long n,
                                                                      float* __restrict__ y) {
    const float a = g ? *g : 1.f;
    for (long long e = (long long)blockIdx.x * LOSS_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * LOSS_THREADS) y[e] = a * x[e];
}

}  // namespace piml

using namespace piml;

PIML_API int piml_collision_pred_loss_blocks(long long n, int nframes) {
    long long b = (n + 4 * LOSS_THREADS - 1) / (4 * LOSS_THREADS);
    b = b < 1 ? 1 : (b > 64 ? 64 : b);
    return (int)b * (nframes < 1 ? 1 : nframes);
}

PIML_API int piml_collision_pred_loss(const float* const* pred_frames, const float* const* feature_frames, int nframes, long long n,
                                      int k, int feature_ld, const float* gates, int t_start, int T_total, float weight, float* out,
                                      float* grad, float* partial, unsigned* ticket, void* stream) {
    if (!pred_frames || !feature_frames || nframes < 1 || nframes > 32 || n < 1 || k < 1 || feature_ld < 4 || !gates || t_start < 0 ||
        t_start + nframes > T_total || !out || !grad)
        return hipErrorInvalidValue;
    const int blocks = piml_collision_pred_loss_blocks(n, nframes);
    if (blocks > 1 && (!partial || !ticket)) return hipErrorInvalidValue;
    CplArgs A = {};
    for (int f = 0; f < nframes; ++f) {
        if (!pred_frames[f] || !feature_frames[f]) return hipErrorInvalidValue;
        A.pred[f] = pred_frames[f]; A.feat[f] = feature_frames[f];
    }
    A.nframes = nframes; A.k = k; A.ld = feature_ld; A.t_start = t_start; A.T_total = T_total; A.n = n; A.gates = gates;
    A.weight = weight; A.out = out; A.grad = grad; A.partial = partial; A.ticket = ticket;
    hipLaunchKernelGGL(collision_pred_loss_kernel, dim3((unsigned)(blocks / nframes), (unsigned)nframes), dim3(LOSS_THREADS), 0,
                       as_stream(stream), A);
    return hipGetLastError();
}

PIML_API int piml_pointwise_losses_blocks(long long rows, long long nmsg, int k) {
    const long long n = 2 * rows + (nmsg > 0 ? nmsg : 0) + rows * (k > 0 ? k : 0);
    long long b = (n + 4 * LOSS_THREADS - 1) / (4 * LOSS_THREADS);
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

PIML_API int piml_pointwise_losses(const float* pred, const float* labels, long long labels_ld, long long rows, const float* msgs,
                                   long long nmsg, float reg_weight, const float* coll_pred, int k, float* out, float* grad, float* partial,
                                   unsigned* ticket, void* stream) {
    if (!pred || !labels || rows < 1 || labels_ld < 6 || !out || !grad || (msgs && nmsg < 1) || (coll_pred && (k < 1 || labels_ld < 6 + k)))
        return hipErrorInvalidValue;
    const int blocks = piml_pointwise_losses_blocks(rows, msgs ? nmsg : 0, coll_pred ? k : 0);
    if (blocks > 1 && (!partial || !ticket)) return hipErrorInvalidValue;
    PwlArgs A = {};
    A.pred = pred; A.lab = labels; A.ld = labels_ld; A.rows = rows; A.msgs = msgs; A.nmsg = msgs ? nmsg : 0; A.reg_weight = reg_weight;
    A.coll = coll_pred; A.k = coll_pred ? k : 0; A.out = out; A.grad = grad; A.partial = partial; A.ticket = ticket;
    hipLaunchKernelGGL(pointwise_losses_kernel, dim3((unsigned)blocks), dim3(LOSS_THREADS), 0, as_stream(stream), A);
    return hipGetLastError();
}

PIML_API int piml_collision_pred_loss_bwd(const float* g_loss, const float* grad, long long n, float* g_pred, void* stream) {
    if (!grad || !g_pred || n < 1) return hipErrorInvalidValue;
    long long b = (n + LOSS_THREADS - 1) / LOSS_THREADS;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)(b > 256 ? 256 : b)), dim3(LOSS_THREADS), 0, as_stream(stream), g_loss, grad,
                       n, g_pred);
    return hipGetLastError();
}

PIML_API int piml_rollout_losses_blocks(int C, int N) {
    const long long pairs = (long long)C * N;
    long long b = (pairs + LOSS_THREADS - 1) / LOSS_THREADS;
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

PIML_API int piml_rollout_losses(const float* p, const float* labels, long long labels_ld, const long long* mask_pred,
                                 const unsigned char* gates, const float* collisions, const float* hard_collisions,
                                 const float* abnormal_mask, int C, int T, int N, float time_decay, float* out, float* g_mse,
                                 float* g_coll, float* g_hard, float* partial, unsigned* ticket, void* stream) {
    if (!p || !labels || !mask_pred || !gates || !out || !g_mse || !g_coll || !g_hard || C < 1 || T < 1 || N < 1 || labels_ld < 2)
        return hipErrorInvalidValue;
    const int blocks = piml_rollout_losses_blocks(C, N);
    if (blocks > 1 && (!partial || !ticket)) return hipErrorInvalidValue;
    LossArgs A = {};
    A.p = p; A.lab = labels; A.ld = labels_ld; A.keep = mask_pred; A.gate = gates; A.coll = collisions; A.hard = hard_collisions;
    A.abn = abnormal_mask; A.C = C; A.T = T; A.N = N; A.time_decay = time_decay; A.out = out; A.g_mse = g_mse; A.g_coll = g_coll;
    A.g_hard = g_hard; A.partial = partial; A.ticket = ticket;
    hipLaunchKernelGGL(rollout_losses_kernel, dim3(blocks), dim3(LOSS_THREADS), 0, as_stream(stream), A);
    return hipGetLastError();
}

PIML_API int piml_rollout_losses_frames(const float* p, const float* labels, long long labels_ld, const long long* mask_pred,
                                        const unsigned char* gates, const float* const* count_frames, int focus,
                                        const float* abnormal_mask, int C, int T, int N, float time_decay, float w_coll,
                                        float w_hard, float* out, float* g_mse, float* g_coll, float* g_hard, float* partial,
                                        unsigned* ticket, void* stream) {
    if (!p || !labels || !mask_pred || !gates || !count_frames || !out || !g_mse || !g_coll || !g_hard || C < 1 || T < 1 || T > 32 ||
        N < 1 || labels_ld < 2)
        return hipErrorInvalidValue;
    const int blocks = piml_rollout_losses_blocks(C, N);
    if (blocks > 1 && (!partial || !ticket)) return hipErrorInvalidValue;
    LossArgs A = {};
    A.p = p; A.lab = labels; A.ld = labels_ld; A.keep = mask_pred; A.gate = gates; A.abn = abnormal_mask; A.C = C; A.T = T; A.N = N;
    A.time_decay = time_decay; A.out = out; A.g_mse = g_mse; A.g_coll = g_coll; A.g_hard = g_hard; A.partial = partial;
    A.ticket = ticket;
    for (int t = 0; t < T; ++t) A.frames[t] = count_frames[t];
    A.use_frames = 1; A.focus = focus; A.stats = 1; A.w_coll = w_coll; A.w_hard = w_hard;
    hipLaunchKernelGGL(rollout_losses_kernel, dim3(blocks), dim3(LOSS_THREADS), 0, as_stream(stream), A);
    return hipGetLastError();
}

PIML_API int piml_rollout_losses_bwd(const float* g_out0, const float* g_out1, const float* g_out2, const float* g_mse,
                                     const float* g_coll, const float* g_hard, long long n, float* g_p, void* stream) {
    if (!g_mse || !g_coll || !g_hard || !g_p || n < 1) return hipErrorInvalidValue;
    long long b = (n + LOSS_THREADS - 1) / LOSS_THREADS;
    hipLaunchKernelGGL(rollout_losses_bwd_kernel, dim3((unsigned)(b > 1024 ? 1024 : b)), dim3(LOSS_THREADS), 0, as_stream(stream),
                       g_out0, g_out1, g_out2, g_mse, g_coll, g_hard, n, g_p, (const float*)nullptr, 1.f, 1.f);
    return hipGetLastError();
}

PIML_API int piml_rollout_losses_frames_bwd(const float* g_mse_out, const float* g_collw_out, const float* g_hardw_out,
                                            const float* g_total_out, float w_coll, float w_hard, const float* g_mse,
                                            const float* g_coll, const float* g_hard, long long n, float* g_p, void* stream) {
    // (the weighted form is selected by a non-NULL total gradient: a caller without one passes a device zero)
    if (!g_mse || !g_coll || !g_hard || !g_p || !g_total_out || n < 1) return hipErrorInvalidValue;
    long long b = (n + LOSS_THREADS - 1) / LOSS_THREADS;
    hipLaunchKernelGGL(rollout_losses_bwd_kernel, dim3((unsigned)(b > 1024 ? 1024 : b)), dim3(LOSS_THREADS), 0, as_stream(stream),
                       g_mse_out, g_collw_out, g_hardw_out, g_mse, g_coll, g_hard, n, g_p, g_total_out, w_coll, w_hard);
    return hipGetLastError();
}

// ---- one launch for a list of device-to-device copies (the batch of a training step into the static inputs of its captured
// graph: fifteen tensors of a few KB each were fifteen copy launches of ~2.6 us, src/models/simulators.py:699-779's
// `data` fields) ----
namespace piml {
constexpr int kMultiCopyMax = 24;
struct MultiCopy {
    void* dst[kMultiCopyMax];
    const void* src[kMultiCopyMax];
    unsigned long long bytes[kMultiCopyMax];
    int n;
};
__global__ __launch_bounds__(256) void multi_copy_kernel(MultiCopy M) {
    const int e = blockIdx.y;
    const unsigned long long nb = M.bytes[e];
    char* d = static_cast<char*>(M.dst[e]);
    const char* s = static_cast<const char*>(M.src[e]);
    const unsigned long long t0 = (unsigned long long)blockIdx.x * 256 + threadIdx.x, stride = (unsigned long long)gridDim.x * 256;
    if (((reinterpret_cast<unsigned long long>(d) | reinterpret_cast<unsigned long long>(s)) & 15ull) == 0) {
        const unsigned long long n16 = nb >> 4;
        for (unsigned long long i = t0; i < n16; i += stride) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        for (unsigned long long i = (n16 << 4) + t0; i < nb; i += stride) d[i] = s[i];
    } else {
        for (unsigned long long i = t0; i < nb; i += stride) d[i] = s[i];
    }
}
}  // namespace piml

PIML_API int piml_multi_copy(void* const* dst, const void* const* src, const size_t* bytes, int n, void* stream) {
    if (n < 0 || (n > 0 && (!dst || !src || !bytes))) return hipErrorInvalidValue;
    for (int base = 0; base < n; base += piml::kMultiCopyMax) {
        piml::MultiCopy M = {};
        size_t most = 0;
        M.n = n - base < piml::kMultiCopyMax ? n - base : piml::kMultiCopyMax;
        for (int i = 0; i < M.n; ++i) {
            if (bytes[base + i] && (!dst[base + i] || !src[base + i])) return hipErrorInvalidValue;
            M.dst[i] = dst[base + i]; M.src[i] = src[base + i]; M.bytes[i] = bytes[base + i];
            if (bytes[base + i] > most) most = bytes[base + i];
        }
        if (most == 0) continue;
        size_t gx = (most / 16 + 255) / 256;
        gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
        hipLaunchKernelGGL(piml::multi_copy_kernel, dim3((unsigned)gx, (unsigned)M.n), dim3(256), 0, piml::as_stream(stream), M);
    }
    return hipGetLastError();
}

// ---- the prologue of the differentiable training rollout in ONE launch (src/models/simulators.py:672-697 of
// test_multiple_rollouts_for_training: the clones of frame t_start, `new_peds_flag = (mask_p - mask_p_pred).long() == 1`,
// `mask_p_pred.long()`, the per-frame gates `torch.sum(mask_p_pred[:, t]) > 0` of :707, the desired speeds) -- on torch
// operators fourteen launches of a few KB each in front of every fine-tuning step.  One workgroup: the arrays are (C, T, N)
// with C T N of a few thousand. ----
namespace piml {
struct PrologueArgs {
    const float *position, *velocity, *acceleration, *destination;      // (C, T, N, 2)
    const long long* dest_idx;                                            // (C, T, N)
    const float *mask_p, *mask_p_pred;                                    // (C, T, N)
    const float* self_features;                                           // (C, T, N, 7)
    int C, T, N, t_start;
    float *p0, *v0, *a0, *d0;                                             // (C, N, 2)
    long long* di0;                                                       // (C, N)
    unsigned char* new_flag;                                              // (C, T, N)
    long long* mask_pred;                                                 // (C, T, N)
    unsigned char* gates;                                                 // (T)
    float* gates_f;                                                       // (T)
    float* speed;                                                         // (C, N)
    int* nan_flag;
};
__global__ __launch_bounds__(1024) void rollout_prologue_kernel(PrologueArgs A) {
    __shared__ long long red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long CN = (long)A.C * A.N, CTN = CN * A.T;
    for (long e = tid; e < CTN; e += 1024) {
        const float mp = A.mask_p[e], mq = A.mask_p_pred[e];
        A.new_flag[e] = (long long)(mp - mq) == 1 ? 1 : 0;
        A.mask_pred[e] = (long long)mq;
    }
    for (long e = tid; e < CN; e += 1024) {
        const long c = e / A.N, n = e - c * A.N;
        const long src = (c * A.T + A.t_start) * A.N + n;
        reinterpret_cast<float2*>(A.p0)[e] = reinterpret_cast<const float2*>(A.position)[src];
        reinterpret_cast<float2*>(A.v0)[e] = reinterpret_cast<const float2*>(A.velocity)[src];
        reinterpret_cast<float2*>(A.a0)[e] = reinterpret_cast<const float2*>(A.acceleration)[src];
        reinterpret_cast<float2*>(A.d0)[e] = reinterpret_cast<const float2*>(A.destination)[src];
        A.di0[e] = A.dest_idx[src];
        A.speed[e] = A.self_features[src * 7 + 6];
    }
    if (tid == 0) *A.nan_flag = 0;
    for (int t = 0; t < A.T; ++t) {                       // gate of frame t: sum over (c, n) of long(mask_p_pred) > 0
        long long sum = 0;
        for (long e = tid; e < CN; e += 1024) {
            const long c = e / A.N, n = e - c * A.N;
            sum += (long long)A.mask_p_pred[(c * A.T + t) * A.N + n];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)sum, o, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)((unsigned long long)sum >> 32), o, 64);
            sum += (long long)(((unsigned long long)hi << 32) | lo);
        }
        __syncthreads();
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        if (tid == 0) {
            long long tot = 0;
            for (int w = 0; w < 16; ++w) tot += red[w];
            A.gates[t] = tot > 0 ? 1 : 0;
            A.gates_f[t] = tot > 0 ? 1.f : 0.f;
        }
    }
}
}  // namespace piml

PIML_API int piml_rollout_prologue(const float* position, const float* velocity, const float* acceleration, const float* destination,
                                   const long long* dest_idx, const float* mask_p, const float* mask_p_pred,
                                   const float* self_features, int C, int T, int N, int t_start, float* p0, float* v0, float* a0,
                                   float* d0, long long* di0, unsigned char* new_flag, long long* mask_pred, unsigned char* gates,
                                   float* gates_f, float* speed, int* nan_flag, void* stream) {
    if (C < 0 || T < 1 || N < 0 || t_start < 0 || t_start >= T) return hipErrorInvalidValue;
    if ((long)C * N == 0) return hipSuccess;
    if (!position || !velocity || !acceleration || !destination || !dest_idx || !mask_p || !mask_p_pred || !self_features || !p0 ||
        !v0 || !a0 || !d0 || !di0 || !new_flag || !mask_pred || !gates || !gates_f || !speed || !nan_flag)
        return hipErrorInvalidValue;
    piml::PrologueArgs A = {position, velocity, acceleration, destination, dest_idx, mask_p, mask_p_pred, self_features, C, T, N, t_start,
                            p0, v0, a0, d0, di0, new_flag, mask_pred, gates, gates_f, speed, nan_flag};
    hipLaunchKernelGGL(piml::rollout_prologue_kernel, dim3(1), dim3(1024), 0, piml::as_stream(stream), A);
    return hipGetLastError();
}

