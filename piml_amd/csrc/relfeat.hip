// Relative-feature kernels (top-k in-view neighbours of every focal agent) for gfx950.
//
// Replaces Pedestrians.get_relative_features (reference src/data/data.py:466-512), which
// materialises (N,N,6)/(N,M,6) tensors and fully sorts every row.  Here one 64-lane
// wavefront owns one focal agent and streams all source positions through an LDS tile:
//
//   phase 1 (every pair, ~7 VALU ops / 64 pairs): d2 = fma(dy,dy,dx*dx) against a running
//            cut-off; survivors are compacted (ballot + mbcnt) into a per-wave LDS ring;
//   phase 2 (survivors only, 64 at a time): the exact float32 arithmetic PyTorch's CPU
//            kernels use for the distance and the view-cone cosine, so the neighbour sets
//            are bit-identical to the reference's;
//   phase 3 insertion into a sorted top-k list held one entry per lane (key =
//            distance bits << 32 | source index), which also tightens the phase-1 cut-off
//            to the current k-th distance.
//
// No N x N intermediate exists; v / a are touched only for the k selected neighbours.
#include "common.hpp"
#include "../../include/piml_hip.h"

#include <cmath>

namespace piml {

constexpr int kTile = 4096;        // source points per LDS tile (32 KiB)
constexpr int kRing = 128;         // per-wave candidate ring (entries), power of two

struct RelfeatArgs {
    const float2* p; const float2* hd; const float2* v; const float2* a; const float2* dest;
    const float2* obs;
    int C, N, M, f0, fcnt, kp, ko;
    float cos_p, cos_o, cut2_p, cut2_o, dthr_p, dthr_o;
    float* ped_feat; float* obs_feat; float2* dest_feat; int* ped_idx; int* obs_idx;
};

// Insert `nk` (known to be < the current k-th key) into the ascending list held one key per
// lane.  Lanes >= k carry don't-care values.
__device__ __forceinline__ u64 list_insert(u64 list, u64 nk, int lane) {
    const u64 up = shift_up1(list);
    const int pos = __ffsll((long long)__ballot(list > nk)) - 1;   // first lane whose key is larger
    return lane > pos ? up : (lane == pos ? nk : list);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void relfeat_fwd_kernel(const RelfeatArgs A) {
    __shared__ float2 tile[kTile];
    __shared__ unsigned short ring_all[WAVES][kRing];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    volatile unsigned short* ring = ring_all[wave];

    const int bpc = (A.fcnt + WAVES - 1) / WAVES;          // blocks per slice
    const int c = blockIdx.x / bpc;
    const int fl = (blockIdx.x - c * bpc) * WAVES + wave;   // focal row of this wave
    const bool has = fl < A.fcnt;
    const int i = A.f0 + (has ? fl : 0);
    const size_t ci = (size_t)c * A.N + i;

    // focal state: identical in every lane -> scalar registers
    const float2 pi2 = A.p[ci];
    const float pix = uniform(pi2.x), piy = uniform(pi2.y);
    const float2 vi2 = A.v[ci], ai2 = A.a[ci];
    const float vix = uniform(nan_to_zero(vi2.x)), viy = uniform(nan_to_zero(vi2.y));
    const float aix = uniform(nan_to_zero(ai2.x)), aiy = uniform(nan_to_zero(ai2.y));
    const bool alive = has && pix == pix && piy == piy;     // NaN focal: every distance is inf

    // heading (data.py:391-394), then cosine_similarity's own re-normalisation of it
    float hx, hy;
    if (A.hd) {
        const float2 h = A.hd[ci];
        hx = uniform(h.x); hy = uniform(h.y);
    } else {
        float hn = norm2(vix, viy);
        if (hn == 0.f) hn = hn + 0.1f;
        hx = __fdiv_rn(vix, hn); hy = __fdiv_rn(viy, hn);
    }
    const float n2c = fmaxf(norm2(hx, hy), 1e-8f);
    const float h0 = __fdiv_rn(hx, n2c), h1 = __fdiv_rn(hy, n2c);

    u64 lists[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float2* __restrict__ src = pass == 0 ? A.p + (size_t)c * A.N : A.obs;
        const int cnt = pass == 0 ? A.N : A.M;
        const int k = pass == 0 ? A.kp : A.ko;
        const float cos_thr = pass == 0 ? A.cos_p : A.cos_o;
        const float dthr = pass == 0 ? A.dthr_p : A.dthr_o;
        float cut2 = pass == 0 ? A.cut2_p : A.cut2_o;

        u64 list = kEmptyKey, kth = kEmptyKey;
        unsigned head = 0, tail = 0;                       // wave-uniform ring cursors

        for (int base = 0; base < cnt; base += kTile) {
            const int tn = min(kTile, cnt - base);
            __syncthreads();                                // previous tile fully consumed
            for (int t = threadIdx.x; t < tn; t += WAVES * 64) tile[t] = src[base + t];
            __syncthreads();
            if (!alive || k <= 0) continue;

            // exact evaluation of `n` buffered candidates (lane l takes ring[head + l])
            auto eval_chunk = [&](unsigned n) {
                const bool act = (unsigned)lane < n;
                const int jl = act ? (int)ring[(head + lane) & (kRing - 1)] : 0;
                head += n;
                const float2 q = tile[jl];
                const float rx = q.x - pix, ry = q.y - piy;
                const float d = norm2(rx, ry);                         // data.py:434
                const float cs = cos_sim_prenorm(rx, ry, d, h0, h1);   // :439-440
                const bool ok = act && cs >= cos_thr && d <= dthr;     // :441-443, :461
                const u64 key = ok ? (((u64)__float_as_uint(d) << 32) | (unsigned)(base + jl)) : kEmptyKey;
                u64 better = __ballot(key < kth);
                while (better) {
                    const int s = __ffsll((long long)better) - 1;
                    better &= better - 1;
                    const u64 nk = readlane64(key, s);
                    if (nk < kth) {
                        list = list_insert(list, nk, lane);
                        kth = readlane64(list, k - 1);
                    }
                }
                if (kth != kEmptyKey) {
                    // any source that can still enter the list has dist <= d_k, hence
                    // d2 <= d_k^2 (1 + 2^-20) whatever the rounding of sqrt and the product
                    const float dk = __uint_as_float((unsigned)(kth >> 32));
                    cut2 = fminf(cut2, dk * dk * 1.00000095367431640625f);
                }
            };

            for (int j0 = 0; j0 < tn; j0 += 64) {
                const int jl = j0 + lane;
                const float2 q = tile[min(jl, tn - 1)];
                const float rx = q.x - pix, ry = q.y - piy;
                const float d2 = sq2(rx, ry);
                const bool cand = jl < tn && d2 <= cut2;              // NaN / inf never pass
                const u64 m = __ballot(cand);
                if (m) {
                    if (cand) ring[(tail + mbcnt(m)) & (kRing - 1)] = (unsigned short)jl;
                    tail += (unsigned)__popcll(m);
                    if (tail - head >= 64u) eval_chunk(64u);
                }
            }
            if (tail != head) eval_chunk(tail - head);       // ring holds tile-local indices
        }
        lists[pass] = list;
    }

    if (!has) return;

    // ---- epilogue: gather the k selected sources, write features / indices ----
    const int kpe = min(A.kp, A.N), koe = min(A.ko, A.M);
    const size_t row = (size_t)c * A.fcnt + fl;
    if (lane < kpe) {
        const u64 key = lists[0];
        const int j = key == kEmptyKey ? -1 : (int)(unsigned)key;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f, f5 = 0.f;
        if (j >= 0) {
            const size_t cj = (size_t)c * A.N + j;
            const float2 pj = A.p[cj], vj = A.v[cj], aj = A.a[cj];
            f0 = pj.x - pix; f1 = pj.y - piy;                           // data.py:491-492
            f2 = nan_to_zero(vj.x) - vix; f3 = nan_to_zero(vj.y) - viy;
            f4 = nan_to_zero(aj.x) - aix; f5 = nan_to_zero(aj.y) - aiy;
        }
        float2* out = reinterpret_cast<float2*>(A.ped_feat + (row * kpe + lane) * 6);
        out[0] = make_float2(f0, f1); out[1] = make_float2(f2, f3); out[2] = make_float2(f4, f5);
        A.ped_idx[row * kpe + lane] = j;
    }
    if (lane < koe) {
        const u64 key = lists[1];
        const int j = key == kEmptyKey ? -1 : (int)(unsigned)key;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f, f5 = 0.f;
        if (j >= 0) {
            const float2 oj = A.obs[j];
            f0 = oj.x - pix; f1 = oj.y - piy;                           // data.py:506-508
            f2 = 0.f - vix; f3 = 0.f - viy; f4 = 0.f - aix; f5 = 0.f - aiy;
        }
        float2* out = reinterpret_cast<float2*>(A.obs_feat + (row * koe + lane) * 6);
        out[0] = make_float2(f0, f1); out[1] = make_float2(f2, f3); out[2] = make_float2(f4, f5);
        A.obs_idx[row * koe + lane] = j;
    }
    if (lane == 0) {
        const float2 d = A.dest[ci];
        A.dest_feat[row] = make_float2(nan_to_zero(d.x - pix), nan_to_zero(d.y - piy));   // :496-497
    }
}

// One thread per focal row.  The scatter into the selected sources uses float atomics
// (<= (kp+1)*6 per row, a few hundred KB in total); the caller zeroes g_state first.
__global__ void relfeat_bwd_kernel(const float* __restrict__ g_ped, const float* __restrict__ g_obs,
                                   const float2* __restrict__ g_destf, const int* __restrict__ ped_idx,
                                   const int* __restrict__ obs_idx, const float2* __restrict__ p,
                                   const float2* __restrict__ dest, int C, int N, int f0, int fcnt,
                                   int kpe, int koe, float* g_state, float2* g_dest) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C * fcnt) return;
    const int c = t / fcnt, fl = t - c * fcnt;
    const size_t row = (size_t)t;
    const size_t ci = (size_t)c * N + f0 + fl;
    float own[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < kpe; ++s) {
        const int j = ped_idx[row * kpe + s];
        if (j < 0) continue;
        const float* g = g_ped + (row * kpe + s) * 6;
        float* dst = g_state + ((size_t)c * N + j) * 6;
#pragma unroll
        for (int q = 0; q < 6; ++q) { const float gv = g[q]; atomicAdd(dst + q, gv); own[q] -= gv; }
    }
    for (int s = 0; s < koe; ++s) {
        if (obs_idx[row * koe + s] < 0) continue;
        const float* g = g_obs + (row * koe + s) * 6;
#pragma unroll
        for (int q = 0; q < 6; ++q) own[q] -= g[q];
    }
    const float2 pp = p[ci], dd = dest[ci], gd = g_destf[row];
    const float dx = dd.x - pp.x, dy = dd.y - pp.y;
    const float gx = dx != dx ? 0.f : gd.x, gy = dy != dy ? 0.f : gd.y;
    g_dest[row] = make_float2(gx, gy);
    own[0] -= gx; own[1] -= gy;
    float* dst = g_state + ci * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) atomicAdd(dst + q, own[q]);
}

// One thread per (slice, agent): two sweeps over time (data.py:363-389), then normalise.
__global__ void heading_kernel(const float2* __restrict__ vel, int C, int T, int N, float2* __restrict__ out) {
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 >= C * N) return;
    const int c = t0 / N, i = t0 - c * N;
    const size_t base = (size_t)c * T * N + i;
    float tx = 0.f, ty = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        float2 h = vel[base + (size_t)t * N];
        if (norm2(h.x, h.y) == 0.f) { h.x = tx; h.y = ty; } else { tx = h.x; ty = h.y; }
        out[base + (size_t)t * N] = h;
    }
    for (int t = 0; t < T; ++t) {
        float2 h = out[base + (size_t)t * N];
        if (norm2(h.x, h.y) == 0.f) { h.x = tx; h.y = ty; } else { tx = h.x; ty = h.y; }
        float n = norm2(h.x, h.y);
        if (n == 0.f) n = n + 0.1f;
        out[base + (size_t)t * N] = make_float2(__fdiv_rn(h.x, n), __fdiv_rn(h.y, n));
    }
}

// Largest float x with sqrtf(x) <= thr: "dist > thr" is then exactly "d2 > x" for the
// correctly rounded sqrt both sides use.
static float dist2_cutoff(float thr) {
    if (!(thr >= 0.f)) return -1.f;
    if (std::isinf(thr)) return INFINITY;
    float x = thr * thr;
    while (sqrtf(x) > thr) x = nextafterf(x, -INFINITY);
    while (sqrtf(nextafterf(x, INFINITY)) <= thr) x = nextafterf(x, INFINITY);
    return x;
}

}  // namespace piml

using namespace piml;

PIML_API int piml_relfeat_fwd(const float* position, const float* heading, const float* velocity,
                              const float* acceleration, const float* destination,
                              const float* obstacles, int C, int N, int M, int focal_begin,
                              int focal_count, int topk_ped, int topk_obs, float cos_thr_ped,
                              float cos_thr_obs, float dist_thr_ped, float dist_thr_obs,
                              float* ped_feat, float* obs_feat, float* dest_feat,
                              int32_t* ped_idx, int32_t* obs_idx, void* stream) {
    if (C < 0 || N < 0 || M < 0 || focal_begin < 0 || focal_count < 0 || focal_begin + focal_count > N ||
        topk_ped < 0 || topk_obs < 0 || topk_ped > PIML_MAX_TOPK || topk_obs > PIML_MAX_TOPK)
        return hipErrorInvalidValue;
    if (C == 0 || focal_count == 0) return hipSuccess;
    if (!position || !velocity || !acceleration || !destination || !dest_feat || (M > 0 && !obstacles))
        return hipErrorInvalidValue;
    RelfeatArgs A;
    A.p = (const float2*)position; A.hd = (const float2*)heading; A.v = (const float2*)velocity;
    A.a = (const float2*)acceleration; A.dest = (const float2*)destination; A.obs = (const float2*)obstacles;
    A.C = C; A.N = N; A.M = M; A.f0 = focal_begin; A.fcnt = focal_count;
    A.kp = topk_ped < N ? topk_ped : N; A.ko = topk_obs < M ? topk_obs : M;
    A.cos_p = cos_thr_ped; A.cos_o = cos_thr_obs;
    A.cut2_p = dist2_cutoff(dist_thr_ped); A.cut2_o = dist2_cutoff(dist_thr_obs);
    A.dthr_p = dist_thr_ped; A.dthr_o = dist_thr_obs;
    A.ped_feat = ped_feat; A.obs_feat = obs_feat; A.dest_feat = (float2*)dest_feat;
    A.ped_idx = ped_idx; A.obs_idx = obs_idx;
    const long rows = (long)C * focal_count;
    // 16 waves (one workgroup per CU at 4 waves/SIMD) once the launch fills the 256 CUs,
    // 4-wave workgroups for small scenes so the rows spread over more CUs.
    if (rows >= 16 * 256) {
        const int bpc = (focal_count + 15) / 16;
        hipLaunchKernelGGL(relfeat_fwd_kernel<16>, dim3(C * bpc), dim3(1024), 0, as_stream(stream), A);
    } else {
        const int bpc = (focal_count + 3) / 4;
        hipLaunchKernelGGL(relfeat_fwd_kernel<4>, dim3(C * bpc), dim3(256), 0, as_stream(stream), A);
    }
    return hipGetLastError();
}

PIML_API int piml_relfeat_bwd(const float* g_ped_feat, const float* g_obs_feat, const float* g_dest_feat,
                              const int32_t* ped_idx, const int32_t* obs_idx, const float* position,
                              const float* destination, int C, int N, int focal_begin, int focal_count,
                              int kp_eff, int ko_eff, float* g_state, float* g_destination, void* stream) {
    if (C < 0 || N < 0 || focal_begin < 0 || focal_count < 0 || focal_begin + focal_count > N ||
        kp_eff < 0 || ko_eff < 0)
        return hipErrorInvalidValue;
    if (C == 0 || focal_count == 0) return hipSuccess;
    if (!g_dest_feat || !position || !destination || !g_state || !g_destination) return hipErrorInvalidValue;
    const long rows = (long)C * focal_count;
    hipLaunchKernelGGL(relfeat_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream),
                       g_ped_feat, g_obs_feat, (const float2*)g_dest_feat, ped_idx, obs_idx,
                       (const float2*)position, (const float2*)destination, C, N, focal_begin, focal_count,
                       kp_eff, ko_eff, g_state, (float2*)g_destination);
    return hipGetLastError();
}

PIML_API int piml_heading_fwd(const float* velocity, int C, int T, int N, float* heading, void* stream) {
    if (C < 0 || T < 0 || N < 0) return hipErrorInvalidValue;
    if ((long)C * T * N == 0) return hipSuccess;
    if (!velocity || !heading) return hipErrorInvalidValue;
    const long n = (long)C * N;
    hipLaunchKernelGGL(heading_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, as_stream(stream),
                       (const float2*)velocity, C, T, N, (float2*)heading);
    return hipGetLastError();
}

// Diagnostic: the exact distance / cosine arithmetic of the selection predicates, exposed so
// a test can pin it bit-for-bit against the CPU restatement on millions of pairs.
__global__ void probe_arith_kernel(const float* rx, const float* ry, const float* hx, const float* hy,
                                   float* dist, float* cosv, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float n2c = fmaxf(norm2(hx[t], hy[t]), 1e-8f);
    const float h0 = __fdiv_rn(hx[t], n2c), h1 = __fdiv_rn(hy[t], n2c);
    const float d = norm2(rx[t], ry[t]);
    dist[t] = d;
    cosv[t] = cos_sim_prenorm(rx[t], ry[t], d, h0, h1);
}

PIML_API int piml_probe_arith(const float* rx, const float* ry, const float* hx, const float* hy,
                              float* dist, float* cosv, int n, void* stream) {
    if (n <= 0) return n < 0 ? hipErrorInvalidValue : hipSuccess;
    hipLaunchKernelGGL(probe_arith_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream),
                       rx, ry, hx, hy, dist, cosv, n);
    return hipGetLastError();
}

PIML_API int piml_abi_version(void) { return PIML_HIP_ABI_VERSION; }

PIML_API const char* piml_error_string(int err) { return hipGetErrorString((hipError_t)err); }
