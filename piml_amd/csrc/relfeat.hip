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
//   phase 3 each chunk is drained smallest-distance-first (DPP row reductions) into a sorted
//            top-k list held one entry per lane, at most k insertions per chunk; the k-th
//            distance tightens the phase-1 cut-off.
//
// No N x N intermediate exists; v / a are touched only for the k selected neighbours.
#include "common.hpp"
#include "trace.hpp"
#include "reduce.hpp"
#include "../../include/piml_hip.h"

#include <cmath>
#include <cstdlib>

namespace piml {

constexpr int kTile = 4096;        // source points per LDS tile (32 KiB)
constexpr int kRing = 1024;        // per-wave candidate ring (entries), power of two; a 512-group
                                   // can append up to 512 candidates on top of < 64 pending ones

struct RelfeatArgs {
    const float* p; const float* v; const float* a;   // per-agent records, `ld` floats apart
    const float2* hd; const float2* dest; const float2* obs;
    int ld;
    int C, N, M, f0, fcnt, kp, ko;
    float cos_p, cos_o, cut2_p, cut2_o, dthr_p, dthr_o;
    float* ped_feat; float* obs_feat; float* dest_feat; int dest_ld; int* ped_idx; int* obs_idx;
    // agent-block sharding runs the launch in two parts so that the first needs no remote data (DESIGN.md "exchange"):
    //   part LOCAL   agent sources [a_lo[0], a_hi[0]) = the rank's own block, the obstacle pass, dest / self features;
    //                the pedestrian list found so far goes to ped_idx (no features yet)
    //   part REMOTE  agent sources = the two ranges either side of the block, the list restored from ped_idx
    //                (distances recomputed with the same arithmetic), then the pedestrian features
    // The selection is the k smallest by (distance, index), whatever order the sources arrive in: both parts together
    // are bit-identical to one launch over [0, N).
    int a_lo[2], a_hi[2];   // agent-pass source ranges (the second may be empty)
    int flags;              // kRfPedFeat | kRfObs | kRfDest | kRfPedList | kRfInit
    int split;              // 1: the launch has twice the workgroups -- the first half runs the pedestrian pass (+ dest / self features) of its rows,
                            // the second half the obstacle pass of the same rows (relfeat_fwd_kernel<W, false> only)
    const float* speed;   // non-NULL: dest_feat rows are the model's self_features rows [dest - p, v, a, v0] (dest_ld >= 7)
    float* zero; long zero_n;   // optional: buffer this launch clears (the state gradient its backward accumulates into)
    long long* tick;            // optional: device-side frame counter this launch advances by one (it does not read it)
    int* stats;   // PIML_RELFEAT_STATS builds only: per focal row {evals, drain rounds, insertions, candidates,
                  // 7 phase stamps (cycles): entry, [tile staged, pass done] per pass ..., + 1 pad}
};

constexpr int kRfPedFeat = 1;    // write pedestrian features + indices
constexpr int kRfObs = 2;        // obstacle pass + its features / indices
constexpr int kRfDest = 4;       // destination (self) features
constexpr int kRfPedList = 8;    // write the pedestrian index list only
constexpr int kRfInit = 16;      // start the pedestrian list from ped_idx

// ---- sorted top-k list, one entry per lane: (distance bits, source index) ----
// Entries ascend by (distance, index); exact distance ties resolve to the lower index.
constexpr unsigned kEmptyDist = 0xffffffffu;   // > bits of every finite non-negative float

template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, false);
}
// minimum over the 64 lanes, returned wave-uniform.  Round 4: six DPP minima (quad_perm x 2, row_half_mirror, row_mirror, then
// row_bcast:15 / row_bcast:31 carry the row minima into lane 63) and ONE readlane, written as asm -- from the builtins hipcc
// made v_mov + s_nop + v_mov_dpp + v_min per step and four readlanes + a v_min3 behind them (19 vector instructions; a
// drain round was 45 of them, fifteen rounds per focal row 41 % of the kernel's vector work).  Two wait states between a
// vector write and a DPP read of the same register (s_nop 1).
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
    unsigned r;
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(r)
        : "v"(x));
    return (unsigned)__builtin_amdgcn_readlane((int)r, 63);
}
// Insert the wave-uniform entry (nd, ni), which precedes the current k-th entry.  The lower
// neighbour's entry arrives by a DPP wave_shr:1 move (lane 0 reads (0, 0), which never
// follows the new entry, so lane 0 can only be the receiving slot).  (distance, index) compared as ONE 64-bit key.
__device__ __forceinline__ void list_insert(unsigned& ld, unsigned& li, unsigned nd, unsigned ni) {
    const unsigned up_d = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ld, 0x138, 0xf, 0xf, true);
    const unsigned up_i = (unsigned)__builtin_amdgcn_update_dpp(0, (int)li, 0x138, 0xf, 0xf, true);
    const u64 nk = ((u64)nd << 32) | ni;
    const bool moves = (((u64)ld << 32) | li) > nk;
    const bool below_moves = (((u64)up_d << 32) | up_i) > nk;
    ld = moves ? (below_moves ? up_d : nd) : ld;
    li = moves ? (below_moves ? up_i : ni) : li;
}

// ---- epilogue of a focal row: gather the k selected sources, write features / indices (lane s = slot s) ----
__device__ __forceinline__ void relfeat_epilogue(const RelfeatArgs& A, int flags, int c, int fl, int lane, const u64 (&lists)[2], float pix,
                                                 float piy, float vix, float viy, float aix, float aiy, float2 vi2, float2 ai2) {
    const int ld = A.ld;
    const int kpe = min(A.kp, A.N), koe = min(A.ko, A.M);
    const size_t row = (size_t)c * A.fcnt + fl;
    if (lane < kpe && (flags & kRfPedList)) {
        const u64 key = lists[0];
        A.ped_idx[row * kpe + lane] = key == kEmptyKey ? -1 : (int)(unsigned)key;
    }
    if (lane < kpe && (flags & kRfPedFeat)) {
        const u64 key = lists[0];
        const int j = key == kEmptyKey ? -1 : (int)(unsigned)key;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f, f5 = 0.f;
        if (j >= 0) {
            const size_t cj = ((size_t)c * A.N + j) * ld;
            const float2 pj = *reinterpret_cast<const float2*>(A.p + cj);
            const float2 vj = *reinterpret_cast<const float2*>(A.v + cj);
            const float2 aj = *reinterpret_cast<const float2*>(A.a + cj);
            f0 = pj.x - pix; f1 = pj.y - piy;                           // data.py:491-492
            f2 = nan_to_zero(vj.x) - vix; f3 = nan_to_zero(vj.y) - viy;
            f4 = nan_to_zero(aj.x) - aix; f5 = nan_to_zero(aj.y) - aiy;
        }
        float2* out = reinterpret_cast<float2*>(A.ped_feat + (row * kpe + lane) * 6);
        out[0] = make_float2(f0, f1); out[1] = make_float2(f2, f3); out[2] = make_float2(f4, f5);
        A.ped_idx[row * kpe + lane] = j;
    }
    if (lane < koe && (flags & kRfObs)) {
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
    if (lane == 0 && (flags & kRfDest)) {
        const float2 d = A.dest[row];
        float* df = A.dest_feat + row * A.dest_ld;          // row stride: 2, or the width of a self_features row
        df[0] = nan_to_zero(d.x - pix); df[1] = nan_to_zero(d.y - piy);                    // :496-497
        if (A.speed) {                                      // self_features = [dest - p, v, a, v0], raw v and a
            df[2] = vi2.x; df[3] = vi2.y; df[4] = ai2.x; df[5] = ai2.y;
            df[6] = A.speed[row];
        }
    }
}

constexpr int kObsTile = 2048;     // RES: obstacle points resident next to the agent tile (16 KiB)

// LDS (dynamic): agent tile x | y (kTile floats each), the waves' candidate rings, and -- RES -- the obstacle tile x | y.
template <int WAVES, bool RES>
constexpr int relfeat_lds_bytes() { return (2 * kTile + WAVES * kRing / 2 + (RES ? 2 * kObsTile : 0)) * 4; }

// RES ("both tiles resident"): when the agent sources fit one tile and the obstacle points fit kObsTile, both are staged up
// front and the only barrier of the launch is the one behind that staging.  Otherwise every pass re-stages the shared tile
// behind a workgroup barrier, i.e. all 16 waves wait for the slowest pedestrian pass before the obstacle pass can start:
// measured at the 4096-agent scene (in-kernel stamps, tools/relfeat_stats.py) a median wave spent 7.5 k of its 32.6 k cycles
// in that wait (per-row work is data dependent: 18 k median, 26 k max for the pedestrian pass).
template <int WAVES, bool RES>
__global__ __launch_bounds__(WAVES * 64) void relfeat_fwd_kernel(const RelfeatArgs A, const PackWork PK) {
    // a deferred weight pack (PIML_DEFER_PACK, reduce.hpp) rides as the launch's trailing workgroups: they reach a CU when
    // the short obstacle workgroups have left, under the tail of the pedestrian passes
    extern __shared__ __attribute__((aligned(16))) float rf_lds[];
    if (PK.first_block >= 0 && (int)blockIdx.x >= PK.first_block) {
        pack_block(PK.A, (int)blockIdx.x - PK.first_block, WAVES * 64, reinterpret_cast<double*>(rf_lds));
        return;
    }
    // source tiles, structure-of-arrays so that a lane fetches 4 consecutive points per
    // ds_read_b128 (x) + ds_read_b128 (y)
    float* const agent_x = rf_lds;
    float* const agent_y = rf_lds + kTile;
    unsigned short* const ring_base = reinterpret_cast<unsigned short*>(rf_lds + 2 * kTile);
    float* const obs_x = RES ? rf_lds + 2 * kTile + WAVES * kRing / 2 : agent_x;
    float* const obs_y = RES ? obs_x + kObsTile : agent_y;

    const int lane = threadIdx.x & 63;
    const int wave = uniform((int)(threadIdx.x >> 6));
    unsigned short* ring = ring_base + wave * kRing;        // stays in the LDS address space

    if (A.tick && blockIdx.x == 0 && threadIdx.x == 0) *A.tick += 1;
    if (A.zero)                                             // a few 100 KB, spread over the whole grid
        for (long e = (long)blockIdx.x * (WAVES * 64) + threadIdx.x; e < A.zero_n;
             e += (long)(PK.first_block >= 0 ? PK.first_block : (int)gridDim.x) * (WAVES * 64))
            A.zero[e] = 0.f;

    const int bpc = (A.fcnt + WAVES - 1) / WAVES;          // blocks per slice
    // split launch: workgroups [0, C * bpc) run the pedestrian pass of their rows, [C * bpc, 2 C * bpc) the obstacle pass of the
    // same rows -- a row's two passes are independent, and a launch of a few thousand rows is bound by the LATENCY of a row (a
    // chain of dependent drains), not by issue slots: side by side the chain is one pass long instead of two
    const int nb = A.C * bpc;
    const int kind = (!RES && A.split) ? ((int)blockIdx.x >= nb ? 2 : 1) : 0;
    const int bid = (int)blockIdx.x - (kind == 2 ? nb : 0);
    const int flags = kind == 1 ? (A.flags & ~kRfObs) : (kind == 2 ? kRfObs : A.flags);
    const int c = bid / bpc;
    const int fl = (bid - c * bpc) * WAVES + wave;          // focal row of this wave
    const bool has = fl < A.fcnt;
    const int i = A.f0 + (has ? fl : 0);
    const size_t ci = (size_t)c * A.N + i;

    // focal state: identical in every lane -> scalar registers
    const int ld = A.ld;
    const float2 pi2 = *reinterpret_cast<const float2*>(A.p + ci * ld);
    const float pix = uniform(pi2.x), piy = uniform(pi2.y);
    const float2 vi2 = *reinterpret_cast<const float2*>(A.v + ci * ld);
    const float2 ai2 = *reinterpret_cast<const float2*>(A.a + ci * ld);
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
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f pix2 = {pix, pix}, piy2 = {piy, piy};

#ifdef PIML_RELFEAT_STATS
    int st_evals = 0, st_rounds = 0, st_ins = 0, st_cand = 0;
    unsigned long long st_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int st_n = 0;
    st_t[st_n++] = __builtin_amdgcn_s_memtime();
#define PIML_STAT(x) x
#else
#define PIML_STAT(x)
#endif
    const float qnan = __uint_as_float(0x7fc00000u);
    // global -> LDS tile (x | y), rows past `tn` padded with NaN (never pass the cut-off)
    auto stage = [&](const float* __restrict__ src, int sld, int base, int tn, int tn_pad, float* tx, float* ty) {
        // (issuing all of a thread's loads before the first LDS write, as the encoder's weight staging does, was
        // measured here and is slower: 23.0 vs 22.4 us at cfg3, 10.2 vs 8.8 us at N = 122 -- the tile is small and
        // the extra registers / redundant clamped loads cost more than the serial round trips)
        for (int t = threadIdx.x; t < tn_pad; t += WAVES * 64) {
            float2 q = make_float2(qnan, qnan);
            if (t < tn) q = *reinterpret_cast<const float2*>(src + (size_t)(base + t) * sld);
            tx[t] = q.x; ty[t] = q.y;
        }
    };
    if (RES) {
        const int n0 = A.a_hi[0] - A.a_lo[0];
        stage(A.p + (size_t)c * A.N * ld, ld, A.a_lo[0], n0, (n0 + 511) & ~511, agent_x, agent_y);
        stage((const float*)A.obs, 2, 0, A.M, (A.M + 511) & ~511, obs_x, obs_y);
        __syncthreads();
        PIML_STAT(if (st_n < 7) st_t[st_n++] = __builtin_amdgcn_s_memtime();)
    }
    u64 lists[2] = {kEmptyKey, kEmptyKey};
    // Runtime loop and a single evaluation site: the exact-evaluation / drain code exists once
    // in the binary, so it stays resident in the instruction cache (inlining it at every
    // append site made each execution an instruction-fetch miss chain).
#pragma unroll 1
    for (int pass = kind == 2 ? 1 : 0; pass < (kind == 1 ? 1 : 2); ++pass) {
        const float* __restrict__ src = pass == 0 ? A.p + (size_t)c * A.N * ld : (const float*)A.obs;
        const int sld = pass == 0 ? ld : 2;
        const int cnt = pass == 0 ? A.N : A.M;
        const int k = pass == 0 ? A.kp : A.ko;
        const float cos_thr = pass == 0 ? A.cos_p : A.cos_o;
        const float dthr = pass == 0 ? A.dthr_p : A.dthr_o;
        float cut2 = pass == 0 ? A.cut2_p : A.cut2_o;      // wave-uniform

        if (pass == 1 && !(flags & kRfObs)) break;         // (uniform over the workgroup: no barrier is skipped unevenly)
        float* const tile_x = pass == 0 ? agent_x : obs_x;
        float* const tile_y = pass == 0 ? agent_y : obs_y;

        unsigned list_d = kEmptyDist, list_i = 0;          // lane s: s-th nearest in-view source so far
        unsigned kth_d = kEmptyDist, kth_i = 0;            // wave-uniform copy of lane k-1's entry
        unsigned head = 0, tail = 0;                       // wave-uniform ring cursors
        if (pass == 0 && (flags & kRfInit) && alive && k > 0) {
            // the list the LOCAL part left in ped_idx (sorted, -1 = empty slot); distances by the arithmetic of phase 2
            const int j = lane < k ? A.ped_idx[((size_t)c * A.fcnt + fl) * k + lane] : -1;
            if (j >= 0) {
                const float2 pj = *reinterpret_cast<const float2*>(src + (size_t)j * sld);
                list_d = __float_as_uint(norm2(pj.x - pix, pj.y - piy));
                list_i = (unsigned)j;
            }
            kth_d = (unsigned)__builtin_amdgcn_readlane((int)list_d, k - 1);
            kth_i = (unsigned)__builtin_amdgcn_readlane((int)list_i, k - 1);
            if (kth_d != kEmptyDist) {
                const float dk = __uint_as_float(kth_d);
                cut2 = fminf(cut2, dk * dk * 1.00000095367431640625f);
            }
        }

#pragma unroll 1
        for (int rg = 0; rg < 2; ++rg) {
        const int lo = pass == 0 ? A.a_lo[rg] : 0, hi = pass == 0 ? A.a_hi[rg] : (rg == 0 ? cnt : 0);
#pragma unroll 1
        for (int base = lo; base < hi; base += kTile) {
            const int tn = min(kTile, hi - base);
            const int tn_pad = (tn + 511) & ~511;          // phase 1 reads whole 512-groups
            if (!RES) {
                __syncthreads();                            // previous tile fully consumed
                stage(src, sld, base, tn, tn_pad, tile_x, tile_y);
                __syncthreads();
                PIML_STAT(if (st_n < 7) st_t[st_n++] = __builtin_amdgcn_s_memtime();)
            }
            if (!alive || k <= 0) continue;

            // one extra trip (j0 == tn_pad) appends nothing and flushes the ring, because the
            // ring holds tile-local indices
#pragma unroll 1
            for (int j0 = 0; j0 <= tn_pad; j0 += 512) {
                const bool flush = j0 == tn_pad;
                if (!flush) {
                    // (round 6, built bit-exact and dropped: the filter on the EXPANDED square s + (-2 px) x + (-2 py) y with s = x^2 + y^2
                    // staged as a third LDS array -- two packed FMAs per pair of points instead of four packed operations, threshold
                    // widened by the rounding bound 4e-6 (|p|^2 + cut2): 20.6 vs 20.8 us at the 4096-agent scene, 139 vs 115 at 16384
                    // (eight-wave workgroups lose their third resident workgroup to the 16 KB), 28.5 vs 29.7 for a rank's share.  Phase 1
                    // is ~6 of the launch's 20 us and trades vector instructions for LDS reads one for one.)
                    // phase 1: lane owns points j0 + 4*lane + {0..3} and j0 + 256 + 4*lane + {0..3}
                    // packed fp32 (v_pk_add/mul/fma): two points per instruction
                    // (round 4, measured and dropped: the next group's points fetched one trip ahead -- 16 more registers,
                    // 23.0 vs 21.8 us at the 4096-agent scene, 130 vs 115 at 16384: the SIMDs are short of issue slots, not
                    // waiting for the LDS)
                    float d2[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 x = *reinterpret_cast<const float4*>(&tile_x[j0 + h * 256 + 4 * lane]);
                        const float4 y = *reinterpret_cast<const float4*>(&tile_y[j0 + h * 256 + 4 * lane]);
                        const v2f rx0 = v2f{x.x, x.y} - pix2, ry0 = v2f{y.x, y.y} - piy2;
                        const v2f rx1 = v2f{x.z, x.w} - pix2, ry1 = v2f{y.z, y.w} - piy2;
                        const v2f q0 = __builtin_elementwise_fma(ry0, ry0, rx0 * rx0);
                        const v2f q1 = __builtin_elementwise_fma(ry1, ry1, rx1 * rx1);
                        d2[h * 4 + 0] = q0.x; d2[h * 4 + 1] = q0.y; d2[h * 4 + 2] = q1.x; d2[h * 4 + 3] = q1.y;
                    }
                    // one compare for all 8 (fminf ignores the NaNs of padding / absent agents)
                    const float dmin = fminf(fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3])),
                                             fminf(fminf(d2[4], d2[5]), fminf(d2[6], d2[7])));
                    if (__builtin_amdgcn_ballot_w64(dmin <= cut2) == 0) continue;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const u64 m = __builtin_amdgcn_ballot_w64(d2[u] <= cut2);
                        if (m == 0) continue;                  // the common case once the cut-off is tight
                        if (d2[u] <= cut2)
                            ring[mbcnt_add(m, tail) & (kRing - 1)] =
                                (unsigned short)(j0 + (u >> 2) * 256 + 4 * lane + (u & 3));
                        tail += (unsigned)__builtin_popcountll(m);
                    }
                }
                // phase 2 + 3: exact evaluation of buffered candidates, 64 per trip
#pragma unroll 1
                // (round 4, measured and dropped: the first evaluation of a pass already at 32 / 16 candidates, so that the cut-off
                // tightens sooner -- 21.3 / 21.7 vs 20.6 us at the 4096-agent scene: the extra evaluation costs more than the
                // appends it saves)
                while (tail - head >= (flush ? 1u : 64u)) {
                    const unsigned n = min(64u, tail - head);
                    PIML_STAT(++st_evals; st_cand += (int)n;)
                    const bool act = (unsigned)lane < n;
                    // the ring is written and read by different lanes of THIS wave only: LDS operations of a
                    // wave complete in order, the barrier below just pins the compiler's ordering
                    __builtin_amdgcn_wave_barrier();
                    const int jl = act ? (int)ring[(head + lane) & (kRing - 1)] : 0;
                    head = uniform((int)(head + n));
                    const float rx = tile_x[jl] - pix, ry = tile_y[jl] - piy;
                    const float d = norm2(rx, ry);                         // data.py:434
                    const float cs = cos_sim_prenorm(rx, ry, d, h0, h1);   // :439-440
                    const bool ok = act && cs >= cos_thr && d <= dthr;     // :441-443, :461
                    unsigned cd = ok ? __float_as_uint(d) : kEmptyDist;
                    // PIML_RELFEAT_LANE_DRAIN (round 4, measured and NOT the default): candidates that precede the current k-th
                    // entry enter the list in LANE order -- the list is the k smallest by (distance, index) of everything
                    // offered to it whatever the order of the offers, so the result is bit-identical (62 tests) -- for two
                    // readlanes + a scalar comparison + the insertion per round instead of a wave-wide minimum + tie walk +
                    // insertion.  But a chunk offered in lane order inserts entries that a later, nearer one pushes out again
                    // (k ln(64 / k) insertions for the first chunk instead of k): 25.4 vs 22.5 us at the 4096-agent scene,
                    // 39.6 vs 35.6 for a rank's 2048 x 16384 share of the 16384-agent scene.
#ifdef PIML_RELFEAT_LANE_DRAIN
                    for (u64 q = __builtin_amdgcn_ballot_w64(cd <= kth_d && cd != kEmptyDist); q; q &= q - 1) {
                        PIML_STAT(++st_rounds;)
                        const int s = __builtin_ctzll(q);
                        const unsigned nd = (unsigned)__builtin_amdgcn_readlane((int)cd, s);
                        const unsigned ni = (unsigned)(base + __builtin_amdgcn_readlane(jl, s));
                        if (nd > kth_d || (nd == kth_d && ni > kth_i)) continue;       // the k-th entry has moved on
                        PIML_STAT(++st_ins;)
                        list_insert(list_d, list_i, nd, ni);
                        kth_d = (unsigned)__builtin_amdgcn_readlane((int)list_d, k - 1);
                        kth_i = (unsigned)__builtin_amdgcn_readlane((int)list_i, k - 1);
                    }
#else
                    // drain the chunk in (distance, index) order: at most k entries can enter
#pragma unroll 1
                    for (;;) {
                        PIML_STAT(++st_rounds;)
                        const unsigned md = wave_min_u32(cd);
                        if (md > kth_d || md == kEmptyDist) break;
                        u64 tied = __builtin_amdgcn_ballot_w64(cd == md);
                        int s = __builtin_ctzll(tied);
                        int nj = __builtin_amdgcn_readlane(jl, s);
                        for (tied &= tied - 1; tied; tied &= tied - 1) {   // exact ties: lowest index first
                            const int s2 = __builtin_ctzll(tied);
                            const int j2 = __builtin_amdgcn_readlane(jl, s2);
                            if (j2 < nj) { nj = j2; s = s2; }
                        }
                        const unsigned ni = (unsigned)(base + nj);
                        if (md == kth_d && ni > kth_i) break;      // every remaining candidate follows the k-th
                        PIML_STAT(++st_ins;)
                        list_insert(list_d, list_i, md, ni);
                        kth_d = (unsigned)__builtin_amdgcn_readlane((int)list_d, k - 1);
                        kth_i = (unsigned)__builtin_amdgcn_readlane((int)list_i, k - 1);
                        if (lane == s) cd = kEmptyDist;
                    }
#endif
                    if (kth_d != kEmptyDist) {
                        // any source that can still enter the list has dist <= d_k, hence
                        // d2 <= d_k^2 (1 + 2^-20) whatever the rounding of sqrt and the product
                        const float dk = __uint_as_float(kth_d);
                        cut2 = fminf(cut2, dk * dk * 1.00000095367431640625f);
                    }
                }
            }
        }
        }
        PIML_STAT(if (st_n < 7) st_t[st_n++] = __builtin_amdgcn_s_memtime();)
        const u64 mine = list_d == kEmptyDist ? kEmptyKey : (u64)list_i;
        if (pass == 0) lists[0] = mine; else lists[1] = mine;
    }

    if (!has) return;
#ifdef PIML_RELFEAT_STATS
    if (A.stats && lane == 0) {
        int* o = A.stats + ((size_t)c * A.fcnt + fl + (kind == 2 ? (size_t)A.C * A.fcnt : 0)) * 12;   // (split: the obstacle workgroups' rows behind the others)
        o[0] = st_evals; o[1] = st_rounds; o[2] = st_ins; o[3] = st_cand;
        for (int q = 0; q < 7; ++q) o[4 + q] = (int)(st_t[q] - st_t[0]);   // cycles since kernel entry of this wave
    }
#endif

    relfeat_epilogue(A, flags, c, fl, lane, lists, pix, piy, vix, viy, aix, aiy, vi2, ai2);
}

// One wavefront per focal row; lane = slot * 8 + component (components 6, 7 idle), so a
// row's gradient block is read nearly coalesced and the sum over slots is three xor-shuffles.
// The scatter into the selected sources uses float atomics (a few hundred KB in total); the
// row's own term is added atomically too because other rows scatter into it concurrently.
__device__ __forceinline__ void relfeat_bwd_rows(
        int bid, const float* __restrict__ g_ped, const float* __restrict__ g_obs, const float2* __restrict__ g_destf,
        const int* __restrict__ ped_idx, const int* __restrict__ obs_idx, const float* __restrict__ p, int ld,
        const float2* __restrict__ dest, int C, int N, int f0, int fcnt, int kpe, int koe, float* g_state,
        float2* g_dest, int gld, float* __restrict__ g_speed) {
    // gld = 2: g_destf rows are d/d(dest_feat).  gld = 7: they are d/d(self_features) rows; their v / a columns join
    // the row's own term and column 6 is d/d(desired speed) (g_speed, may be NULL)
    const int lane = threadIdx.x & 63;
    const long row = (long)bid * 4 + (threadIdx.x >> 6);
    if (row >= (long)C * fcnt) return;
    const int c = (int)(row / fcnt), fl = (int)(row - (long)c * fcnt);
    const size_t ci = (size_t)c * N + f0 + fl;
    const int q = lane & 7, s0 = lane >> 3;
    float own = 0.f;
    // the row's own terms: requested up front too (they were three more round trips behind the neighbour sums)
    const bool head = s0 == 0 && q < 6;
    float d_q = 0.f, p_q = 0.f, gdf = 0.f;
    if (head) {
        if (q < 2) {
            d_q = reinterpret_cast<const float*>(dest)[row * 2 + q];
            p_q = p[ci * ld + q];
            gdf = reinterpret_cast<const float*>(g_destf)[row * gld + q];
        } else if (gld == 7) {
            gdf = reinterpret_cast<const float*>(g_destf)[row * 7 + q];
        }
    }
    if (q < 6) {
        // every index and gradient of the lane requested at once (k <= PIML_MAX_TOPK: TR trips of 8 slots at most) -- as loops
        // that looked at an index before they asked for its gradient a row was four dependent round trips to L2
        constexpr int TR = (PIML_MAX_TOPK + 7) / 8;
        int jp[TR], jo[TR];
        float gp[TR], go[TR];
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            const int s = s0 + 8 * t;
            const bool inp = s < kpe, ino = s < koe;
            jp[t] = inp ? ped_idx[row * kpe + s] : -1;
            gp[t] = inp ? g_ped[(row * kpe + s) * 6 + q] : 0.f;
            jo[t] = ino ? obs_idx[row * koe + s] : -1;
            go[t] = ino ? g_obs[(row * koe + s) * 6 + q] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < TR; ++t)
            if (jp[t] >= 0) {
                atomicAdd(g_state + ((size_t)c * N + jp[t]) * 6 + q, gp[t]);
                own -= gp[t];
            }
#pragma unroll
        for (int t = 0; t < TR; ++t)
            if (jo[t] >= 0) own -= go[t];
    }
    own += __shfl_xor(own, 8, 64);
    own += __shfl_xor(own, 16, 64);
    own += __shfl_xor(own, 32, 64);
    if (head) {
        if (q < 2) {
            const float dq = d_q - p_q;
            const float gd = dq != dq ? 0.f : gdf;
            reinterpret_cast<float*>(g_dest)[row * 2 + q] = gd;
            own -= gd;
        } else if (gld == 7) {
            own += gdf;
        }
        atomicAdd(g_state + ci * 6 + q, own);
    }
    if (gld == 7 && g_speed && lane == 6) g_speed[row] = reinterpret_cast<const float*>(g_destf)[row * 7 + 6];
}

__global__ __launch_bounds__(256) void relfeat_bwd_kernel(
        const float* __restrict__ g_ped, const float* __restrict__ g_obs, const float2* __restrict__ g_destf,
        const int* __restrict__ ped_idx, const int* __restrict__ obs_idx, const float* __restrict__ p, int ld,
        const float2* __restrict__ dest, int C, int N, int f0, int fcnt, int kpe, int koe, float* g_state,
        float2* g_dest, int gld, float* __restrict__ g_speed) {
    relfeat_bwd_rows((int)blockIdx.x, g_ped, g_obs, g_destf, ped_idx, obs_idx, p, ld, dest, C, N, f0, fcnt, kpe, koe, g_state, g_dest,
                     gld, g_speed);
}

// The same rows behind the DEFERRED slot sums of the network's backward pass (reduce.hpp): workgroups [0, nred) sum weight-
// gradient slots (bandwidth: they go first), the rest are the relfeat backward's -- two independent small kernels in one launch.
__global__ __launch_bounds__(256) void relfeat_bwd_reduce_kernel(
        const ReduceAll R, int nred, const float* __restrict__ g_ped, const float* __restrict__ g_obs,
        const float2* __restrict__ g_destf, const int* __restrict__ ped_idx, const int* __restrict__ obs_idx,
        const float* __restrict__ p, int ld, const float2* __restrict__ dest, int C, int N, int f0, int fcnt, int kpe, int koe,
        float* g_state, float2* g_dest, float* __restrict__ g_speed) {
    if ((int)blockIdx.x < nred) {
        reduce_block(R, (int)blockIdx.x);
        return;
    }
    relfeat_bwd_rows((int)blockIdx.x - nred, g_ped, g_obs, g_destf, ped_idx, obs_idx, p, ld, dest, C, N, f0, fcnt, kpe, koe, g_state,
                     g_dest, 7, g_speed);
}

// the relfeat backward over d/d(self_features) rows (gld = 7), carrying the slot sums a piml_pinnsf_bwd(PIML_DEFER_SLOT_SUMS) left on
// this stream as its leading workgroups (and the unfold of PIML_POOL_TRAIN behind them)
static int relfeat_bwd_self_launch(const float* g_ped_feat, const float* g_obs_feat, const float* g_self, const int* ped_idx,
                                   const int* obs_idx, const float* position, int state_ld, const float* destination, int C, int N,
                                   int focal_begin, int focal_count, int kp_eff, int ko_eff, float* g_state, float* g_destination,
                                   float* g_speed, void* stream) {
    const long rows = (long)C * focal_count;
    ReduceAll R;
    if (pending_slot_sums_take(as_stream(stream), &R)) {
        const int nred = R.gx * R.nsets;
        hipLaunchKernelGGL(relfeat_bwd_reduce_kernel, dim3((unsigned)(nred + (rows + 3) / 4)), dim3(256), 0, as_stream(stream),
                           R, nred, g_ped_feat, g_obs_feat, (const float2*)g_self, ped_idx, obs_idx, position, state_ld,
                           (const float2*)destination, C, N, focal_begin, focal_count, kp_eff, ko_eff, g_state,
                           (float2*)g_destination, g_speed);
        trace_mark("relfeat_bwd", as_stream(stream));
        if (R.nunf > 0) return launch_unfold(R, as_stream(stream));
    } else {
        hipLaunchKernelGGL(relfeat_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), g_ped_feat,
                           g_obs_feat, (const float2*)g_self, ped_idx, obs_idx, position, state_ld, (const float2*)destination, C, N,
                           focal_begin, focal_count, kp_eff, ko_eff, g_state, (float2*)g_destination, 7, g_speed);
        trace_mark("relfeat_bwd", as_stream(stream));
    }
    return hipGetLastError();
}

// Deterministic variant of relfeat_bwd (no atomics, bit-reproducible): one thread per (source agent, component).
// The scatter part is a GATHER over the entries of the neighbour lists that name this source: `sorted_keys` are the
// keys slice * N + source of all (row, slot) entries sorted ascending (empty slots carry the sentinel C * N), `order`
// their positions row * kpe + slot in the same (stable) order, so every sum runs in one fixed order.
__global__ __launch_bounds__(256) void relfeat_bwd_det_kernel(
        const float* __restrict__ g_ped, const float* __restrict__ g_obs, const float2* __restrict__ g_destf,
        const int* __restrict__ ped_idx, const int* __restrict__ obs_idx, const long long* __restrict__ sorted_keys,
        const long long* __restrict__ order, long long nkeys, const float* __restrict__ p, int ld,
        const float2* __restrict__ dest, int C, int N, int f0, int fcnt, int kpe, int koe, int accumulate,
        float* g_state, float2* g_dest) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long src = t >> 3;
    const int q = (int)(t & 7);
    if (src >= (long long)C * N || q >= 6) return;
    const int c = (int)(src / N), j = (int)(src - (long long)c * N);
    float sum = 0.f;
    if (j >= f0 && j < f0 + fcnt) {                       // the row's own term: -(sum of its upstream gradients)
        const long long row = (long long)c * fcnt + (j - f0);
        for (int s = 0; s < kpe; ++s)
            if (ped_idx[row * kpe + s] >= 0) sum -= g_ped[(row * kpe + s) * 6 + q];
        for (int s = 0; s < koe; ++s)
            if (obs_idx[row * koe + s] >= 0) sum -= g_obs[(row * koe + s) * 6 + q];
        if (q < 2) {
            const float dq = reinterpret_cast<const float*>(dest)[row * 2 + q] - p[src * ld + q];
            const float gd = dq != dq ? 0.f : reinterpret_cast<const float*>(g_destf)[row * 2 + q];
            reinterpret_cast<float*>(g_dest)[row * 2 + q] = gd;
            sum -= gd;
        }
    }
    long long lo = 0, hi = nkeys;                         // first entry with key >= src
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (sorted_keys[mid] < src) lo = mid + 1; else hi = mid;
    }
    for (long long e = lo; e < nkeys && sorted_keys[e] == src; ++e) sum += g_ped[order[e] * 6 + q];
    float* o = g_state + src * 6 + q;
    *o = accumulate ? *o + sum : sum;
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

template <int W, bool R>
static void relfeat_go(dim3 grid, dim3 block, void* stream, const RelfeatArgs& A, bool forward_of_a_step = false) {
    constexpr int bytes = relfeat_lds_bytes<W, R>();
    PackWork PK;
    PK.first_block = -1;
    if (forward_of_a_step && pending_pack_take(as_stream(stream), &PK.A)) {     // a pack left by piml_pinnsf_pack(PIML_DEFER_PACK)
        PK.first_block = (int)grid.x;
        grid.x += (unsigned)pack_blocks_total(PK.A, W * 64);
    }
    relfeat_fwd_kernel<W, R><<<grid, block, bytes, as_stream(stream)>>>(A, PK);
}
template <int W>
static int relfeat_attr() {
    constexpr int bytes = relfeat_lds_bytes<W, true>();
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(relfeat_fwd_kernel<W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

static int relfeat_launch(const float* position, const float* heading, const float* velocity,
                          const float* acceleration, int state_ld, const float* destination,
                          const float* obstacles, int C, int N, int M, int focal_begin,
                          int focal_count, int topk_ped, int topk_obs, float cos_thr_ped,
                          float cos_thr_obs, float dist_thr_ped, float dist_thr_obs,
                          float* ped_feat, float* obs_feat, float* dest_feat, int dest_feat_ld,
                          int32_t* ped_idx, int32_t* obs_idx, const float* speed, float* zero, long zero_n, void* stream,
                          int part = 0, long long* tick = nullptr) {
    if (C < 0 || N < 0 || M < 0 || focal_begin < 0 || focal_count < 0 || focal_begin + focal_count > N ||
        topk_ped < 0 || topk_obs < 0 || topk_ped > PIML_MAX_TOPK || topk_obs > PIML_MAX_TOPK ||
        state_ld < 2 || (state_ld & 1) || dest_feat_ld < 2)
        return hipErrorInvalidValue;
    if (C == 0 || focal_count == 0) return hipSuccess;
    const int kpe_ = topk_ped < N ? topk_ped : N, koe_ = topk_obs < M ? topk_obs : M;     // an output with k = 0 may be NULL
    if (part < 0 || part > 2 || !position || !velocity || !acceleration || (kpe_ > 0 && !ped_idx)) return hipErrorInvalidValue;
    if (part != 2 && (!destination || !dest_feat || (M > 0 && !obstacles) || (koe_ > 0 && (!obs_feat || !obs_idx))))
        return hipErrorInvalidValue;
    if (part != 1 && kpe_ > 0 && !ped_feat) return hipErrorInvalidValue;
    RelfeatArgs A;
    A.a_lo[0] = 0; A.a_hi[0] = N; A.a_lo[1] = 0; A.a_hi[1] = 0;
    A.flags = kRfPedFeat | kRfObs | kRfDest;
    A.split = 0;
    if (part == 1) {            // LOCAL: the focal block's own agents as sources + everything that needs no remote record
        A.a_lo[0] = focal_begin; A.a_hi[0] = focal_begin + focal_count;
        A.flags = kRfObs | kRfDest | kRfPedList;
    } else if (part == 2) {     // REMOTE: the agents either side of the block, continuing the list of the LOCAL part
        A.a_lo[0] = 0; A.a_hi[0] = focal_begin; A.a_lo[1] = focal_begin + focal_count; A.a_hi[1] = N;
        A.flags = kRfPedFeat | kRfInit;
    }
    A.p = position; A.v = velocity; A.a = acceleration; A.ld = state_ld;
    A.hd = (const float2*)heading; A.dest = (const float2*)destination; A.obs = (const float2*)obstacles;
    A.C = C; A.N = N; A.M = M; A.f0 = focal_begin; A.fcnt = focal_count;
    A.kp = topk_ped < N ? topk_ped : N; A.ko = topk_obs < M ? topk_obs : M;
    A.cos_p = cos_thr_ped; A.cos_o = cos_thr_obs;
    A.cut2_p = dist2_cutoff(dist_thr_ped); A.cut2_o = dist2_cutoff(dist_thr_obs);
    A.dthr_p = dist_thr_ped; A.dthr_o = dist_thr_obs;
    A.ped_feat = ped_feat; A.obs_feat = obs_feat; A.dest_feat = dest_feat; A.dest_ld = dest_feat_ld;
    A.ped_idx = ped_idx; A.obs_idx = obs_idx;
    A.speed = speed; A.zero = zero; A.zero_n = zero ? zero_n : 0; A.tick = tick;
    A.stats = nullptr;
#ifdef PIML_RELFEAT_STATS
    if (const char* e = getenv("PIML_RELFEAT_STATS_PTR")) A.stats = (int*)strtoull(e, nullptr, 0);
#endif
    const long rows = (long)C * focal_count;
    // Workgroup size: every workgroup stages the whole source array once, so bigger groups cut
    // L2->LDS traffic, while smaller groups shorten the barrier tails (the per-wave work is
    // data dependent) and spread small launches over more CUs.
    int waves = rows >= 16384 ? 8 : (rows >= 4096 ? 16 : (rows >= 1024 ? 8 : 4));   // measured (tools/time_*.py)
    // more sources than one tile (every pass re-stages the tile behind two barriers: the shapes of a sharded scene): eight waves
    // -- round 6, tools/time_sharded_shapes.py: 4096 focal rows x 8192 / 16384 / 32768 sources 26.4 / 36.5 / 57.7 us against
    // 29.1 / 40.3 / 59.1 with sixteen (and 32.5 / 45.5 / 67.9 with four); 2048 x 16384: 29.7 / 32.8 / 34.0
    if (N > kTile && rows >= 1024) waves = 8;
    if (const char* e = getenv("PIML_RELFEAT_WAVES")) waves = atoi(e);
    const int bpc = (focal_count + waves - 1) / waves;
    const dim3 grid((unsigned)(C * bpc)), block((unsigned)(waves * 64));
    // both tiles resident (one barrier per launch) when the agent sources are one tile and the obstacle pass exists and fits
    static const bool res_off = getenv("PIML_RELFEAT_RESIDENT") && atoi(getenv("PIML_RELFEAT_RESIDENT")) == 0;
    const bool res = !res_off && (A.flags & kRfObs) && M > 0 && M <= kObsTile && A.a_hi[1] == A.a_lo[1] &&
                     A.a_hi[0] - A.a_lo[0] <= kTile && A.ko > 0;
    if (res) {
        static int attr = -1;      // dynamic LDS above 64 KB has to be enabled per kernel once per process
        if (attr < 0) {
            attr = relfeat_attr<16>();
            if (!attr) attr = relfeat_attr<8>();
            if (!attr) attr = relfeat_attr<4>();
        }
        if (attr) return attr;
    }
    // split launch (pedestrian and obstacle passes of a row in different workgroups): whenever the obstacle pass exists
    static const bool split_off = getenv("PIML_RELFEAT_SPLIT") && atoi(getenv("PIML_RELFEAT_SPLIT")) == 0;
    // (measured, tools/time_relfeat.py: 8.4 -> 6.2 us at 122 agents, 10.6 -> 7.7 at 1024, 22.3 -> 21.8 at 4096 + 2000 points,
    // 34.6 -> 29.3 for a rank's 2048 x 16384 share, 120 -> 115 at 16384; 64 slices of 128 agents -- 8192 short rows, bound by
    // issue slots -- 17.9 -> 22.5: not split)
    if (!split_off && (A.flags & kRfObs) && M > 0 && A.ko > 0 && A.kp > 0 && (C == 1 || rows <= 4096)) {
        A.split = 1;
        const dim3 grid2(grid.x * 2);
        switch (waves) {
            case 16: relfeat_go<16, false>(grid2, block, stream, A, true); break;
            case 8: relfeat_go<8, false>(grid2, block, stream, A, true); break;
            case 4: relfeat_go<4, false>(grid2, block, stream, A, true); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (waves * 2 + (res ? 1 : 0)) {
        case 33: relfeat_go<16, true>(grid, block, stream, A, true); break;
        case 32: relfeat_go<16, false>(grid, block, stream, A, true); break;
        case 17: relfeat_go<8, true>(grid, block, stream, A, true); break;
        case 16: relfeat_go<8, false>(grid, block, stream, A, true); break;
        case 9: relfeat_go<4, true>(grid, block, stream, A, true); break;
        case 8: relfeat_go<4, false>(grid, block, stream, A, true); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

PIML_API int piml_relfeat_fwd(const float* position, const float* heading, const float* velocity,
                              const float* acceleration, int state_ld, const float* destination,
                              const float* obstacles, int C, int N, int M, int focal_begin,
                              int focal_count, int topk_ped, int topk_obs, float cos_thr_ped,
                              float cos_thr_obs, float dist_thr_ped, float dist_thr_obs,
                              float* ped_feat, float* obs_feat, float* dest_feat, int dest_feat_ld,
                              int32_t* ped_idx, int32_t* obs_idx, void* stream) {
    return relfeat_launch(position, heading, velocity, acceleration, state_ld, destination, obstacles, C, N, M, focal_begin,
                          focal_count, topk_ped, topk_obs, cos_thr_ped, cos_thr_obs, dist_thr_ped, dist_thr_obs, ped_feat,
                          obs_feat, dest_feat, dest_feat_ld, ped_idx, obs_idx, nullptr, nullptr, 0, stream);
}

// piml_relfeat_fwd whose third output is the model's self_features rows [dest - p, v, a, v0] (C, n, 7) instead of the
// destination features (C, n, 2): the per-frame torch.cat of the training rollout (src/models/simulators.py:778-779) inside
// the launch.  desired_speed (C, n): v0 of the focal rows.  g_state_zero (C * N * 6 floats, may be NULL) is cleared on the way
// for piml_relfeat_bwd_self to accumulate into.
PIML_API int piml_relfeat_fwd_self(const float* position, const float* heading, const float* velocity,
                                   const float* acceleration, int state_ld, const float* destination, const float* obstacles,
                                   const float* desired_speed, int C, int N, int M, int focal_begin, int focal_count,
                                   int topk_ped, int topk_obs, float cos_thr_ped, float cos_thr_obs, float dist_thr_ped,
                                   float dist_thr_obs, float* ped_feat, float* obs_feat, float* self_features, int32_t* ped_idx,
                                   int32_t* obs_idx, float* g_state_zero, void* stream) {
    if (C > 0 && focal_count > 0 && (!desired_speed || !self_features)) return hipErrorInvalidValue;
    return relfeat_launch(position, heading, velocity, acceleration, state_ld, destination, obstacles, C, N, M, focal_begin,
                          focal_count, topk_ped, topk_obs, cos_thr_ped, cos_thr_obs, dist_thr_ped, dist_thr_obs, ped_feat,
                          obs_feat, self_features, 7, ped_idx, obs_idx, desired_speed, g_state_zero, (long)C * N * 6, stream);
}

// Backward of piml_relfeat_fwd_self: g_self (C, n, 7) carries d/d(dest_feat) in columns 0-1, d/d(v, a) of the focal rows in
// 2-5 and d/d(desired speed) in 6.  ACCUMULATES into g_state (C, N, 6) = d/d(p, v, a), which the caller has cleared;
// g_destination (C, n, 2) and g_speed (C, n; may be NULL) are written.
PIML_API int piml_relfeat_bwd_self(const float* g_ped_feat, const float* g_obs_feat, const float* g_self, const int32_t* ped_idx,
                                   const int32_t* obs_idx, const float* position, int state_ld, const float* destination, int C,
                                   int N, int focal_begin, int focal_count, int kp_eff, int ko_eff, float* g_state,
                                   float* g_destination, float* g_speed, void* stream) {
    if (C < 0 || N < 0 || focal_begin < 0 || focal_count < 0 || focal_begin + focal_count > N || kp_eff < 0 || ko_eff < 0 ||
        state_ld < 2 || (state_ld & 1))
        return hipErrorInvalidValue;
    if (C == 0 || focal_count == 0) return hipSuccess;
    if (!g_self || !position || !destination || !g_state || !g_destination || (kp_eff > 0 && (!g_ped_feat || !ped_idx)) ||
        (ko_eff > 0 && (!g_obs_feat || !obs_idx)))
        return hipErrorInvalidValue;
    return relfeat_bwd_self_launch(g_ped_feat, g_obs_feat, g_self, ped_idx, obs_idx, position, state_ld, destination, C, N, focal_begin,
                                   focal_count, kp_eff, ko_eff, g_state, g_destination, g_speed, stream);
}

// piml_relfeat_fwd that also advances a device-side frame counter by one (the captured inference-rollout frame ends with
// this launch: the increment was a 4 us launch of its own).  The kernel never reads the counter.
PIML_API int piml_relfeat_fwd_tick(const float* position, const float* heading, const float* velocity,
                                   const float* acceleration, int state_ld, const float* destination,
                                   const float* obstacles, int C, int N, int M, int focal_begin,
                                   int focal_count, int topk_ped, int topk_obs, float cos_thr_ped,
                                   float cos_thr_obs, float dist_thr_ped, float dist_thr_obs,
                                   float* ped_feat, float* obs_feat, float* dest_feat, int dest_feat_ld,
                                   int32_t* ped_idx, int32_t* obs_idx, long long* tick, void* stream) {
    return relfeat_launch(position, heading, velocity, acceleration, state_ld, destination, obstacles, C, N, M, focal_begin,
                          focal_count, topk_ped, topk_obs, cos_thr_ped, cos_thr_obs, dist_thr_ped, dist_thr_obs, ped_feat,
                          obs_feat, dest_feat, dest_feat_ld, ped_idx, obs_idx, nullptr, nullptr, 0, stream, 0, tick);
}

// One scene of packed (N, 6) = (p, v, a) records: the features of the focal rows AND their self_features rows
// [dest - p, v, a, v0] (n, 7) in the same launch; `g_state_zero` (N * 6 floats, may be NULL) is cleared on the way for
// piml_relfeat_self_bwd to accumulate into.
PIML_API int piml_relfeat_self_fwd(const float* state, const float* destination_rows, const float* obstacles,
                                   const float* desired_speed, int N, int M, int focal_begin, int focal_count,
                                   int topk_ped, int topk_obs, float cos_thr_ped, float cos_thr_obs, float dist_thr_ped,
                                   float dist_thr_obs, float* ped_feat, float* obs_feat, float* self_features,
                                   int32_t* ped_idx, int32_t* obs_idx, float* g_state_zero, void* stream) {
    if (focal_count > 0 && (!state || !desired_speed || !self_features)) return hipErrorInvalidValue;
    const int e = relfeat_launch(state, nullptr, state + 2, state + 4, 6, destination_rows, obstacles, 1, N, M, focal_begin,
                                 focal_count, topk_ped, topk_obs, cos_thr_ped, cos_thr_obs, dist_thr_ped, dist_thr_obs, ped_feat,
                                 obs_feat, self_features, 7, ped_idx, obs_idx, desired_speed, g_state_zero, (long)N * 6, stream);
    trace_mark("relfeat_fwd", as_stream(stream));
    return e;
}

// piml_relfeat_self_fwd in two launches for agent-block sharding (part 1 = PIML_RELFEAT_LOCAL, 2 = PIML_RELFEAT_REMOTE,
// 0 = the whole thing): LOCAL reads only the focal block's own records (and the obstacles) and can run while the
// all-gather of the other blocks is in flight; REMOTE continues from the list LOCAL left in ped_idx.  Together the
// outputs are bit-identical to one launch.
PIML_API int piml_relfeat_self_fwd_part(int part, const float* state, const float* destination_rows, const float* obstacles,
                                        const float* desired_speed, int N, int M, int focal_begin, int focal_count,
                                        int topk_ped, int topk_obs, float cos_thr_ped, float cos_thr_obs,
                                        float dist_thr_ped, float dist_thr_obs, float* ped_feat, float* obs_feat,
                                        float* self_features, int32_t* ped_idx, int32_t* obs_idx, float* g_state_zero,
                                        void* stream) {
    if (focal_count > 0 && (!state || (part != 2 && (!desired_speed || !self_features)))) return hipErrorInvalidValue;
    return relfeat_launch(state, nullptr, state + 2, state + 4, 6, destination_rows, obstacles, 1, N, M, focal_begin,
                          focal_count, topk_ped, topk_obs, cos_thr_ped, cos_thr_obs, dist_thr_ped, dist_thr_obs, ped_feat,
                          obs_feat, self_features, 7, ped_idx, obs_idx, part == 2 ? nullptr : desired_speed,
                          part == 2 ? nullptr : g_state_zero, (long)N * 6, stream, part);
}

// Backward of piml_relfeat_self_fwd in one launch: g_self (n, 7) carries d/d(dest_feat) in columns 0-1, d/d(v, a) of the
// focal rows in 2-5 and d/d(desired speed) in 6.  ACCUMULATES into g_state (N, 6), which the caller (or the forward's
// g_state_zero) has cleared; g_destination (n, 2) and g_speed (n, may be NULL) are written.
PIML_API int piml_relfeat_self_bwd(const float* g_ped_feat, const float* g_obs_feat, const float* g_self, const int* ped_idx,
                                   const int* obs_idx, const float* state, const float* destination_rows, int N,
                                   int focal_begin, int focal_count, int kp_eff, int ko_eff, float* g_state,
                                   float* g_destination, float* g_speed, void* stream) {
    if (N < 0 || focal_begin < 0 || focal_count < 0 || focal_begin + focal_count > N || kp_eff < 0 || ko_eff < 0)
        return hipErrorInvalidValue;
    if (focal_count == 0) return hipSuccess;
    if (!g_self || !state || !destination_rows || !g_state || !g_destination || (kp_eff > 0 && (!g_ped_feat || !ped_idx)) ||
        (ko_eff > 0 && (!g_obs_feat || !obs_idx)))
        return hipErrorInvalidValue;
    return relfeat_bwd_self_launch(g_ped_feat, g_obs_feat, g_self, ped_idx, obs_idx, state, 6, destination_rows, 1, N, focal_begin,
                                   focal_count, kp_eff, ko_eff, g_state, g_destination, g_speed, stream);
}

PIML_API int piml_relfeat_bwd(const float* g_ped_feat, const float* g_obs_feat, const float* g_dest_feat,
                              const int32_t* ped_idx, const int32_t* obs_idx, const float* position,
                              int state_ld, const float* destination, int C, int N, int focal_begin, int focal_count,
                              int kp_eff, int ko_eff, float* g_state, float* g_destination, void* stream) {
    if (C < 0 || N < 0 || focal_begin < 0 || focal_count < 0 || focal_begin + focal_count > N ||
        kp_eff < 0 || ko_eff < 0 || state_ld < 2 || (state_ld & 1))
        return hipErrorInvalidValue;
    if (C == 0 || focal_count == 0) return hipSuccess;
    if (!g_dest_feat || !position || !destination || !g_state || !g_destination) return hipErrorInvalidValue;
    const long rows = (long)C * focal_count;
    hipLaunchKernelGGL(relfeat_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream),
                       g_ped_feat, g_obs_feat, (const float2*)g_dest_feat, ped_idx, obs_idx,
                       position, state_ld, (const float2*)destination, C, N, focal_begin, focal_count,
                       kp_eff, ko_eff, g_state, (float2*)g_destination, 2, (float*)nullptr);
    return hipGetLastError();
}

PIML_API int piml_relfeat_bwd_det(const float* g_ped_feat, const float* g_obs_feat, const float* g_dest_feat,
                                  const int* ped_idx, const int* obs_idx, const long long* sorted_keys,
                                  const long long* order, const float* position, int state_ld, const float* destination,
                                  int C, int N, int focal_begin, int focal_count, int kp_eff, int ko_eff, int accumulate,
                                  float* g_state, float* g_destination, void* stream) {
    if (C < 0 || N < 0 || focal_begin < 0 || focal_count < 0 || focal_begin + focal_count > N ||
        kp_eff < 0 || ko_eff < 0 || state_ld < 2 || (state_ld & 1))
        return hipErrorInvalidValue;
    if (C == 0 || N == 0) return hipSuccess;
    if (!g_dest_feat || !position || !destination || !g_state || !g_destination) return hipErrorInvalidValue;
    const long long nkeys = (long long)C * focal_count * kp_eff;
    if (nkeys > 0 && (!sorted_keys || !order || !g_ped_feat || !ped_idx)) return hipErrorInvalidValue;
    const long long threads = (long long)C * N * 8;
    hipLaunchKernelGGL(relfeat_bwd_det_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream),
                       g_ped_feat, g_obs_feat, (const float2*)g_dest_feat, ped_idx, obs_idx, sorted_keys, order, nkeys,
                       position, state_ld, (const float2*)destination, C, N, focal_begin, focal_count, kp_eff, ko_eff,
                       accumulate, g_state, (float2*)g_destination);
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
