// Backward of the PINNSF encoder on the agents' sums of h2 (PIML_POOL_TRAIN) as TWO CREWS of four waves (round 6).
//
// Reference arithmetic: the autograd of MLP(in, [128, 128, 128]) (src/models/model.py:40-65) below the neighbour-axis sum
// (:1279-1283) with the last layer folded into the decoder:
//     G2 = g_sum[row / k] * [h2 > 0]     dW2 = G2^T H1, db2 = colsum G2        H1 = relu(W1 x + b1)
//     G1 = (G2 W2) * [h1 > 0]            dW1 = G1^T X,  db1 = colsum G1        g_x = G1 W1
//
// encoder_bwd3.hip runs this as ONE wave per SIMD (368 registers): every vector, LDS and scalar instruction of a tile is issued
// by the wave that also issues its 96 matrix instructions, a lone wave issues one instruction per four cycles whatever its
// kind, and its stamps read 8.5 k cycles per tile for 3.1 k cycles of products.  Here a workgroup is EIGHT waves, two per SIMD
// (a SIMD issues the vector instructions of two waves at twice the rate of one, and one wave's LDS / memory latency is the
// other's issue time), cut by ROLE, not by data -- the persistent state of a tile's work (weight fragments 96 registers, dW2
// accumulators 128) does not fit one wave of 256 registers, but it falls into two halves that never meet in a register:
//   crew A (waves 0-3, wave w = feature block w): the CHAIN.  Gathers g_sum of the next tile, masks it with the signs of h2,
//     splits it into bf16 pieces and lays them into the G2 image (LDS); layer B = 48 products on the W2^T fragments it holds
//     in registers for the whole slab (hi, mid AND lo: no weight traffic per tile); masks G1, lays it into the G1 tile (LDS);
//     recomputes H1 of the next tile (f32 matrix instruction) and lays its pieces into the H1 image; db2, db1.
//   crew B (waves 4-7): the WEIGHT GRADIENTS.  dW2 += G2^T H1 = 48 products from the two images into 128 accumulator
//     registers; one tile behind, from the G1 tile: dW1 (vector FMAs) and g_x (the per-wave partials through LDS, fixed
//     order).
// Every buffer between the crews is double-buffered (the x rows three deep), so a tile costs ONE workgroup barrier -- a bare
// s_barrier behind lgkmcnt(0): __syncthreads() would also wait for the loads that are in flight for the next tile.
// Everything is plain VGPRs and builtin matrix instructions (<= 256 registers: no accumulator-file operands, no asm hazards).
// Arithmetic, product order and summation order of every WEIGHT gradient are those of enc_bwd_fused_x3_kernel<..., SUMS>: the two
// kernels agree BITWISE there; g_x is the same f32 arithmetic in another summation order (f32 matrix instruction, below)
// (tests/test_sums_gpu.py::test_two_crew_backward_against_the_one_wave_kernel).
// Fragment element order, images and the slot layout: encoder_bwd3.hip / pack.hpp.
#include "common.hpp"
#include "encoder.hpp"
#include "x3.hpp"

namespace piml {

typedef short f5_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f5_s16x4 f5_lds_s16x4;

constexpr int F5_THREADS = 512;
constexpr int F5_IMG = 3 * 128 * 64;                         // one G2 image [piece 3][feature 128][row 32] bf16 = one H1 image = 24576 B
constexpr int F5_M = 0;                                      // G2 images [parity 2]
constexpr int F5_H = F5_M + 2 * F5_IMG;                      // H1 images [parity 2]: B fragments [block 4][k-step 2][piece 3][lane 64] u32x4
constexpr int F5_TROW = 36;                                  // floats per feature of the G1 tile (32 rows + 4: conflict-free both ways)
constexpr int F5_TBYTES = 128 * F5_TROW * 4;
constexpr int F5_T = F5_H + 2 * F5_IMG;                      // G1 tiles [parity 2][feature 128][36] floats
constexpr int F5_XS = F5_T + 2 * F5_TBYTES;                  // x rows [ring 3][32][8] floats
constexpr int F5_MK = F5_XS + 3 * 1024;                      // sign words of h1 [parity 2][128 dwords]
constexpr int F5_W1 = F5_MK + 2 * 512;                       // W1 rows [128][8] floats
constexpr int F5_GXP = F5_W1 + 4096;                         // g_x partials [parity 2][wave 4][row 32][8] floats
// 1: complementary halves -- crew A's products beside crew B's vector work, then crew A's vector work beside crew B's products;
// 0: every wave interleaves its vector work with its own products (both waves of a SIMD want both pipes all the time)
#ifndef PIML_F5_PHASED
#define PIML_F5_PHASED 1
#endif
// 1: crew A's requests of the tile after the next between its products (second register set, 21 moves); 0: behind its vector
// half.  Measured level-to-worse (30.2 - 30.5 against 29.5 - 29.7 us): what crew A saves, crew B's products lose -- the SIMD issues
// about one instruction per six cycles whichever wave it comes from
#ifndef PIML_F5_REQ_EARLY
#define PIML_F5_REQ_EARLY 0
#endif
// 1: g_x = G1 W1 on v_mfma_f32_16x16x4_f32 -- the W1 operand is EIGHT registers per lane for the whole slab and the G1 operand one
// conflict-light ds_read_b32 per product (16 per tile); 0: vector FMAs on W1 rows broadcast from LDS (32 ds_read_b128 per wave and
// tile, which the stamps price at ~1000 cycles of the tile: the LDS pipe, not the FMAs, was crew B's vector half)
#ifndef PIML_F5_GX_MFMA
#define PIML_F5_GX_MFMA 1
#endif
// 1: dW1 = G1^T X on the same instruction (two chains of eight products per tile into eight persistent registers; operands: one
// ds_read_b32 of the G1 tile and one of the x rows per product) instead of 96 FMAs on 36 LDS reads per wave and tile.  Built,
// green, and SLOWER (31.5 against 29.5 us): with it crew B has no vector half left, its 80 products start beside crew A's chain
// and the tile becomes the matrix pipe's 134 x 32 cycles.  Off; the vector form is the default.
#ifndef PIML_F5_DW1_MFMA
#define PIML_F5_DW1_MFMA 0
#endif
static_assert(!PIML_F5_GX_MFMA || PIML_F5_PHASED, "the g_x products ride in the phased schedule");
static_assert(!PIML_F5_DW1_MFMA || PIML_F5_GX_MFMA, "the dW1 products need the zero region the g_x products freed");
constexpr int F5_GXP_BYTES = PIML_F5_GX_MFMA ? 4 * 32 * 16 * 4 : 4096;      // one parity: [wave 4][row 32][16 | 8] floats
constexpr int F5_TAB = F5_GXP + 2 * F5_GXP_BYTES;            // gather table [rem < 16][half 2][register 16] byte offsets
constexpr int F5_KMAX = 16;
constexpr int F5_LDS_BYTES = F5_TAB + F5_KMAX * 2 * 16 * 4;
constexpr int F5_LDS_LAUNCH = F5_LDS_BYTES;
static_assert(F5_LDS_LAUNCH <= 160 * 1024, "fits the CU");
constexpr int F5_PART1 = EH * EH + 1024 + 2 * EH;            // = F3_PART1 (encoder_bwd3.hip): dW2 | dW1 (1024-float field) | db2 | db1

struct F5Args {
    EncArgs A;
    int nA[2];          // workgroups of branch 0 / branch 1 (grid = their sum)
};

#ifdef PIML_F5_STAMPS
// diagnostic build only (tools/f5_stamps.py): cycles between the stamps of every wave, summed over the workgroup's tiles
__device__ unsigned long long g_f5_stamps[256 * 8 * 16];      // [workgroup][wave 8][stamp 16]
#define F5_STAMP(i)                                                        \
    do {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                 \
        {                                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
            st[i] += t_ - tprev;                                           \
            tprev = t_;                                                    \
        }                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                 \
    } while (0)
#else
#define F5_STAMP(i)
#endif

// the workgroup barrier of the tile loop: this wave's LDS operations done, nothing said about its loads in flight
#define F5_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ float f5_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ constexpr int f5_rho(int r) { return (r & 3) + 8 * (r >> 2); }      // row of accumulator register r in lane half 0 (half 1: + 4)

// diagnostic builds (tools/r6_f5skip.sh; RESULTS WRONG ON PURPOSE): PIML_F5_SKIP = bits of work left out, to see what it costs
//   1: crew B's lagging vector work (g_x, dW1)   2: g_x only   4: crew A's G2 image of the next tile   8: crew A's H1
//   16: crew A's G1 mask + tile   32: crew B's products   64: crew A's products   128: crew A's requests
#ifndef PIML_F5_SKIP
#define PIML_F5_SKIP 0
#endif

#define F5_SLOT(MF, FILL)                        \
    do {                                         \
        MF;                                      \
        __builtin_amdgcn_sched_barrier(0);       \
        FILL;                                    \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

// INC: the columns of x the vector loops run over -- 6 when in_dim == 6 (every PINNSF encoder of the reference:
// ped_feature_dim = obs_feature_dim = 6, src/main.py:60-62), else 8 (in_dim <= 8, padded with zeros)
template <bool GX, int INC>
__global__ __launch_bounds__(F5_THREADS) void enc_bwd_sums2_kernel(F5Args F) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int crew = wv >> 2, w = wv & 3;                     // waves w and w + 4 share a SIMD
    const int ctid = tid & 255;                               // thread index within the crew
    const int n = lane & 31, h = lane >> 5;
    int bx = (int)blockIdx.x, b = 0;
    if (bx >= F.nA[0]) { b = 1; bx -= F.nA[0]; }
    const piml_encoder_branch J = b ? F.A.br[1] : F.A.br[0];
    const int nwg = F.nA[b];
    const unsigned R = (unsigned)J.rows;                      // rows < 2^22 (checked on the host): byte offsets fit 32 bits
    const unsigned IN = __builtin_amdgcn_readfirstlane((unsigned)J.in_dim), K = __builtin_amdgcn_readfirstlane((unsigned)J.k);
    const unsigned kmagic = __builtin_amdgcn_readfirstlane((unsigned)((0x100000000ull + K - 1) / K));      // row / K == umulhi(row, kmagic)
    const int ntiles = (int)((R + 31) >> 5);
    const int nit = bx < ntiles ? (ntiles - bx + nwg - 1) / nwg : 0;      // tiles of this workgroup: bx, bx + nwg, ...
    float* P = J.partials + (size_t)bx * F5_PART1;
    constexpr int NS = INC == 6 ? 3 : 4;                      // column pairs of x

    // Buffer resources: an offset past the range reads as zero / is not stored (rows and tiles past the end cost no branch)
    auto rsrc = [&](const void* base, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    };
    constexpr unsigned kOut = 0x80000000u;       // (+ a table offset: no wrap)
    const __amdgpu_buffer_rsrc_t rs_gp = rsrc(J.g_pooled, (R / K) * EH * 4);
    const __amdgpu_buffer_rsrc_t rs_x = rsrc(J.x, R * IN * 4);
    const __amdgpu_buffer_rsrc_t rs_mk = rsrc(J.relu_mask, (unsigned)ntiles * 1024);
    const __amdgpu_buffer_rsrc_t rs_gx = rsrc(J.g_x, GX ? R * IN * 4 : 0u);
    auto ld1 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned soff) {
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0);
    };

#ifdef PIML_F5_STAMPS
    unsigned long long st[16], tprev = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0;
#endif

    // ---- LDS: the buffers the lagging work of the first iterations reads must hold zeros; the gather table; W1 rows ----
    {
        float4* z = reinterpret_cast<float4*>(smem + F5_T);
        constexpr int NZ = (F5_W1 - F5_T) / 16;               // G1 tiles | x ring | sign words
#pragma unroll
        for (int i = 0; i < (NZ + F5_THREADS - 1) / F5_THREADS; ++i)
            if (i * F5_THREADS + tid < NZ) z[i * F5_THREADS + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 2 * F5_GXP_BYTES / 16 / F5_THREADS; ++i) reinterpret_cast<float4*>(smem + F5_GXP)[i * F5_THREADS + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        // agent of tile row m relative to the tile's first agent: (rem + m) / K with rem = (32 tile) % K; entry = byte offset of
        // that agent's 512-byte row of g_sum; register r of lane half hh holds row rho(r) + 4 hh
        if ((unsigned)tid < K * 32u) {
            const unsigned rem = (unsigned)tid >> 5, hh = ((unsigned)tid >> 4) & 1u, r = (unsigned)tid & 15u;
            reinterpret_cast<unsigned*>(smem + F5_TAB)[tid] = ((rem + 4u * hh + (unsigned)f5_rho((int)r)) / K) * (EH * 4);
        }
        if (tid < 256 && PIML_F5_GX_MFMA) {                  // no W1 rows in LDS: the region is the zero operand of the dW1 products' idle columns
            reinterpret_cast<float4*>(smem + F5_W1)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (tid < 256) {
            const float* W1r = J.packed + PACK_FWD + 32768;     // W1 rows padded to 8 columns
            if (INC == 6) {                                   // g_x: lane half h takes columns 3 h .. 3 h + 2 -> [feature][half][c, c, c, 0]
                const float* src = W1r + (tid >> 1) * 8 + 3 * (tid & 1);
                reinterpret_cast<float4*>(smem + F5_W1)[tid] = make_float4(src[0], src[1], src[2], 0.f);
            } else {
                reinterpret_cast<float4*>(smem + F5_W1)[tid] = reinterpret_cast<const float4*>(W1r)[tid];
            }
        }
    }

    if (crew == 0) {
        // =================================================== crew A: the chain ===================================================
        // W2^T fragments of block w, all eight k-blocks, three pieces: 96 registers for the whole slab
        u32x4 wh[8], wm[8], wl[8];
        {
            const u32x4* imgB = reinterpret_cast<const u32x4*>(J.packed + PACK_F32 + 3 * X3_IMG) + lane;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = w * 8 + kb;
                wh[kb] = imgB[(fb * 2) * 64];
                wm[kb] = imgB[(fb * 2 + 1) * 64];
                wl[kb] = imgB[X3_HM / 4 + fb * 64];
            }
        }
        const float* W1rows = J.packed + PACK_FWD + 32768;
        float w1v[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) w1v[s] = W1rows[(32 * w + n) * 8 + 2 * s + h];
        const float b1v = J.b1[32 * w + n];

        // ---- per-lane LDS addresses (encoder_bwd3.hip) ----
        const int fw = 32 * w + n;
        const int swz_w = (fw >> 1) & 7;
        const int mw_off = F5_M + fw * 64;                     // writer of the G2 image: feature fw, chunk c at slot (c ^ swz_w)
        int mr[2];                                             // reader (ds_read_b64_tr_b16): + kb * 1024 + piece * 8192
        {
            const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g16 >> 1;
            const int chunk = 4 * (g16 & 1) + pp;
#pragma unroll
            for (int half2 = 0; half2 < 2; ++half2)
                mr[half2] = F5_M + (8 * half2 + 4 * hh + q) * 64 + ((chunk ^ (4 * half2 + 2 * hh + (q >> 1))) * 8);
        }
        // sign words of h1: lane (n, h), register r needs bit bp of word (w >> 1) of source lane rho(r) + 4 h + 32 h', h' = (n >> 2) & 1
        const int bp = 16 * (w & 1) + (n & 3) + 4 * (n >> 3);
        const int mk_off = F5_MK + ((4 * h + 32 * ((n >> 2) & 1)) * 2 + (w >> 1)) * 4;
        const int t_off = F5_T + fw * (F5_TROW * 4) + (4 * h) * 4;       // + parity * F5_TBYTES + 32 g: rows 8 g + 4 h .. + 3
        const int h_off = F5_H + ((w * 2) * 3 * 64 + lane) * 16;         // + parity * F5_IMG + ((s * 3 + piece) * 64) * 16
        const unsigned gbase = (unsigned)(32 * w + n) * 4u;              // this lane's feature within a row of g_sum

        // requests of a tile: g_sum rows through the table (16 loads), the lane's sign word of h2, its x values for H1, and
        // (threads 0 .. 127 of the crew) the tile's sign words of h1
        float g2[16];
        unsigned m2 = 0, mkv = 0;
        float xa[NS];
        // (every offset that must be range-checked travels in the VECTOR offset: the scalar offset of a buffer access is not part
        // of the check)
        auto req_g = [&](int tile) {
            const unsigned t32 = __builtin_amdgcn_readfirstlane((unsigned)tile * 32u);
            const unsigned a0 = __builtin_amdgcn_readfirstlane(__umulhi(t32, kmagic)), rem = t32 - a0 * K;
            const bool live = tile < ntiles;
            const unsigned gb = live ? gbase + a0 * (EH * 4) : kOut;      // (an agent past the end is out of the resource's range)
            const uint4* tab = reinterpret_cast<const uint4*>(smem + F5_TAB + (rem * 32u + 16u * (unsigned)h) * 4u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 o = tab[i];
                g2[4 * i + 0] = __uint_as_float(ld1(rs_gp, gb + o.x, 0u));
                g2[4 * i + 1] = __uint_as_float(ld1(rs_gp, gb + o.y, 0u));
                g2[4 * i + 2] = __uint_as_float(ld1(rs_gp, gb + o.z, 0u));
                g2[4 * i + 3] = __uint_as_float(ld1(rs_gp, gb + o.w, 0u));
            }
            m2 = ld1(rs_mk, live ? (unsigned)tile * 1024u + (128u + 2u * (unsigned)lane + (unsigned)(w >> 1)) * 4u : kOut, 0u);
        };
        auto req_x = [&](int tile) {
            const unsigned row = (unsigned)tile * 32u + (unsigned)n;
            const bool valid = (tile < ntiles) & (row < R);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned cx = 2u * s + h;
                xa[s] = __uint_as_float(ld1(rs_x, (valid & (cx < IN)) ? (row * IN + cx) * 4u : kOut, 0u));
            }
            mkv = ld1(rs_mk, tile < ntiles ? (unsigned)tile * 1024u + (unsigned)(ctid & 127) * 4u : kOut, 0u);
        };
        // The same requests cut into six steps that ride between the products (PIML_F5_REQ_EARLY): a vector-memory instruction costs
        // the lone issuer 15 - 30 cycles, the 22 of a tile were ~950 cycles of crew A's vector half; between products they are
        // covered.  They land in a second register set (the first still holds the tile the vector half is about to consume).
        float g2n[16];
        unsigned m2n = 0, mkvn = 0;
        float xan[NS];
        uint4 tabv;
        auto req_step = [&](int i, int tile) {
            const unsigned t32 = __builtin_amdgcn_readfirstlane((unsigned)tile * 32u);
            const unsigned a0 = __builtin_amdgcn_readfirstlane(__umulhi(t32, kmagic)), rem = t32 - a0 * K;
            const bool live = tile < ntiles;
            const uint4* tab = reinterpret_cast<const uint4*>(smem + F5_TAB + (rem * 32u + 16u * (unsigned)h) * 4u);
            if (i == 0) { tabv = tab[0]; return; }
            if (i <= 4) {
                const unsigned gb = live ? gbase + a0 * (EH * 4) : kOut;
                const uint4 o = tabv;
                if (i < 4) tabv = tab[i];
                g2n[4 * i - 4] = __uint_as_float(ld1(rs_gp, gb + o.x, 0u));
                g2n[4 * i - 3] = __uint_as_float(ld1(rs_gp, gb + o.y, 0u));
                g2n[4 * i - 2] = __uint_as_float(ld1(rs_gp, gb + o.z, 0u));
                g2n[4 * i - 1] = __uint_as_float(ld1(rs_gp, gb + o.w, 0u));
                return;
            }
            m2n = ld1(rs_mk, live ? (unsigned)tile * 1024u + (128u + 2u * (unsigned)lane + (unsigned)(w >> 1)) * 4u : kOut, 0u);
            const unsigned row = (unsigned)tile * 32u + (unsigned)n;
            const bool valid = live & (row < R);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned cx = 2u * s + h;
                xan[s] = __uint_as_float(ld1(rs_x, (valid & (cx < IN)) ? (row * IN + cx) * 4u : kOut, 0u));
            }
            mkvn = ld1(rs_mk, live ? (unsigned)tile * 1024u + (unsigned)(ctid & 127) * 4u : kOut, 0u);
        };
        auto req_take = [&]() {                                // the incoming set becomes the current one
#pragma unroll
            for (int r = 0; r < 16; ++r) g2[r] = g2n[r];
#pragma unroll
            for (int s = 0; s < NS; ++s) xa[s] = xan[s];
            m2 = m2n; mkv = mkvn;
        };
        // G2 of the requested tile = g2 * [h2 > 0] -> db2 -> bf16 pieces -> image `pm`; 16 steps (k-step s = j >> 3)
        float db1 = 0.f, db2 = 0.f;
        unsigned phi[4], pmid[4], plo[4];
        float sp_ra = 0.f, sp_rb = 0.f;
        auto split_a = [&](float a_, float b_, unsigned& hi) {
            hi = bf16_pair(a_, b_);
            sp_ra = a_ - __uint_as_float(hi << 16);
            sp_rb = b_ - __uint_as_float(hi & 0xffff0000u);
        };
        auto split_b = [&](unsigned& mid, unsigned& lo) {
            mid = bf16_pair(sp_ra, sp_rb);
            lo = bf16_pair(sp_ra - __uint_as_float(mid << 16), sp_rb - __uint_as_float(mid & 0xffff0000u));
        };
        auto g2_step = [&](int j, int pm) {
            const int s = j >> 3, q = j & 7;
            if (q < 2) {
                const unsigned m2w = m2 >> (16 * (w & 1));
#pragma unroll
                for (int r = 8 * s + 4 * q; r < 8 * s + 4 * q + 4; ++r) {
                    const int t = __builtin_amdgcn_sbfe(m2w, r, 1);
                    g2[r] = __uint_as_float(__float_as_uint(g2[r]) & (unsigned)t);
                    db2 += g2[r];
                }
            } else if (q < 6) {
                const int d = q - 2;
                split_a(g2[8 * s + 2 * d], g2[8 * s + 2 * d + 1], phi[d]);
                split_b(pmid[d], plo[d]);
            } else {
                const int g = 2 * s + (q - 6), d = 2 * (g & 1);
                unsigned char* dst = smem + mw_off + pm * F5_IMG + (((2 * g + h) ^ swz_w) * 8);
                *reinterpret_cast<uint2*>(dst) = make_uint2(phi[d], phi[d + 1]);
                *reinterpret_cast<uint2*>(dst + 8192) = make_uint2(pmid[d], pmid[d + 1]);
                *reinterpret_cast<uint2*>(dst + 16384) = make_uint2(plo[d], plo[d + 1]);
            }
        };
        // H1 = relu(W1 x + b1) of the requested tile -> pieces -> image `pm`; h1_mma, then 10 steps (k-step s = j / 5)
        f32x16 hacc;
        auto h1_mma = [&]() {
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = b1v;
#pragma unroll
            for (int s = 0; s < NS; ++s) hacc = mfma32(xa[s], w1v[s], hacc);
        };
        auto h1_step = [&](int j, int pm) {
            const int s = j / 5, q = j % 5;
            if (q < 4) {
                split_a(f5_relu(hacc[8 * s + 2 * q]), f5_relu(hacc[8 * s + 2 * q + 1]), phi[q]);
                split_b(pmid[q], plo[q]);
            } else {
                u32x4* dst = reinterpret_cast<u32x4*>(smem + h_off + pm * F5_IMG + (s * 3 * 64) * 16);
                dst[0] = (u32x4){phi[0], phi[1], phi[2], phi[3]};
                dst[64] = (u32x4){pmid[0], pmid[1], pmid[2], pmid[3]};
                dst[128] = (u32x4){plo[0], plo[1], plo[2], plo[3]};
            }
        };
        // the x values (wave s: column pair s) and the sign words of the requested tile -> LDS
        auto stage = [&](int xslot, int pm) {
            if (w < NS) reinterpret_cast<float*>(smem + F5_XS + xslot * 1024)[n * 8 + 2 * w + h] = w == 0 ? xa[0] : (w == 1 ? xa[1] : (w == 2 ? xa[2] : xa[NS - 1]));
            if (ctid < 128) reinterpret_cast<unsigned*>(smem + F5_MK + pm * 512)[ctid] = mkv;
        };
        auto load_b = [&](u32x4 (&o)[3], int kb, int pm) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const f5_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f5_lds_s16x4*)(smem + mr[0] + pm * F5_IMG + kb * 1024 + p * 8192));
                const f5_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f5_lds_s16x4*)(smem + mr[1] + pm * F5_IMG + kb * 1024 + p * 8192));
                const uint2 x = __builtin_bit_cast(uint2, lo4), y = __builtin_bit_cast(uint2, hi4);
                o[p] = (u32x4){x.x, x.y, y.x, y.y};
            }
        };

        F5_BARRIER();                                          // the table
        // ---- prologue: the first tile's images, the second tile's requests ----
        req_g(bx);
        req_x(bx);
#pragma unroll
        for (int j = 0; j < 16; ++j) g2_step(j, 0);
        h1_mma();
#pragma unroll
        for (int j = 0; j < 10; ++j) h1_step(j, 0);
        stage(0, 0);
        req_g(bx + nwg);
        req_x(bx + nwg);
        F5_STAMP(15);
        F5_BARRIER();

        for (int it = 0; it < nit; ++it) {
            const int par = it & 1, tile = bx + it * nwg;
            F5_STAMP(0);
            // layer B: G1 = (G2 W2) * [h1 > 0], 48 products into one accumulator (the five small products of every k-block first,
            // the eight hi x hi on top: encoder_bwd3.hip); between them the NEXT tile's G2 and H1 images and the requests of the
            // tile after it
            f32x16 acc;
            u32x4 opa[2][3], ahi[8];                          // (ahi: the activations' hi pieces, kept for the eight hi x hi products)
            unsigned mkw[16];
            const unsigned* mk1 = reinterpret_cast<const unsigned*>(smem + mk_off + par * 512);
            auto fill_a = [&](int sl) {
                if (sl < 40 && sl % 5 == 0) {                  // the next k-block's operands
                    const int kb = sl / 5 + 1;
                    if (kb < 8) load_b(opa[kb & 1], kb, par);
                    return;
                }
                if (sl == 38 || sl == 39) return;
                const int f = sl < 40 ? sl - sl / 5 - 1 : sl - 10;     // free steps: 0 .. 29 under the small products, 30 .. 37 under the hi x hi ones
                if (PIML_F5_PHASED) {                          // the requests of the tile after the next; this tile's sign words (late: used right behind)
                    if (PIML_F5_REQ_EARLY && f >= 2 && f < 14 && !(f & 1) && !(PIML_F5_SKIP & 128)) req_step(f / 2 - 1, tile + 2 * nwg);
                    if (f >= 30 && f < 34) {
#pragma unroll
                        for (int i = 4 * (f - 30); i < 4 * (f - 30) + 4; ++i) mkw[i] = mk1[2 * f5_rho(i)];
                    }
                    return;
                }
                if (f < 16) { if (!(PIML_F5_SKIP & 4)) g2_step(f, par ^ 1); }
                else if (f < 20) {
#pragma unroll
                    for (int i = 4 * (f - 16); i < 4 * (f - 16) + 4; ++i) mkw[i] = mk1[2 * f5_rho(i)];
                }
                else if (f == 20) { if (!(PIML_F5_SKIP & 8)) h1_mma(); }
                else if (f >= 23 && f < 33) { if (!(PIML_F5_SKIP & 8)) h1_step(f - 23, par ^ 1); }
                else if (f == 33) stage((it + 1) % 3, par ^ 1);
                else if (f == 34) { if (!(PIML_F5_SKIP & 128)) req_g(tile + 2 * nwg); }
                else if (f == 35) { if (!(PIML_F5_SKIP & 128)) req_x(tile + 2 * nwg); }
            };
            load_b(opa[0], 0, par);
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const u32x4 (&o)[3] = opa[kb & 1];
                ahi[kb] = o[0];
#define F5_MA(X) do { if (!(PIML_F5_SKIP & 64)) { X; } else if (kb == 0) acc = zero16; } while (0)
                F5_SLOT(F5_MA(acc = mfma_bf(o[2], wh[kb], kb == 0 ? zero16 : acc)), fill_a(kb * 5 + 0));
                F5_SLOT(F5_MA(acc = mfma_bf(o[1], wm[kb], acc)), fill_a(kb * 5 + 1));
                F5_SLOT(F5_MA(acc = mfma_bf(o[0], wl[kb], acc)), fill_a(kb * 5 + 2));
                F5_SLOT(F5_MA(acc = mfma_bf(o[1], wh[kb], acc)), fill_a(kb * 5 + 3));
                F5_SLOT(F5_MA(acc = mfma_bf(o[0], wm[kb], acc)), fill_a(kb * 5 + 4));
            }
#ifdef PIML_F5_ACC2
            f32x16 acc2 = zero16;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) F5_SLOT(if (kb & 1) acc2 = mfma_bf(ahi[kb], wh[kb], acc2); else acc = mfma_bf(ahi[kb], wh[kb], acc), fill_a(40 + kb));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
#else
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) F5_SLOT(F5_MA(acc = mfma_bf(ahi[kb], wh[kb], acc)), fill_a(40 + kb));
#endif
            F5_STAMP(1);
            // G1: mask, db1, the lane's 16 rows of its feature -> the G1 tile
            if (!(PIML_F5_SKIP & 16))
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = __builtin_amdgcn_sbfe(mkw[r], bp, 1);
                acc[r] = __uint_as_float(__float_as_uint(acc[r]) & (unsigned)t);
                db1 += acc[r];
            }
            if (!(PIML_F5_SKIP & 16)) {
                float4* dst = reinterpret_cast<float4*>(smem + t_off + par * F5_TBYTES);
#pragma unroll
                for (int g = 0; g < 4; ++g) dst[2 * g] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            }
            F5_STAMP(2);
            if (PIML_F5_PHASED) {
                // the vector half of the tile, while crew B's products hold the matrix pipe: the NEXT tile's H1 (its three f32
                // products first: they queue behind crew B's) and G2 images, the requests of the tile after it
                if (!(PIML_F5_SKIP & 8)) h1_mma();
                if (!(PIML_F5_SKIP & 4)) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) g2_step(j, par ^ 1);
                }
                F5_STAMP(6);
                if (!(PIML_F5_SKIP & 8)) {
#pragma unroll
                    for (int j = 0; j < 10; ++j) h1_step(j, par ^ 1);
                }
                F5_STAMP(7);
                stage((it + 1) % 3, par ^ 1);
                if (!(PIML_F5_SKIP & 128)) {
                    if (PIML_F5_REQ_EARLY) req_take();
                    else { req_g(tile + 2 * nwg); req_x(tile + 2 * nwg); }
                }
                F5_STAMP(5);
            }
            F5_BARRIER();
        }
        F5_STAMP(3);
        F5_BARRIER();                                          // (crew B's two trailing barriers)
        F5_BARRIER();
        db1 += __shfl_xor(db1, 32, 64);
        db2 += __shfl_xor(db2, 32, 64);
        if (h == 0) {
            P[EH * EH + 1024 + 32 * w + n] = db2;
            P[EH * EH + 1024 + EH + 32 * w + n] = db1;
        }
        for (unsigned cc = IN * 128 + ctid; cc < 1024; cc += 256) P[EH * EH + cc] = 0.f;        // the unused tail of the dW1 field
        F5_STAMP(11);
    } else {
        // =========================================== crew B: the weight gradients ===========================================
        f32x16 c[4], sm[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { c[jb][r] = 0.f; sm[jb][r] = 0.f; }
        float w1acc[INC];
#pragma unroll
        for (int cc = 0; cc < INC; ++cc) w1acc[cc] = 0.f;
        float w1g[8];                                          // B operand of the g_x products: W1[32 w + 4 j + (lane >> 4)][lane & 15]
        {
            const float* W1r = J.packed + PACK_FWD + 32768;     // W1 rows padded to 8 columns (zeros beyond in_dim)
#pragma unroll
            for (int j = 0; j < 8; ++j) w1g[j] = (lane & 15) < 8 ? W1r[(32 * w + 4 * j + (lane >> 4)) * 8 + (lane & 15)] : 0.f;
        }
        const int fw = 32 * w + n;
        const int swz_w = (fw >> 1) & 7;
        const int mw_off = F5_M + fw * 64;                     // this lane's own feature of the G2 image (A fragments of dW2)
        const int t_row = F5_T + fw * (F5_TROW * 4) + (4 * h) * 4;        // this lane's feature of the G1 tile, rows 4 h ..
        const int t_col = F5_T + (32 * w) * (F5_TROW * 4) + n * 4;        // g_x: feature 32 w + f, row n: + f * 144
        const int w1_off = F5_W1 + (32 * w) * 32 + h * 16;                // g_x: W1[32 w + f][4 h .. 4 h + 3]: + f * 32
        const int xs_off = F5_XS + (4 * h) * 32;                          // dW1: x rows 4 h ..

        auto load_g2 = [&](u32x4 (&g)[3], int s_, int pm) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint2 x = *reinterpret_cast<const uint2*>(smem + mw_off + pm * F5_IMG + p * 8192 + (((4 * s_ + h) ^ swz_w) * 8));
                const uint2 y = *reinterpret_cast<const uint2*>(smem + mw_off + pm * F5_IMG + p * 8192 + (((4 * s_ + 2 + h) ^ swz_w) * 8));
                g[p] = (u32x4){x.x, x.y, y.x, y.y};
            }
        };
        auto load_h = [&](u32x4 (&o)[3], int u_, int pm) {
            const int s_ = u_ >> 2, jb_ = u_ & 3;
            const u32x4* hr = reinterpret_cast<const u32x4*>(smem + F5_H + pm * F5_IMG) + lane;
#pragma unroll
            for (int p = 0; p < 3; ++p) o[p] = hr[((jb_ * 2 + s_) * 3 + p) * 64];
        };
        // g_x of a finished tile: the four waves' partials, fixed order; crew thread (row ctid >> 3, column ctid & 7)
        auto gx_store = [&](int tile, int pm) {
            const unsigned grow = (unsigned)tile * 32u + (unsigned)(ctid >> 3), cs = (unsigned)ctid & 7u;
            float v;
            if (PIML_F5_GX_MFMA) {                           // [wave][row][16]: column = slot
                const float* gp = reinterpret_cast<const float*>(smem + F5_GXP + pm * F5_GXP_BYTES) + (ctid >> 3) * 16 + (ctid & 7);
                v = ((gp[0] + gp[512]) + gp[1024]) + gp[1536];
            } else {
                const float* gp = reinterpret_cast<const float*>(smem + F5_GXP + pm * F5_GXP_BYTES) + ctid;
                v = ((gp[0] + gp[256]) + gp[512]) + gp[768];
            }
            // (vector form, INC == 6: slot 4 h + j of a row's eight holds column 3 h + j, slots 3 and 7 nothing)
            const unsigned cx = (INC == 6 && !PIML_F5_GX_MFMA) ? 3u * (cs >> 2) + (cs & 3u) : cs;
            const bool cok = (INC == 6 && !PIML_F5_GX_MFMA) ? (cs & 3u) != 3u : cx < IN;
            const unsigned off = (grow * IN + cx) * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_gx, (int)(((tile >= 0) & (tile < ntiles) & (grow < R) & cok) ? off : kOut), 0, 0);
        };
        // the lagging vector work on the G1 tile `pm` and x ring slot `xslot`
#ifndef PIML_F5_GXD
#define PIML_F5_GXD 1
#endif
        constexpr int GXR = PIML_F5_GXD + 1;                   // ring: loads PIML_F5_GXD steps ahead of their use
        float gx[4];
        float tv[GXR][2];
        float4 tw[GXR][2];
        auto gx_load = [&](int k, int pm) {                    // features 2 k, 2 k + 1 of the wave's block
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (PIML_F5_SKIP & 512) tv[k % GXR][e] = (float)k; else
                tv[k % GXR][e] = *reinterpret_cast<const float*>(smem + t_col + pm * F5_TBYTES + (2 * k + e) * (F5_TROW * 4));
                if (PIML_F5_SKIP & 256) tw[k % GXR][e] = make_float4(1.f, 2.f, 3.f, (float)k); else
                tw[k % GXR][e] = *reinterpret_cast<const float4*>(smem + w1_off + (2 * k + e) * 32);
            }
        };
        auto gx_fma = [&](int k) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float v = tv[k % GXR][e];
                const float4 wv = tw[k % GXR][e];
                gx[0] = __fmaf_rn(wv.x, v, gx[0]); gx[1] = __fmaf_rn(wv.y, v, gx[1]);
                gx[2] = __fmaf_rn(wv.z, v, gx[2]);
                if (INC == 8) gx[3] = __fmaf_rn(wv.w, v, gx[3]);
            }
        };
        float g1[16];
        float4 xv[2][4];
        auto g1_load = [&](int half_, int pm) {
            const float4* src = reinterpret_cast<const float4*>(smem + t_row + pm * F5_TBYTES);
#pragma unroll
            for (int g = 2 * half_; g < 2 * half_ + 2; ++g) {
                const float4 v = src[2 * g];
                g1[4 * g] = v.x; g1[4 * g + 1] = v.y; g1[4 * g + 2] = v.z; g1[4 * g + 3] = v.w;
            }
        };
        auto x_load = [&](int k, int xslot) {                  // rows (registers) 2 k, 2 k + 1
            const float4* xr = reinterpret_cast<const float4*>(smem + xs_off + xslot * 1024);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                xv[k & 1][2 * e] = xr[2 * f5_rho(2 * k + e)];
                if (INC == 8) xv[k & 1][2 * e + 1] = xr[2 * f5_rho(2 * k + e) + 1];
                else {
                    const float2 t2 = *reinterpret_cast<const float2*>(&xr[2 * f5_rho(2 * k + e) + 1]);
                    xv[k & 1][2 * e + 1] = make_float4(t2.x, t2.y, 0.f, 0.f);
                }
            }
        };
        auto x_fma = [&](int k) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float4 xa4 = xv[k & 1][2 * e], xb4 = xv[k & 1][2 * e + 1];
                const float g = g1[2 * k + e];
                w1acc[0] = __fmaf_rn(g, xa4.x, w1acc[0]); w1acc[1] = __fmaf_rn(g, xa4.y, w1acc[1]);
                w1acc[2] = __fmaf_rn(g, xa4.z, w1acc[2]); w1acc[3] = __fmaf_rn(g, xa4.w, w1acc[3]);
                w1acc[4] = __fmaf_rn(g, xb4.x, w1acc[4]); w1acc[5] = __fmaf_rn(g, xb4.y, w1acc[5]);
                if (INC == 8) { w1acc[INC - 2] = __fmaf_rn(g, xb4.z, w1acc[INC - 2]); w1acc[INC - 1] = __fmaf_rn(g, xb4.w, w1acc[INC - 1]); }
            }
        };
        // 27 steps: g_x (features two by two, loads a step ahead), its partial, dW1 (rows two by two, loads a step ahead)
        // g_x on the f32 matrix instruction: D[row][c] = sum over the wave's 32 features of G1[row][f] W1[f][c], two chains (rows
        // 0 .. 15 / 16 .. 31) of eight products (four features each); operand A: lane (m = row & 15, k = lane >> 4) reads
        // G1[feature 4 j + k][row] from the tile, operand B: lane (k, n = column) holds W1[4 j + k][n] (0 beyond column 7)
        f32x4 gd[2];
        float ga[8][2];                                        // (all sixteen operands requested up front: 8 ds_read2_b32)
        const int ga_off = F5_T + (32 * w + (lane >> 4)) * (F5_TROW * 4) + (lane & 15) * 4;      // + 4 j features, + 16 r rows
        auto gm_load = [&](int j, int pm) {
#pragma unroll
            for (int r = 0; r < 2; ++r) ga[j][r] = *reinterpret_cast<const float*>(smem + ga_off + pm * F5_TBYTES + j * (4 * F5_TROW * 4) + r * 64);
        };
        // dW1 the same way: D[feature][c] += sum over the tile's rows of G1[row][feature] x[row][c]; chain c2 = features 16 c2 .. + 15
        // of the wave's block; operand A: lane (m = feature & 15, k = lane >> 4) reads G1[feature][row 4 j + k], operand B: lane
        // (k, n = column) reads x[row 4 j + k][n] -- columns 8 .. 15 read zeros (the former W1 region)
        f32x4 dw1m[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        float da[8][2], dbv[8];
        const int da_off = F5_T + (32 * w + (lane & 15)) * (F5_TROW * 4) + (lane >> 4) * 4;      // + 16 c2 features, + 4 j rows
        const int db_off = (lane & 15) < 8 ? F5_XS + (lane >> 4) * 32 + (lane & 15) * 4 : F5_W1 + (lane >> 4) * 32;
        auto dwm_step = [&](int f, int pm, int xslot) {       // 0 .. 9
            if (f < 2) {
#pragma unroll
                for (int j = 4 * f; j < 4 * f + 4; ++j) {
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) da[j][c2] = *reinterpret_cast<const float*>(smem + da_off + pm * F5_TBYTES + c2 * (16 * F5_TROW * 4) + j * 16);
                    dbv[j] = *reinterpret_cast<const float*>(smem + db_off + xslot * 1024 + j * 128);
                }
            } else {
                const int j = f - 2;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) dw1m[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[j][c2], dbv[j], dw1m[c2], 0, 0, 0);
            }
        };
        // (these products ride between crew B's OWN products, not in its vector half: there crew A's chain of 51 dependent products
        // owns the matrix pipe -- the older wave wins the arbitration every time -- and crew B's in-order stream stood behind its
        // first g_x product for the whole of crew A's chain: 4.5 k cycles for the vector half instead of 3.5 k)
        auto gxm_step = [&](int f, int pm) {                  // 0 .. 9; pm: the G1 tile
            if (!GX) return;
            if (f == 0) {
                gd[0] = gd[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) gm_load(j, pm);
            } else if (f <= 8) {
                const int j = f - 1;
#pragma unroll
                for (int r = 0; r < 2; ++r) gd[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[j][r], w1g[j], gd[r], 0, 0, 0);
            } else if (f == 9) {                              // rows 16 r + 4 (lane >> 4) + i, column lane & 15
                float* dst = reinterpret_cast<float*>(smem + F5_GXP + (pm ^ 1) * F5_GXP_BYTES) + (w * 32 + 4 * (lane >> 4)) * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[(16 * r + i) * 16] = gd[r][i];
            }
        };
        auto lag_step_m = [&](int f, int pm, int xslot) {     // dW1 alone: 10 steps
            if (PIML_F5_DW1_MFMA) return;
            if (f == 0) { g1_load(0, pm); g1_load(1, pm); x_load(0, xslot); }
            else if (f == 1) { x_load(1, xslot); x_fma(0); }
            else if (f < 9) { if (f < 8) x_load(f, xslot); x_fma(f - 1); }
        };
        auto lag_step_v = [&](int f, int pm, int xslot) {
            if (f == 0) {
                gx[0] = gx[1] = gx[2] = gx[3] = 0.f;
                if (GX && !(PIML_F5_SKIP & 2)) {
#pragma unroll
                    for (int k = 0; k < PIML_F5_GXD; ++k) gx_load(k, pm);
                }
            } else if (f <= 16) {
                if (GX && !(PIML_F5_SKIP & 2)) { if (f - 1 + PIML_F5_GXD < 16) gx_load(f - 1 + PIML_F5_GXD, pm); gx_fma(f - 1); }
                if (f == 15) g1_load(0, pm);
                if (f == 16) { g1_load(1, pm); x_load(0, xslot); }
            } else if (f == 17) {
                if (GX) reinterpret_cast<float4*>(smem + F5_GXP + (pm ^ 1) * F5_GXP_BYTES)[(w * 32 + n) * 2 + h] = make_float4(gx[0], gx[1], gx[2], gx[3]);
                x_load(1, xslot); x_fma(0);
            } else if (f < 25) {
                if (f < 24) x_load(f - 16, xslot);
                x_fma(f - 17);
            }
        };
        auto lag_step = [&](int f, int pm, int xslot) {
            if (PIML_F5_GX_MFMA) lag_step_m(f, pm, xslot); else lag_step_v(f, pm, xslot);
        };

#ifdef PIML_F5_PRIO_B
        __builtin_amdgcn_s_setprio(PIML_F5_PRIO_B);
#endif
        F5_BARRIER();                                          // the table (crew A's prologue barrier)
        F5_STAMP(15);
        F5_BARRIER();
        for (int it = 0; it < nit; ++it) {
            const int par = it & 1, tile = bx + it * nwg;
            F5_STAMP(0);
            // dW2 += G2^T H1: 48 products (8 groups u = 4 s + jb of six) into the slab's accumulators; between them the vector work
            // on the tile BEFORE (its G1 tile is in the other buffer, its x rows two ring slots back) and the g_x store of the one
            // before that
            u32x4 g2f[3], opb[2][3];
            const int lpm = par ^ 1, xslot = (it + 2) % 3;
            auto fill_b = [&](int sl) {
                if (sl % 6 == 0) {
                    const int u_ = sl / 6 + 1;
                    if (u_ < 8) load_h(opb[u_ & 1], u_, par);
                    return;
                }
                if (sl == 23) { load_g2(g2f, 1, par); return; }   // (behind the last product of k-step 0)
                const int f = sl - sl / 6 - 1 - (sl > 23);         // 39 free steps
                if (PIML_F5_PHASED) {
                    if (PIML_F5_GX_MFMA && !(PIML_F5_SKIP & 2)) {
                        if (f == 1) gxm_step(0, lpm);
                        else if (f >= 6 && f < 14) gxm_step(f - 5, lpm);
                        else if (f == 22) gxm_step(9, lpm);
                    }
                    if (PIML_F5_DW1_MFMA && !(PIML_F5_SKIP & 1)) {
                        if (f == 2 || f == 3) dwm_step(f - 2, lpm, xslot);
                        else if (f >= 14 && f < 22) dwm_step(f - 12, lpm, xslot);
                    }
                    return;
                }
                if (PIML_F5_SKIP & 1) return;
                if (f == 0) { if (GX) gx_store(tile - 2 * nwg, par ^ 1); }
                else if (f < 28) lag_step(f - 1, lpm, xslot);
            };
            if (PIML_F5_PHASED && !(PIML_F5_SKIP & 1)) {
                // the vector half first, while crew A's products hold the matrix pipe
                if (GX) gx_store(tile - 2 * nwg, par ^ 1);
#pragma unroll
                for (int f = 0; f < 17; ++f) lag_step(f, lpm, xslot);
                F5_STAMP(6);
#pragma unroll
                for (int f = 17; f < 27; ++f) lag_step(f, lpm, xslot);
                F5_STAMP(5);
            }
#ifdef PIML_F5_PRIO_PROD
            __builtin_amdgcn_s_setprio(PIML_F5_PRIO_PROD);      // the products' issue slots in front of crew A's vector stream (the older wave)
#endif
            load_g2(g2f, 0, par);
            load_h(opb[0], 0, par);
#pragma unroll
            for (int u = 0; u < 8; ++u) {                          // u = 4 s + jb
                const int jb = u & 3;
                const u32x4 (&o)[3] = opb[u & 1];
#define F5_MB(X) do { if (!(PIML_F5_SKIP & 32)) { X; } } while (0)
                F5_SLOT(F5_MB(sm[jb] = mfma_bf(g2f[2], o[0], sm[jb])), fill_b(u * 6 + 0));
                F5_SLOT(F5_MB(sm[jb] = mfma_bf(g2f[1], o[1], sm[jb])), fill_b(u * 6 + 1));
                F5_SLOT(F5_MB(sm[jb] = mfma_bf(g2f[0], o[2], sm[jb])), fill_b(u * 6 + 2));
                F5_SLOT(F5_MB(sm[jb] = mfma_bf(g2f[1], o[0], sm[jb])), fill_b(u * 6 + 3));
                F5_SLOT(F5_MB(sm[jb] = mfma_bf(g2f[0], o[1], sm[jb])), fill_b(u * 6 + 4));
                F5_SLOT(F5_MB(c[jb] = mfma_bf(g2f[0], o[0], c[jb])), fill_b(u * 6 + 5));
            }
#ifdef PIML_F5_PRIO_PROD
            __builtin_amdgcn_s_setprio(0);
#endif
            F5_STAMP(1);
            F5_BARRIER();
        }
        F5_STAMP(3);
        // the last tile's vector work, the last two g_x stores
        {
            const int it = nit, par = it & 1, tile = bx + it * nwg;
            if (GX) gx_store(tile - 2 * nwg, par ^ 1);
#pragma unroll
            for (int f = 0; f < 27; ++f) lag_step(f, par ^ 1, (it + 2) % 3);
            if (PIML_F5_GX_MFMA) {
#pragma unroll
                for (int f = 0; f < 10; ++f) gxm_step(f, par ^ 1);
            }
            if (PIML_F5_DW1_MFMA) {
#pragma unroll
                for (int f = 0; f < 10; ++f) dwm_step(f, par ^ 1, (it + 2) % 3);
            }
            F5_BARRIER();
            if (GX) gx_store(tile - nwg, par);
            F5_BARRIER();
        }
        F5_STAMP(4);
        // ---- the slot: dW2 | dW1 ----
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[(size_t)(32 * w + f5_rho(r) + 4 * h) * EH + 32 * jb + n] = c[jb][r] + sm[jb][r];
        if (PIML_F5_DW1_MFMA) {                                // register i of chain c2: feature 32 w + 16 c2 + 4 (lane >> 4) + i, column lane & 15
            if ((unsigned)(lane & 15) < IN) {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int i = 0; i < 4; ++i) P[EH * EH + (size_t)(32 * w + 16 * c2 + 4 * (lane >> 4) + i) * IN + (lane & 15)] = dw1m[c2][i];
            }
        } else {
#pragma unroll
            for (int cc = 0; cc < INC; ++cc) w1acc[cc] += __shfl_xor(w1acc[cc], 32, 64);
            if (h == 0) {
                float* o = P + EH * EH + (size_t)(32 * w + n) * IN;
#pragma unroll
                for (int cc = 0; cc < INC; ++cc)
                    if ((unsigned)cc < IN) o[cc] = w1acc[cc];
            }
        }
        F5_STAMP(11);
    }
#ifdef PIML_F5_STAMPS
    if (lane == 0)
        for (int i = 0; i < 16; ++i) g_f5_stamps[(blockIdx.x * 8 + wv) * 16 + i] = st[i];
#endif
}

#ifdef PIML_F5_STAMPS
extern "C" __attribute__((visibility("default"))) int piml_f5_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f5_stamps), sizeof(unsigned long long) * 256 * 8 * 16);
}
#endif

int enc_f5_set_attributes() {
    auto set = [&](const void* f) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, F5_LDS_LAUNCH); };
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_sums2_kernel<false, 6>))) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_sums2_kernel<true, 6>))) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_sums2_kernel<false, 8>))) return e;
    return set(reinterpret_cast<const void*>(enc_bwd_sums2_kernel<true, 8>));
}

// the PIML_POOL_TRAIN backward (g_pooled = the gradient of the agents' sums; checked by the caller); false: a shape this kernel
// does not take (k > 16) -- the caller launches enc_f3_launch(..., sums = true)
bool enc_f5_launch(const EncArgs& A, const int* nA, hipStream_t s) {
    F5Args F;
    F.A = A;
    F.nA[0] = nA[0]; F.nA[1] = A.nbr > 1 ? nA[1] : 0;
    bool in6 = true;
    for (int i = 0; i < A.nbr; ++i) {
        if (A.br[i].k > F5_KMAX) return false;
        in6 = in6 && A.br[i].in_dim == 6;
    }
    const bool gx = A.br[0].g_x != nullptr;
    const dim3 g((unsigned)(F.nA[0] + F.nA[1])), blk(F5_THREADS);
    if (in6) {
        if (gx) hipLaunchKernelGGL((enc_bwd_sums2_kernel<true, 6>), g, blk, F5_LDS_LAUNCH, s, F);
        else hipLaunchKernelGGL((enc_bwd_sums2_kernel<false, 6>), g, blk, F5_LDS_LAUNCH, s, F);
    } else {
        if (gx) hipLaunchKernelGGL((enc_bwd_sums2_kernel<true, 8>), g, blk, F5_LDS_LAUNCH, s, F);
        else hipLaunchKernelGGL((enc_bwd_sums2_kernel<false, 8>), g, blk, F5_LDS_LAUNCH, s, F);
    }
    return true;
}

}  // namespace piml
