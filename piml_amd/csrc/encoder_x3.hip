// The PINNSF encoder stages on SPLIT bf16 products: every f32 product w * x of the two 128 x 128 layers is evaluated as
//     (w_hi + w_mid + w_lo) * (x_hi + x_mid + x_lo),   each piece a bf16, the three summing to the f32 value EXACTLY
// (pack.hpp: split3), keeping the six partial products whose magnitude is >= 2^-16 of the full one:
//     w_lo x_hi + w_mid x_mid + w_hi x_lo  +  w_mid x_hi + w_hi x_mid  +  w_hi x_hi.
// A bf16 x bf16 product is exact in f32 and the matrix core accumulates in f32, so what is dropped (w_mid x_lo, w_lo x_mid,
// w_lo x_lo) is <= 3 * 2^-24 of |w x| per product: the rounding of ONE f32 multiply.  The results sit as close to the
// float64 product as those of the f32 matrix-core kernels in encoder.hip do (tests/test_encoder_gpu.py measures both) --
// this is f32 arithmetic carried by v_mfma_f32_32x32x16_bf16, which retires a 32x32x16 block in 32 cycles where
// v_mfma_f32_32x32x2_f32 needs 8 x 64: six of them per k-block = 192 cycles against 512.
//
// Same formulation as encoder.hip (transposed product, a tile's activations stay in registers between layers):
// accumulator registers 8 s .. 8 s + 7 of block bp, split into their three pieces and packed pairwise, ARE the B operand
// of k-block kb = 2 bp + s of the next layer (k order inside the block: element t of lane half h = feature
// 16 kb + 8 (t >> 2) + 4 h + (t & 3)); the weights are packed once per step in that k order, already split.
// Reference arithmetic: src/models/model.py:40-65, :1271-1283 (see encoder.hip).
#include <cstdlib>

#include "common.hpp"
#include "encoder.hpp"
#include "philox.hpp"
#include "x3.hpp"

namespace piml {

// the three pieces of a whole 32 x 128 activation tile: B operands of the 8 k-blocks
struct Pieces {
    u32x4 hi[8], mid[8], lo[8];
};

__device__ __forceinline__ void split_tile(const f32x16 (&in)[4], Pieces& P) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
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

// bit 16 u + r = (register r of block u is positive); two instructions per value (compare into vcc, add-with-carry
// m = 2 m + vcc), the last register first
__device__ __forceinline__ unsigned sign_bits(const f32x16& b0, const f32x16& b1) {
    unsigned m = 0;
#pragma unroll
    for (int r = 15; r >= 0; --r) asm volatile("v_cmp_gt_f32 vcc, %1, 0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(b1[r]) : "vcc");
#pragma unroll
    for (int r = 15; r >= 0; --r) asm volatile("v_cmp_gt_f32 vcc, %1, 0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(b0[r]) : "vcc");
    return m;
}
// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// LDS (u32x4 units unless noted): W2 image whole [HM 4096 | LO 2048] | W3 HM of fragments 0 .. 28 [29][2][64] |
// W1 fragments 1024 floats | b1 b2 b3 384 floats   = 163 328 B of the CU's 163 840.  The W3 LO pieces (one of the six
// products reads them) and the hi / mid pieces of W3's last three fragments come from the packed image in L2, requested
// one output block ahead.
constexpr int X3_FB3 = 29;
constexpr int X3_FWD_W3 = X3_IMG / 4;                        // u32x4 offset of the W3 HM part
constexpr int X3_FWD_F32 = X3_IMG + X3_FB3 * 2 * 64 * 4;     // float offset of W1 fragments + biases
constexpr int X3_FWD_LDS_BYTES = (X3_FWD_F32 + 1024 + 384) * 4;
static_assert(X3_FWD_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

constexpr int W3_N4 = X3_FB3 * 2 * 64, W3_ROUNDS = (W3_N4 + ENC_THREADS - 1) / ENC_THREADS;

__device__ __forceinline__ void load_x(float (&xb)[4], const float* __restrict__ x, long long tile, long long ntiles, long long R,
                                       int IN, int lane) {
    const long long row = tile * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 2 * s + (lane >> 5);
        xb[s] = (tile < ntiles && row < R && c < IN) ? x[row * IN + c] : 0.f;
    }
}

__device__ __forceinline__ void land_w3(const u32x4 (&w3r)[W3_ROUNDS], float* lds, int tid) {
    u32x4* dst = reinterpret_cast<u32x4*>(lds) + X3_FWD_W3;
#pragma unroll
    for (int r = 0; r < W3_ROUNDS; ++r) {
        const int e = r * ENC_THREADS + tid;
        if (e < W3_N4) dst[e] = w3r[r];
    }
    __syncthreads();
}

#ifdef PIML_ENC_STAMPS
// diagnostic build only (tools/enc_stamps_fwd.py): shader-clock stamps of thread 0 of every workgroup of the last enc_fwd_sum launch
__device__ unsigned long long g_enc_stamps[512 * 16];
#define ENC_STAMP(i, wait)                                                                         \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (wait) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                      \
        if (threadIdx.x == 0 && blockIdx.x < 512) g_enc_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define ENC_STAMP(i, wait)
#endif

template <int K, int PH>
__device__ __forceinline__ void pool_rows(const f32x16 (&a)[4], long long tile, long long agents, int lane, float* __restrict__ part_a,
                                          float* __restrict__ part_b);
template <bool ROWS>
__device__ __forceinline__ void store_h2_rows(const f32x16 (&a)[4], float* __restrict__ h2, long long tile, long long R, int lane);

__device__ __forceinline__ float add_halves(float x);

// ONE 32-feature block of pool_rows: the sums of the block's 16 registers per agent, stored at once (enc_fwd_x3_kernel<DROP, true>
// cannot hold all four output blocks until the end: its last layer's operands fill the register file)
template <int K, int PH>
__device__ __forceinline__ void pool_block(const f32x16& ab, int blk, long long tile, long long agents, int lane, float* __restrict__ part_a,
                                           float* __restrict__ part_b) {
    constexpr int o = (PH * 32) % K;
    constexpr int S = (o + 31) / K + 1;
    const int i = lane & 31;
    const bool h = lane >= 32;
    const long long a0 = (tile * 32) / K;
    float sum[S];
#pragma unroll
    for (int s = 0; s < S; ++s) sum[s] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m0 = (r & 3) + 8 * (r >> 2), s0 = (o + m0) / K, s1 = (o + m0 + 4) / K;
        if (s0 == s1) sum[s0] += ab[r];
        else {
            sum[s0] += h ? 0.f : ab[r];
            sum[s1] += h ? ab[r] : 0.f;
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) sum[s] = add_halves(sum[s]);
    float* __restrict__ pa = part_a + a0 * EH + 32 * blk + i;
    float* __restrict__ pb = part_b + a0 * EH + 32 * blk + i;
    const int left = (int)(agents - a0 < S ? agents - a0 : S);
    if (!h) {
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s < left) ((s == 0 && o != 0) ? pb : pa)[s * EH] = sum[s];
    }
}
__device__ __forceinline__ void pool_block_k(const f32x16& ab, int blk, int k, long long tile, long long agents, int lane,
                                             float* __restrict__ part_a, float* __restrict__ part_b) {
    if (k == 6) {
        switch ((int)(tile % 3)) {
            case 0: pool_block<6, 0>(ab, blk, tile, agents, lane, part_a, part_b); break;
            case 1: pool_block<6, 1>(ab, blk, tile, agents, lane, part_a, part_b); break;
            default: pool_block<6, 2>(ab, blk, tile, agents, lane, part_a, part_b); break;
        }
    } else if (k == 10) {
        switch ((int)(tile % 5)) {
            case 0: pool_block<10, 0>(ab, blk, tile, agents, lane, part_a, part_b); break;
            case 1: pool_block<10, 1>(ab, blk, tile, agents, lane, part_a, part_b); break;
            case 2: pool_block<10, 2>(ab, blk, tile, agents, lane, part_a, part_b); break;
            case 3: pool_block<10, 3>(ab, blk, tile, agents, lane, part_a, part_b); break;
            default: pool_block<10, 4>(ab, blk, tile, agents, lane, part_a, part_b); break;
        }
    } else {
        pool_block<2, 0>(ab, blk, tile, agents, lane, part_a, part_b);
    }
}
// the block's 16 rows of one lane half, 128 contiguous bytes per row (store_h2_rows for one block)
__device__ __forceinline__ void store_rows_block(const f32x16& ab, int blk, float* __restrict__ out, long long tile, long long R, int lane) {
    const int j = lane & 31, h = lane >> 5;
    float* __restrict__ base = out + (tile * 32 + 4 * h) * EH + 32 * blk + j;
    const long long left = R - tile * 32 - 4 * h;
    if (R - tile * 32 >= 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * EH] = ab[r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((r & 3) + 8 * (r >> 2) < left) base[((r & 3) + 8 * (r >> 2)) * EH] = ab[r];
    }
}

// the register -> agent map of a tile's rows is chosen by (tile mod k): wave-uniform
__device__ __forceinline__ void pool_rows_k(const f32x16 (&a)[4], int k, long long tile, long long agents, int lane, float* __restrict__ part_a,
                                            float* __restrict__ part_b) {
    if (k == 6) {
        switch ((int)(tile % 3)) {
            case 0: pool_rows<6, 0>(a, tile, agents, lane, part_a, part_b); break;
            case 1: pool_rows<6, 1>(a, tile, agents, lane, part_a, part_b); break;
            default: pool_rows<6, 2>(a, tile, agents, lane, part_a, part_b); break;
        }
    } else if (k == 10) {
        switch ((int)(tile % 5)) {
            case 0: pool_rows<10, 0>(a, tile, agents, lane, part_a, part_b); break;
            case 1: pool_rows<10, 1>(a, tile, agents, lane, part_a, part_b); break;
            case 2: pool_rows<10, 2>(a, tile, agents, lane, part_a, part_b); break;
            case 3: pool_rows<10, 3>(a, tile, agents, lane, part_a, part_b); break;
            default: pool_rows<10, 4>(a, tile, agents, lane, part_a, part_b); break;
        }
    } else {
        pool_rows<2, 0>(a, tile, agents, lane, part_a, part_b);
    }
}

// DROP: 0 = no dropout, 1 = keep_bits given, 2 = the kernel draws the p = 0.5 mask itself (one Philox call per row, philox.hpp)
// and leaves it in keep_bits for the backward
// EXCH (round 5, PIML_POOL_MSGS): the LAST layer with exchanged operands (kblock_x3_t: D'[row][feature], lane = feature, registers
// = the tile's rows, as enc_fwd_sum_x3_kernel's layer 2), so that the neighbour-axis sum of the MESSAGES -- which a dropout mask
// keeps from moving in front of this layer -- is additions between registers too: the agents' sums go to sum_a / sum_b, the message
// rows are stored only for a branch that carries `msgs` (the collision head's input), 128 contiguous bytes per (row, block).  A
// row's keep words are drawn / loaded by the lane that owns the row and handed to the lanes that own its features by ds_bpermute.
template <int DROP, bool EXCH = false>
__global__ __launch_bounds__(ENC_THREADS) void enc_fwd_x3_kernel(EncArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(threadIdx.x >> 6));
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int nwg = b ? (int)gridDim.x - A.wg_split : (A.nbr > 1 ? A.wg_split : (int)gridDim.x);
    const long long R = J.rows;
    const int IN = J.in_dim;
    const long long ntiles = (R + 31) >> 5;
    // wave-major: tile t of the first nwg tiles goes to wave 0 of workgroup t, the next nwg tiles to the workgroups' waves 1, ... --
    // below nwg x 8 tiles the work spreads over the CUs (and over their SIMDs) before any SIMD gets a second wave (round 5: at the
    // fine-tuning loop's 488 agents the block-major order kept 31 workgroups of eight waves busy and 225 CUs idle)
    const long long first = (long long)wave * nwg + ((int)blockIdx.x - wg0);
    const long long stride = (long long)nwg * ENC_WAVES;
    if (A.zero)
        for (int e = blockIdx.x * ENC_THREADS + tid; e < A.zero_n; e += gridDim.x * ENC_THREADS) A.zero[e] = 0.f;
    unsigned long long gseed = 0, goff = 0;
    if (DROP == 2) { gseed = A.gen_state[0]; goff = A.gen_state[1]; }
    if ((long long)((int)blockIdx.x - wg0) >= ntiles) {               // whole workgroup idle (not even wave 0 has a tile)
        if (DROP == 2 && tid == 0) dropout_advance(A.gen_state, goff, gridDim.x);
        return;
    }

    const float* x3 = J.packed + PACK_F32;
    // the first tile's input row: requested before the staging loads (vmcnt retires in order)
    float xb[4];
    ENC_STAMP(0, false);
    load_x(xb, J.x, first, ntiles, R, IN, lane);
    stage_linear<X3_IMG>(lds, x3, tid);
    stage_linear<1024 + 384>(lds + X3_FWD_F32, J.packed + 32768, tid);
    __syncthreads();
    // W3's LDS part stays in flight (in registers) behind layers 1 and 2 of the first tile.  (EXCH: landed at once -- with the
    // gather of the keep words those 32 registers no longer fit, and hipcc spilled exactly them)
    u32x4 w3r[W3_ROUNDS];
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(x3 + X3_IMG);
#pragma unroll
        for (int r = 0; r < W3_ROUNDS; ++r) {
            const int e = r * ENC_THREADS + tid;
            w3r[r] = src[e < W3_N4 ? e : 0];
        }
    }
    bool w3_pending = true;
    ENC_STAMP(1, false);
    if (EXCH && DROP) {
        land_w3(w3r, lds, tid);
        w3_pending = false;
    }
    ENC_STAMP(2, false);

    for (long long tile = first; tile < ntiles; tile += stride) {
        // the operand addresses are made opaque per tile: as loop invariants the compiler hoists the bias and fragment
        // reads of the whole tile out of the loop and spills them
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const u32x4* W2hm = reinterpret_cast<const u32x4*>(lds) + lane_t;
        const u32x4* W2lo = W2hm + X3_HM / 4;
        const u32x4* W3hm = reinterpret_cast<const u32x4*>(lds) + X3_FWD_W3 + lane_t;
        const u32x4* W3hm_g = reinterpret_cast<const u32x4*>(x3 + X3_IMG) + lane_t;
        const u32x4* W3lo_g = W3hm_g + X3_HM / 4;
        const float* W1f = lds + X3_FWD_F32;
        const float* bias = W1f + 1024;
        const int j = lane_t & 31, h = lane_t >> 5;
        const long long row = tile * 32 + j;
        const bool valid = row < R;
        f32x16 a[4];
        Pieces P;
        // ---- layer 1: K = in_dim (padded to 8), f32 products (3 % of the tile's matrix cycles) ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + feat0(blk, q, h));
                a[blk][4 * q + 0] = bq.x; a[blk][4 * q + 1] = bq.y; a[blk][4 * q + 2] = bq.z; a[blk][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) a[blk] = mfma32(W1f[(blk * 4 + s) * 64 + lane_t], xb[s], a[blk]);
#pragma unroll
            for (int r = 0; r < 16; ++r) a[blk][r] = relu1(a[blk][r]);
        }
        if (J.h1 && valid) {
            float* o = J.h1 + row * EH;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4_stream(o + feat0(blk, q, h), a[blk][4 * q], a[blk][4 * q + 1], a[blk][4 * q + 2], a[blk][4 * q + 3]);
        }
        // sign bits for the dX chain (relu_mask): lane (row, h), layer L: bit 16 blk + r of a 64-bit word = register r of block blk > 0
        uint2* mrow = J.relu_mask ? reinterpret_cast<uint2*>(J.relu_mask) + (tile * 2) * 64 + lane_t : nullptr;
        if (mrow) mrow[0] = make_uint2(sign_bits(a[0], a[1]), sign_bits(a[2], a[3]));
        ENC_STAMP(3, false);
        split_tile(a, P);
        // ---- layer 2 (a is dead: reused for the outputs) ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x16 acc, sm;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + 128 + feat0(blk, q, h));
                acc[4 * q + 0] = bq.x; acc[4 * q + 1] = bq.y; acc[4 * q + 2] = bq.z; acc[4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = blk * 8 + kb;
                kblock_x3(acc, sm, W2hm[(fb * 2) * 64], W2hm[(fb * 2 + 1) * 64], W2lo[fb * 64], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += sm[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
            a[blk] = acc;
            if (J.h2 && valid) {
                float* o = J.h2 + row * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4_stream(o + feat0(blk, q, h), acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        }
        if (mrow) mrow[64] = make_uint2(sign_bits(a[0], a[1]), sign_bits(a[2], a[3]));
        ENC_STAMP(4, false);
        if (w3_pending) {
            land_w3(w3r, lds, tid);
            w3_pending = false;
        }
        // ---- layer 3 (no activation), msgs = scale * output ----
        u32x4 lw[2][8];              // LO pieces of W3, one output block ahead
        u32x4 tail[6];               // hi / mid of fragments 29 .. 31 (block 3, k-blocks 5 .. 7)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) lw[0][kb] = W3lo_g[kb * 64];
        if (!EXCH) {
#pragma unroll
            for (int u = 0; u < 6; ++u) tail[u] = W3hm_g[(X3_FB3 * 2 + u) * 64];
        }
        load_x(xb, J.x, tile + stride, ntiles, R, IN, lane);       // the next tile's input row
        uint4 kw = make_uint4(0u, 0u, 0u, 0u);
        if (DROP == 1 && valid) kw = reinterpret_cast<const uint4*>(J.keep_bits)[row];
        // EXCH: bit r of keepx[blk] = keep (row rho(r) + 4 h, this lane's feature of block blk) -- the rows' words come from the lanes
        // that own the rows (ds_bpermute), gathered into one word per block IN FRONT of the split, while the layer's 96 operand
        // registers do not exist yet (the draw too: behind the split, where the plain form has it, the two do not fit)
        unsigned keepx[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (EXCH && DROP) {
            if (DROP == 2) {
                const PhiloxOut r = keep_words_fair(gseed, goff, (unsigned)row, (unsigned)b);
                kw = make_uint4(r.x, r.y, r.z, r.w);
                if (valid && h == 0) reinterpret_cast<uint4*>(J.keep_bits)[row] = kw;
            }
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const unsigned kwb = word_of(kw, blk);
                unsigned t = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned wd = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ((r & 3) + 8 * (r >> 2) + 4 * h), (int)kwb);
                    t |= ((wd >> (lane_t & 31)) & 1u) << r;
                }
                keepx[blk] = t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        split_tile(a, P);
        if (!EXCH && DROP == 2) {          // `a` is dead here (64 free registers).  Both lane halves of a row draw the same words; half 0 records them
            const PhiloxOut r = keep_words_fair(gseed, goff, (unsigned)row, (unsigned)b);
            kw = make_uint4(r.x, r.y, r.z, r.w);
            if (valid && h == 0) reinterpret_cast<uint4*>(J.keep_bits)[row] = kw;
        }
        const float scale = J.scale;
        ENC_STAMP(5, false);
        if (EXCH) {
            // (register budget: the LO pieces one block ahead, the last three fragments' hi / mid pieces only inside the block that uses
            // them -- there is no block ahead of it -- and a row's sixteen keep bits gathered into ONE word in front of the products)
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                if (blk < 3) {
#pragma unroll
                    for (int kb = 0; kb < 8; ++kb) lw[(blk + 1) & 1][kb] = W3lo_g[((blk + 1) * 8 + kb) * 64];
                }
                u32x4 tl[6];
                if (blk == 3) {
#pragma unroll
                    for (int u = 0; u < 6; ++u) tl[u] = W3hm_g[(X3_FB3 * 2 + u) * 64];
                }
                const unsigned keepw = keepx[blk];
                f32x16 acc, sm;
                const float bv = bias[256 + 32 * blk + (lane_t & 31)];
#pragma unroll
                for (int r = 0; r < 16; ++r) { sm[r] = 0.f; acc[r] = bv; }
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    const int fb = blk * 8 + kb;
                    const u32x4 wh = fb < X3_FB3 ? W3hm[(fb * 2) * 64] : tl[fb < X3_FB3 ? 0 : (fb - X3_FB3) * 2];
                    const u32x4 wm = fb < X3_FB3 ? W3hm[(fb * 2 + 1) * 64] : tl[fb < X3_FB3 ? 0 : (fb - X3_FB3) * 2 + 1];
                    kblock_x3_t(acc, sm, wh, wm, lw[blk & 1][kb], P.hi[kb], P.mid[kb], P.lo[kb]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[r] + sm[r];
                    if (DROP) v = keep_if(v, keepw, r);
                    acc[r] = scale * v;
                }
                if (J.msgs) store_rows_block(acc, blk, J.msgs, tile, R, lane_t);
                pool_block_k(acc, blk, J.k, tile, R / J.k, lane_t, J.sum_a, J.sum_b);
            }
        } else
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            if (blk < 3) {
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) lw[(blk + 1) & 1][kb] = W3lo_g[((blk + 1) * 8 + kb) * 64];
            }
            f32x16 acc, sm;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + 256 + feat0(blk, q, h));
                acc[4 * q + 0] = bq.x; acc[4 * q + 1] = bq.y; acc[4 * q + 2] = bq.z; acc[4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = blk * 8 + kb;
                const u32x4 wh = fb < X3_FB3 ? W3hm[(fb * 2) * 64] : tail[fb < X3_FB3 ? 0 : (fb - X3_FB3) * 2];
                const u32x4 wm = fb < X3_FB3 ? W3hm[(fb * 2 + 1) * 64] : tail[fb < X3_FB3 ? 0 : (fb - X3_FB3) * 2 + 1];
                kblock_x3(acc, sm, wh, wm, lw[blk & 1][kb], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += sm[r];
            if (DROP) keep_block(acc, word_of(kw, blk), h);
            if (valid) {
                float* o = J.msgs + row * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4_stream(o + feat0(blk, q, h), scale * acc[4 * q], scale * acc[4 * q + 1], scale * acc[4 * q + 2], scale * acc[4 * q + 3]);
            }
        }
    }
    ENC_STAMP(6, false);
    ENC_STAMP(7, false);
    ENC_STAMP(8, true);
    if (w3_pending) land_w3(w3r, lds, tid);       // a wave without a tile: the barrier still counts it
    if (DROP == 2 && tid == 0) dropout_advance(A.gen_state, goff, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------
// forward for INFERENCE (no gradient, no dropout, nobody reads the per-row messages): the neighbour-axis sum taken BEFORE the
// last layer.  msgs = scale (W3 h2 + b3) is linear in h2, so sum_r msgs[r] = scale (W3 sum_r h2[r] + k b3): this kernel stops
// after layer 2 and leaves the agents' sums of h2; the caller folds scale W3 into the decoder's first layer
// (W_d1' = scale W_d1 W3, b_d1' = b_d1 + scale k W_d1 b3: once per weight pack), so layer 3 -- half of the kernel's matrix
// work -- and the 33 MB of messages are gone (rollout frame at 4096 agents 77 -> 58 us: this kernel 17 us against 30).
// Layer 2 runs with its operands exchanged (kblock_x3_t): the output block is D'[row][feature] -- lane = feature, registers
// = rows rho(r) + 4 h of the tile -- so an agent's sum is additions between REGISTERS (no transposition through LDS, the cost
// that sank the round-3 attempt at summing inside the forward), one exchange between the lane halves per agent and block,
// and 128-byte stores.  Which registers belong to which agent depends on (first row of the tile) mod k: three cases for
// k = 6, five for k = 10, each compiled with the register -> agent map as constants (pool_rows<K, PH>; other k: the caller
// keeps the message path).  An agent whose rows straddle two tiles gets its two parts into two buffers (`msgs` = the part
// from the tile its first row lies in, `h2` = the rest; both (agents, 128)): no atomics, no cleared buffer, and the decoder adds
// them ((a k) >> 5 != (a k + k - 1) >> 5 says which agents have a second part).
// ---------------------------------------------------------------------------------------------------------
// x[l] + x[l ^ 32] in every lane: v_permlane32_swap (gfx950) exchanges the upper half of one register with the lower half of
// another -- one vector instruction where __shfl_xor(x, 32) is a ds_bpermute and its wait
__device__ __forceinline__ float add_halves(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int K, int PH>
__device__ __forceinline__ void pool_rows(const f32x16 (&a)[4], long long tile, long long agents, int lane, float* __restrict__ part_a,
                                          float* __restrict__ part_b) {
    constexpr int o = (PH * 32) % K;                 // rows of the tile's first agent that lie in earlier tiles
    constexpr int S = (o + 31) / K + 1;              // agents with rows in this tile
    const int i = lane & 31;
    const bool h = lane >= 32;
    const long long a0 = (tile * 32) / K;
    // Round 5 (in-kernel stamps: the sums cost a wave 4.5 - 5 k clocks): every sum of the tile first, the stores behind them in ONE
    // lane-half region -- as a store per sum each sat in an exec-mask region of its own, which kept the compiler from moving the
    // next sum's lane exchange over it (24 exposed LDS round trips), and the index was 64-bit arithmetic with an `agent < agents`
    // test per store.  `tile` is wave-uniform in the callers, so the addresses are scalar + a lane offset.
    float tot[4][S];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        float sum[S];
#pragma unroll
        for (int s = 0; s < S; ++s) sum[s] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m0 = (r & 3) + 8 * (r >> 2), s0 = (o + m0) / K, s1 = (o + m0 + 4) / K;     // the register's row in half 0 / half 1
            if (s0 == s1) sum[s0] += a[blk][r];
            else {
                sum[s0] += h ? 0.f : a[blk][r];
                sum[s1] += h ? a[blk][r] : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) tot[blk][s] = add_halves(sum[s]);
    }
    float* __restrict__ pa = part_a + a0 * EH + i;
    float* __restrict__ pb = part_b + a0 * EH + i;
    const int left = (int)(agents - a0 < S ? agents - a0 : S);          // agents of the tile that exist (wave-uniform)
    if (!h) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s < left) {
                float* dst = (s == 0 && o != 0) ? pb : pa;
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) dst[s * EH + 32 * blk] = tot[blk][s];
            }
        }
    }
}

__global__ __launch_bounds__(ENC_THREADS) void enc_fwd_pool_x3_kernel(EncArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(threadIdx.x >> 6));      // in an SGPR: `tile` and every address built on it stay scalar
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int nwg = b ? (int)gridDim.x - A.wg_split : (A.nbr > 1 ? A.wg_split : (int)gridDim.x);
    const long long R = J.rows;
    const int IN = J.in_dim;
    const long long ntiles = (R + 31) >> 5;
    // wave-major: tile t of the first nwg tiles goes to wave 0 of workgroup t, the next nwg tiles to the workgroups' waves 1, ... --
    // below nwg x 8 tiles the work spreads over the CUs (and over their SIMDs) before any SIMD gets a second wave (round 5: at the
    // fine-tuning loop's 488 agents the block-major order kept 31 workgroups of eight waves busy and 225 CUs idle)
    const long long first = (long long)wave * nwg + ((int)blockIdx.x - wg0);
    const long long stride = (long long)nwg * ENC_WAVES;
    if (A.zero)
        for (int e = blockIdx.x * ENC_THREADS + tid; e < A.zero_n; e += gridDim.x * ENC_THREADS) A.zero[e] = 0.f;
    if ((long long)((int)blockIdx.x - wg0) >= ntiles) return;        // whole workgroup idle (not even wave 0 has a tile)
    const float* x3 = J.packed + PACK_F32;
    float xb[4];
    load_x(xb, J.x, first, ntiles, R, IN, lane);
    stage_linear<X3_IMG>(lds, x3, tid);                                         // W2's image; W3 is not needed
    stage_linear<1024 + 384>(lds + X3_FWD_F32, J.packed + 32768, tid);
    __syncthreads();
    const int k = J.k;
    const long long agents = R / k;
    for (long long tile = first; tile < ntiles; tile += stride) {
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const u32x4* W2hm = reinterpret_cast<const u32x4*>(lds) + lane_t;
        const u32x4* W2lo = W2hm + X3_HM / 4;
        const float* W1f = lds + X3_FWD_F32;
        const float* bias = W1f + 1024;
        const int h = lane_t >> 5;
        f32x16 a[4];
        Pieces P;
        // ---- layer 1 as in enc_fwd_x3_kernel ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + feat0(blk, q, h));
                a[blk][4 * q + 0] = bq.x; a[blk][4 * q + 1] = bq.y; a[blk][4 * q + 2] = bq.z; a[blk][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) a[blk] = mfma32(W1f[(blk * 4 + s) * 64 + lane_t], xb[s], a[blk]);
#pragma unroll
            for (int r = 0; r < 16; ++r) a[blk][r] = relu1(a[blk][r]);
        }
        split_tile(a, P);
        load_x(xb, J.x, tile + stride, ntiles, R, IN, lane);       // the next tile's input row
        // ---- layer 2, transposed: a[blk] = relu(h2)[row rho(r) + 4 h][feature 32 blk + (lane & 31)] ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x16 acc, sm;
            const float bv = bias[128 + 32 * blk + (lane_t & 31)];
#pragma unroll
            for (int r = 0; r < 16; ++r) { sm[r] = 0.f; acc[r] = bv; }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = blk * 8 + kb;
                kblock_x3_t(acc, sm, W2hm[(fb * 2) * 64], W2hm[(fb * 2 + 1) * 64], W2lo[fb * 64], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) a[blk][r] = relu1(acc[r] + sm[r]);
        }
        // ---- the agents' sums (wave-uniform choice of the register -> agent map) ----
        pool_rows_k(a, k, tile, agents, lane_t, J.msgs, J.h2);
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward for TRAINING on the agents' sums of h2 (PIML_POOL_TRAIN, include/piml_hip.h): enc_fwd_pool_x3_kernel's two layers and
// register sums, plus what a backward pass needs -- the signs of h1 (the row-per-lane words of enc_fwd_x3_kernel) and of h2 in
// the layout of the exchanged layer, which is the layout the one-pass backward masks in (lane = feature, registers = the tile's
// rows: no transposition on either side) -- and, where the branch carries `h2`, the h2 rows themselves for the collision head
// (from the exchanged layout a (row, block) segment is 32 lanes x 4 B = 128 contiguous bytes).
// Reference arithmetic: src/models/model.py:40-65 (layers 1 - 2), :1279-1283 (the sum; layer 3 and the processor scale are folded
// into the decoder's first layer, pack.hpp: fold_w).
// ---------------------------------------------------------------------------------------------------------

template <bool ROWS>
__device__ __forceinline__ void store_h2_rows(const f32x16 (&a)[4], float* __restrict__ h2, long long tile, long long R, int lane) {
    if (!ROWS) return;
    const int j = lane & 31, h = lane >> 5;
    // (round 5, in-kernel stamps: with a 64-bit row index and a bounds test per store, REQUESTING the 64 stores took the pedestrian
    // workgroups -- the launch's critical path -- 8.4 k clocks; now a base pointer, compile-time offsets, a full tile without tests)
    float* __restrict__ base = h2 + (tile * 32 + 4 * h) * EH + j;
    const long long left = R - tile * 32 - 4 * h;          // rows of this lane half's first row on
    if (R - tile * 32 >= 32) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * EH + 32 * blk] = a[blk][r];
    } else {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) < left) base[((r & 3) + 8 * (r >> 2)) * EH + 32 * blk] = a[blk][r];
    }
}

__global__ __launch_bounds__(ENC_THREADS) void enc_fwd_sum_x3_kernel(EncArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(threadIdx.x >> 6));      // in an SGPR: `tile` and every address built on it stay scalar
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int nwg = b ? (int)gridDim.x - A.wg_split : (A.nbr > 1 ? A.wg_split : (int)gridDim.x);
    const long long R = J.rows;
    const int IN = J.in_dim;
    const long long ntiles = (R + 31) >> 5;
    // wave-major: tile t of the first nwg tiles goes to wave 0 of workgroup t, the next nwg tiles to the workgroups' waves 1, ... --
    // below nwg x 8 tiles the work spreads over the CUs (and over their SIMDs) before any SIMD gets a second wave (round 5: at the
    // fine-tuning loop's 488 agents the block-major order kept 31 workgroups of eight waves busy and 225 CUs idle)
    const long long first = (long long)wave * nwg + ((int)blockIdx.x - wg0);
    const long long stride = (long long)nwg * ENC_WAVES;
    if (A.zero)
        for (int e = blockIdx.x * ENC_THREADS + tid; e < A.zero_n; e += gridDim.x * ENC_THREADS) A.zero[e] = 0.f;
    if ((long long)((int)blockIdx.x - wg0) >= ntiles) return;        // whole workgroup idle (not even wave 0 has a tile)
    const float* x3 = J.packed + PACK_F32;
    float xb[4];
    ENC_STAMP(0, false);
    load_x(xb, J.x, first, ntiles, R, IN, lane);
    stage_linear<X3_IMG>(lds, x3, tid);                                         // W2's image; W3 is not needed
    stage_linear<1024 + 384>(lds + X3_FWD_F32, J.packed + 32768, tid);
    ENC_STAMP(1, true);
    __syncthreads();
    ENC_STAMP(2, false);
    const int k = J.k;
    const long long agents = R / k;
    for (long long tile = first; tile < ntiles; tile += stride) {
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const u32x4* W2hm = reinterpret_cast<const u32x4*>(lds) + lane_t;
        const u32x4* W2lo = W2hm + X3_HM / 4;
        const float* W1f = lds + X3_FWD_F32;
        const float* bias = W1f + 1024;
        const int h = lane_t >> 5;
        f32x16 a[4];
        Pieces P;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + feat0(blk, q, h));
                a[blk][4 * q + 0] = bq.x; a[blk][4 * q + 1] = bq.y; a[blk][4 * q + 2] = bq.z; a[blk][4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) a[blk] = mfma32(W1f[(blk * 4 + s) * 64 + lane_t], xb[s], a[blk]);
#pragma unroll
            for (int r = 0; r < 16; ++r) a[blk][r] = relu1(a[blk][r]);
        }
        uint2* mrow = reinterpret_cast<uint2*>(J.relu_mask) + (tile * 2) * 64 + lane_t;
        mrow[0] = make_uint2(sign_bits(a[0], a[1]), sign_bits(a[2], a[3]));       // h1: lane = row, bits = features (enc_fwd_x3_kernel's)
        ENC_STAMP(3, false);
        split_tile(a, P);
        ENC_STAMP(4, false);
        load_x(xb, J.x, tile + stride, ntiles, R, IN, lane);       // the next tile's input row
        // ---- layer 2, exchanged operands: a[blk] = relu(h2)[row rho(r) + 4 h][feature 32 blk + (lane & 31)] ----
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x16 acc, sm;
            const float bv = bias[128 + 32 * blk + (lane_t & 31)];
#pragma unroll
            for (int r = 0; r < 16; ++r) { sm[r] = 0.f; acc[r] = bv; }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = blk * 8 + kb;
                kblock_x3_t(acc, sm, W2hm[(fb * 2) * 64], W2hm[(fb * 2 + 1) * 64], W2lo[fb * 64], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) a[blk][r] = relu1(acc[r] + sm[r]);
        }
        ENC_STAMP(5, false);
        mrow[64] = make_uint2(sign_bits(a[0], a[1]), sign_bits(a[2], a[3]));      // h2: lane = feature, bits = the tile's rows
        if (J.h2) store_h2_rows<true>(a, J.h2, tile, R, lane_t);
        ENC_STAMP(6, false);
        pool_rows_k(a, k, tile, agents, lane_t, J.sum_a, J.sum_b);
        ENC_STAMP(7, false);
    }
    ENC_STAMP(8, true);
}

// ---------------------------------------------------------------------------------------------------------
// forward for FEW rows (rollouts of real clips: 100 .. 1000 agents): four waves per tile like enc_fwd_split_kernel
// (encoder.hip; a lone wave per SIMD is bound by the latency of its chain of dependent matrix instructions, here
// 16 + 48 + 48 of them instead of 400).  Wave (t, blk) computes output block blk of every layer of tile t, two tiles per
// workgroup; it splits ITS 16 registers of a layer's output and the pieces travel through LDS as ready-made B operands
// (k-block 2 blk + s = registers 8 s .. 8 s + 7).  The fragments of the wave's output block come straight from the packed
// image, the next layer's under this layer's products.  Every accumulator sees the k-blocks and the six products in the
// order of enc_fwd_x3_kernel: bitwise identical outputs.
// ---------------------------------------------------------------------------------------------------------
constexpr int X3_SPLIT_LDS_BYTES = 2 * 2 * 4 * 3 * 2 * 64 * 16;       // [layer 2][tile 2][block 4][piece 3][s 2][lane 64] u32x4

template <int DROP>        // as in enc_fwd_x3_kernel
__global__ __launch_bounds__(512) void enc_fwd_split_x3_kernel(EncArgs A, int pairs0) {
    extern __shared__ __align__(16) float lds[];
    u32x4* exch = reinterpret_cast<u32x4*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x >= pairs0 ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int t = wave >> 2, blk = wave & 3;
    const long long R = J.rows;
    const int IN = J.in_dim;
    const long long tile = ((long long)blockIdx.x - (b ? pairs0 : 0)) * 2 + t;
    const int j = lane & 31, h = lane >> 5;
    const long long row = tile * 32 + j;
    const bool valid = row < R;
    if (A.zero)
        for (int e = blockIdx.x * 512 + tid; e < A.zero_n; e += gridDim.x * 512) A.zero[e] = 0.f;
    const unsigned long long goff = DROP == 2 ? A.gen_state[1] : 0ull;
    const u32x4* W2hm = reinterpret_cast<const u32x4*>(J.packed + PACK_F32) + lane;
    const u32x4* W2lo = W2hm + X3_HM / 4;
    const u32x4* W3hm = reinterpret_cast<const u32x4*>(J.packed + PACK_F32 + X3_IMG) + lane;
    const u32x4* W3lo = W3hm + X3_HM / 4;
    const float* W1g = J.packed + 32768;
    const float* bias = J.packed + 32768 + 1024;
    u32x4 wf[8][3];                                   // this wave's fragments of the layer: k-block, (hi, mid, lo)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {                  // in flight during layer 1
        const int fb = blk * 8 + kb;
        wf[kb][0] = W2hm[(fb * 2) * 64]; wf[kb][1] = W2hm[(fb * 2 + 1) * 64]; wf[kb][2] = W2lo[fb * 64];
    }
    float xb[4], w1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 2 * s + h;
        xb[s] = (valid && c < IN) ? J.x[(valid ? row : 0) * IN + c] : 0.f;
        w1[s] = W1g[(blk * 4 + s) * 64 + lane];
    }
    // the biases of one layer at a time, the next layer's requested under this layer's products (all three held at once were 48
    // registers: with the dropout forms' mask words the kernel spilled 4)
    float4 bq[4], bqn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bias + feat0(blk, q, h));
    f32x16 acc, sm;
    auto init = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[4 * q] = bq[q].x; acc[4 * q + 1] = bq[q].y; acc[4 * q + 2] = bq[q].z; acc[4 * q + 3] = bq[q].w; }
#pragma unroll
        for (int r = 0; r < 16; ++r) sm[r] = 0.f;
    };
    auto store = [&](float* dst, float sc) {
        if (dst && valid) {
            float* o = dst + row * EH;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                store4_stream(o + feat0(blk, q, h), sc * acc[4 * q], sc * acc[4 * q + 1], sc * acc[4 * q + 2], sc * acc[4 * q + 3]);
        }
    };
    auto hand_over = [&](int l) {                     // this wave's 16 registers, split, as the B operands of k-blocks 2 blk, 2 blk + 1
        u32x4* dst = exch + ((((l * 2 + t) * 4 + blk) * 3) * 2) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) split3(acc[8 * s + 2 * d], acc[8 * s + 2 * d + 1], hi[d], mid[d], lo[d]);
            dst[(0 * 2 + s) * 64] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
            dst[(1 * 2 + s) * 64] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
            dst[(2 * 2 + s) * 64] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        }
    };
    // ---- layer 1 (f32 instruction, as in enc_fwd_x3_kernel) ----
    init();
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bias + 128 + feat0(blk, q, h));
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma32(w1[s], xb[s], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
    store(J.h1, 1.f);
    hand_over(0);
    __syncthreads();
    // ---- layers 2 and 3 ----
#pragma unroll
    for (int l = 1; l < 3; ++l) {
        init();
        u32x4 wn[8][2];                                // (hi, mid) of the next layer; its lo pieces follow once wf's are dead (as in the dX form)
        if (l == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bqn[q] = *reinterpret_cast<const float4*>(bias + 256 + feat0(blk, q, h));
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {          // next layer's, under this layer's products
                const int fb = blk * 8 + kb;
                wn[kb][0] = W3hm[(fb * 2) * 64]; wn[kb][1] = W3hm[(fb * 2 + 1) * 64];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const u32x4* src = exch + (((l - 1) * 2 + t) * 4 * 3 * 2) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const u32x4* e = src + ((kb >> 1) * 3 * 2 + (kb & 1)) * 64;
            kblock_x3(acc, sm, wf[kb][0], wf[kb][1], wf[kb][2], e[0], e[2 * 64], e[4 * 64]);
        }
        if (l == 1) {
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) wf[kb][2] = W3lo[(blk * 8 + kb) * 64];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += sm[r];
        if (l == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
            store(J.h2, 1.f);
            hand_over(1);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) { wf[kb][0] = wn[kb][0]; wf[kb][1] = wn[kb][1]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = bqn[q];
            __syncthreads();
        } else {
            if (DROP == 1) keep_block(acc, valid ? J.keep_bits[row * 4 + blk] : 0u, h);
            if (DROP == 2) {                     // the four waves of a tile draw the same call; wave blk takes and records word blk
                const PhiloxOut r = keep_words_fair(A.gen_state[0], goff, (unsigned)row, (unsigned)b);
                const unsigned word = blk == 0 ? r.x : (blk == 1 ? r.y : (blk == 2 ? r.z : r.w));
                if (valid && h == 0) J.keep_bits[row * 4 + blk] = word;
                keep_block(acc, word, h);
            }
            store(J.msgs, J.scale);
        }
    }
    if (DROP == 2 && tid == 0) dropout_advance(A.gen_state, goff, gridDim.x);
}

// (Round 4, measured and removed: the four-waves-per-tile cut made PERSISTENT for many rows -- two crews of four waves per
// workgroup, each wave's (hi, mid) weight fragments of both layers in 128 AGPRs as asm operands, the lo pieces in LDS, no weight
// traffic per tile at all; bitwise equal to enc_fwd_x3_kernel, 112 tests green.  36.4 us against 27.2 at the 4096-agent scene:
// a tile is three phases separated by workgroup barriers and the two crews of a SIMD march through them in lockstep.  As
// INDEPENDENT workgroups of one crew each (48 KB of LDS, the lo pieces from L2; two per CU, their barriers unrelated) 41.9 us,
// one per CU 46.9.  One wave per tile keeps its activations in registers between the layers and pays for that with the
// weights' trip through LDS, which is the cheaper of the two: the hand-over of a tile's activations through LDS and a barrier
// per layer is what the cut costs, whatever happens to the weights.)

// ---------------------------------------------------------------------------------------------------------
// backward, part 1: the dX chain (see enc_bwd_dx_kernel in encoder.hip for the arithmetic)
// ---------------------------------------------------------------------------------------------------------
// LDS: W3^T image whole [HM 4096 | LO 2048 u32x4] | W2^T HM of fragments 0 .. 29 [30][2][64] u32x4 | W1 rows [f 128][8]
// floats = 163 840 B, all of the CU's.  W2^T's LO pieces and the hi / mid of its last two fragments come from L2.
constexpr int X3_FB2T = 30;
constexpr int X3_DX_W2T = X3_IMG / 4;                         // u32x4 offset of the W2^T HM part
constexpr int X3_DX_F32 = X3_IMG + X3_FB2T * 2 * 64 * 4;      // float offset of the W1 rows
constexpr int X3_DX_LDS_BYTES = (X3_DX_F32 + 1024) * 4;
static_assert(X3_DX_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
constexpr int W2T_N4 = X3_FB2T * 2 * 64, W2T_ROUNDS = (W2T_N4 + ENC_THREADS - 1) / ENC_THREADS;

__device__ __forceinline__ void land_w2t(const u32x4 (&w)[W2T_ROUNDS], float* lds, int tid) {
    u32x4* dst = reinterpret_cast<u32x4*>(lds) + X3_DX_W2T;
#pragma unroll
    for (int r = 0; r < W2T_ROUNDS; ++r) {
        const int e = r * ENC_THREADS + tid;
        if (e < W2T_N4) dst[e] = w[r];
    }
    __syncthreads();
}

template <bool MASK, bool DROP>
__global__ __launch_bounds__(ENC_THREADS) void enc_bwd_dx_x3_kernel(EncArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int nwg = b ? (int)gridDim.x - A.wg_split : (A.nbr > 1 ? A.wg_split : (int)gridDim.x);
    const long long R = J.rows;
    const int IN = J.in_dim, K = J.k;
    const long long ntiles = (R + 31) >> 5;
    const long long first = (long long)((int)blockIdx.x - wg0) * ENC_WAVES + wave;
    const long long stride = (long long)nwg * ENC_WAVES;
    if ((long long)((int)blockIdx.x - wg0) * ENC_WAVES >= ntiles) return;

    const float* x3 = J.packed + PACK_F32;
    const bool want_gx = J.g_x != nullptr;
    stage_linear<X3_IMG>(lds, x3 + 2 * X3_IMG, tid);
    stage_linear<1024>(lds + X3_DX_F32, J.packed + PACK_FWD + 32768, tid);
    __syncthreads();
    // W2^T's LDS part.  (Rounds 2 - 3 kept it in flight in 32 registers behind the first tile's first layer: those registers
    // were what the kernel spilled -- 9 / 13 VGPRs.  Since round 4 this kernel is the fallback of the one-pass backward and
    // stages everything up front: ~1 us of prologue, no scratch.)
    {
        u32x4 w2r[W2T_ROUNDS];
        const u32x4* src = reinterpret_cast<const u32x4*>(x3 + 3 * X3_IMG);
#pragma unroll
        for (int r = 0; r < W2T_ROUNDS; ++r) {
            const int e = r * ENC_THREADS + tid;
            w2r[r] = src[e < W2T_N4 ? e : 0];
        }
        land_w2t(w2r, lds, tid);
    }

    const float scale = J.scale;
    for (long long tile = first; tile < ntiles; tile += stride) {
        int lane_t = lane;           // opaque per tile (see enc_fwd_x3_kernel)
        asm volatile("" : "+v"(lane_t));
        const u32x4* W3hm = reinterpret_cast<const u32x4*>(lds) + lane_t;
        const u32x4* W3lo = W3hm + X3_HM / 4;
        const u32x4* W2hm = reinterpret_cast<const u32x4*>(lds) + X3_DX_W2T + lane_t;
        const u32x4* W2hm_g = reinterpret_cast<const u32x4*>(x3 + 3 * X3_IMG) + lane_t;
        const u32x4* W2lo_g = W2hm_g + X3_HM / 4;
        const float4* W1r = reinterpret_cast<const float4*>(lds + X3_DX_F32);      // row f = float4 2 f, 2 f + 1
        const int j = lane_t & 31, h = lane_t >> 5;
        const long long row = tile * 32 + j;
        const bool valid = row < R;
        f32x16 g[4];
        Pieces P;
        uint2 m2 = make_uint2(0u, 0u), m1 = make_uint2(0u, 0u);      // sign bits of h2, h1 (forward: sign_bits)
        if (MASK) {
            const uint2* mrow = reinterpret_cast<const uint2*>(J.relu_mask) + (tile * 2) * 64 + lane_t;
            m1 = mrow[0];
            m2 = mrow[64];
        }
        // ---- g3 in registers ----
        {
            const float* gp = (J.g_pooled && valid) ? J.g_pooled + (row / K) * EH : nullptr;
            const float* gm = (J.g_msgs && valid) ? J.g_msgs + row * EH : nullptr;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gp) v = *reinterpret_cast<const float4*>(gp + feat0(blk, q, h));
                    if (gm) {
                        const float4 m = *reinterpret_cast<const float4*>(gm + feat0(blk, q, h));
                        v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
                    }
                    g[blk][4 * q + 0] = scale * v.x; g[blk][4 * q + 1] = scale * v.y;
                    g[blk][4 * q + 2] = scale * v.z; g[blk][4 * q + 3] = scale * v.w;
                }
            if (DROP) {
                uint4 kw = make_uint4(0u, 0u, 0u, 0u);
                if (valid) kw = reinterpret_cast<const uint4*>(J.keep_bits)[row];
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) keep_block(g[blk], word_of(kw, blk), h);
            }
        }
        split_tile(g, P);
        // ---- g_h2 = W3^T g3, masked by h2 -> g2 (g is dead: reused) ----
        const float* hp2 = J.h2 + (valid ? row : 0) * EH;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            __builtin_amdgcn_sched_barrier(0);
            float4 hv[4];                                   // this block's h2 values: in flight during the MFMAs
            if (!MASK) {
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4*>(hp2 + feat0(blk, q, h));
            }
            f32x16 acc, sm;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = blk * 8 + kb;
                kblock_x3(acc, sm, W3hm[(fb * 2) * 64], W3hm[(fb * 2 + 1) * 64], W3lo[fb * 64], P.hi[kb], P.mid[kb], P.lo[kb]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += sm[r];
            if (MASK) {       // (rows past the end: their forward inputs were zeros, their gradients are not stored)
                const unsigned mw = blk < 2 ? m2.x : m2.y;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = valid ? keep_if(acc[r], mw, 16 * (blk & 1) + r) : 0.f;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = hv[q];
                    acc[4 * q + 0] = (valid && a.x > 0.f) ? acc[4 * q + 0] : 0.f;
                    acc[4 * q + 1] = (valid && a.y > 0.f) ? acc[4 * q + 1] : 0.f;
                    acc[4 * q + 2] = (valid && a.z > 0.f) ? acc[4 * q + 2] : 0.f;
                    acc[4 * q + 3] = (valid && a.w > 0.f) ? acc[4 * q + 3] : 0.f;
                }
            }
            g[blk] = acc;
            if (valid) {
                float* o = J.g2 + row * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4_stream(o + feat0(blk, q, h), acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        }
        // ---- g_h1 = W2^T g2, masked by h1 -> g1; g_x = W1^T g1 block by block on the vector pipe ----
        // LO pieces of W2^T: HALF an output block ahead (k-blocks 4 .. 7 of this block and 0 .. 3 of the next are requested
        // under the products of k-blocks 0 .. 3 / 4 .. 7).  A whole block ahead in two register sets was 64 registers, and
        // the kernel spilled 9 - 13 of its 256; since round 4 this kernel is the fallback of the one-pass backward (no
        // sign bits, PIML_ENC_FUSED_BWD=0, branches with different kinds of upstream gradient).
        u32x4 lw[8];
        u32x4 tail[4];               // hi / mid of fragments 30, 31 (block 3, k-blocks 6, 7)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) lw[kb] = W2lo_g[kb * 64];
#pragma unroll
        for (int u = 0; u < 4; ++u) tail[u] = W2hm_g[(X3_FB2T * 2 + u) * 64];
        split_tile(g, P);
        const float* hp1 = J.h1 + (valid ? row : 0) * EH;
        float gx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) gx[c] = 0.f;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            __builtin_amdgcn_sched_barrier(0);
            float4 hv[4];
            if (!MASK) {
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4*>(hp1 + feat0(blk, q, h));
            }
            f32x16 acc, sm;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = blk * 8 + kb;
                const u32x4 wh = fb < X3_FB2T ? W2hm[(fb * 2) * 64] : tail[fb < X3_FB2T ? 0 : (fb - X3_FB2T) * 2];
                const u32x4 wm = fb < X3_FB2T ? W2hm[(fb * 2 + 1) * 64] : tail[fb < X3_FB2T ? 0 : (fb - X3_FB2T) * 2 + 1];
                kblock_x3(acc, sm, wh, wm, lw[kb], P.hi[kb], P.mid[kb], P.lo[kb]);
                if (kb == 3 && blk < 3) {          // k-blocks 0 .. 3 of the next output block: their registers are free now
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) lw[k2] = W2lo_g[((blk + 1) * 8 + k2) * 64];
                }
            }
            if (blk < 3) {
#pragma unroll
                for (int k2 = 4; k2 < 8; ++k2) lw[k2] = W2lo_g[((blk + 1) * 8 + k2) * 64];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += sm[r];
            if (MASK) {
                const unsigned mw = blk < 2 ? m1.x : m1.y;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = valid ? keep_if(acc[r], mw, 16 * (blk & 1) + r) : 0.f;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = hv[q];
                    acc[4 * q + 0] = (valid && a.x > 0.f) ? acc[4 * q + 0] : 0.f;
                    acc[4 * q + 1] = (valid && a.y > 0.f) ? acc[4 * q + 1] : 0.f;
                    acc[4 * q + 2] = (valid && a.z > 0.f) ? acc[4 * q + 2] : 0.f;
                    acc[4 * q + 3] = (valid && a.w > 0.f) ? acc[4 * q + 3] : 0.f;
                }
            }
            if (valid) {
                float* o = J.g1 + row * EH;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store4_stream(o + feat0(blk, q, h), acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
            // lane (row, h) holds half of the row's g1 features: 8 partial dot products over them (explicit FMAs, the
            // order of enc_bwd_dx_kernel), the other half arrives with one cross-half exchange after the last block
            if (want_gx) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_barrier(0);      // 8 LDS reads in flight per group
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int f = feat0(blk, q, h) + u;
                        const float4 wa = W1r[2 * f], wb = W1r[2 * f + 1];
                        const float v = acc[4 * q + u];
                        gx[0] = __fmaf_rn(wa.x, v, gx[0]); gx[1] = __fmaf_rn(wa.y, v, gx[1]);
                        gx[2] = __fmaf_rn(wa.z, v, gx[2]); gx[3] = __fmaf_rn(wa.w, v, gx[3]);
                        gx[4] = __fmaf_rn(wb.x, v, gx[4]); gx[5] = __fmaf_rn(wb.y, v, gx[5]);
                        gx[6] = __fmaf_rn(wb.z, v, gx[6]); gx[7] = __fmaf_rn(wb.w, v, gx[7]);
                    }
                }
            }
        }
        if (want_gx) {
#pragma unroll
            for (int c = 0; c < 8; ++c) gx[c] += __shfl_xor(gx[c], 32, 64);
            if (valid) {       // half h stores input features 4 h .. 4 h + 3
                float* o = J.g_x + row * IN + 4 * h;
                const int left = IN - 4 * h;       // scalar selects (an array select goes through scratch)
                const float s0 = h ? gx[4] : gx[0], s1 = h ? gx[5] : gx[1], s2 = h ? gx[6] : gx[2], s3 = h ? gx[7] : gx[3];
                if (left > 0) o[0] = s0;
                if (left > 1) o[1] = s1;
                if (left > 2) o[2] = s2;
                if (left > 3) o[3] = s3;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// dX chain for FEW rows (the fine-tuning loop on real clips, src/models/simulators.py:659-832: 100 .. 1000 agents): four
// waves per tile like enc_fwd_split_x3_kernel.  Wave (t, blk) owns block blk of g3, g2 and g1 of tile t: it builds ITS 16
// registers of g3 (loads, scale, keep mask), splits them and hands the pieces over through LDS as ready-made B operands;
// after the barrier every wave runs the eight k-blocks of its output block of W3^T g3, masks with h2, stores g2, hands
// the split block over again, and likewise for g1 = (W2^T g2) * [h1 > 0].  g_x = W1^T g1 by the blk = 0 wave of the tile
// from all four g1 blocks (f32, through LDS) in enc_bwd_dx_x3_kernel's order.  Every accumulator sees the k-blocks and
// the six products in the order of enc_bwd_dx_x3_kernel: bitwise identical gradients.
// ---------------------------------------------------------------------------------------------------------
constexpr int X3_DXS_G1 = X3_SPLIT_LDS_BYTES;                             // byte offset of the g1 exchange [tile 2][block 4][16][64] floats
constexpr int X3_DXS_LDS_BYTES = X3_SPLIT_LDS_BYTES + 2 * 4 * 16 * 64 * 4;
static_assert(X3_DXS_LDS_BYTES <= 160 * 1024, "fits the CU");

template <bool DROP>
__global__ __launch_bounds__(512) void enc_bwd_dx_split_x3_kernel(EncArgs A, int pairs0) {
    extern __shared__ __align__(16) float lds[];
    u32x4* exch = reinterpret_cast<u32x4*>(lds);
    float* g1x = lds + X3_DXS_G1 / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x >= pairs0 ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int t = wave >> 2, blk = wave & 3;
    const long long R = J.rows;
    const int IN = J.in_dim, K = J.k;
    const long long tile = ((long long)blockIdx.x - (b ? pairs0 : 0)) * 2 + t;
    const int j = lane & 31, h = lane >> 5;
    const long long row = tile * 32 + j;
    const bool valid = row < R;
    const long long rr = valid ? row : 0;
    const float scale = J.scale;
    const float* x3 = J.packed + PACK_F32;
    const u32x4* W3hm = reinterpret_cast<const u32x4*>(x3 + 2 * X3_IMG) + lane;
    const u32x4* W3lo = W3hm + X3_HM / 4;
    const u32x4* W2hm = reinterpret_cast<const u32x4*>(x3 + 3 * X3_IMG) + lane;
    const u32x4* W2lo = W2hm + X3_HM / 4;
    const float4* W1r = reinterpret_cast<const float4*>(J.packed + PACK_FWD + 32768);      // row f = float4 2 f, 2 f + 1
    u32x4 wf[8][3];                                   // this wave's fragments of the layer: k-block, (hi, mid, lo)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const int fb = blk * 8 + kb;
        wf[kb][0] = W3hm[(fb * 2) * 64]; wf[kb][1] = W3hm[(fb * 2 + 1) * 64]; wf[kb][2] = W3lo[fb * 64];
    }
    float4 hv[4];                                     // this block's h2 values (ReLU mask of the first layer of the chain)
#pragma unroll
    for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4*>(J.h2 + rr * EH + feat0(blk, q, h));
    f32x16 acc, sm;
    // ---- this wave's block of g3 ----
    {
        const float* gp = J.g_pooled ? J.g_pooled + (rr / K) * EH : nullptr;
        const float* gm = J.g_msgs ? J.g_msgs + rr * EH : nullptr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gp) v = *reinterpret_cast<const float4*>(gp + feat0(blk, q, h));
            if (gm) {
                const float4 m = *reinterpret_cast<const float4*>(gm + feat0(blk, q, h));
                v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
            }
            acc[4 * q + 0] = valid ? scale * v.x : 0.f; acc[4 * q + 1] = valid ? scale * v.y : 0.f;
            acc[4 * q + 2] = valid ? scale * v.z : 0.f; acc[4 * q + 3] = valid ? scale * v.w : 0.f;
        }
        if (DROP) keep_block(acc, valid ? J.keep_bits[rr * 4 + blk] : 0u, h);
    }
    auto hand_over = [&](int l) {                     // this wave's 16 registers, split, as the B operands of k-blocks 2 blk, 2 blk + 1
        u32x4* dst = exch + ((((l * 2 + t) * 4 + blk) * 3) * 2) * 64 + lane;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) split3(acc[8 * s_ + 2 * d], acc[8 * s_ + 2 * d + 1], hi[d], mid[d], lo[d]);
            dst[(0 * 2 + s_) * 64] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
            dst[(1 * 2 + s_) * 64] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
            dst[(2 * 2 + s_) * 64] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        }
    };
    hand_over(0);
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 2; ++l) {                      // l = 0: g2 = (W3^T g3) * [h2 > 0];  l = 1: g1 = (W2^T g2) * [h1 > 0]
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; sm[r] = 0.f; }
        u32x4 wn[8][2];                                // (hi, mid) of the next layer; its lo pieces follow once wf's are dead
        float4 hn[4];
        if (l == 0) {                                  // the next layer's operands travel under this layer's products
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int fb = blk * 8 + kb;
                wn[kb][0] = W2hm[(fb * 2) * 64]; wn[kb][1] = W2hm[(fb * 2 + 1) * 64];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) hn[q] = *reinterpret_cast<const float4*>(J.h1 + rr * EH + feat0(blk, q, h));
        }
        __builtin_amdgcn_sched_barrier(0);
        const u32x4* src = exch + ((l * 2 + t) * 4 * 3 * 2) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const u32x4* e = src + ((kb >> 1) * 3 * 2 + (kb & 1)) * 64;
            kblock_x3(acc, sm, wf[kb][0], wf[kb][1], wf[kb][2], e[0], e[2 * 64], e[4 * 64]);
        }
        if (l == 0) {
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) wf[kb][2] = W2lo[(blk * 8 + kb) * 64];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += sm[r];
        float* dst = (l == 0 ? J.g2 : J.g1) + rr * EH;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = hv[q];
            acc[4 * q + 0] = (valid && a.x > 0.f) ? acc[4 * q + 0] : 0.f;
            acc[4 * q + 1] = (valid && a.y > 0.f) ? acc[4 * q + 1] : 0.f;
            acc[4 * q + 2] = (valid && a.z > 0.f) ? acc[4 * q + 2] : 0.f;
            acc[4 * q + 3] = (valid && a.w > 0.f) ? acc[4 * q + 3] : 0.f;
            if (valid) store4_stream(dst + feat0(blk, q, h), acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
        if (l == 0) {
            hand_over(1);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) { wf[kb][0] = wn[kb][0]; wf[kb][1] = wn[kb][1]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[q] = hn[q];
        } else if (J.g_x) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g1x[((t * 4 + blk) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
    }
    // ---- g_x = W1^T g1, by one wave of the tile, in enc_bwd_dx_x3_kernel's order ----
    if (blk == 0 && J.g_x) {
        float gx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) gx[c] = 0.f;
#pragma unroll
        for (int bp = 0; bp < 4; ++bp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = feat0(bp, q, h) + u;
                    const float4 wa = W1r[2 * f], wb = W1r[2 * f + 1];
                    const float v = g1x[((t * 4 + bp) * 16 + 4 * q + u) * 64 + lane];
                    gx[0] = __fmaf_rn(wa.x, v, gx[0]); gx[1] = __fmaf_rn(wa.y, v, gx[1]);
                    gx[2] = __fmaf_rn(wa.z, v, gx[2]); gx[3] = __fmaf_rn(wa.w, v, gx[3]);
                    gx[4] = __fmaf_rn(wb.x, v, gx[4]); gx[5] = __fmaf_rn(wb.y, v, gx[5]);
                    gx[6] = __fmaf_rn(wb.z, v, gx[6]); gx[7] = __fmaf_rn(wb.w, v, gx[7]);
                }
            }
#pragma unroll
        for (int c = 0; c < 8; ++c) gx[c] += __shfl_xor(gx[c], 32, 64);
        if (valid) {
            float* o = J.g_x + row * IN + 4 * h;
            const int left = IN - 4 * h;
            const float s0 = h ? gx[4] : gx[0], s1 = h ? gx[5] : gx[1], s2 = h ? gx[6] : gx[2], s3 = h ? gx[7] : gx[3];
            if (left > 0) o[0] = s0;
            if (left > 1) o[1] = s1;
            if (left > 2) o[2] = s2;
            if (left > 3) o[3] = s3;
        }
    }
}

// (backward, part 2 -- the weight gradients dW = G^T H on split products -- lives in encoder_dww.hip (row slabs, both layers per
// workgroup: few rows) and encoder_dw2.hip (layer-split workgroups of producer / consumer waves: above the few-rows bound).)

int enc_x3_set_attributes() {
    auto set = [](const void* f, int bytes) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_dx_split_x3_kernel<false>), X3_DXS_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_dx_split_x3_kernel<true>), X3_DXS_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_split_x3_kernel<0>), X3_SPLIT_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_split_x3_kernel<1>), X3_SPLIT_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_split_x3_kernel<2>), X3_SPLIT_LDS_BYTES)) return e;
    const void* dx[4] = {reinterpret_cast<const void*>(enc_bwd_dx_x3_kernel<false, false>),
                         reinterpret_cast<const void*>(enc_bwd_dx_x3_kernel<true, false>),
                         reinterpret_cast<const void*>(enc_bwd_dx_x3_kernel<false, true>),
                         reinterpret_cast<const void*>(enc_bwd_dx_x3_kernel<true, true>)};
    for (const void* f : dx)
        if (int e = set(f, X3_DX_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_pool_x3_kernel), X3_FWD_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_sum_x3_kernel), X3_FWD_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_x3_kernel<2, true>), X3_FWD_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_x3_kernel<1, true>), X3_FWD_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_x3_kernel<0, true>), X3_FWD_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_x3_kernel<2>), X3_FWD_LDS_BYTES)) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_fwd_x3_kernel<1>), X3_FWD_LDS_BYTES)) return e;
    return set(reinterpret_cast<const void*>(enc_fwd_x3_kernel<0>), X3_FWD_LDS_BYTES);
}

// `drop`: every branch of the launch carries keep_bits (checked by the callers: all or none)
void enc_x3_launch_bwd_dx(const EncArgs& A, int total, bool mask, bool drop, hipStream_t s) {
    const dim3 g(total), b(ENC_THREADS);
    if (mask && drop) hipLaunchKernelGGL((enc_bwd_dx_x3_kernel<true, true>), g, b, X3_DX_LDS_BYTES, s, A);
    else if (mask) hipLaunchKernelGGL((enc_bwd_dx_x3_kernel<true, false>), g, b, X3_DX_LDS_BYTES, s, A);
    else if (drop) hipLaunchKernelGGL((enc_bwd_dx_x3_kernel<false, true>), g, b, X3_DX_LDS_BYTES, s, A);
    else hipLaunchKernelGGL((enc_bwd_dx_x3_kernel<false, false>), g, b, X3_DX_LDS_BYTES, s, A);
}

void enc_x3_launch_bwd_dx_split(const EncArgs& A, int pairs0, int pairs1, bool drop, hipStream_t s) {
    const dim3 g((unsigned)(pairs0 + pairs1));
    if (drop) hipLaunchKernelGGL(enc_bwd_dx_split_x3_kernel<true>, g, dim3(512), X3_DXS_LDS_BYTES, s, A, pairs0);
    else hipLaunchKernelGGL(enc_bwd_dx_split_x3_kernel<false>, g, dim3(512), X3_DXS_LDS_BYTES, s, A, pairs0);
}

void enc_x3_launch_fwd_split(const EncArgs& A, int pairs0, int pairs1, bool drop, hipStream_t s) {
    const dim3 g((unsigned)(pairs0 + pairs1));
    if (drop && A.gen_state) hipLaunchKernelGGL(enc_fwd_split_x3_kernel<2>, g, dim3(512), X3_SPLIT_LDS_BYTES, s, A, pairs0);
    else if (drop) hipLaunchKernelGGL(enc_fwd_split_x3_kernel<1>, g, dim3(512), X3_SPLIT_LDS_BYTES, s, A, pairs0);
    else hipLaunchKernelGGL(enc_fwd_split_x3_kernel<0>, g, dim3(512), X3_SPLIT_LDS_BYTES, s, A, pairs0);
}

void enc_x3_launch_fwd_pool(const EncArgs& A, int total, hipStream_t s) {
    hipLaunchKernelGGL(enc_fwd_pool_x3_kernel, dim3(total), dim3(ENC_THREADS), X3_FWD_LDS_BYTES, s, A);
}

void enc_x3_launch_fwd_sum(const EncArgs& A, int total, hipStream_t s) {
    hipLaunchKernelGGL(enc_fwd_sum_x3_kernel, dim3(total), dim3(ENC_THREADS), X3_FWD_LDS_BYTES, s, A);
}

void enc_x3_launch_fwd(const EncArgs& A, int total, bool drop, hipStream_t s, bool exch) {
    const dim3 g(total), b(ENC_THREADS);
    if (exch) {          // PIML_POOL_MSGS: the last layer with exchanged operands, the agents' sums of the messages from registers
        if (drop && A.gen_state) hipLaunchKernelGGL((enc_fwd_x3_kernel<2, true>), g, b, X3_FWD_LDS_BYTES, s, A);
        else if (drop) hipLaunchKernelGGL((enc_fwd_x3_kernel<1, true>), g, b, X3_FWD_LDS_BYTES, s, A);
        else hipLaunchKernelGGL((enc_fwd_x3_kernel<0, true>), g, b, X3_FWD_LDS_BYTES, s, A);
        return;
    }
    if (drop && A.gen_state) hipLaunchKernelGGL(enc_fwd_x3_kernel<2>, g, b, X3_FWD_LDS_BYTES, s, A);
    else if (drop) hipLaunchKernelGGL(enc_fwd_x3_kernel<1>, g, b, X3_FWD_LDS_BYTES, s, A);
    else hipLaunchKernelGGL(enc_fwd_x3_kernel<0>, g, b, X3_FWD_LDS_BYTES, s, A);
}

}  // namespace piml

#ifdef PIML_ENC_STAMPS
extern "C" __attribute__((visibility("default"))) int piml_enc_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(piml::g_enc_stamps), sizeof(unsigned long long) * 512 * 16);
}
#endif
