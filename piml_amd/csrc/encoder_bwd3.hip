// Backward of the PINNSF encoder in ONE pass over the rows: the dX chain AND the weight gradients of the two lower layers,
// without the round trip of the pre-activation gradients g2 / g1 through memory (round 4).
//
// Reference arithmetic: the autograd of MLP(in, [128, 128, 128]) (src/models/model.py:40-65) under the processor
// Dropout_p(2 x) and the neighbour-axis sum (:82-119, :1279-1283):
//     G3 = keep * scale * (g_pooled[row / k] + g_msgs[row])
//     G2 = (G3 W3) * [h2 > 0]        dW2 = G2^T H1, db2 = colsum G2        H1 = relu(W1 x + b1)
//     G1 = (G2 W2) * [h1 > 0]        dW1 = G1^T X,  db1 = colsum G1        g_x = G1 W1
// (dW3 = G3^T H2 / db3 need nothing of the chain: they stay with the layer-0 workgroups of encoder_dw2.hip.)
//
// Until round 3 enc_bwd_dx_x3_kernel wrote g2 / g1 (2 x 33 MB at the 4096-agent scene) and enc_bwd_dw2_x3_kernel read them
// back: a third of the step's memory traffic.  Here a 32-row tile's G2 never leaves the CU:
//   * a workgroup is FOUR waves, one per SIMD, 512 registers each; wave w owns feature block w (32 of 128 features) of
//     every layer of the tile (the cut of enc_bwd_dx_split_x3_kernel) and walks the workgroup's tiles one after another;
//   * its operand fragments of W3^T and W2^T -- 8 k-blocks x (hi, mid, lo) x 2 layers = 192 registers -- are loaded ONCE
//     and stay in registers for every tile: no weight image in LDS, no weight traffic per tile;
//   * the chain runs in the NON-transposed orientation, D[row][feature] = sum_k act[row][k] W[k][feature] (activations = A
//     operand, weights = B operand): the result has its feature on the lane and the tile's rows in the 16 registers, which
//     IS the operand layout of a product that contracts over the rows -- registers 8 s .. 8 s + 7, split and packed
//     pairwise, are the A fragment of k-step s of dW2 = G2^T H1.  H1 is recomputed in the same orientation (eight f32
//     matrix instructions per wave and tile) and travels through LDS as the B fragments; wave w accumulates the four output
//     blocks (w, 0 .. 3) of dW2 in 128 registers for the whole slab of the workgroup;
//   * the next layer of the chain contracts over the FEATURES (the lane index of the result): that one transposition rides
//     on the hand-over between the waves, which goes through LDS anyway -- each lane stores its feature's 32 rows as bf16
//     pieces into a [feature][row] image (XOR-swizzled 8-byte chunks: conflict-free both ways) and the readers fetch
//     [row][8 features] fragments with ds_read_b64_tr_b16;
//   * dW1 / db1 / db2 are sums over registers (rows) on the vector pipe; g_x = G1 W1 contracts over the lanes: the wave's
//     G1 block is transposed through a private f32 LDS tile and summed per row, the four waves' partials meet in LDS and
//     are added in a fixed order (no atomics: bit-reproducible);
//   * one wave per SIMD has nobody to hide its latencies behind, so the tile loop is software-pipelined by hand: the vector
//     work that does not depend on a burst of matrix instructions is written between them -- H1's split under the products of
//     layer A, the NEXT tile's G3 under layer B, G1's sums under dW2 -- the loads of the next tile travel a whole tile ahead,
//     and every access to memory is a buffer access whose range check replaces the branch.
// Element order of every fragment: element t of lane half h of k-block kb = index 16 kb + 8 (t >> 2) + 4 h + (t & 3), the
// order of the packed images (pack.hpp), so the transposed weight images of the dX kernels serve as B operands unchanged.
#include "common.hpp"
#include "encoder.hpp"
#include "x3.hpp"

namespace piml {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int F3_THREADS = 256;
// LDS (bytes)
constexpr int F3_BUFA = 0;                                   // G3 pieces, A fragments: [kb 8][piece 3][lane 64] u32x4
constexpr int F3_M = F3_BUFA + 8 * 3 * 64 * 16;              // G2 pieces, [piece 3][feature 128][row 32] bf16, swizzled 8-byte chunks
constexpr int F3_BUFH = F3_M + 3 * 128 * 64;                 // H1 pieces, B fragments: [block 4][k-step 2][piece 3][lane 64] u32x4
constexpr int F3_XS = F3_BUFH + 4 * 2 * 3 * 64 * 16;         // the tile's x rows [parity 2][32][8] floats
constexpr int F3_MK = F3_XS + 2 * 1024;                      // the tile's sign words [parity 2][layer 2][lane 64] uint2
constexpr int F3_W1 = F3_MK + 2 * 1024;                      // W1 rows [128][8] floats
constexpr int F3_GX = F3_W1 + 4096;                          // g_x partials [wave 4][row 32][8] floats
constexpr int F3_WLO = F3_GX + 4 * 32 * 8 * 4;               // LO pieces of the wave's weight fragments [wave 4][layer 2][kb 8][lane 64] u32x4
constexpr int F3_T = F3_WLO + 4 * 2 * 8 * 64 * 16;           // G1 half blocks for g_x: [wave 4][feature 16][36] floats
constexpr int F3_TROW = 36;
constexpr int F3_TDUMMY = F3_T + 4 * 16 * F3_TROW * 4;        // [wave 4][lane 32] float4: where the lanes outside a pass store instead
constexpr int F3_LDS_BYTES = F3_TDUMMY + 4 * 32 * 16;
static_assert(F3_LDS_BYTES <= 160 * 1024, "fits the CU");

// diagnostic builds (tools/r4_ab.sh; RESULTS WRONG ON PURPOSE): PIML_F3_SKIP = bits of work left out, to see what it costs
//   1: the steps between layer A's products   2: those of layer B   4: those of dW2   8: G2's mask / split / hand-over
//   16: the products of dW2   32: the products of both chain layers
#ifndef PIML_F3_SKIP
#define PIML_F3_SKIP 0
#endif

#ifdef PIML_F3_STAMPS
// diagnostic build only (tools/f3_stamps.py): cycles of wave 0 between the stamps, summed over the workgroup's tiles
__device__ unsigned long long g_f3_stamps[256 * 64];          // [workgroup][wave 4][stamp 16]
#define F3_STAMP(i)                                                        \
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
#define F3_STAMP(i)
#endif

// The workgroup barrier of the tile loops: this wave's LDS operations done, nothing said about its loads in flight (the requests of
// the next tile travel across it; __syncthreads() also waits for vmcnt(0) -- the stamps read 400 - 1040 cycles at B1 for that).
// PIML_F3_SYNCTHREADS=1: the plain barrier (A/B)
#ifndef PIML_F3_SYNCTHREADS
#define PIML_F3_SYNCTHREADS 0
#endif
#if PIML_F3_SYNCTHREADS
#define F3_BARRIER() __syncthreads()
#else
#define F3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// 1: g_x = G1 W1 on v_mfma_f32_16x16x4_f32 (exact f32 products): per pass of 16 features two chains (rows 0 .. 15 / 16 .. 31) of four
// products; the G1 operand is one ds_read_b32 of the wave's half tile per product, the W1 operand EIGHT registers for the whole
// slab -- instead of 32 ds_read_b32 + 32 broadcast ds_read_b128 of W1 rows + 128 FMAs per wave and tile (the LDS pipe, not the
// FMAs, was the price there: encoder_bwd5.hip).  Built, green (246 tests) and LEVEL here (58.3 against 58.4 us in the dropout step): the
// lone wave's vector form already sits in the shadow of the dW2 products.  0 (default): the vector form
#ifndef PIML_F3_GX_MFMA
#define PIML_F3_GX_MFMA 0
#endif

struct F3Args {
    EncArgs A;
    int nA[2];          // workgroups of branch 0 / branch 1 (grid = their sum)
    int slot0[2];       // layer-0 slots (DW2_PART0 floats each) in front of this kernel's slots in the branch's `partials`
    int with_dw3;       // phase 2: this launch also writes the layer-0 slots (dW3 | db3), slot = workgroup index within the branch
};

constexpr int F3_PART0 = EH * EH + EH;                       // = DW2_PART0 (encoder_dw2.hip): dW3 | db3
constexpr int F3_PART1 = EH * EH + 1024 + 2 * EH;            // = DW2_PART1: dW2 | dW1 (1024-float field) | db2 | db1

// ReLU as ONE integer maximum: a negative float (and -0) is a negative int, a positive one keeps its bits.  (fmed3 / fmax on a
// value that comes out of an asm statement cost a canonicalising v_max in front.)
__device__ __forceinline__ float relu_i(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// row of the tile held by accumulator register r in lane half 0 (half 1: + 4)
__device__ __forceinline__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

// 16 registers -> the three bf16 pieces of both k-steps: element t of k-step s = register 8 s + t
struct Pieces2 {
    u32x4 hi[2], mid[2], lo[2];
};
__device__ __forceinline__ void split_half(const f32x16& a, Pieces2& P, int s) {
    unsigned hi[4], mid[4], lo[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) split3(a[8 * s + 2 * d], a[8 * s + 2 * d + 1], hi[d], mid[d], lo[d]);
    P.hi[s] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
    P.mid[s] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
    P.lo[s] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
}

// ---- matrix instructions with the weight operand in the ACCUMULATOR half of the register file ----
// hipcc gives a builtin MFMA's A / B operands VGPRs only: 128 registers of weight fragments held across the tile loop end
// up parked in AGPRs and are copied back in front of every use (330 v_accvgpr_read per tile, and scratch beyond that).  As
// `asm` operands with the "a" constraint they ARE the B operand.  What hipcc does not do for an asm statement is done here
// (cdna_hip_programming.md 5.7): results in VGPRs, `=&v` where the chain starts from zero (the destination must not land on
// an operand), two wait states in front of a product whose A operand may come from a move (s_nop 1), and f3_settle() --
// sixteen states -- between a chain's last product and the first vector instruction that reads it.
#ifndef PIML_F3_PAD
#define PIML_F3_PAD ""
#endif
__device__ __forceinline__ void f3_mfma0(f32x16& d, const u32x4& a, const u32x4& b_acc) {          // d = a x B(agpr)
    asm volatile(PIML_F3_PAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void f3_mfma0v(f32x16& d, const u32x4& a, const u32x4& b) {             // d = a x b(vgpr)
    asm volatile(PIML_F3_PAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f3_mfma(f32x16& d, const u32x4& a, const u32x4& b_acc) {           // d += a x B(agpr)
    asm volatile(PIML_F3_PAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void f3_mfmav(f32x16& d, const u32x4& a, const u32x4& b) {              // d += a x b(vgpr)
    asm volatile(PIML_F3_PAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f3_mfma32(f32x16& d, float a, float b) {                            // f32 instruction, all VGPRs
    asm volatile(PIML_F3_PAD "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f3_settle(f32x16& d) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(d)); }
// inputs of a tile that come from memory, requested one tile ahead
struct F3Pre {
    float4 gp[2][2], gm[2][2];     // [k-step s][half2]: features 32 w + 16 s + 8 half2 + 4 h .. + 3 of the lane's row
    unsigned kw;                   // keep word w of the row
    float xa[4];                   // x[row][2 s + h]: A operand of the H1 recomputation
    float xs;                      // staging: x[tile row tid >> 3][tid & 7]
    unsigned mk;                   // staging: dword tid of the tile's 256 sign dwords
    float g2[16];                  // SUMS: d/d(sum)[agent of row rho(r) + 4 h][feature 32 w + n] -- G2 before its mask
    unsigned m2;                   // SUMS: the lane's sign word of h2 (blocks 2 (w >> 1), + 1: 16 rows each)
};

// SUMS (PIML_POOL_TRAIN, include/piml_hip.h): the branch was trained on the agents' sums of h2 -- every row of an agent sees the
// same upstream gradient g = d/d(sum), so G2 = g[agent] * [h2 > 0] is loaded (one dword per row and lane: the lane's feature of
// the row's agent, 128 contiguous bytes per lane half) instead of computed: no W3^T fragments, no G3 staging, no layer A, and no
// phase 2 (dW3 / db3 follow from the decoder's folded first layer, network.hip: unfold).  The signs of h2 come in the layout of
// the exchanged forward layer (enc_fwd_sum_x3_kernel), which is this kernel's accumulator layout: one dword per lane and tile.
template <bool POOL, bool MSGS, bool DROP, bool GX, bool SUMS = false>
__global__ __launch_bounds__(F3_THREADS) void enc_bwd_fused_x3_kernel(F3Args F) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = (int)blockIdx.x, b = 0;
    if (bx >= F.nA[0]) { b = 1; bx -= F.nA[0]; }
    const piml_encoder_branch J = b ? F.A.br[1] : F.A.br[0];
    const int nwg = F.nA[b];
    const unsigned R = (unsigned)J.rows;                      // rows < 2^22 (checked on the host): byte offsets fit 32 bits
    const unsigned IN = __builtin_amdgcn_readfirstlane((unsigned)J.in_dim), K = __builtin_amdgcn_readfirstlane((unsigned)J.k);
    const unsigned kmagic = __builtin_amdgcn_readfirstlane((unsigned)((0x100000000ull + K - 1) / K));      // row / K == umulhi(row, kmagic)
    const int ntiles = (int)((R + 31) >> 5);
    const int n = lane & 31, h = lane >> 5;
    const float scale = J.scale;
    float* P = J.partials + (size_t)F.slot0[b] * F3_PART0 + (size_t)bx * F3_PART1;

    // Buffer resources: an offset past the range reads as zero / is not stored, so rows past the end and tiles past the last
    // cost no branch (kOut = an offset that is out of every range).
    auto rsrc = [&](const void* base, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    };
    constexpr unsigned kOut = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rs_gp = rsrc(J.g_pooled, POOL ? (R / K) * EH * 4 : 0u);
    const __amdgpu_buffer_rsrc_t rs_gm = rsrc(J.g_msgs, MSGS ? R * EH * 4 : 0u);
    const __amdgpu_buffer_rsrc_t rs_kb = rsrc(J.keep_bits, DROP ? R * 16 : 0u);
    const __amdgpu_buffer_rsrc_t rs_x = rsrc(J.x, R * IN * 4);
    const __amdgpu_buffer_rsrc_t rs_mk = rsrc(J.relu_mask, (unsigned)ntiles * 1024);
    const __amdgpu_buffer_rsrc_t rs_gx = rsrc(J.g_x, GX ? R * IN * 4 : 0u);
    auto ld1 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0); };
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    };

    // ---- this wave's weight fragments: block w of W3^T and W2^T, all eight k-blocks ----
    // (hi, mid) in registers, 128 of them; the lo pieces -- one of the six products reads them -- in a private part of LDS
    u32x4 wfA[8][2], wfB[8][2];
    u32x4* const wlo = reinterpret_cast<u32x4*>(smem + F3_WLO) + w * (2 * 8 * 64) + lane;       // + (layer * 8 + kb) * 64
    {
        const u32x4* imgA = reinterpret_cast<const u32x4*>(J.packed + PACK_F32 + 2 * X3_IMG) + lane;
        const u32x4* imgB = reinterpret_cast<const u32x4*>(J.packed + PACK_F32 + 3 * X3_IMG) + lane;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const int fb = w * 8 + kb;
            if (!SUMS) { wfA[kb][0] = imgA[(fb * 2) * 64]; wfA[kb][1] = imgA[(fb * 2 + 1) * 64]; }
            wfB[kb][0] = imgB[(fb * 2) * 64]; wfB[kb][1] = imgB[(fb * 2 + 1) * 64];
        }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const int fb = w * 8 + kb;
            if (!SUMS) wlo[kb * 64] = imgA[X3_HM / 4 + fb * 64];
            wlo[(8 + kb) * 64] = imgB[X3_HM / 4 + fb * 64];
        }
    }
    // W1 rows (padded to 8 columns) -> LDS for g_x; this lane's W1 / b1 values for the H1 recomputation
    const float* W1rows = J.packed + PACK_FWD + 32768;
    reinterpret_cast<float4*>(smem + F3_W1)[tid] = reinterpret_cast<const float4*>(W1rows)[tid];
    float w1v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) w1v[s] = W1rows[(32 * w + n) * 8 + 2 * s + h];
    const float b1v = J.b1[32 * w + n];
    float w1g[8];                                             // B operand of the g_x products: W1[32 w + 16 p + 4 j + (lane >> 4)][lane & 15]
#pragma unroll
    for (int q = 0; q < 8; ++q) w1g[q] = (GX && PIML_F3_GX_MFMA && (lane & 15) < 8) ? W1rows[(32 * w + 4 * q + (lane >> 4)) * 8 + (lane & 15)] : 0.f;

    // ---- per-lane LDS addresses ----
    u32x4* const bufA = reinterpret_cast<u32x4*>(smem + F3_BUFA) + lane;
    u32x4* const bufH = reinterpret_cast<u32x4*>(smem + F3_BUFH) + lane;
    // M, writer: feature f = 32 w + n, chunk 2 g + h of its 64-byte row at slot (chunk ^ ((f >> 1) & 7))
    const int fw = 32 * w + n;
    unsigned char* const Mw = smem + F3_M + fw * 64;
    const int swz_w = (fw >> 1) & 7;
    // M, reader (ds_read_b64_tr_b16): lane 4 q + pp of 16-lane group g16 supplies row (f0 + q), columns c0 + 4 pp .. + 3 with
    // c0 = 16 (g16 & 1), f0 = 16 kb + 8 half2 + 4 (g16 >> 1); ((f0 + q) >> 1) & 7 = 4 half2 + 2 (g16 >> 1) + (q >> 1)
    int mr[2];
    {
        const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g16 >> 1;
        const int chunk = 4 * (g16 & 1) + pp;
#pragma unroll
        for (int half2 = 0; half2 < 2; ++half2)
            mr[half2] = F3_M + (8 * half2 + 4 * hh + q) * 64 + ((chunk ^ (4 * half2 + 2 * hh + (q >> 1))) * 8);
    }
    // sign words: lane (n, h), register r needs bit bp of word (w >> 1) of source lane rho(r) + 4 h + 32 h', h' = (n >> 2) & 1
    const int bp = 16 * (w & 1) + (n & 3) + 4 * (n >> 3);
    const int mk_off = F3_MK + ((4 * h + 32 * ((n >> 2) & 1)) * 2 + (w >> 1)) * 4;
    const int xs_off = F3_XS + (4 * h) * 32;
    // G1 half block of pass p (features 16 p .. 16 p + 15 of the wave's block): written by the lanes that hold those features,
    // read by every lane (row n) for the columns 4 h .. 4 h + 3 of g_x
    float* const Tbase = reinterpret_cast<float*>(smem + F3_T) + w * (16 * F3_TROW);
    // (a divergent branch around the stores costs the register allocation of the whole loop: 256 VGPRs + 17 spills against 170.
    // The lanes outside the pass store to a slot of their own instead: an address select, no branch.)
    float* const Tw_in = Tbase + (n & 15) * F3_TROW + 4 * h;     // + 8 g: registers 4 g .. 4 g + 3
    float* const Tw_out = reinterpret_cast<float*>(smem + F3_TDUMMY) + (w * 32 + (lane & 15) + 16 * h) * 4;
    const float* const Tr = Tbase + n;                           // + i rows: feature 16 p + i, row n
    const float4* const W1l = reinterpret_cast<const float4*>(smem + F3_W1) + (32 * w) * 2 + h;      // + 2 f: columns 4 h .. 4 h + 3 of feature 32 w + f

    // ---- accumulators that live for the whole slab ----
    f32x16 c[4], sm[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c[jb][r] = 0.f; sm[jb][r] = 0.f; }
    float w1acc[8], db1 = 0.f, db2 = 0.f;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) w1acc[cc] = 0.f;

    // ---- the vector work that rides between the matrix instructions, cut into steps of about six instructions (a product holds
    // the issue port for 8 of its 32 cycles: what fits the other 24 is free).  A step's index is a constant once the loops are
    // unrolled; sched_barrier pins every step between its two products (F3_SLOT below). ----
    F3Pre S;                                                 // the NEXT tile's requests (a tile past the last one reads zeros)
    float pf_sc = 0.f;                                       // scale, or 0 for a row past the end: G3 = 0, and with it G2, G1, every sum
    auto pf_step = [&](int i, int tile) {
        const unsigned row = (unsigned)tile * 32 + n;
        const bool live = tile < ntiles;                     // (bitwise combinations: a short-circuit would branch)
        const bool valid = live & (row < R);
        if (SUMS && i < 4) {                               // registers 4 i .. 4 i + 3: rows rho(r) + 4 h of the tile, this lane's feature
            // agent of row 32 tile + m = a0 + (rem + m) / K with a0, rem = (32 tile) / K, % K on the scalar unit and (rem + m) < K + 32
            // divided by a 16-bit reciprocal (exact below 2^10 for K < 64: one 24-bit multiply + shift; v_mul_hi_u32 is quarter
            // rate).  No row test: R = agents * K, so a row or tile past the end is an agent past the end -- out of the resource's range
            const unsigned t32 = __builtin_amdgcn_readfirstlane((unsigned)tile * 32u);
            const unsigned a0 = __builtin_amdgcn_readfirstlane(__umulhi(t32, kmagic)), rem = t32 - a0 * K;
            const unsigned base = (a0 * EH + 32 * w + n) * 4, x0 = rem + 4 * h, rcp = (65536u + K - 1) / K;
#pragma unroll
            for (int r = 4 * i; r < 4 * i + 4; ++r) {
                const unsigned q = ((x0 + rho(r)) * rcp) >> 16;
                S.g2[r] = __uint_as_float(ld1(rs_gp, base + q * (EH * 4)));
            }
        } else if (SUMS && i == 4) {
            S.m2 = ld1(rs_mk, live ? ((unsigned)tile * 256 + 128 + 2 * lane + (w >> 1)) * 4 : kOut);
        } else if (i < 4) {
            const int s = i >> 1, half2 = i & 1;
            const unsigned f = 32 * w + 16 * s + 8 * half2 + 4 * h;
            S.gp[s][half2] = make_float4(0.f, 0.f, 0.f, 0.f);
            S.gm[s][half2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (POOL) S.gp[s][half2] = ld4(rs_gp, valid ? (__umulhi(row, kmagic) * EH + f) * 4 : kOut);
            if (MSGS) S.gm[s][half2] = ld4(rs_gm, valid ? (row * EH + f) * 4 : kOut);
        } else if (i == 4) {
            S.kw = 0xffffffffu;
            if (DROP) S.kw = ld1(rs_kb, valid ? (row * 4 + w) * 4 : kOut);
            pf_sc = valid ? scale : 0.f;
        } else if (i == 5) {
            // (SUMS: H1 of tile t + 1 is built under layer B of tile t, so its x rows are requested TWO tiles ahead)
            const unsigned xrow = SUMS ? row + 32u * (unsigned)nwg : row;
            const bool xvalid = SUMS ? ((tile + nwg < ntiles) & (xrow < R)) : valid;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned cx = 2u * s + h;
                const unsigned off = (xrow * IN + cx) * 4;
                S.xa[s] = __uint_as_float(ld1(rs_x, (xvalid & (cx < IN)) ? off : kOut));
            }
        } else {
            const unsigned srow = (unsigned)tile * 32 + (tid >> 3), cx = tid & 7;
            const unsigned off = (srow * IN + cx) * 4;
            S.xs = __uint_as_float(ld1(rs_x, (live & (srow < R) & (cx < IN)) ? off : kOut));
            const unsigned moff = ((unsigned)tile * 256 + tid) * 4;
            S.mk = ld1(rs_mk, live ? moff : kOut);
        }
    };
    // split3 (pack.hpp) in two halves
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
    // the tile's x rows and sign words -> LDS (parity buffers)
    auto stage = [&](int par_) {
        reinterpret_cast<float*>(smem + F3_XS + par_ * 1024)[tid] = S.xs;
        reinterpret_cast<unsigned*>(smem + F3_MK + par_ * 1024)[tid] = S.mk;
    };
    // this wave's two k-blocks of the next tile's G3 -> bufA, 22 steps: part j (k-step j >> 1, half j & 1) = values, then two
    // split halves for each of its two register pairs; a store step behind parts 1 and 3
    unsigned g3hi[4], g3mid[4], g3lo[4];
    float g3v[4];
    auto g3_step = [&](int st) {
        if (st == 10 || st == 21) {
            const int kb = 2 * w + (st == 21);
            bufA[(kb * 3 + 0) * 64] = (u32x4){g3hi[0], g3hi[1], g3hi[2], g3hi[3]};
            bufA[(kb * 3 + 1) * 64] = (u32x4){g3mid[0], g3mid[1], g3mid[2], g3mid[3]};
            bufA[(kb * 3 + 2) * 64] = (u32x4){g3lo[0], g3lo[1], g3lo[2], g3lo[3]};
            return;
        }
        const int t = st > 10 ? st - 1 : st, j = t / 5, sub = t % 5, s = j >> 1, half2 = j & 1;
        if (sub == 0) {
            const float4 a_ = S.gp[s][half2], g_ = S.gm[s][half2];
            g3v[0] = (a_.x + g_.x) * pf_sc; g3v[1] = (a_.y + g_.y) * pf_sc; g3v[2] = (a_.z + g_.z) * pf_sc; g3v[3] = (a_.w + g_.w) * pf_sc;
            if (DROP) {
                const unsigned m = S.kw >> (4 * h);
#pragma unroll
                for (int u = 0; u < 4; ++u) g3v[u] = keep_if(g3v[u], m, 16 * s + 8 * half2 + u);
            }
        } else if (sub == 1) split_a(g3v[0], g3v[1], g3hi[2 * half2]);
        else if (sub == 2) split_b(g3mid[2 * half2], g3lo[2 * half2]);
        else if (sub == 3) split_a(g3v[2], g3v[3], g3hi[2 * half2 + 1]);
        else split_b(g3mid[2 * half2 + 1], g3lo[2 * half2 + 1]);
    };
    // H1 = relu(...) -> pieces -> bufH, 18 steps: two split halves per register pair, a store step behind each k-step
    f32x16 hacc;
    unsigned hhi[4], hmid[4], hlo[4];
    // SUMS: H1 is double-buffered (the G3 buffer is free there): buffer (par) holds the running tile's, buffer (par ^ 1) is written
    // under layer B for the next tile
    u32x4* const bufH2 = SUMS ? reinterpret_cast<u32x4*>(smem + F3_BUFA) + lane : bufH;
    u32x4* hw = bufH;                                        // where h1_step writes
    auto h1_step = [&](int st) {
        if (st == 8 || st == 17) {
            const int s = st == 17;
            hw[((w * 2 + s) * 3 + 0) * 64] = (u32x4){hhi[0], hhi[1], hhi[2], hhi[3]};
            hw[((w * 2 + s) * 3 + 1) * 64] = (u32x4){hmid[0], hmid[1], hmid[2], hmid[3]};
            hw[((w * 2 + s) * 3 + 2) * 64] = (u32x4){hlo[0], hlo[1], hlo[2], hlo[3]};
            return;
        }
        const int t = st > 8 ? st - 1 : st, q = t >> 1, d = q & 3;       // register pair q = (2 q, 2 q + 1), dword d of its k-step
        if ((t & 1) == 0) split_a(relu_i(hacc[2 * q]), relu_i(hacc[2 * q + 1]), hhi[d]);
        else split_b(hmid[d], hlo[d]);
    };
    // g_x of a finished tile: the four waves' partials, fixed order; thread (row tid >> 3, column tid & 7)
    auto gx_store = [&](int tile) {
        const float* gp = reinterpret_cast<const float*>(smem + F3_GX) + tid;
        const float v = ((gp[0] + gp[256]) + gp[512]) + gp[768];
        const unsigned grow = (unsigned)tile * 32 + (tid >> 3), cx = tid & 7;
        const unsigned off = (grow * IN + cx) * 4;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_gx, (int)(((tile >= 0) & (grow < R) & (cx < IN)) ? off : kOut), 0, 0);
    };
    unsigned mkw[16];                                        // sign words of the 16 rows this lane masks, requested under the products
    auto mk_step = [&](int i, const unsigned* mk) {
        mkw[2 * i] = mk[2 * rho(2 * i)];
        mkw[2 * i + 1] = mk[2 * rho(2 * i + 1)];
    };

#define F3_SLOT(MF, FILL)                        \
    do {                                         \
        if (!(PIML_F3_SKIP & 32)) { MF; }        \
        __builtin_amdgcn_sched_barrier(0);       \
        FILL;                                    \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#define F3_SLOTW(MF, FILL)                       \
    do {                                         \
        if (!(PIML_F3_SKIP & 16)) { MF; }        \
        __builtin_amdgcn_sched_barrier(0);       \
        FILL;                                    \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
    // A chain layer = 48 products into ONE accumulator: the five small products of every k-block first (kblock_x3's order,
    // x3.hpp; their sum, <= 2^-8 of the result, forms exactly as in a second accumulator), then the eight hi x hi products on top
    // -- the matrix core cuts the low bits of the aligned addends, and this way only those eight additions happen at the
    // magnitude of the result, as with kblock_x3's two accumulators, without the 16 registers and the 16 additions of the second.
    // The activations' hi pieces stay in registers for the second pass (ahi).  Step fill(slot) behind product `slot`.
#define F3_KSMALL(FIRST, acc_, o_, Wh_, Wm_, fill_, slot0_)                                               \
    do {                                                                                                  \
        F3_SLOT(if (FIRST) f3_mfma0(acc_, o_[2], Wh_); else f3_mfma(acc_, o_[2], Wh_), fill_((slot0_) + 0)); \
        F3_SLOT(f3_mfma(acc_, o_[1], Wm_), fill_((slot0_) + 1));                                          \
        F3_SLOT(f3_mfmav(acc_, o_[0], o_[3]), fill_((slot0_) + 2));                                       \
        F3_SLOT(f3_mfma(acc_, o_[1], Wh_), fill_((slot0_) + 3));                                          \
        F3_SLOT(f3_mfma(acc_, o_[0], Wm_), fill_((slot0_) + 4));                                          \
    } while (0)

#ifdef PIML_F3_STAMPS
    unsigned long long st[16], tprev = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0;
#endif
    int tile = bx, par = 0, prev_tile = -1;
    float xa[4];
    {
#pragma unroll
        for (int i = 0; i < 7; ++i) pf_step(i, tile);
        stage(0);
        if (!SUMS) {
#pragma unroll
            for (int i = 0; i < 22; ++i) g3_step(i);
#pragma unroll
            for (int s = 0; s < 4; ++s) xa[s] = S.xa[s];
        } else {
            // H1 of the FIRST tile into buffer 0 (x of this tile: one plain load, the pipeline's requests run two tiles ahead), and
            // the x rows of the second tile as `xa` (pf_step(5, tile - nwg) asks for tile + ... = the tile after `tile - nwg + nwg`)
            float x0[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned cx = 2u * s + h, row0 = (unsigned)tile * 32 + n;
                x0[s] = __uint_as_float(ld1(rs_x, ((tile < ntiles) & (row0 < R) & (cx < IN)) ? (row0 * IN + cx) * 4 : kOut));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) xa[s] = S.xa[s];         // (pf_step(5, tile) above: the rows of tile + nwg)
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = b1v;
#pragma unroll
            for (int s = 0; s < 4; ++s) f3_mfma32(hacc, x0[s], w1v[s]);
            f3_settle(hacc);
#pragma unroll
            for (int st_ = 0; st_ < 18; ++st_) h1_step(st_);
        }
    }
    F3_STAMP(15);
    for (; tile < ntiles; tile += nwg, par ^= 1) {
        F3_BARRIER();                                                                      // B1: bufA, x rows, sign words, g_x partials
        F3_STAMP(0);
        // ============ region X: layer A (G2 = (G3 W3) * [h2 > 0]); in its shadow the next tile's requests, the g_x store of
        // ============ the previous tile and H1 = relu(W1 x + b1) with its split ============
        const int ntile = tile + nwg;
        if (!SUMS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = b1v;
#pragma unroll
            for (int s = 0; s < 4; ++s) f3_mfma32(hacc, xa[s], w1v[s]);
        }
        f32x16 acc;
        u32x4 opa[2][4], ahi[8];                               // operands of a k-block: pieces hi, mid, lo of the activations + LO of the weights
        unsigned m2w = 0;
        if (SUMS) {                                            // G2 before its mask: loaded a tile ahead; then the next tile's requests
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = S.g2[r];
            m2w = S.m2 >> (16 * (w & 1));
        } else {
#pragma unroll
            for (int p = 0; p < 3; ++p) opa[0][p] = bufA[p * 64];
            opa[0][3] = wlo[0];
        }
        const unsigned* mk2 = reinterpret_cast<const unsigned*>(smem + mk_off + par * 1024 + 512);       // layer 1 of the pair: h2
        auto fill_x = [&](int sl) {
            if (sl < 40 && sl % 5 == 0) {                      // the next k-block's operands
                const int kb = sl / 5 + 1;
                if (kb < 8) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) opa[kb & 1][p] = bufA[(kb * 3 + p) * 64];
                    opa[kb & 1][3] = wlo[kb * 64];
                }
                return;
            }
            const int f = sl < 40 ? sl - sl / 5 - 1 : sl - 8;  // 40 free steps
            if ((PIML_F3_SKIP & 1) && f >= 7) return;
            if (f < 7) pf_step(f, ntile);
            else if (f == 7) { if (GX) gx_store(prev_tile); }
            else if (f == 9) f3_settle(hacc);
            else if (f >= 10 && f < 28) h1_step(f - 10);
            else if (f >= 30 && f < 38) mk_step(f - 30, mk2);
        };
        if (SUMS) {                                            // no layer A: the next tile's requests and the g_x store; H1 rides under layer B
#pragma unroll
            for (int f = 0; f < 7; ++f) pf_step(f, ntile);
            if (GX) gx_store(prev_tile);
        } else {
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                ahi[kb] = opa[kb & 1][0];
                if (kb == 0) F3_KSMALL(true, acc, opa[0], wfA[0][0], wfA[0][1], fill_x, 0);
                else F3_KSMALL(false, acc, opa[kb & 1], wfA[kb][0], wfA[kb][1], fill_x, kb * 5);
            }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) F3_SLOT(f3_mfma(acc, ahi[kb], wfA[kb][0]), fill_x(40 + kb));
        }
        F3_STAMP(1);
        if (!SUMS) f3_settle(acc);
        if (!(PIML_F3_SKIP & 8)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = SUMS ? __builtin_amdgcn_sbfe(m2w, r, 1) : __builtin_amdgcn_sbfe(mkw[r], bp, 1);
                acc[r] = __uint_as_float(__float_as_uint(acc[r]) & (unsigned)t);
                db2 += acc[r];
            }
            Pieces2 G2;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                split_half(acc, G2, s);
#pragma unroll
                for (int g = 2 * s; g < 2 * s + 2; ++g) {
                    const int slot = ((2 * g + h) ^ swz_w) * 8;
                    const int d = 2 * (g & 1);
                    *reinterpret_cast<uint2*>(Mw + slot) = make_uint2(G2.hi[s][d], G2.hi[s][d + 1]);
                    *reinterpret_cast<uint2*>(Mw + 8192 + slot) = make_uint2(G2.mid[s][d], G2.mid[s][d + 1]);
                    *reinterpret_cast<uint2*>(Mw + 16384 + slot) = make_uint2(G2.lo[s][d], G2.lo[s][d + 1]);
                }
            }
        }
        F3_STAMP(2);
        F3_BARRIER();                                                                      // B2: M, bufH
        F3_STAMP(3);
        // ============ region Y: layer B (G1 = (G2 W2) * [h1 > 0]) with the NEXT tile's G3 in its shadow (bufA is free: every
        // ============ wave has passed B2), then dW2 += G2^T H1 with G1's sums in its shadow ============
        auto load_b = [&](u32x4 (&o)[4], int kb) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + mr[0] + kb * 1024 + p * 8192));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + mr[1] + kb * 1024 + p * 8192));
                const uint2 x = __builtin_bit_cast(uint2, lo4), y = __builtin_bit_cast(uint2, hi4);
                o[p] = (u32x4){x.x, x.y, y.x, y.y};
            }
            o[3] = wlo[(8 + kb) * 64];
        };
        load_b(opa[0], 0);
        if (SUMS) {                                            // H1 of the NEXT tile (its x rows arrived a tile ago) -> the other buffer
            hw = par ? bufH : bufH2;
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = b1v;
#pragma unroll
            for (int s = 0; s < 4; ++s) f3_mfma32(hacc, xa[s], w1v[s]);
        }
        const unsigned* mk1 = reinterpret_cast<const unsigned*>(smem + mk_off + par * 1024);             // layer 0 of the pair: h1
        auto fill_y = [&](int sl) {
            if (sl < 40 && sl % 5 == 0) {
                const int kb = sl / 5 + 1;
                if (kb < 8) load_b(opa[kb & 1], kb);
                return;
            }
            const int f = sl < 40 ? sl - sl / 5 - 1 : sl - 8;
            if ((PIML_F3_SKIP & 2) && f >= 1) return;
            if (f == 0) stage(par ^ 1);
            else if (f >= 1 && f < 23) {
                if (!SUMS) g3_step(f - 1);
                else if (f == 3) f3_settle(hacc);              // (three products behind the f32 instructions of H1)
                else if (f >= 4 && f < 22) h1_step(f - 4);
            }
            else if (f >= 30 && f < 38) mk_step(f - 30, mk1);
        };
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            ahi[kb] = opa[kb & 1][0];
            if (kb == 0) F3_KSMALL(true, acc, opa[0], wfB[0][0], wfB[0][1], fill_y, 0);
            else F3_KSMALL(false, acc, opa[kb & 1], wfB[kb][0], wfB[kb][1], fill_y, kb * 5);
        }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) F3_SLOT(f3_mfma(acc, ahi[kb], wfB[kb][0]), fill_y(40 + kb));
        f3_settle(acc);
#pragma unroll
        for (int s = 0; s < 4; ++s) xa[s] = S.xa[s];
        F3_STAMP(4);
        // dW2: 48 products (8 groups u = 4 s + jb of six) into the slab's accumulators; between them, pinned like above:
        //   the next group's operands | G1's mask and db1 (8 steps) | pass 0 of g_x: the lanes that hold features 0 .. 15 of the
        //   block lay their 32 rows into the wave's tile [feature][row], then every lane (row n, half h) adds those features'
        //   terms of g_x[row][4 h .. 4 h + 3] (reads one step ahead of their use) | pass 1 | dW1 (two rows per step)
        float gx[4] = {0.f, 0.f, 0.f, 0.f};
        u32x4 g2f[3], opb[2][3];
        const float4* xr = reinterpret_cast<const float4*>(smem + xs_off + par * 1024);
        auto load_g2 = [&](int s_) {                         // this lane's own G2 pieces of k-step s, back from the image it wrote them into
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint2 x = *reinterpret_cast<const uint2*>(Mw + p * 8192 + (((4 * s_ + h) ^ swz_w) * 8));
                const uint2 y = *reinterpret_cast<const uint2*>(Mw + p * 8192 + (((4 * s_ + 2 + h) ^ swz_w) * 8));
                g2f[p] = (u32x4){x.x, x.y, y.x, y.y};
            }
        };
        const u32x4* const hr = (SUMS && par) ? bufH2 : bufH;  // this tile's H1 pieces
        auto load_h = [&](u32x4 (&o)[3], int u_) {
            const int s_ = u_ >> 2, jb_ = u_ & 3;
#pragma unroll
            for (int p = 0; p < 3; ++p) o[p] = hr[((jb_ * 2 + s_) * 3 + p) * 64];
        };
        float tv[2][2];                                      // g_x: two features' G1 values of this lane's row, one step ahead
        float4 tw[2][2];                                     //      and their W1 columns
        float4 xv[2][4];                                     // dW1: two rows' x values, one step ahead
        auto gx_load = [&](int p_, int k) {                  // features 2 k, 2 k + 1 of pass p_
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                tv[k & 1][e] = Tr[(2 * k + e) * F3_TROW];
                tw[k & 1][e] = W1l[2 * (16 * p_ + 2 * k + e)];
            }
        };
        auto gx_fma = [&](int k) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float v = tv[k & 1][e];
                const float4 wv = tw[k & 1][e];
                gx[0] = __fmaf_rn(wv.x, v, gx[0]); gx[1] = __fmaf_rn(wv.y, v, gx[1]);
                gx[2] = __fmaf_rn(wv.z, v, gx[2]); gx[3] = __fmaf_rn(wv.w, v, gx[3]);
            }
        };
        auto t_write = [&](int p_) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float* dst = (n >> 4) == p_ ? Tw_in + 8 * g : Tw_out;
                *reinterpret_cast<float4*>(dst) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            }
        };
        auto x_load = [&](int k) {                           // rows (registers) 2 k, 2 k + 1
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                xv[k & 1][2 * e] = xr[2 * rho(2 * k + e)];
                xv[k & 1][2 * e + 1] = xr[2 * rho(2 * k + e) + 1];
            }
        };
        auto x_fma = [&](int k) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float4 xa4 = xv[k & 1][2 * e], xb4 = xv[k & 1][2 * e + 1];
                const float g = acc[2 * k + e];
                w1acc[0] = __fmaf_rn(g, xa4.x, w1acc[0]); w1acc[1] = __fmaf_rn(g, xa4.y, w1acc[1]);
                w1acc[2] = __fmaf_rn(g, xa4.z, w1acc[2]); w1acc[3] = __fmaf_rn(g, xa4.w, w1acc[3]);
                w1acc[4] = __fmaf_rn(g, xb4.x, w1acc[4]); w1acc[5] = __fmaf_rn(g, xb4.y, w1acc[5]);
                w1acc[6] = __fmaf_rn(g, xb4.z, w1acc[6]); w1acc[7] = __fmaf_rn(g, xb4.w, w1acc[7]);
            }
        };
        // g_x on the matrix instruction: operand A = lane (m = row & 15, k = lane >> 4) reads G1[feature 4 j + k of the pass][row 16 r + m]
        f32x4 gd[2];
        float ga[4][2];
        const float* const Tm = Tbase + (lane >> 4) * F3_TROW + (lane & 15);        // + 4 j features, + 16 r rows
        auto gm_load = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 2; ++r) ga[j][r] = Tm[4 * j * F3_TROW + 16 * r];
        };
        auto gm_mma = [&](int p_, int j) {
#pragma unroll
            for (int r = 0; r < 2; ++r) gd[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[j][r], w1g[4 * p_ + j], gd[r], 0, 0, 0);
        };
        auto gm_store = [&]() {                              // rows 16 r + 4 (lane >> 4) + i, column lane & 15 (columns >= 8: a slot of the lane's own)
            float* dst = (lane & 15) < 8 ? reinterpret_cast<float*>(smem + F3_GX) + (w * 32 + 4 * (lane >> 4)) * 8 + (lane & 15)
                                          : reinterpret_cast<float*>(smem + F3_TDUMMY) + (w * 32 + (lane & 31)) * 4 - 0;
            const int stride = (lane & 15) < 8 ? 8 : 0;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[(16 * r + i) * stride] = gd[r][i];
        };
        auto fill_w = [&](int sl) {
            if (sl % 6 == 0) {
                const int u_ = sl / 6 + 1;
                if (u_ < 8) load_h(opb[u_ & 1], u_);
                return;
            }
            if (sl == 23) { load_g2(1); return; }             // (behind the last product of k-step 0)
            const int f = sl - sl / 6 - 1 - (sl > 23);        // 39 free steps
            if (PIML_F3_SKIP & 4) return;
            if (f < 8) {                                       // G1 = (G2 W2) * [h1 > 0], two rows a step
#pragma unroll
                for (int r = 2 * f; r < 2 * f + 2; ++r) {
                    const int t = __builtin_amdgcn_sbfe(mkw[r], bp, 1);
                    acc[r] = __uint_as_float(__float_as_uint(acc[r]) & (unsigned)t);
                    db1 += acc[r];
                }
            } else if (GX && PIML_F3_GX_MFMA && f >= 8 && f < 26) {
                if (f == 8) { gd[0] = gd[1] = (f32x4){0.f, 0.f, 0.f, 0.f}; t_write(0); }
                else if (f == 9) gm_load();
                else if (f >= 11 && f < 15) gm_mma(0, f - 11);
                else if (f == 15) t_write(1);                 // (behind the pass-0 reads: a wave's LDS operations complete in order)
                else if (f == 16) gm_load();
                else if (f >= 18 && f < 22) gm_mma(1, f - 18);
                else if (f == 25) gm_store();
            } else if (GX && f == 8) { t_write(0); gx_load(0, 0); }
            else if (GX && f >= 9 && f < 17) { if (f < 16) gx_load(0, f - 8); gx_fma(f - 9); }
            else if (GX && f == 17) { t_write(1); gx_load(1, 0); }
            else if (GX && f >= 18 && f < 26) { if (f < 25) gx_load(1, f - 17); gx_fma(f - 18); }
            else if (f == 26) {
                if (GX && !PIML_F3_GX_MFMA) reinterpret_cast<float4*>(smem + F3_GX)[(w * 32 + n) * 2 + h] = make_float4(gx[0], gx[1], gx[2], gx[3]);
                x_load(0);
            } else if (f >= 27 && f < 35) { if (f < 34) x_load(f - 26); x_fma(f - 27); }
        };
        load_g2(0);
        load_h(opb[0], 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {                          // u = 4 s + jb
            const int jb = u & 3;
            const u32x4 (&o)[3] = opb[u & 1];
            F3_SLOTW(sm[jb] = mfma_bf(g2f[2], o[0], sm[jb]), fill_w(u * 6 + 0));
            F3_SLOTW(sm[jb] = mfma_bf(g2f[1], o[1], sm[jb]), fill_w(u * 6 + 1));
            F3_SLOTW(sm[jb] = mfma_bf(g2f[0], o[2], sm[jb]), fill_w(u * 6 + 2));
            F3_SLOTW(sm[jb] = mfma_bf(g2f[1], o[0], sm[jb]), fill_w(u * 6 + 3));
            F3_SLOTW(sm[jb] = mfma_bf(g2f[0], o[1], sm[jb]), fill_w(u * 6 + 4));
            F3_SLOTW(c[jb] = mfma_bf(g2f[0], o[0], c[jb]), fill_w(u * 6 + 5));
        }
        F3_STAMP(6);
        prev_tile = tile;
    }
    __syncthreads();
    if (GX) gx_store(prev_tile);
    // ---- the slot: dW2 | dW1 | db2 | db1 ----
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) P[(size_t)(32 * w + rho(r) + 4 * h) * EH + 32 * jb + n] = c[jb][r] + sm[jb][r];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) w1acc[cc] += __shfl_xor(w1acc[cc], 32, 64);
    db1 += __shfl_xor(db1, 32, 64);
    db2 += __shfl_xor(db2, 32, 64);
    if (h == 0) {
        float* o = P + EH * EH + (size_t)(32 * w + n) * IN;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
            if ((unsigned)cc < IN) o[cc] = w1acc[cc];
        P[EH * EH + 1024 + 32 * w + n] = db2;
        P[EH * EH + 1024 + EH + 32 * w + n] = db1;
    }
    for (unsigned cc = IN * 128 + tid; cc < 1024; cc += F3_THREADS) P[EH * EH + cc] = 0.f;      // the unused tail of the dW1 field
    F3_STAMP(11);

    // =====================================================================================================================
    // phase 2: dW3 = G3^T H2 and db3 over the same tiles, into the same 128 accumulator registers (the layer-0 slot of this
    // workgroup; until round 4 a launch of its own -- 22 us for 6 us of products: prologue, epilogue and launch boundary of
    // a second kernel).  Both operands contract over the ROWS.  They come from memory as rows: a lane loads four consecutive
    // features of a row with one 16-byte load (4-byte loads -- a feature per lane -- ran this phase at a quarter of the
    // speed), splits them and lays the pieces into a [row][feature] image in LDS, 8-byte chunks XOR-swizzled by the row;
    // ds_read_b64_tr_b16 then hands every lane its feature's rows: operand fragments with the feature on the lane and the
    // rows as the k index, for G3 (A: block w of this wave) and H2 (B: all four blocks).  Wave w loads rows 8 w .. 8 w + 7 of
    // the tile; rows past the end are agents past the end and read as zeros by the buffer range check.  Two image pairs: one
    // barrier per tile.  scale is applied once, to the sums.
    // =====================================================================================================================
    if (!SUMS && F.with_dw3) {
        float* P0 = J.partials + (size_t)bx * F3_PART0;
        // accumulators pinned to AGPRs through asm operands (left to itself hipcc moves all 128 registers to VGPRs and back in
        // every iteration of this loop)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { c[jb][r] = 0.f; sm[jb][r] = 0.f; }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) asm volatile("" : "+a"(c[jb]), "+a"(sm[jb]));
        float d3[4] = {0.f, 0.f, 0.f, 0.f};                  // db3 of features 4 n .. 4 n + 3 over this lane's rows
        const __amdgpu_buffer_rsrc_t rs_h2 = rsrc(J.h2, R * EH * 4);
        struct P2Pre { float4 hv[4], gp[4], gm[4]; unsigned kw[4]; };
        constexpr int P2_IMG = 3 * 32 * 256;                  // bytes of one image: [piece][row 32][feature 128] bf16
        // images: H2 / G3 of even tiles over bufA | M | bufH (72 KB of phase 1), of odd tiles over the weights' lo pieces (64 KB)
        auto img = [&](int pb, int which) { return smem + (pb ? F3_WLO : F3_BUFA) + which * P2_IMG; };
        static_assert(2 * P2_IMG <= F3_XS - F3_BUFA && 2 * P2_IMG <= 4 * 2 * 8 * 64 * 16, "the images fit the dead buffers");
        const unsigned v_row = (unsigned)(4 * h) * (EH * 4) + (unsigned)n * 16;       // lane part of a (rows, 128) offset
        auto p2_load = [&](P2Pre& Q, int tile_) {
            const bool live = tile_ < ntiles;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned row0 = (unsigned)tile_ * 32 + 8 * w + i;                          // scalar: the row of lane half 0 (half 1: + 4)
                const unsigned so = live ? row0 * (EH * 4) : kOut;
                Q.hv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_h2, (int)v_row, (int)so, 0));
                Q.gm[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MSGS) Q.gm[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_gm, (int)v_row, (int)so, 0));
                Q.gp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (POOL) {      // the agents of both lane halves on the scalar unit
                    const unsigned a0 = __builtin_amdgcn_readfirstlane(__umulhi(row0, kmagic)), a1 = __builtin_amdgcn_readfirstlane(__umulhi(row0 + 4, kmagic));
                    Q.gp[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_gp, (int)((unsigned)n * 16 + (h ? (a1 - a0) * (EH * 4) : 0u)),
                                                                                                 (int)(live ? a0 * (EH * 4) : kOut), 0));
                }
                Q.kw[i] = 0u;
                if (DROP) Q.kw[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_kb, (int)((4 * h) * 16 + (n >> 3) * 4), (int)(live ? row0 * 16 : kOut), 0);
            }
        };
        auto mfma_aa = [&](f32x16& d, const u32x4& a_, const u32x4& b_) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a_), "v"(b_));
        };
        // reader: lane 4 q + p of 16-lane group g16 supplies row r0 + q, chunk 8 blk + 4 (g16 & 1) + p; r0 = 16 s + 8 half2 + 4 (g16 >> 1)
        int tr_off;
        {
            const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
            tr_off = (4 * (g16 >> 1) + q) * 256 + (4 * (g16 & 1) + pp) * 8;                  // + (16 s + 8 half2) * 256 + ((8 blk) ^ (8 q)) * 8 + piece * 8192
        }
        const int trq = (lane >> 2) & 3;
        auto frag = [&](const unsigned char* im, int blk, int s_, int piece) -> u32x4 {
            const int o0 = tr_off + (16 * s_) * 256 + ((8 * blk) ^ (8 * trq)) * 8 + piece * 8192;
            const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(im + o0));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(im + o0 + 8 * 256));
            const uint2 x = __builtin_bit_cast(uint2, lo4), y = __builtin_bit_cast(uint2, hi4);
            return (u32x4){x.x, x.y, y.x, y.y};
        };
        // (two register sets addressed by a compile-time index: a run-time index would put them into scratch)
        P2Pre Q0, Q1;
        // laying a tile's rows into the images, 24 steps (row i, array G / H, three steps each: values + first half of the first
        // pair's split | its second half + first half of the second pair's | second half + the three stores): they ride between
        // the products of the tile BEFORE (the other image pair), pinned like the steps of phase 1
        float lv[4];
        unsigned lh0, lm0, ll0, lh1, lm1, ll1;
        auto lay_step = [&](const P2Pre& C, int pb, int st) {
            const int i = st / 6, arr = (st / 3) & 1, sub = st % 3;         // arr 0: G3, 1: H2
            if (sub == 0) {
                if (arr == 0) {
                    lv[0] = C.gp[i].x + C.gm[i].x; lv[1] = C.gp[i].y + C.gm[i].y; lv[2] = C.gp[i].z + C.gm[i].z; lv[3] = C.gp[i].w + C.gm[i].w;
                    if (DROP) {
                        const unsigned m = C.kw[i] >> ((4 * n) & 31);
#pragma unroll
                        for (int u = 0; u < 4; ++u) lv[u] = keep_if(lv[u], m, u);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) d3[u] += lv[u];
                } else {
                    lv[0] = C.hv[i].x; lv[1] = C.hv[i].y; lv[2] = C.hv[i].z; lv[3] = C.hv[i].w;
                }
                split_a(lv[0], lv[1], lh0);
            } else if (sub == 1) {
                split_b(lm0, ll0);
                split_a(lv[2], lv[3], lh1);
            } else {
                split_b(lm1, ll1);
                unsigned char* d = img(pb, arr == 0 ? 1 : 0) + (8 * w + 4 * h + i) * 256 + ((n ^ (8 * i)) * 8);        // (row & 3 = i)
                *reinterpret_cast<uint2*>(d) = make_uint2(lh0, lh1);
                *reinterpret_cast<uint2*>(d + 8192) = make_uint2(lm0, lm1);
                *reinterpret_cast<uint2*>(d + 16384) = make_uint2(ll0, ll1);
            }
        };
        auto mma_tile = [&](int pb, const P2Pre& Cn) {         // 48 products on image pair pb; the NEXT tile (Cn) is laid into the other pair
            const unsigned char* imH = img(pb, 0);
            const unsigned char* imG = img(pb, 1);
            u32x4 ga[2][3], ob[2][3];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int p = 0; p < 3; ++p) ga[s_][p] = frag(imG, w, s_, p);
#pragma unroll
            for (int p = 0; p < 3; ++p) ob[0][p] = frag(imH, 0, 0, p);
            auto fill = [&](int sl) {
                if (sl % 6 == 0) {                             // the next group's operands
                    const int u_ = sl / 6 + 1;
                    if (u_ < 8) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) ob[u_ & 1][p] = frag(imH, u_ >> 1, u_ & 1, p);
                    }
                    return;
                }
                const int f = sl - sl / 6 - 1;                 // 40 free steps
                if (f < 24) lay_step(Cn, pb ^ 1, f);
            };
#pragma unroll
            for (int u = 0; u < 8; ++u) {                      // u = 2 jb + s
                const int jb = u >> 1, s_ = u & 1;
                const u32x4 (&o)[3] = ob[u & 1];
                F3_SLOT(mfma_aa(sm[jb], ga[s_][2], o[0]), fill(u * 6 + 0));
                F3_SLOT(mfma_aa(sm[jb], ga[s_][1], o[1]), fill(u * 6 + 1));
                F3_SLOT(mfma_aa(sm[jb], ga[s_][0], o[2]), fill(u * 6 + 2));
                F3_SLOT(mfma_aa(sm[jb], ga[s_][1], o[0]), fill(u * 6 + 3));
                F3_SLOT(mfma_aa(sm[jb], ga[s_][0], o[1]), fill(u * 6 + 4));
                F3_SLOT(mfma_aa(c[jb], ga[s_][0], o[0]), fill(u * 6 + 5));
            }
        };
        int t2 = bx;
        p2_load(Q0, t2);
        p2_load(Q1, t2 + nwg);
        __syncthreads();                                       // (phase 1's last reads of the buffers under the images are done)
#pragma unroll
        for (int st_ = 0; st_ < 24; ++st_) lay_step(Q0, 0, st_);
        __syncthreads();
        for (; t2 < ntiles; t2 += 2 * nwg) {
            // image pair 0 holds tile t2, Q1 the requests of tile t2 + nwg, Q0 is free
            p2_load(Q0, t2 + 2 * nwg);
            mma_tile(0, Q1);
            F3_BARRIER();
            if (t2 + nwg >= ntiles) break;
            p2_load(Q1, t2 + 3 * nwg);
            mma_tile(1, Q0);
            F3_BARRIER();
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(c[jb]), "+a"(sm[jb]));
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) P0[(size_t)(32 * w + rho(r) + 4 * h) * EH + 32 * jb + n] = (c[jb][r] + sm[jb][r]) * scale;
        // db3: this lane's sums of features 4 n .. 4 n + 3; the two lane halves and the four waves meet in LDS (fixed order)
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(smem + F3_XS);                               // [wave 4][half 2][n 32] float4 = 4 KB
        red[(w * 2 + h) * 32 + n] = make_float4(d3[0], d3[1], d3[2], d3[3]);
        __syncthreads();
        if (tid < 128) {
            const float* rf = reinterpret_cast<const float*>(smem + F3_XS) + tid;            // feature tid of [slot 8][128]
            float acc3 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc3 += rf[q * 128];
            P0[EH * EH + tid] = acc3 * scale;
        }
    }
#ifdef PIML_F3_STAMPS
    F3_STAMP(12);
    if ((tid & 63) == 0)
        for (int i = 0; i < 16; ++i) g_f3_stamps[blockIdx.x * 64 + w * 16 + i] = st[i];
#endif
}

#ifdef PIML_F3_STAMPS
extern "C" __attribute__((visibility("default"))) int piml_f3_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f3_stamps), sizeof(unsigned long long) * 256 * 64);
}
#endif

template <bool P_, bool M_>
static int f3_set(int bytes) {
    auto set = [&](const void* f) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_fused_x3_kernel<P_, M_, false, false>))) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_fused_x3_kernel<P_, M_, false, true>))) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_fused_x3_kernel<P_, M_, true, false>))) return e;
    return set(reinterpret_cast<const void*>(enc_bwd_fused_x3_kernel<P_, M_, true, true>));
}

int enc_f3_set_attributes() {
    if (int e = f3_set<true, true>(F3_LDS_BYTES)) return e;
    if (int e = f3_set<true, false>(F3_LDS_BYTES)) return e;
    if (int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_bwd_fused_x3_kernel<true, false, false, false, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES)) return e;
    if (int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_bwd_fused_x3_kernel<true, false, false, true, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES)) return e;
    return f3_set<false, true>(F3_LDS_BYTES);
}

template <bool P_, bool M_>
static void f3_go(const F3Args& F, dim3 g, bool drop, bool gx, hipStream_t s) {
    const dim3 b(F3_THREADS);
    if (drop && gx) hipLaunchKernelGGL((enc_bwd_fused_x3_kernel<P_, M_, true, true>), g, b, F3_LDS_BYTES, s, F);
    else if (drop) hipLaunchKernelGGL((enc_bwd_fused_x3_kernel<P_, M_, true, false>), g, b, F3_LDS_BYTES, s, F);
    else if (gx) hipLaunchKernelGGL((enc_bwd_fused_x3_kernel<P_, M_, false, true>), g, b, F3_LDS_BYTES, s, F);
    else hipLaunchKernelGGL((enc_bwd_fused_x3_kernel<P_, M_, false, false>), g, b, F3_LDS_BYTES, s, F);
}

// A: the launch's branches (both with the same kinds of upstream gradients, keep bits and g_x: checked by the caller);
// nA[b] workgroups and slot0[b] layer-0 slots in front for branch b
void enc_f3_launch(const EncArgs& A, const int* nA, const int* slot0, bool with_dw3, hipStream_t s, bool sums) {
    F3Args F;
    F.A = A;
    F.with_dw3 = with_dw3 ? 1 : 0;
    F.nA[0] = nA[0]; F.nA[1] = A.nbr > 1 ? nA[1] : 0;
    F.slot0[0] = slot0[0]; F.slot0[1] = A.nbr > 1 ? slot0[1] : 0;
    const bool pool = A.br[0].g_pooled != nullptr, msgs = A.br[0].g_msgs != nullptr, drop = A.br[0].keep_bits != nullptr;
    const bool gx = A.br[0].g_x != nullptr;
    const dim3 g((unsigned)(F.nA[0] + F.nA[1]));
    if (sums) {          // (checked by the caller: g_pooled only, no keep bits)
        if (gx) hipLaunchKernelGGL((enc_bwd_fused_x3_kernel<true, false, false, true, true>), g, dim3(F3_THREADS), F3_LDS_BYTES, s, F);
        else hipLaunchKernelGGL((enc_bwd_fused_x3_kernel<true, false, false, false, true>), g, dim3(F3_THREADS), F3_LDS_BYTES, s, F);
        return;
    }
    if (pool && msgs) f3_go<true, true>(F, g, drop, gx, s);
    else if (pool) f3_go<true, false>(F, g, drop, gx, s);
    else f3_go<false, true>(F, g, drop, gx, s);
}

}  // namespace piml
