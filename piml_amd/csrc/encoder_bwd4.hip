// Backward of the PINNSF encoder in one launch, EIGHT waves per workgroup (two per SIMD): the dX chain, every weight
// gradient, g_x -- the pre-activation gradients g2 / g1 never leave the CU (round 4).
//
// Reference arithmetic: the autograd of MLP(in, [128, 128, 128]) (src/models/model.py:40-65) under the processor
// Dropout_p(2 x) and the neighbour-axis sum (:82-119, :1279-1283):
//     G3 = keep * scale * (g_pooled[row / k] + g_msgs[row])               dW3 = G3^T H2, db3 = colsum G3     (phase 2)
//     G2 = (G3 W3) * [h2 > 0]        dW2 = G2^T H1, db2 = colsum G2        H1 = relu(W1 x + b1)               (phase 1)
//     G1 = (G2 W2) * [h1 > 0]        dW1 = G1^T X,  db1 = colsum G1        g_x = G1 W1
//
// encoder_bwd3.hip is the same algorithm with FOUR waves of 512 registers (one per SIMD, 32-feature blocks on
// v_mfma_f32_32x32x16_bf16).  Measured there (tools/probe_mfma_slots.hip, variant builds): a lone wave per SIMD issues in
// order, so its time is the SUM of its matrix, vector and LDS instructions whatever the interleaving -- 19 us of products
// became 45.  Here a wave owns a 16-feature block and the products run on v_mfma_f32_16x16x32_bf16:
//   * weight fragments of its block, (hi, mid) of both layers: 64 registers, held in AGPRs as `asm` operands for the whole
//     slab (hipcc gives a builtin's A / B operands VGPRs only and copies parked AGPRs back in front of every use); the lo
//     pieces -- one of the six products reads them -- in a private part of LDS;
//   * its 8 output blocks of dW2 (16 x 16 each, main + small accumulator): 64 registers;
//   * that is 128 of a wave's 256 registers at two waves per SIMD: while one wave of a SIMD waits for its products' pipe or
//     an LDS read, the other issues its vector work -- the overlap no instruction order gave the four-wave kernel.
// Orientation as in encoder_bwd3.hip: D[row][feature] = sum_k act[row][k] W[k][feature] (activations = A operand), so a layer's
// result has its feature on the lane (l & 15) and the rows in the registers (row 16 rh + 4 (l >> 4) + i for register i of
// row half rh) -- the operand layout of every product that contracts over the ROWS (dW2 = G2^T H1: the 8 registers, split and
// packed pairwise, are the A fragment; H1 is recomputed in the same layout and travels through LDS as the B fragments).
// The next chain layer contracts over the features: that transposition rides on the hand-over through LDS ([feature][row]
// image of bf16 pieces, 8-byte chunks XOR-swizzled, read back with ds_read_b64_tr_b16).
// k order of every 32-deep fragment: element j of lane group g = l >> 4 is index 32 ks + 16 (g >> 1) + 8 (j >> 2) + 4 (g & 1)
// + (j & 3), which makes lane (c, g) of this kernel lane (16 (v & 1) + c, g & 1) of k-block 2 ks + (g >> 1) of the packed
// 32x32x16 images (pack.hpp): the transposed weight images of the dX kernels serve unchanged.
#include "common.hpp"
#include "encoder.hpp"
#include "x3.hpp"

namespace piml {

typedef short s16x4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_ lds_s16x4_;
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));

constexpr int F4_THREADS = 512;
// LDS (bytes)
constexpr int F4_BUFA = 0;                                   // G3 pieces, A fragments: [ks 4][rh 2][piece 3][lane 64] u32x4
constexpr int F4_M = F4_BUFA + 4 * 2 * 3 * 64 * 16;          // G2 pieces, [piece 3][feature 128][row 32] bf16, swizzled 8-byte chunks
constexpr int F4_BUFH = F4_M + 3 * 128 * 64;                 // H1 pieces, B fragments: [block 8][piece 3][lane 64] u32x4
constexpr int F4_XS = F4_BUFH + 8 * 3 * 64 * 16;             // the tile's x rows [parity 2][32][8] floats
constexpr int F4_MK = F4_XS + 2 * 1024;                      // the tile's sign words [parity 2][layer 2][lane 64] uint2
constexpr int F4_W1 = F4_MK + 2 * 1024;                      // W1 rows [128][8] floats
constexpr int F4_GX = F4_W1 + 4096;                          // g_x partials [wave 8][row 32][8] floats
constexpr int F4_WLO = F4_GX + 8 * 32 * 8 * 4;               // LO pieces of the wave's weight fragments [wave 8][layer 2][ks 4][lane 64] u32x4
constexpr int F4_LDS_BYTES = F4_WLO + 8 * 2 * 4 * 64 * 16;
static_assert(F4_LDS_BYTES <= 160 * 1024, "fits the CU");
// G1 block for g_x, [feature 16][36] floats per wave, over the wave's OWN fragment of bufA (3 KB): written by that wave alone
// (top of a tile), read by all waves between the tile's two barriers; the G1 block is written and read by the wave behind
// the second barrier, in front of its own next bufA write.
constexpr int F4_TROW = 36;
static_assert(16 * F4_TROW * 4 <= 3 * 64 * 16, "a G1 block fits the wave's part of bufA");

struct F4Args {
    EncArgs A;
    int nA[2];          // workgroups of branch 0 / branch 1 (grid = their sum)
    int slot0[2];       // layer-0 slots (F4_PART0 floats each) in front of the layer-1 slots in the branch's `partials`
    int with_dw3;       // phase 2: this launch also writes the layer-0 slots (dW3 | db3), slot = workgroup index within the branch
};

constexpr int F4_PART0 = EH * EH + EH;                       // = DW2_PART0 (encoder_dw2.hip): dW3 | db3
constexpr int F4_PART1 = EH * EH + 1024 + 2 * EH;            // = DW2_PART1: dW2 | dW1 (1024-float field) | db2 | db1

typedef float f32x4_ __attribute__((ext_vector_type(4)));

// ---- matrix instructions as asm statements (see encoder_bwd3.hip): the weight operand in an AGPR ("a"), results in VGPRs for
// the chain (read by vector instructions right away) and in AGPRs for the slab accumulators.  hipcc pads nothing inside an asm
// statement.  The chain's operands come straight from LDS loads and an accumulator's first product starts from the literal 0,
// so no vector instruction writes a register a chain product reads: no wait states in front (PIML_F4_PAD for A/B); the dW
// products read pieces a vector instruction made (two wait states in front) and f4_settle() stands between a chain's last
// product and the first vector instruction that reads the result. ----
#ifndef PIML_F4_PAD
#define PIML_F4_PAD ""
#endif
__device__ __forceinline__ void f4_mfma(f32x4_& d, const u32x4& a, const u32x4& b_acc) {
    asm volatile(PIML_F4_PAD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void f4_mfma0(f32x4_& d, const u32x4& a, const u32x4& b_acc) {        // d = a x B: a chain's first product
    asm volatile(PIML_F4_PAD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void f4_mfmav(f32x4_& d, const u32x4& a, const u32x4& b) {
    asm volatile(PIML_F4_PAD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f4_mfma_acc(f32x4_& d, const u32x4& a, const u32x4& b) {        // slab accumulator in an AGPR
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f4_mfma32(f32x4_& d, float a, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f4_settle(f32x4_& d) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(d)); }
__device__ __forceinline__ void f4_settle_acc(f32x4_& d) { asm volatile("s_nop 7\n\ts_nop 7" : "+a"(d)); }
__device__ __forceinline__ float f4_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// 8 values (two row halves x four registers: rows 16 rh + 4 g + i) -> the three bf16 pieces, element j = 4 rh + i
struct F4Pieces {
    u32x4 hi, mid, lo;
};
__device__ __forceinline__ void f4_split(const f32x4_& a0, const f32x4_& a1, F4Pieces& P) {
    unsigned hi[4], mid[4], lo[4];
    split3(a0[0], a0[1], hi[0], mid[0], lo[0]);
    split3(a0[2], a0[3], hi[1], mid[1], lo[1]);
    split3(a1[0], a1[1], hi[2], mid[2], lo[2]);
    split3(a1[2], a1[3], hi[3], mid[3], lo[3]);
    P.hi = (u32x4){hi[0], hi[1], hi[2], hi[3]};
    P.mid = (u32x4){mid[0], mid[1], mid[2], mid[3]};
    P.lo = (u32x4){lo[0], lo[1], lo[2], lo[3]};
}

// inputs of a tile that come from memory, requested one tile ahead
struct F4Pre {
    float4 gp[2], gm[2];           // [half2]: features f0 + 8 half2 .. + 3 of the lane's row (this wave's fragment of G3)
    unsigned kw;                   // keep word of the row that holds those features
    float xa[2][2];                // x[16 rh + c][4 s + g]: A operand of the H1 recomputation
    unsigned st;                   // staging: x[tile row][col] (threads 0 .. 255) or a sign dword (threads 256 .. 511)
};

template <bool POOL, bool MSGS, bool DROP, bool GX>
__global__ __launch_bounds__(F4_THREADS) void enc_bwd_fused8_x3_kernel(F4Args F) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);              // wave = 16-feature block
    int bx = (int)blockIdx.x, b = 0;
    if (bx >= F.nA[0]) { b = 1; bx -= F.nA[0]; }
    const piml_encoder_branch J = b ? F.A.br[1] : F.A.br[0];
    const int nwg = F.nA[b];
    const unsigned R = (unsigned)J.rows;                      // rows < 2^22 (checked on the host): byte offsets fit 32 bits
    const unsigned IN = __builtin_amdgcn_readfirstlane((unsigned)J.in_dim), K = __builtin_amdgcn_readfirstlane((unsigned)J.k);
    const unsigned kmagic = __builtin_amdgcn_readfirstlane((unsigned)((0x100000000ull + K - 1) / K));      // row / K == umulhi(row, kmagic)
    const int ntiles = (int)((R + 31) >> 5);
    const int c = lane & 15, g = lane >> 4;
    const float scale = J.scale;
    float* P = J.partials + (size_t)F.slot0[b] * F4_PART0 + (size_t)bx * F4_PART1;

    auto rsrc = [&](const void* base, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    };
    constexpr unsigned kOut = 0xfffffff0u;                    // an offset out of every range: reads zero, is not stored
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

    // ---- this wave's weight fragments: features 16 v .. 16 v + 15 of W3^T and W2^T, four 32-deep k-steps ----
    u32x4 wfA[4][2], wfB[4][2];
    u32x4* const wlo = reinterpret_cast<u32x4*>(smem + F4_WLO) + v * (2 * 4 * 64) + lane;       // + (layer * 4 + ks) * 64
    {
        const u32x4* imgA = reinterpret_cast<const u32x4*>(J.packed + PACK_F32 + 2 * X3_IMG);
        const u32x4* imgB = reinterpret_cast<const u32x4*>(J.packed + PACK_F32 + 3 * X3_IMG);
        const int il = 16 * (v & 1) + c + 32 * (g & 1);       // lane of the 32x32x16 image
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int fb = (v >> 1) * 8 + 2 * ks + (g >> 1);
            wfA[ks][0] = imgA[(fb * 2) * 64 + il]; wfA[ks][1] = imgA[(fb * 2 + 1) * 64 + il];
            wfB[ks][0] = imgB[(fb * 2) * 64 + il]; wfB[ks][1] = imgB[(fb * 2 + 1) * 64 + il];
            wlo[ks * 64] = imgA[X3_HM / 4 + fb * 64 + il];
            wlo[(4 + ks) * 64] = imgB[X3_HM / 4 + fb * 64 + il];
        }
    }
    const float* W1rows = J.packed + PACK_FWD + 32768;        // [f][8], zero-padded columns
    if (tid < 256) reinterpret_cast<float4*>(smem + F4_W1)[tid] = reinterpret_cast<const float4*>(W1rows)[tid];
    float w1v[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) w1v[s] = W1rows[(16 * v + c) * 8 + 4 * s + g];
    const float b1v = J.b1[16 * v + c];

    // ---- per-lane LDS addresses ----
    u32x4* const bufA = reinterpret_cast<u32x4*>(smem + F4_BUFA) + lane;       // + ((ks * 2 + rh) * 3 + piece) * 64
    u32x4* const bufH = reinterpret_cast<u32x4*>(smem + F4_BUFH) + lane;       // + (block * 3 + piece) * 64
    const int ks_b = v >> 1, rh_b = v & 1;                                     // the fragment of G3 this wave builds
    const int fw = 16 * v + c;                                                 // this lane's feature
    unsigned char* const Mw = smem + F4_M + fw * 64;                           // chunk 4 rh + g at slot (chunk ^ ((fw >> 1) & 7))
    const int swz_w = (fw >> 1) & 7;
    // M, reader: lane 4 q + p of group g supplies row (f0 + q), data rows 16 rh + 4 p .. + 3, f0 = 32 ks + 16 (g >> 1) + 4 (g & 1) + 8 half2;
    // ((f0 + q) >> 1) & 7 = (2 (g & 1) + 4 half2 + (q >> 1)) & 7
    int mr[2][2];                                                              // [rh][half2], + ks * 2048 + piece * 8192
    {
        const int q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int half2 = 0; half2 < 2; ++half2)
                mr[rh][half2] = F4_M + (16 * (g >> 1) + 4 * (g & 1) + 8 * half2 + q) * 64 + (((4 * rh + pp) ^ ((2 * (g & 1) + 4 * half2 + (q >> 1)) & 7)) * 8);
    }
    // sign words: lane (c, g), register (rh, i) = row 16 rh + 4 g + i, feature 16 v + c: bit bp of dword (v >> 2) of source lane row + 32 h'
    const int fl = 16 * (v & 1) + c;
    const int bp = 16 * ((v >> 1) & 1) + (fl & 3) + 4 * (fl >> 3);
    const int mk_off = F4_MK + ((4 * g + 32 * ((fl >> 2) & 1)) * 2 + (v >> 2)) * 4;         // + (16 rh + i) * 8 + layer * 512 + par * 1024
    const int xs_off = F4_XS + (4 * g) * 32;                                                // + (16 rh + i) * 32 + par * 1024
    float* const Tbase = reinterpret_cast<float*>(smem + F4_BUFA + ((ks_b * 2 + rh_b) * 3) * 64 * 16);
    float* const Tw = Tbase + c * F4_TROW + 4 * g;                                          // + 16 rh: registers of row half rh
    const int rown = lane & 31, hh = lane >> 5;
    const float* const Tr = Tbase + rown;                                                   // + f * TROW: feature f of the block, row rown
    const float4* const W1l = reinterpret_cast<const float4*>(smem + F4_W1) + (16 * v) * 2 + hh;      // + 2 f: columns 4 hh .. + 3

    // ---- accumulators that live for the whole slab ----
    f32x4_ cacc[8], sacc[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int i = 0; i < 4; ++i) { cacc[jb][i] = 0.f; sacc[jb][i] = 0.f; }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) asm volatile("" : "+a"(cacc[jb]), "+a"(sacc[jb]));
    float w1acc[8], db1 = 0.f, db2 = 0.f;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) w1acc[cc] = 0.f;

    auto prefetch = [&](int tile) -> F4Pre {
        F4Pre S;
        const bool live = tile < ntiles;                     // (bitwise combinations: a short-circuit would branch)
        const unsigned row = (unsigned)tile * 32 + 16 * rh_b + c;
        const bool valid = live & (row < R);
#pragma unroll
        for (int half2 = 0; half2 < 2; ++half2) {
            const unsigned f = 32 * ks_b + 16 * (g >> 1) + 4 * (g & 1) + 8 * half2;
            S.gp[half2] = make_float4(0.f, 0.f, 0.f, 0.f);
            S.gm[half2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (POOL) S.gp[half2] = ld4(rs_gp, valid ? (__umulhi(row, kmagic) * EH + f) * 4 : kOut);
            if (MSGS) S.gm[half2] = ld4(rs_gm, valid ? (row * EH + f) * 4 : kOut);
        }
        S.kw = 0u;
        if (DROP) S.kw = ld1(rs_kb, valid ? (row * 4 + ks_b) * 4 : kOut);
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned xrow = (unsigned)tile * 32 + 16 * rh + c, cx = 4u * s + g;
                const unsigned off = (xrow * IN + cx) * 4;
                S.xa[rh][s] = __uint_as_float(ld1(rs_x, (live & (xrow < R) & (cx < IN)) ? off : kOut));
            }
        if (tid < 256) {
            const unsigned srow = (unsigned)tile * 32 + (tid >> 3), cx = tid & 7;
            const unsigned off = (srow * IN + cx) * 4;
            S.st = ld1(rs_x, (live & (srow < R) & (cx < IN)) ? off : kOut);
        } else {
            S.st = ld1(rs_mk, live ? ((unsigned)tile * 256 + (tid - 256)) * 4 : kOut);
        }
        return S;
    };
    auto gx_store = [&](int tile) {                          // thread (row tid >> 3, column tid & 7), tid < 256: the eight waves' partials, fixed order
        const float* gp = reinterpret_cast<const float*>(smem + F4_GX) + (tid & 255);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += gp[q * 256];
        const unsigned grow = (unsigned)tile * 32 + ((tid & 255) >> 3), cx = tid & 7;
        const unsigned off = (grow * IN + cx) * 4;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), rs_gx, (int)(((tile >= 0) & (tid < 256) & (grow < R) & (cx < IN)) ? off : kOut), 0, 0);
    };

    int tile = bx, par = 0, prev_tile = -1;
    F4Pre S = prefetch(tile);
    for (; tile < ntiles; tile += nwg, par ^= 1) {
        // ============ top: stage the tile's x rows / sign words, build this wave's fragment of G3 ============
        if (tid < 256) reinterpret_cast<unsigned*>(smem + F4_XS + par * 1024)[tid] = S.st;
        else reinterpret_cast<unsigned*>(smem + F4_MK + par * 1024)[tid - 256] = S.st;
        {
            unsigned hi[4], mid[4], lo[4];
            const unsigned m = S.kw >> (16 * (g >> 1) + 4 * (g & 1));
#pragma unroll
            for (int half2 = 0; half2 < 2; ++half2) {
                const float4 a_ = S.gp[half2], g_ = S.gm[half2];
                float x[4] = {(a_.x + g_.x) * scale, (a_.y + g_.y) * scale, (a_.z + g_.z) * scale, (a_.w + g_.w) * scale};
                if (DROP) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = keep_if(x[u], m, 8 * half2 + u);
                }
                split3(x[0], x[1], hi[2 * half2], mid[2 * half2], lo[2 * half2]);
                split3(x[2], x[3], hi[2 * half2 + 1], mid[2 * half2 + 1], lo[2 * half2 + 1]);
            }
            const int fr = (ks_b * 2 + rh_b) * 3;
            bufA[(fr + 0) * 64] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
            bufA[(fr + 1) * 64] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
            bufA[(fr + 2) * 64] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        }
        float xa[2][2];
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int s = 0; s < 2; ++s) xa[rh][s] = S.xa[rh][s];
        __syncthreads();                                                                   // B1: bufA, x rows, sign words, g_x partials
        S = prefetch(tile + nwg);                                                          // the next tile's requests
        if (GX) gx_store(prev_tile);
        // ============ H1 = relu(W1 x + b1) of this wave's features -> bufH ============
        {
            f32x4_ hacc[2];
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hacc[rh][i] = b1v;
#pragma unroll
                for (int s = 0; s < 2; ++s) f4_mfma32(hacc[rh], xa[rh][s], w1v[s]);
            }
            f4_settle(hacc[0]);
            f4_settle(hacc[1]);
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int i = 0; i < 4; ++i) hacc[rh][i] = f4_relu(hacc[rh][i]);
            F4Pieces H;
            f4_split(hacc[0], hacc[1], H);
            bufH[(v * 3 + 0) * 64] = H.hi;
            bufH[(v * 3 + 1) * 64] = H.mid;
            bufH[(v * 3 + 2) * 64] = H.lo;
        }
        // ============ layer A: G2 = (G3 W3) * [h2 > 0] ============
        F4Pieces G2;
        {
            f32x4_ acc[2], sma[2];
            u32x4 op[2][3], wl[2];                             // operands of one (k-step, row half); the next one in flight under its six products
#pragma unroll
            for (int p = 0; p < 3; ++p) op[0][p] = bufA[p * 64];
            wl[0] = wlo[0];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                      // u = 2 ks + rh
                const int ks = u >> 1, rh = u & 1;
                if (u < 7) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) op[(u + 1) & 1][p] = bufA[((u + 1) * 3 + p) * 64];
                    if (rh == 1) wl[(ks + 1) & 1] = wlo[(ks + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&o)[3] = op[u & 1];
                if (ks == 0) f4_mfma0(sma[rh], o[2], wfA[ks][0]);
                else f4_mfma(sma[rh], o[2], wfA[ks][0]);
                f4_mfma(sma[rh], o[1], wfA[ks][1]);
                f4_mfmav(sma[rh], o[0], wl[ks & 1]);
                f4_mfma(sma[rh], o[1], wfA[ks][0]);
                f4_mfma(sma[rh], o[0], wfA[ks][1]);
                if (ks == 0) f4_mfma0(acc[rh], o[0], wfA[ks][0]);
                else f4_mfma(acc[rh], o[0], wfA[ks][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
            f4_settle(acc[0]); f4_settle(acc[1]); f4_settle(sma[0]); f4_settle(sma[1]);
            const unsigned* mk = reinterpret_cast<const unsigned*>(smem + mk_off + par * 1024 + 512);     // layer 1 of the pair: h2
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = __builtin_amdgcn_sbfe(mk[(16 * rh + i) * 2], bp, 1);
                    acc[rh][i] = __uint_as_float(__float_as_uint(acc[rh][i] + sma[rh][i]) & (unsigned)t);
                    db2 += acc[rh][i];
                }
            f4_split(acc[0], acc[1], G2);
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int slot = ((4 * rh + g) ^ swz_w) * 8;
                *reinterpret_cast<uint2*>(Mw + slot) = make_uint2(G2.hi[2 * rh], G2.hi[2 * rh + 1]);
                *reinterpret_cast<uint2*>(Mw + 8192 + slot) = make_uint2(G2.mid[2 * rh], G2.mid[2 * rh + 1]);
                *reinterpret_cast<uint2*>(Mw + 16384 + slot) = make_uint2(G2.lo[2 * rh], G2.lo[2 * rh + 1]);
            }
        }
        __syncthreads();                                                                   // B2: M, bufH
        // ============ dW2 += G2^T H1: this wave's 16 features of G2 against all eight blocks of H1 ============
        {
            u32x4 ob[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) ob[0][p] = bufH[p * 64];
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) {
                if (jb < 7) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) ob[(jb + 1) & 1][p] = bufH[((jb + 1) * 3 + p) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&o)[3] = ob[jb & 1];
                f4_mfma_acc(sacc[jb], G2.lo, o[0]);
                f4_mfma_acc(sacc[jb], G2.mid, o[1]);
                f4_mfma_acc(sacc[jb], G2.hi, o[2]);
                f4_mfma_acc(sacc[jb], G2.mid, o[0]);
                f4_mfma_acc(sacc[jb], G2.hi, o[1]);
                f4_mfma_acc(cacc[jb], G2.hi, o[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ============ layer B: G1 = (G2 W2) * [h1 > 0] ============
        f32x4_ acc[2];
        {
            f32x4_ sma[2];
            u32x4 op[2][3], wl[2];
            auto load_b = [&](u32x4 (&o)[3], int ks, int rh) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const s16x4_ lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_*)(smem + mr[rh][0] + ks * 2048 + p * 8192));
                    const s16x4_ hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_*)(smem + mr[rh][1] + ks * 2048 + p * 8192));
                    const uint2 x = __builtin_bit_cast(uint2, lo4), y = __builtin_bit_cast(uint2, hi4);
                    o[p] = (u32x4){x.x, x.y, y.x, y.y};
                }
            };
            load_b(op[0], 0, 0);
            wl[0] = wlo[4 * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                      // u = 2 ks + rh
                const int ks = u >> 1, rh = u & 1;
                if (u < 7) {
                    load_b(op[(u + 1) & 1], (u + 1) >> 1, (u + 1) & 1);
                    if (rh == 1) wl[(ks + 1) & 1] = wlo[(4 + ks + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&o)[3] = op[u & 1];
                if (ks == 0) f4_mfma0(sma[rh], o[2], wfB[ks][0]);
                else f4_mfma(sma[rh], o[2], wfB[ks][0]);
                f4_mfma(sma[rh], o[1], wfB[ks][1]);
                f4_mfmav(sma[rh], o[0], wl[ks & 1]);
                f4_mfma(sma[rh], o[1], wfB[ks][0]);
                f4_mfma(sma[rh], o[0], wfB[ks][1]);
                if (ks == 0) f4_mfma0(acc[rh], o[0], wfB[ks][0]);
                else f4_mfma(acc[rh], o[0], wfB[ks][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
            f4_settle(acc[0]); f4_settle(acc[1]); f4_settle(sma[0]); f4_settle(sma[1]);
            const unsigned* mk = reinterpret_cast<const unsigned*>(smem + mk_off + par * 1024);           // layer 0 of the pair: h1
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = __builtin_amdgcn_sbfe(mk[(16 * rh + i) * 2], bp, 1);
                    acc[rh][i] = __uint_as_float(__float_as_uint(acc[rh][i] + sma[rh][i]) & (unsigned)t);
                    db1 += acc[rh][i];
                }
        }
        // ============ G1's sums: dW1 (over this lane's 8 rows), g_x (through the wave's private tile) ============
        {
            const float4* xr = reinterpret_cast<const float4*>(smem + xs_off + par * 1024);
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 xa4 = xr[(16 * rh + i) * 2], xb4 = xr[(16 * rh + i) * 2 + 1];
                    const float gv = acc[rh][i];
                    w1acc[0] = __fmaf_rn(gv, xa4.x, w1acc[0]); w1acc[1] = __fmaf_rn(gv, xa4.y, w1acc[1]);
                    w1acc[2] = __fmaf_rn(gv, xa4.z, w1acc[2]); w1acc[3] = __fmaf_rn(gv, xa4.w, w1acc[3]);
                    w1acc[4] = __fmaf_rn(gv, xb4.x, w1acc[4]); w1acc[5] = __fmaf_rn(gv, xb4.y, w1acc[5]);
                    w1acc[6] = __fmaf_rn(gv, xb4.z, w1acc[6]); w1acc[7] = __fmaf_rn(gv, xb4.w, w1acc[7]);
                }
            if (GX) {
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
                    *reinterpret_cast<float4*>(Tw + 16 * rh) = make_float4(acc[rh][0], acc[rh][1], acc[rh][2], acc[rh][3]);
                float gx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int f = 0; f < 16; ++f) {
                    const float tv = Tr[f * F4_TROW];
                    const float4 wv = W1l[2 * f];
                    gx[0] = __fmaf_rn(wv.x, tv, gx[0]); gx[1] = __fmaf_rn(wv.y, tv, gx[1]);
                    gx[2] = __fmaf_rn(wv.z, tv, gx[2]); gx[3] = __fmaf_rn(wv.w, tv, gx[3]);
                }
                reinterpret_cast<float4*>(smem + F4_GX)[(v * 32 + rown) * 2 + hh] = make_float4(gx[0], gx[1], gx[2], gx[3]);
            }
        }
        prev_tile = tile;
    }
    __syncthreads();
    if (GX) gx_store(prev_tile);
    // ---- the layer-1 slot: dW2 | dW1 | db2 | db1 ----
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) { f4_settle_acc(cacc[jb]); f4_settle_acc(sacc[jb]); }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int i = 0; i < 4; ++i) P[(size_t)(16 * v + 4 * g + i) * EH + 16 * jb + c] = cacc[jb][i] + sacc[jb][i];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        w1acc[cc] += __shfl_xor(w1acc[cc], 16, 64);
        w1acc[cc] += __shfl_xor(w1acc[cc], 32, 64);
    }
    db1 += __shfl_xor(db1, 16, 64); db1 += __shfl_xor(db1, 32, 64);
    db2 += __shfl_xor(db2, 16, 64); db2 += __shfl_xor(db2, 32, 64);
    if (g == 0) {
        float* o = P + EH * EH + (size_t)(16 * v + c) * IN;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
            if ((unsigned)cc < IN) o[cc] = w1acc[cc];
        P[EH * EH + 1024 + 16 * v + c] = db2;
        P[EH * EH + 1024 + EH + 16 * v + c] = db1;
    }
    for (unsigned cc = IN * 128 + tid; cc < 1024; cc += F4_THREADS) P[EH * EH + cc] = 0.f;      // the unused tail of the dW1 field

    // =====================================================================================================================
    // phase 2: dW3 = G3^T H2 and db3 over the same tiles, into the same accumulator registers (the layer-0 slot of this
    // workgroup).  Both operands contract over the ROWS and come from memory as rows: a lane loads four consecutive features
    // of a row with one 16-byte load, splits them and lays the pieces into a [row][feature] image in LDS (8-byte chunks
    // XOR-swizzled by the row); ds_read_b64_tr_b16 then hands every lane its feature's rows: fragments with the feature on the
    // lane and the rows as the k index (element j of lane group g = row 8 g + j), for G3 (A: this wave's 16 features) and H2
    // (B: all eight blocks).  Wave v loads rows 4 v .. 4 v + 3 of the tile; rows past the end are agents past the end and read
    // as zeros by the buffer range check.  Two image pairs: one barrier per tile.  scale is applied once, to the sums.
    // =====================================================================================================================
    if (F.with_dw3) {
        float* P0 = J.partials + (size_t)bx * F4_PART0;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
            for (int i = 0; i < 4; ++i) { cacc[jb][i] = 0.f; sacc[jb][i] = 0.f; }
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) asm volatile("" : "+a"(cacc[jb]), "+a"(sacc[jb]));
        float d3[4] = {0.f, 0.f, 0.f, 0.f};                  // db3 of features 4 n .. 4 n + 3 over this lane's rows
        const __amdgpu_buffer_rsrc_t rs_h2 = rsrc(J.h2, R * EH * 4);
        struct P2Pre { float4 hv[2], gp[2], gm[2]; unsigned kw[2]; };
        constexpr int P2_IMG = 3 * 32 * 256;                  // bytes of one image: [piece][row 32][feature 128] bf16
        auto img = [&](int pb, int which) { return smem + (pb ? F4_WLO : F4_BUFA) + which * P2_IMG; };
        static_assert(2 * P2_IMG <= F4_XS - F4_BUFA && 2 * P2_IMG <= 8 * 2 * 4 * 64 * 16, "the images fit the dead buffers");
        const int n = lane & 31;                               // chunk of the row: features 4 n .. 4 n + 3
        const unsigned v_row = (unsigned)(2 * hh) * (EH * 4) + (unsigned)n * 16;       // lane part of a (rows, 128) offset: rows 4 v + 2 hh + i
        auto p2_load = [&](P2Pre& Q, int tile_) {
            const bool live = tile_ < ntiles;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned row0 = (unsigned)tile_ * 32 + 4 * v + i;                          // scalar: the row of lane half 0 (half 1: + 2)
                const unsigned so = live ? row0 * (EH * 4) : kOut;
                Q.hv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_h2, (int)v_row, (int)so, 0));
                Q.gm[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MSGS) Q.gm[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_gm, (int)v_row, (int)so, 0));
                Q.gp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (POOL) {      // the agents of both lane halves on the scalar unit
                    const unsigned a0 = __builtin_amdgcn_readfirstlane(__umulhi(row0, kmagic)), a1 = __builtin_amdgcn_readfirstlane(__umulhi(row0 + 2, kmagic));
                    Q.gp[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_gp, (int)((unsigned)n * 16 + (hh ? (a1 - a0) * (EH * 4) : 0u)),
                                                                                                 (int)(live ? a0 * (EH * 4) : kOut), 0));
                }
                Q.kw[i] = 0u;
                if (DROP) Q.kw[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_kb, (int)((2 * hh) * 16 + (n >> 3) * 4), (int)(live ? row0 * 16 : kOut), 0);
            }
        };
        // four values of row 4 v + 2 hh + i -> three 8-byte chunks of pieces at [row][chunk n] of an image
        auto lay = [&](unsigned char* im, int i, float x, float y, float z, float w_) {
            unsigned hi0, mid0, lo0, hi1, mid1, lo1;
            split3(x, y, hi0, mid0, lo0);
            split3(z, w_, hi1, mid1, lo1);
            const int rw = 4 * v + 2 * hh + i;                 // (rw & 3 = 2 hh + i)
            unsigned char* d = im + rw * 256 + ((n ^ (8 * (2 * hh + i))) * 8);
            *reinterpret_cast<uint2*>(d) = make_uint2(hi0, hi1);
            *reinterpret_cast<uint2*>(d + 8192) = make_uint2(mid0, mid1);
            *reinterpret_cast<uint2*>(d + 16384) = make_uint2(lo0, lo1);
        };
        // reader: lane 4 q + p of group g supplies image row r0 + q (r0 = 8 g + 4 half2), chunk 4 blk + p of 16-feature block blk
        const int trq = (lane >> 2) & 3;
        const int tr_off = (8 * g + trq) * 256 + (lane & 3) * 8;                       // + half2 * 1024 + ((4 blk) ^ (8 q)) * 8 + piece * 8192
        auto frag = [&](const unsigned char* im, int blk, int piece) -> u32x4 {
            const int o0 = tr_off + (((4 * blk) ^ (8 * trq)) & 31) * 8 + piece * 8192;
            const s16x4_ lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_*)(im + o0));
            const s16x4_ hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_*)(im + o0 + 4 * 256));
            const uint2 x = __builtin_bit_cast(uint2, lo4), y = __builtin_bit_cast(uint2, hi4);
            return (u32x4){x.x, x.y, y.x, y.y};
        };
        P2Pre Q0, Q1;                                          // (addressed by a compile-time index: a run-time index would put them into scratch)
        auto p2_tile = [&](const P2Pre& C, int pb) {
            unsigned char* imH = img(pb, 0);
            unsigned char* imG = img(pb, 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float gq[4] = {C.gp[i].x + C.gm[i].x, C.gp[i].y + C.gm[i].y, C.gp[i].z + C.gm[i].z, C.gp[i].w + C.gm[i].w};
                if (DROP) {
                    const unsigned m = C.kw[i] >> ((4 * n) & 31);
#pragma unroll
                    for (int u = 0; u < 4; ++u) gq[u] = keep_if(gq[u], m, u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) d3[u] += gq[u];
                lay(imG, i, gq[0], gq[1], gq[2], gq[3]);
                lay(imH, i, C.hv[i].x, C.hv[i].y, C.hv[i].z, C.hv[i].w);
            }
            __syncthreads();
            u32x4 ga[3], ob[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) ga[p] = frag(imG, v, p);
#pragma unroll
            for (int p = 0; p < 3; ++p) ob[0][p] = frag(imH, 0, p);
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) {
                if (jb < 7) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) ob[(jb + 1) & 1][p] = frag(imH, jb + 1, p);
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&o)[3] = ob[jb & 1];
                f4_mfma_acc(sacc[jb], ga[2], o[0]);
                f4_mfma_acc(sacc[jb], ga[1], o[1]);
                f4_mfma_acc(sacc[jb], ga[0], o[2]);
                f4_mfma_acc(sacc[jb], ga[1], o[0]);
                f4_mfma_acc(sacc[jb], ga[0], o[1]);
                f4_mfma_acc(cacc[jb], ga[0], o[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int t2 = bx;
        p2_load(Q0, t2);
        __syncthreads();                                       // (phase 1's last reads of the buffers under the images are done)
        for (; t2 < ntiles; t2 += 2 * nwg) {
            p2_load(Q1, t2 + nwg);
            p2_tile(Q0, 0);
            if (t2 + nwg < ntiles) {
                p2_load(Q0, t2 + 2 * nwg);
                p2_tile(Q1, 1);
            }
        }
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) { f4_settle_acc(cacc[jb]); f4_settle_acc(sacc[jb]); }
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
            for (int i = 0; i < 4; ++i) P0[(size_t)(16 * v + 4 * g + i) * EH + 16 * jb + c] = (cacc[jb][i] + sacc[jb][i]) * scale;
        // db3: this lane's sums of features 4 n .. 4 n + 3; the two lane halves and the eight waves meet in LDS (fixed order)
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(smem + F4_BUFA);                             // [wave 8][half 2][n 32] float4 = 8 KB
        red[(v * 2 + hh) * 32 + n] = make_float4(d3[0], d3[1], d3[2], d3[3]);
        __syncthreads();
        if (tid < 128) {
            const float* rf = reinterpret_cast<const float*>(smem + F4_BUFA) + tid;          // feature tid of [slot 16][128]
            float acc3 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc3 += rf[q * 128];
            P0[EH * EH + tid] = acc3 * scale;
        }
    }
}

template <bool P_, bool M_>
static int f4_set(int bytes) {
    auto set = [&](const void* f) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_fused8_x3_kernel<P_, M_, false, false>))) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_fused8_x3_kernel<P_, M_, false, true>))) return e;
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_fused8_x3_kernel<P_, M_, true, false>))) return e;
    return set(reinterpret_cast<const void*>(enc_bwd_fused8_x3_kernel<P_, M_, true, true>));
}

int enc_f4_set_attributes() {
    if (int e = f4_set<true, true>(F4_LDS_BYTES)) return e;
    if (int e = f4_set<true, false>(F4_LDS_BYTES)) return e;
    return f4_set<false, true>(F4_LDS_BYTES);
}

template <bool P_, bool M_>
static void f4_go(const F4Args& F, dim3 gr, bool drop, bool gx, hipStream_t s) {
    const dim3 bl(F4_THREADS);
    if (drop && gx) hipLaunchKernelGGL((enc_bwd_fused8_x3_kernel<P_, M_, true, true>), gr, bl, F4_LDS_BYTES, s, F);
    else if (drop) hipLaunchKernelGGL((enc_bwd_fused8_x3_kernel<P_, M_, true, false>), gr, bl, F4_LDS_BYTES, s, F);
    else if (gx) hipLaunchKernelGGL((enc_bwd_fused8_x3_kernel<P_, M_, false, true>), gr, bl, F4_LDS_BYTES, s, F);
    else hipLaunchKernelGGL((enc_bwd_fused8_x3_kernel<P_, M_, false, false>), gr, bl, F4_LDS_BYTES, s, F);
}

// A: the launch's branches (both with the same kinds of upstream gradients, keep bits and g_x: checked by the caller);
// nA[b] workgroups and slot0[b] layer-0 slots in front for branch b
void enc_f4_launch(const EncArgs& A, const int* nA, const int* slot0, bool with_dw3, hipStream_t s) {
    F4Args F;
    F.A = A;
    F.with_dw3 = with_dw3 ? 1 : 0;
    F.nA[0] = nA[0]; F.nA[1] = A.nbr > 1 ? nA[1] : 0;
    F.slot0[0] = slot0[0]; F.slot0[1] = A.nbr > 1 ? slot0[1] : 0;
    const bool pool = A.br[0].g_pooled != nullptr, msgs = A.br[0].g_msgs != nullptr, drop = A.br[0].keep_bits != nullptr;
    const bool gx = A.br[0].g_x != nullptr;
    const dim3 gr((unsigned)(F.nA[0] + F.nA[1]));
    if (pool && msgs) f4_go<true, true>(F, gr, drop, gx, s);
    else if (pool) f4_go<true, false>(F, gr, drop, gx, s);
    else f4_go<false, true>(F, gr, drop, gx, s);
}

}  // namespace piml
