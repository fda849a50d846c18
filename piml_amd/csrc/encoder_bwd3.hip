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
//     are added in a fixed order (no atomics: bit-reproducible).
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
constexpr int F3_LDS_BYTES = F3_WLO + 4 * 2 * 8 * 64 * 16;
static_assert(F3_LDS_BYTES <= 160 * 1024, "fits the CU");
// G1 blocks for g_x, [feature 32][36] floats per wave: wave w's tile lies over ITS OWN two k-blocks of bufA (6 KB).  Those
// are written by wave w alone (phase 1) and read by all waves in phase 2 only; the tile is written and read by wave w in
// phase 4, behind the barrier that ends phase 2, and in front of wave w's own next phase-1 writes.
constexpr int F3_TROW = 36;
static_assert(32 * F3_TROW * 4 <= 2 * 3 * 64 * 16, "a G1 block fits the wave's part of bufA");

struct F3Args {
    EncArgs A;
    int nA[2];          // workgroups of branch 0 / branch 1 (grid = their sum)
    int slot0[2];       // layer-0 slots (DW2_PART0 floats each) in front of this kernel's slots in the branch's `partials`
};

constexpr int F3_PART0 = EH * EH + EH;                       // = DW2_PART0 (encoder_dw2.hip): dW3 | db3
constexpr int F3_PART1 = EH * EH + 1024 + 2 * EH;            // = DW2_PART1: dW2 | dW1 (1024-float field) | db2 | db1

// row of the tile held by accumulator register r in lane half h
__device__ __forceinline__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

// 16 registers -> the three bf16 pieces of both k-steps: element t of k-step s = register 8 s + t
struct Pieces2 {
    u32x4 hi[2], mid[2], lo[2];
};
__device__ __forceinline__ void split_block(const f32x16& a, Pieces2& P) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) split3(a[8 * s + 2 * d], a[8 * s + 2 * d + 1], hi[d], mid[d], lo[d]);
        P.hi[s] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
        P.mid[s] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
        P.lo[s] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
    }
}

// inputs of a tile that come from memory, requested one tile ahead
struct F3Pre {
    float4 gp[2][2], gm[2][2];     // [k-step s][half2]: features 32 w + 16 s + 8 half2 + 4 h .. + 3 of the lane's row
    unsigned kw;                   // keep word w of the row
    float xa[4];                   // x[row][2 s + h]: A operand of the H1 recomputation
    float xs;                      // staging: x[tile row tid >> 3][tid & 7]
    uint2 mk;                      // staging: sign words (threads 0 .. 127)
};

template <bool POOL, bool MSGS, bool DROP>
__global__ __launch_bounds__(F3_THREADS) void enc_bwd_fused_x3_kernel(F3Args F) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = (int)blockIdx.x, b = 0;
    if (bx >= F.nA[0]) { b = 1; bx -= F.nA[0]; }
    const piml_encoder_branch J = b ? F.A.br[1] : F.A.br[0];
    const int nwg = F.nA[b];
    const unsigned R = (unsigned)J.rows;                      // rows < 2^24 (checked on the host)
    const unsigned IN = __builtin_amdgcn_readfirstlane((unsigned)J.in_dim), K = __builtin_amdgcn_readfirstlane((unsigned)J.k);
    const unsigned kmagic = __builtin_amdgcn_readfirstlane((unsigned)((0x100000000ull + K - 1) / K));      // row / K == umulhi(row, kmagic)
    const int ntiles = (int)((R + 31) >> 5);
    const int n = lane & 31, h = lane >> 5;
    const float scale = J.scale;
    float* P = J.partials + (size_t)F.slot0[b] * F3_PART0 + (size_t)bx * F3_PART1;

    // ---- this wave's weight fragments: block w of W3^T and W2^T, all eight k-blocks, three pieces (192 registers) ----
    // (hi, mid) in registers, 128 of them; the lo pieces -- one of the six products reads them -- in a private part of LDS
    u32x4 wfA[8][2], wfB[8][2];
    u32x4* const wlo = reinterpret_cast<u32x4*>(smem + F3_WLO) + w * (2 * 8 * 64) + lane;       // + (layer * 8 + kb) * 64
    {
        const u32x4* imgA = reinterpret_cast<const u32x4*>(J.packed + PACK_F32 + 2 * X3_IMG) + lane;
        const u32x4* imgB = reinterpret_cast<const u32x4*>(J.packed + PACK_F32 + 3 * X3_IMG) + lane;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const int fb = w * 8 + kb;
            wfA[kb][0] = imgA[(fb * 2) * 64]; wfA[kb][1] = imgA[(fb * 2 + 1) * 64];
            wfB[kb][0] = imgB[(fb * 2) * 64]; wfB[kb][1] = imgB[(fb * 2 + 1) * 64];
        }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const int fb = w * 8 + kb;
            wlo[kb * 64] = imgA[X3_HM / 4 + fb * 64];
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
    float* const Tbase = reinterpret_cast<float*>(smem + F3_BUFA + (2 * w) * 3 * 64 * 16);
    float* const Tw = Tbase + n * F3_TROW + 4 * h;               // + 8 g: registers 4 g .. 4 g + 3
    const float* const Tr = Tbase + h * F3_TROW + n;             // + 2 i rows: feature 2 i + h, row n
    const float4* const W1l = reinterpret_cast<const float4*>(smem + F3_W1) + (32 * w + h) * 2;      // + 4 i: feature 32 w + 2 i + h

    // ---- accumulators that live for the whole slab ----
    f32x16 c[4], sm[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c[jb][r] = 0.f; sm[jb][r] = 0.f; }
    float w1acc[8], db1 = 0.f, db2 = 0.f;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) w1acc[cc] = 0.f;

    auto prefetch = [&](int tile) -> F3Pre {
        F3Pre S;
        const bool live = tile < ntiles;
        const unsigned row = (unsigned)tile * 32 + n;
        const bool valid = live && row < R;
        const unsigned rr = valid ? row : 0u;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int half2 = 0; half2 < 2; ++half2) {
                const int f = 32 * w + 16 * s + 8 * half2 + 4 * h;
                S.gp[s][half2] = make_float4(0.f, 0.f, 0.f, 0.f);
                S.gm[s][half2] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (POOL) S.gp[s][half2] = *reinterpret_cast<const float4*>(J.g_pooled + (size_t)__umulhi(rr, kmagic) * EH + f);
                if (MSGS) S.gm[s][half2] = *reinterpret_cast<const float4*>(J.g_msgs + (size_t)rr * EH + f);
            }
        S.kw = 0xffffffffu;
        if (DROP) S.kw = J.keep_bits[(size_t)rr * 4 + w];
        if (!valid) S.kw = 0u;                               // rows past the end: G3 = 0, and with it G2, G1 and every sum
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned cx = 2u * s + h;
            S.xa[s] = (valid && cx < IN) ? J.x[(size_t)rr * IN + cx] : 0.f;
        }
        {
            const unsigned srow = (unsigned)tile * 32 + (tid >> 3), cx = tid & 7;
            S.xs = (live && srow < R && cx < IN) ? J.x[(size_t)srow * IN + cx] : 0.f;
        }
        S.mk = make_uint2(0u, 0u);
        if (live && tid < 128) S.mk = reinterpret_cast<const uint2*>(J.relu_mask)[(size_t)tile * 128 + tid];
        return S;
    };

    int tile = bx;
    F3Pre S = prefetch(tile);
    int par = 0;
    int prev_tile = -1;
    for (; tile < ntiles; tile += nwg, par ^= 1) {
        // ================= phase 1: stage the tile's x rows and sign words, build this wave's two k-blocks of G3 =================
        reinterpret_cast<float*>(smem + F3_XS + par * 1024)[tid] = S.xs;
        if (tid < 128) reinterpret_cast<uint2*>(smem + F3_MK + par * 1024)[tid] = S.mk;
        {
            const unsigned m = S.kw >> (4 * h);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                unsigned hi[4], mid[4], lo[4];
#pragma unroll
                for (int half2 = 0; half2 < 2; ++half2) {
                    const float4 a = S.gp[s][half2], g = S.gm[s][half2];
                    float v[4] = {(a.x + g.x) * scale, (a.y + g.y) * scale, (a.z + g.z) * scale, (a.w + g.w) * scale};
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = keep_if(v[u], m, 16 * s + 8 * half2 + u);
                    split3(v[0], v[1], hi[2 * half2], mid[2 * half2], lo[2 * half2]);
                    split3(v[2], v[3], hi[2 * half2 + 1], mid[2 * half2 + 1], lo[2 * half2 + 1]);
                }
                const int kb = 2 * w + s;
                bufA[(kb * 3 + 0) * 64] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
                bufA[(kb * 3 + 1) * 64] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
                bufA[(kb * 3 + 2) * 64] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
            }
        }
        float xa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xa[s] = S.xa[s];
        __syncthreads();                                                                   // B1
        // g_x of the previous tile: the four waves' partials, fixed order
        if (prev_tile >= 0 && J.g_x) {
            const float* gp = reinterpret_cast<const float*>(smem + F3_GX) + tid;
            const float v = ((gp[0] + gp[256]) + gp[512]) + gp[768];
            const unsigned grow = (unsigned)prev_tile * 32 + (tid >> 3), cx = tid & 7;
            if (grow < R && cx < IN) J.g_x[(size_t)grow * IN + cx] = v;
        }
        // ================= phase 2: G2 = (G3 W3) * [h2 > 0]; H1 = relu(W1 x + b1) =================
        Pieces2 G2;
        {
            f32x16 acc, sma;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; sma[r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                __builtin_amdgcn_sched_barrier(0);
                kblock_x3(acc, sma, bufA[(kb * 3 + 0) * 64], bufA[(kb * 3 + 1) * 64], bufA[(kb * 3 + 2) * 64], wfA[kb][0], wfA[kb][1], wlo[kb * 64]);
            }
            const unsigned* mk = reinterpret_cast<const unsigned*>(smem + mk_off + par * 1024 + 512);     // layer 1 of the pair: h2
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = __builtin_amdgcn_sbfe(mk[2 * rho(r)], bp, 1);
                acc[r] = __uint_as_float(__float_as_uint(acc[r] + sma[r]) & (unsigned)t);
                db2 += acc[r];
            }
            split_block(acc, G2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int slot = ((2 * g + h) ^ swz_w) * 8;
                const int s = g >> 1, d = 2 * (g & 1);
                *reinterpret_cast<uint2*>(Mw + slot) = make_uint2(G2.hi[s][d], G2.hi[s][d + 1]);
                *reinterpret_cast<uint2*>(Mw + 8192 + slot) = make_uint2(G2.mid[s][d], G2.mid[s][d + 1]);
                *reinterpret_cast<uint2*>(Mw + 16384 + slot) = make_uint2(G2.lo[s][d], G2.lo[s][d + 1]);
            }
        }
        {
            f32x16 hacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = b1v;
#pragma unroll
            for (int s = 0; s < 4; ++s) hacc = mfma32(xa[s], w1v[s], hacc);
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = relu1(hacc[r]);
            Pieces2 H;
            split_block(hacc, H);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bufH[((w * 2 + s) * 3 + 0) * 64] = H.hi[s];
                bufH[((w * 2 + s) * 3 + 1) * 64] = H.mid[s];
                bufH[((w * 2 + s) * 3 + 2) * 64] = H.lo[s];
            }
        }
        __syncthreads();                                                                   // B2
        // ================= phase 3: dW2 += G2^T H1  and  G1 = (G2 W2) * [h1 > 0] =================
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                __builtin_amdgcn_sched_barrier(0);
                kblock_x3(c[jb], sm[jb], G2.hi[s], G2.mid[s], G2.lo[s], bufH[((jb * 2 + s) * 3 + 0) * 64], bufH[((jb * 2 + s) * 3 + 1) * 64],
                          bufH[((jb * 2 + s) * 3 + 2) * 64]);
            }
        f32x16 acc, sma;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; sma[r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            __builtin_amdgcn_sched_barrier(0);
            u32x4 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + mr[0] + kb * 1024 + p * 8192));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + mr[1] + kb * 1024 + p * 8192));
                const uint2 x = __builtin_bit_cast(uint2, lo4), y = __builtin_bit_cast(uint2, hi4);
                a[p] = (u32x4){x.x, x.y, y.x, y.y};
            }
            kblock_x3(acc, sma, a[0], a[1], a[2], wfB[kb][0], wfB[kb][1], wlo[(8 + kb) * 64]);
        }
        __builtin_amdgcn_sched_barrier(0);
        S = prefetch(tile + nwg);                                                          // the next tile's loads
        // ================= phase 4: G1's sums on the vector pipe =================
        {
            const unsigned* mk = reinterpret_cast<const unsigned*>(smem + mk_off + par * 1024);           // layer 0 of the pair: h1
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = __builtin_amdgcn_sbfe(mk[2 * rho(r)], bp, 1);
                acc[r] = __uint_as_float(__float_as_uint(acc[r] + sma[r]) & (unsigned)t);
                db1 += acc[r];
            }
            const float4* xr = reinterpret_cast<const float4*>(smem + xs_off + par * 1024);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 xa4 = xr[2 * rho(r)], xb4 = xr[2 * rho(r) + 1];
                const float g = acc[r];
                w1acc[0] = __fmaf_rn(g, xa4.x, w1acc[0]); w1acc[1] = __fmaf_rn(g, xa4.y, w1acc[1]);
                w1acc[2] = __fmaf_rn(g, xa4.z, w1acc[2]); w1acc[3] = __fmaf_rn(g, xa4.w, w1acc[3]);
                w1acc[4] = __fmaf_rn(g, xb4.x, w1acc[4]); w1acc[5] = __fmaf_rn(g, xb4.y, w1acc[5]);
                w1acc[6] = __fmaf_rn(g, xb4.z, w1acc[6]); w1acc[7] = __fmaf_rn(g, xb4.w, w1acc[7]);
            }
            if (J.g_x) {
                // the wave's G1 block through its private tile: [feature][row], then per row over the block's features
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(Tw + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
                float gx[8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) gx[cc] = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = Tr[2 * i * F3_TROW];
                    const float4 wa = W1l[4 * i], wb = W1l[4 * i + 1];
                    gx[0] = __fmaf_rn(wa.x, v, gx[0]); gx[1] = __fmaf_rn(wa.y, v, gx[1]);
                    gx[2] = __fmaf_rn(wa.z, v, gx[2]); gx[3] = __fmaf_rn(wa.w, v, gx[3]);
                    gx[4] = __fmaf_rn(wb.x, v, gx[4]); gx[5] = __fmaf_rn(wb.y, v, gx[5]);
                    gx[6] = __fmaf_rn(wb.z, v, gx[6]); gx[7] = __fmaf_rn(wb.w, v, gx[7]);
                }
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) gx[cc] += __shfl_xor(gx[cc], 32, 64);
                if (h == 0) {
                    float4* o = reinterpret_cast<float4*>(smem + F3_GX) + (w * 32 + n) * 2;
                    o[0] = make_float4(gx[0], gx[1], gx[2], gx[3]);
                    o[1] = make_float4(gx[4], gx[5], gx[6], gx[7]);
                }
            }
        }
        prev_tile = tile;
    }
    __syncthreads();
    if (prev_tile >= 0 && J.g_x) {
        const float* gp = reinterpret_cast<const float*>(smem + F3_GX) + tid;
        const float v = ((gp[0] + gp[256]) + gp[512]) + gp[768];
        const unsigned grow = (unsigned)prev_tile * 32 + (tid >> 3), cx = tid & 7;
        if (grow < R && cx < IN) J.g_x[(size_t)grow * IN + cx] = v;
    }
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
}

int enc_f3_set_attributes() {
    auto set = [](const void* f) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES); };
#define PIML_F3_SET(P_, M_)                                                                      \
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_fused_x3_kernel<P_, M_, false>))) return e; \
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_fused_x3_kernel<P_, M_, true>))) return e;
    PIML_F3_SET(true, true)
    PIML_F3_SET(true, false)
    PIML_F3_SET(false, true)
#undef PIML_F3_SET
    return hipSuccess;
}

// A: the launch's branches (both with the same kinds of upstream gradients and keep bits: checked by the caller);
// nA[b] workgroups and slot0[b] layer-0 slots in front for branch b
void enc_f3_launch(const EncArgs& A, const int* nA, const int* slot0, hipStream_t s) {
    F3Args F;
    F.A = A;
    F.nA[0] = nA[0]; F.nA[1] = A.nbr > 1 ? nA[1] : 0;
    F.slot0[0] = slot0[0]; F.slot0[1] = A.nbr > 1 ? slot0[1] : 0;
    const bool pool = A.br[0].g_pooled != nullptr, msgs = A.br[0].g_msgs != nullptr, drop = A.br[0].keep_bits != nullptr;
    const dim3 g((unsigned)(F.nA[0] + F.nA[1])), b(F3_THREADS);
#define PIML_F3_GO(P_, M_)                                                                                             \
    do {                                                                                                               \
        if (drop) hipLaunchKernelGGL((enc_bwd_fused_x3_kernel<P_, M_, true>), g, b, F3_LDS_BYTES, s, F);               \
        else hipLaunchKernelGGL((enc_bwd_fused_x3_kernel<P_, M_, false>), g, b, F3_LDS_BYTES, s, F);                   \
    } while (0)
    if (pool && msgs) PIML_F3_GO(true, true);
    else if (pool) PIML_F3_GO(true, false);
    else PIML_F3_GO(false, true);
#undef PIML_F3_GO
}

}  // namespace piml
