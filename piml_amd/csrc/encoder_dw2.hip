// Weight gradients of the PINNSF encoder on split bf16 products, LAYER-SPLIT decomposition (round 3).
//
// Reference arithmetic: the autograd of MLP(in, [128, 128, 128]) (src/models/model.py:40-65) under the processor
// Dropout_p(2 x) and the neighbour-axis sum (:82-119, :1279-1283):
//     dW3 = G3^T H2, db3 = colsum G3        G3 = keep * scale * (g_pooled[row / k] + g_msgs[row])
//     dW2 = G2^T H1, db2 = colsum G2        H1 = relu(W1 x + b1)
//     dW1 = G1^T X,  db1 = colsum G1
// enc_bwd_dw_x3_kernel (encoder_x3.hip) gives every workgroup a row slab and BOTH 128 x 128 products: 128 accumulator
// registers per wave, four operand arrays staged per batch, one partial slot of 34 k floats per workgroup (35 MB written and
// read back at the 4096-agent scene), and h1 read from memory (33 MB that the forward only stores for this kernel: the
// dX chain masks with sign bits).  Here a workgroup takes ONE layer of a longer slab:
//     L = 0:  dW3, db3                  stages G3 (built from g_pooled / g_msgs / keep bits) and H2 (loaded)
//     L = 1:  dW2, db2, dW1, db1        stages G2 (loaded) and H1 -- RECOMPUTED from x when the branch carries no h1
// -> 64 accumulator registers per wave (wave w: output blocks (w >> 1, 2 (w & 1) + {0, 1})), two staged arrays, half the
// partial bytes, no h1 traffic at all.  A batch is 32 rows = two k-blocks.  Fragment layout as in encoder_x3.hip (lane
// (f & 31) + 32 hh of feature block f >> 5 holds 8 bf16 = 8 rows), but the rows of a k-block are taken in the order of a
// 32 x 32 accumulator's registers -- element t of lane half hh = row 16 kb + 4 hh + (t & 3) + 8 (t >> 2) -- on BOTH operand
// sides, because that is the layout in which the recomputed H1 arrives: the H-side wave of feature block blk runs
// relu(W1 x + b1) for the batch's 32 rows as four f32 matrix instructions in the TRANSPOSED orientation (A = rows of x,
// B = W1's fragments), result lane (n, h), register r = row (r & 3) + 8 (r >> 2) + 4 h of feature 32 blk + n.
#include "common.hpp"
#include "encoder.hpp"
#include "x3.hpp"

namespace piml {

constexpr int DW2_ROWS = 32;
constexpr int DW2_ARR = 3 * 2 * 256;                 // u32x4 of one array: [piece 3][k-block 2][feature block 4][lane 64]
constexpr int DW2_BUF = 2 * DW2_ARR + 64;            // G | H | the batch's x rows [32][8] floats
constexpr int DW2_LDS_BYTES = 2 * DW2_BUF * 16;      // two buffers
constexpr int DW2_PART0 = EH * EH + EH;              // dW3 | db3
constexpr int DW2_PART1 = EH * EH + 1024 + 2 * EH;   // dW2 | dW1 (128 x in_dim in a 1024-float field) | db2 | db1
static_assert(DW2_PART0 + DW2_PART1 == ENC_PART, "the two slot kinds partition a full slot");
static_assert(4 * 128 * 9 * 4 + 2 * 128 * 4 <= DW2_LDS_BYTES, "final exchange fits");

struct Dw2Args {
    EncArgs A;
    int n0[2], n1[2];        // workgroups of (branch, layer 0) / (branch, layer 1); grid = branch 0: L0 | L1, branch 1: L0 | L1
};

template <bool POOL, bool MSGS, bool DROP, bool H1R>
__global__ __launch_bounds__(ENC_THREADS) void enc_bwd_dw2_x3_kernel(Dw2Args D) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = (int)blockIdx.x;
    int b = 0;
    if (bx >= D.n0[0] + D.n1[0]) { b = 1; bx -= D.n0[0] + D.n1[0]; }
    const int L = __builtin_amdgcn_readfirstlane(bx >= D.n0[b] ? 1 : 0);
    const unsigned p = (unsigned)(L ? bx - D.n0[b] : bx);
    const unsigned nwg = (unsigned)(L ? D.n1[b] : D.n0[b]);
    const piml_encoder_branch J = b ? D.A.br[1] : D.A.br[0];
    const unsigned R = (unsigned)J.rows;                   // rows < 2^24 (checked on the host): 32-bit indexing
    const unsigned IN = __builtin_amdgcn_readfirstlane((unsigned)J.in_dim), K = __builtin_amdgcn_readfirstlane((unsigned)J.k);
    const unsigned kmagic = __builtin_amdgcn_readfirstlane((unsigned)((0x100000000ull + K - 1) / K));      // row / K == umulhi(row, kmagic)
    unsigned slab = (R + nwg - 1) / nwg;
    slab = (slab + 1) & ~1u;
    const unsigned r0 = __builtin_amdgcn_readfirstlane(p * slab < R ? p * slab : R);
    const unsigned r1 = __builtin_amdgcn_readfirstlane(r0 + slab < R ? r0 + slab : R);
    const float scale = J.scale;
    const int ia = wave >> 1, jb0 = 2 * (wave & 1);        // output blocks (ia, jb0), (ia, jb0 + 1)

    f32x16 c[2], sm[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c[u][r] = 0.f; sm[u][r] = 0.f; }
    // staging role: feature sf, lane half sh of the fragment; waves 0-3 the G array, waves 4-7 the H array
    const unsigned sf = tid & 127, sh = (wave >> 1) & 1;
    const bool gside = wave < 4;
    const unsigned slot = (sf >> 5) * 64 + (sf & 31) + 32 * sh;
    const int hblk = (wave - 4) & 3;                       // H1R: feature block of this H-side wave
    // dW1 / db1 role (L = 1): feature sf, rows 8 rg .. 8 rg + 7 of the batch
    const unsigned rg = wave >> 1;
    float sG = 0.f, s1 = 0.f;                              // column sums: of the staged G array (db3 / db2), of g1 (db1)
    float w1[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) w1[cc] = 0.f;

    // Every load goes through a buffer resource over this workgroup's slab with the row as the SCALAR offset (rows are
    // wave-uniform): no address arithmetic on the vector pipe, a row past the slab reads as 0 by the range check.
    const unsigned srows = r1 - r0, sbytes = srows * EH * 4;
    auto rsrc = [&](const void* base, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    };
    const unsigned pbytes = (R / K) * EH * 4;
    // array 0 of this role: L0 G side: g_pooled (whole array, row / k) or g_msgs; L0 H side: h2; L1 G side: g2; L1 H side: h1
    const bool pooled0 = L == 0 && gside && POOL;
    const float* a0 = L == 0 ? (gside ? (POOL ? J.g_pooled : J.g_msgs) : J.h2) : (gside ? J.g2 : J.h1);
    const bool a0_live = !(H1R && L == 1 && !gside);       // H1R: nothing is loaded for H1
    const __amdgpu_buffer_rsrc_t rs0 = pooled0 ? rsrc(a0, pbytes) : rsrc(a0_live ? a0 + (size_t)r0 * EH : nullptr, a0_live ? sbytes : 0u);
    const __amdgpu_buffer_rsrc_t rsm = rsrc((POOL && MSGS) ? J.g_msgs + (size_t)r0 * EH : nullptr, (POOL && MSGS) ? sbytes : 0u);
    const __amdgpu_buffer_rsrc_t rsg = rsrc(J.g1 + (size_t)r0 * EH, sbytes);
    const __amdgpu_buffer_rsrc_t rsx = rsrc(J.x + (size_t)r0 * IN, srows * IN * 4);
    const unsigned kbytes = srows * 16;
    const __amdgpu_buffer_rsrc_t rsk = rsrc(DROP ? J.keep_bits + (size_t)r0 * 4 : nullptr, DROP ? kbytes : 0u);
    const unsigned xvoff = (tid < 256 && (unsigned)(tid & 7) < IN) ? ((tid >> 3) * IN + (tid & 7)) * 4 : 0x7fff0000u;
    auto ld = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
    };
    // H1R: W1 fragments + bias of this wave's feature block, the x operand's lane offsets
    float w1b[4] = {0.f, 0.f, 0.f, 0.f}, b1n = 0.f;
    unsigned xav[4] = {0x7fff0000u, 0x7fff0000u, 0x7fff0000u, 0x7fff0000u};
    if (H1R && L == 1 && !gside) {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            w1b[s_] = J.packed[32768 + (hblk * 4 + s_) * 64 + lane];
            const unsigned cx = 2u * s_ + (lane >> 5);
            if (cx < IN) xav[s_] = ((lane & 31) * IN + cx) * 4;
        }
        b1n = J.packed[32768 + 1024 + 32 * hblk + (lane & 31)];
    }

    // agent (row / k) of the unit's first row, kept incrementally (the batches are requested in row order, 32 rows apart)
    unsigned pidx0 = __umulhi(r0 + 4 * sh, kmagic), prem0 = r0 + 4 * sh - pidx0 * K;
    const unsigned q32 = __umulhi(32u, kmagic), m32 = 32u - q32 * K;
    const unsigned q5 = __umulhi(5u, kmagic), m5 = 5u - q5 * K;
    struct Stage { float a[16], m[16], g1[8], x, xa[4]; unsigned kw[16]; };
    auto stage_load = [&](unsigned rb_) -> Stage {           // issue the loads of the batch starting at row rb
        Stage S;
        const unsigned rb = __builtin_amdgcn_readfirstlane(rb_);
        unsigned pidx = pidx0, prem = prem0;
        if (POOL) {
            pidx0 += q32; prem0 += m32;
            if (prem0 >= K) { prem0 -= K; ++pidx0; }
        }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) S.xa[s_] = 0.f;
        if (H1R)
            if (L == 1 && !gside) {                            // the batch's x rows as the A operand of the h1 product
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) S.xa[s_] = ld(rsx, xav[s_], rb < r1 ? (rb - r0) * IN * 4 : srows * IN * 4);
            }
#pragma unroll
        for (int e = 0; e < 16; ++e) {                         // e = 8 kb + t: row 16 kb + 4 sh + (t & 3) + 8 (t >> 2)
            const unsigned row = rb + 16 * (e >> 3) + 4 * sh + (e & 3) + 8 * ((e >> 2) & 1);       // scalar
            const unsigned rel = row < r1 ? (row - r0) * (EH * 4) : sbytes;
            const unsigned pi = __builtin_amdgcn_readfirstlane(pidx);
            const unsigned off0 = pooled0 ? (row < r1 ? pi * (EH * 4) : pbytes) : rel;
            S.a[e] = 0.f;
            if (a0_live) S.a[e] = ld(rs0, sf * 4, off0);
            if (POOL) {                                         // to the unit's next row: + 1, or + 5 behind every fourth
                if ((e & 3) == 3) {
                    prem += m5; pidx += q5;
                    if (prem >= K) { prem -= K; ++pidx; }
                } else {
                    ++prem;
                    if (prem == K) { prem = 0; ++pidx; }
                }
            }
            S.m[e] = 0.f;
            if (POOL && MSGS)
                if (L == 0 && gside) S.m[e] = ld(rsm, sf * 4, rel);
            S.kw[e] = 0u;
            if (DROP)
                if (L == 0 && gside) S.kw[e] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsk, (int)((sf >> 5) * 4), (int)(row < r1 ? (row - r0) * 16 : kbytes), 0);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            S.g1[t] = 0.f;
            if (L == 1) {
                const unsigned row = rb + 8 * rg + t;
                S.g1[t] = ld(rsg, sf * 4, row < r1 ? (row - r0) * (EH * 4) : sbytes);
            }
        }
        S.x = 0.f;
        if (L == 1) S.x = ld(rsx, xvoff, rb < r1 ? (rb - r0) * IN * 4 : srows * IN * 4);
        return S;
    };
    float gq[8];                                             // g1 values of the batch in the compute phase
    auto write_pieces = [&](u32x4* dst, const float (&v)[16]) {     // dst: the array's entry of this lane; [piece][kb] 256 apart
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) split3(v[8 * kb + 2 * d], v[8 * kb + 2 * d + 1], hi[d], mid[d], lo[d]);
            dst[(0 * 2 + kb) * 256] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
            dst[(1 * 2 + kb) * 256] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
            dst[(2 * 2 + kb) * 256] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        }
    };
    auto stage_write = [&](const Stage& S, float* buf) {     // registers -> split -> LDS
        u32x4* B = reinterpret_cast<u32x4*>(buf);
        float v[16];
        if (H1R && L == 1 && !gside) {
            // h1 of the batch's 32 rows x this wave's 32 features: lane (n, h), register r = row (r & 3) + 8 (r >> 2) + 4 h
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = b1n;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) acc = mfma32(S.xa[s_], w1b[s_], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = relu1(acc[r]);
            write_pieces(B + DW2_ARR + hblk * 64 + lane, v);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = (L == 0 && gside) ? (S.a[e] + S.m[e]) * scale : S.a[e];
            if (DROP && L == 0 && gside) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = ((S.kw[e] >> (sf & 31)) & 1u) ? v[e] : 0.f;
            }
            if (gside) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sG += v[e];
            }
            write_pieces(B + (gside ? 0 : DW2_ARR) + slot, v);
        }
        if (L == 1 && tid < 256) buf[2 * DW2_ARR * 4 + tid] = S.x;
    };
    auto take_g1 = [&](const Stage& S) {
#pragma unroll
        for (int t = 0; t < 8; ++t) gq[t] = S.g1[t];
    };
    auto compute = [&](const float* buf) {
        const u32x4* B = reinterpret_cast<const u32x4*>(buf);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const u32x4* Ap = B + kb * 256 + ia * 64 + lane;
            const u32x4* Bp = B + DW2_ARR + kb * 256 + jb0 * 64 + lane;
            const u32x4 ah = Ap[0], am = Ap[512], al = Ap[1024];
            u32x4 bh[2], bm[2], bl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { bh[u] = Bp[u * 64]; bm[u] = Bp[512 + u * 64]; bl[u] = Bp[1024 + u * 64]; }
#pragma unroll
            for (int u = 0; u < 2; ++u) kblock_x3(c[u], sm[u], ah, am, al, bh[u], bm[u], bl[u]);
        }
        if (L == 1) {                                          // dW1 / db1: rows 8 rg .. 8 rg + 7 of the batch
            const float4* xr = reinterpret_cast<const float4*>(buf + 2 * DW2_ARR * 4 + rg * 64);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float4 xa = xr[2 * t], xb = xr[2 * t + 1];
                const float g = gq[t];
                w1[0] = __fmaf_rn(g, xa.x, w1[0]); w1[1] = __fmaf_rn(g, xa.y, w1[1]);
                w1[2] = __fmaf_rn(g, xa.z, w1[2]); w1[3] = __fmaf_rn(g, xa.w, w1[3]);
                w1[4] = __fmaf_rn(g, xb.x, w1[4]); w1[5] = __fmaf_rn(g, xb.y, w1[5]);
                w1[6] = __fmaf_rn(g, xb.z, w1[6]); w1[7] = __fmaf_rn(g, xb.w, w1[7]);
                s1 += g;
            }
        }
    };
    if (r0 < r1) {
        const unsigned nb = (r1 - r0 + DW2_ROWS - 1) / DW2_ROWS;
        {
            const Stage S = stage_load(r0);
            stage_write(S, lds);
            take_g1(S);
        }
        __syncthreads();
        for (unsigned t = 0; t < nb; ++t) {
            float* cur = lds + (t & 1) * DW2_BUF * 4;
            float* nxt = lds + ((t + 1) & 1) * DW2_BUF * 4;
            // the next batch's loads are issued BETWEEN this batch's products (see enc_bwd_dw_x3_kernel)
            const Stage S2 = stage_load(r0 + (t + 1) * DW2_ROWS);
            compute(cur);
            __builtin_amdgcn_sched_group_barrier(0x100, 18, 0);          // fragment + x reads
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one product
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);       // two loads
            }
            __builtin_amdgcn_sched_barrier(0);
            stage_write(S2, nxt);
            take_g1(S2);
            __syncthreads();
        }
    }
    // ---- the partial slot: L0 [dW3 | db3] at slot p, L1 [dW2 | dW1 | db2 | db1] behind the branch's L0 slots ----
    float* P = L ? J.partials + (size_t)D.n0[b] * DW2_PART0 + (size_t)p * DW2_PART1 : J.partials + (size_t)p * DW2_PART0;
    const int n = lane & 31, h = lane >> 5;
    // accumulator u, register r, lane (n, h): dW[32 ia + (r & 3) + 8 (r >> 2) + 4 h][32 (jb0 + u) + n]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = 32 * ia + (r & 3) + 8 * (r >> 2) + 4 * h;
            P[(size_t)orow * EH + 32 * (jb0 + u) + n] = c[u][r] + sm[u][r];
        }
    // column sums (and dW1): the partial sums of the row groups / lane halves meet in LDS (the batch buffers are dead)
    __syncthreads();
    {
        if (L == 1) {
            float* red = lds + (rg * 128 + sf) * 9;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) red[cc] = w1[cc];
            red[8] = s1;
        }
        float* red2 = lds + 4 * 128 * 9 + sh * 128 + sf;
        if (gside) red2[0] = sG;
    }
    __syncthreads();
    if (tid < 128) {
        const float dbG = lds[4 * 128 * 9 + tid] + lds[4 * 128 * 9 + 128 + tid];
        if (L == 0) {
            P[EH * EH + tid] = dbG;
        } else {
            float acc[9];
#pragma unroll
            for (int cc = 0; cc < 9; ++cc)
                acc[cc] = (lds[(0 * 128 + tid) * 9 + cc] + lds[(1 * 128 + tid) * 9 + cc]) + (lds[(2 * 128 + tid) * 9 + cc] + lds[(3 * 128 + tid) * 9 + cc]);
            float* o = P + EH * EH + tid * IN;                  // dW1 row-major (128, in_dim) at the head of its 1024 floats
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if ((unsigned)cc < IN) o[cc] = acc[cc];
            P[EH * EH + 1024 + tid] = dbG;
            P[EH * EH + 1024 + EH + tid] = acc[8];
        }
    }
    if (L == 1 && tid >= 128 && tid < 128 + 128) {              // the unused tail of the dW1 field (in_dim < 8): defined zeros
        const int f = tid - 128;
        for (unsigned cc = IN * 128 + f; cc < 1024; cc += 128) P[EH * EH + cc] = 0.f;
    }
}

// slots of layer 0 among a branch's `w` workgroups: the layer-1 workgroups move about twice the bytes per row (g2 + g1
// against h2), so they get two thirds of them (PIML_DW2_L0_SHARE: per mille, default 360)
static int dw2_l0_share() {
    static int v = getenv("PIML_DW2_L0_SHARE") ? atoi(getenv("PIML_DW2_L0_SHARE")) : 360;
    return v;
}
void enc_dw2_split(int w, int* n0, int* n1) {
    int a = (int)((long long)w * dw2_l0_share() / 1000);
    if (a < 1) a = 1;
    if (a > w - 1) a = w - 1;
    if (w < 2) a = w;          // (a single workgroup cannot be split: the callers use the other kernel then)
    *n0 = a;
    *n1 = w - a;
}

int enc_dw2_set_attributes() {
    auto set = [](const void* f) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, DW2_LDS_BYTES); };
#define PIML_DW2_SET(P_, M_)                                                                           \
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_dw2_x3_kernel<P_, M_, false, false>))) return e; \
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_dw2_x3_kernel<P_, M_, false, true>))) return e;  \
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_dw2_x3_kernel<P_, M_, true, false>))) return e;  \
    if (int e = set(reinterpret_cast<const void*>(enc_bwd_dw2_x3_kernel<P_, M_, true, true>))) return e;
    PIML_DW2_SET(true, true)
    PIML_DW2_SET(true, false)
    PIML_DW2_SET(false, true)
#undef PIML_DW2_SET
    return hipSuccess;
}

template <bool POOL, bool MSGS>
static void dw2_go(const Dw2Args& D, dim3 g, bool drop, bool h1r, hipStream_t s) {
    const dim3 b(ENC_THREADS);
    if (drop && h1r) hipLaunchKernelGGL((enc_bwd_dw2_x3_kernel<POOL, MSGS, true, true>), g, b, DW2_LDS_BYTES, s, D);
    else if (drop) hipLaunchKernelGGL((enc_bwd_dw2_x3_kernel<POOL, MSGS, true, false>), g, b, DW2_LDS_BYTES, s, D);
    else if (h1r) hipLaunchKernelGGL((enc_bwd_dw2_x3_kernel<POOL, MSGS, false, true>), g, b, DW2_LDS_BYTES, s, D);
    else hipLaunchKernelGGL((enc_bwd_dw2_x3_kernel<POOL, MSGS, false, false>), g, b, DW2_LDS_BYTES, s, D);
}

// A: the launch's branches (A.wg_split = workgroups of branch 0 out of `total`); both branches have the same kinds of
// upstream gradients, keep bits and h1 (checked by the caller)
void enc_dw2_launch(const EncArgs& A, int total, hipStream_t s) {
    Dw2Args D;
    D.A = A;
    const int w0 = A.nbr > 1 ? A.wg_split : total;
    enc_dw2_split(w0, &D.n0[0], &D.n1[0]);
    D.n0[1] = D.n1[1] = 0;
    if (A.nbr > 1) enc_dw2_split(total - w0, &D.n0[1], &D.n1[1]);
    const bool pool = A.br[0].g_pooled != nullptr, msgs = A.br[0].g_msgs != nullptr;
    const bool drop = A.br[0].keep_bits != nullptr, h1r = A.br[0].h1 == nullptr;
    const dim3 g(total);
    if (pool && msgs) dw2_go<true, true>(D, g, drop, h1r, s);
    else if (pool) dw2_go<true, false>(D, g, drop, h1r, s);
    else dw2_go<false, true>(D, g, drop, h1r, s);
}

}  // namespace piml
