// Weight gradients of the PINNSF encoder on split bf16 products: LAYER-SPLIT workgroups, PRODUCER / CONSUMER waves (round 3).
//
// Reference arithmetic: the autograd of MLP(in, [128, 128, 128]) (src/models/model.py:40-65) under the processor
// Dropout_p(2 x) and the neighbour-axis sum (:82-119, :1279-1283):
//     dW3 = G3^T H2, db3 = colsum G3        G3 = keep * scale * (g_pooled[row / k] + g_msgs[row])
//     dW2 = G2^T H1, db2 = colsum G2        H1 = relu(W1 x + b1)
//     dW1 = G1^T X,  db1 = colsum G1
// The slab kernel (enc_bwd_dw_x3w_kernel, encoder_dww.hip) gives every workgroup BOTH 128 x 128 products and lets every
// wave do everything in turn -- load, split into bf16 pieces, write to LDS, barrier, products -- so the phases of a batch
// add up (s_memtime: ~5 000 cycles per 16 rows, of which the matrix pipe works 1 500), whatever the instruction scheduling.
// Here
//   * a workgroup takes ONE layer of a longer slab (L = 0: dW3, db3 from G3 | H2;  L = 1: dW2, db2, dW1, db1 from G2 | H1):
//     64 KB partial slots instead of 136 KB, two staged arrays, 16 output blocks;
//   * waves 0-3 are CONSUMERS: wave (iq, jq) owns the output blocks {2 iq, 2 iq + 1} x {2 jq, 2 jq + 1} (128 accumulator
//     registers), reads operand fragments from LDS and issues the 48 products of a 32-row batch -- and, on L = 1, the
//     dW1 / db1 sums on the otherwise idle vector pipe;
//   * waves 4-7 are PRODUCERS: wave (array, row half) loads four consecutive features of eight rows per lane with 16-byte
//     loads (two batches ahead), builds G3 where that is its array, splits, and writes half fragment entries (ds_write_b64)
//     into the other LDS buffer.  One barrier per batch; a SIMD hosts one wave of each kind, so the matrix pipe and the
//     vector pipe of a SIMD work at the same time by construction, not by scheduling.
//   * H1R (branches without h1): the H-side producers of an L = 1 workgroup compute h1 = relu(W1 x + b1) of the batch with
//     the forward's f32 matrix instructions instead of loading it (33 MB less to write in the forward and to read here).
// Operand features are dealt round-robin as in encoder_dww.hip: feature f <-> block f & 3, slot f >> 2, so that the four
// features of a producer lane land in four blocks at one slot (conflict-free 8-byte writes); element t of lane half g of a
// fragment entry = row 8 g + t of the k-block on both sides.  A recomputed H1 arrives as lane (n, h), register r = row
// (r & 3) + 8 (r >> 2) + 4 h of the batch: registers 4 q .. 4 q + 3 are exactly half h of the entry of k-group q & 1 of
// k-block q >> 1.
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
constexpr int DW2_RED = 8 * 128 * 9;                 // floats of the final dW1 / db1 exchange (8 row groups), then 4 x 128 bias sums
static_assert(DW2_PART0 + DW2_PART1 == ENC_PART, "the two slot kinds partition a full slot");
static_assert((DW2_RED + 4 * 128) * 4 <= DW2_LDS_BYTES, "final exchange fits");

#ifndef PIML_DW2_DEPTH
#define PIML_DW2_DEPTH 2
#endif

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
    const unsigned nb = __builtin_amdgcn_readfirstlane((r1 - r0 + DW2_ROWS - 1) / DW2_ROWS);
    // Buffer resources over this workgroup's slab: the (wave-uniform) row goes into the SCALAR offset, clamped to the range
    // for rows past the slab -- the hardware's range check answers those with zeros --, the lane's place inside two rows into
    // the vector offset.  A role without business with an array gets a resource of zero bytes.
    const unsigned srows = r1 - r0, sbytes = srows * EH * 4;
    auto rsrc = [&](const void* base, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    };
    auto ld = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
    };
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
    };
    const unsigned fi = lane & 31, rsub = lane >> 5;
    const unsigned avoff = rsub * (4 * EH * 4) + fi * 16;        // features 4 fi .. 4 fi + 3 of the row 4 rsub behind the scalar one
    // the partial slot: L0 [dW3 | db3] at slot p, L1 [dW2 | dW1 | db2 | db1] behind the branch's L0 slots
    float* P = L ? J.partials + (size_t)D.n0[b] * DW2_PART0 + (size_t)p * DW2_PART1 : J.partials + (size_t)p * DW2_PART0;

    if (wave < 4) {
        // ================================================ consumers ================================================
        const int iq = wave >> 1, jq = wave & 1;
        f32x16 c[2][2], sm[2][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { c[u >> 1][u & 1][r] = 0.f; sm[u >> 1][u & 1][r] = 0.f; }
        // dW1 / db1 (L = 1): features 4 fi .. 4 fi + 3 of rows 8 wave + 4 rsub .. + 3 of the batch
        float w1[4][8], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[j] = 0.f;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) w1[j][cc] = 0.f;
        }
        const __amdgpu_buffer_rsrc_t rsg = rsrc(J.g1 + (size_t)r0 * EH, L == 1 ? sbytes : 0u);
        struct G1 { float4 v[4]; };
        auto g1_load = [&](unsigned rb_) -> G1 {
            G1 g;
            const unsigned rb = __builtin_amdgcn_readfirstlane(rb_);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned row = rb + 8 * wave + t;
                g.v[t] = ld4(rsg, avoff, row < r1 ? (row - r0) * (EH * 4) : sbytes);
            }
            return g;
        };
        auto compute = [&](const float* buf, const G1& g) {
            const u32x4* B = reinterpret_cast<const u32x4*>(buf);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const u32x4* Ap = B + kb * 256 + (2 * iq) * 64 + lane;
                const u32x4* Bp = B + DW2_ARR + kb * 256 + (2 * jq) * 64 + lane;
                u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ah[u] = Ap[u * 64]; am[u] = Ap[512 + u * 64]; al[u] = Ap[1024 + u * 64];
                    bh[u] = Bp[u * 64]; bm[u] = Bp[512 + u * 64]; bl[u] = Bp[1024 + u * 64];
                }
#pragma unroll
                for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        kblock_x3(c[ia][jb], sm[ia][jb], ah[ia], am[ia], al[ia], bh[jb], bm[jb], bl[jb]);
                    }
            }
            if (L == 1) {
                const float4* xr = reinterpret_cast<const float4*>(buf + 2 * DW2_ARR * 4 + (8 * wave + 4 * rsub) * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 xa = xr[2 * t], xb = xr[2 * t + 1];
                    const float gv[4] = {g.v[t].x, g.v[t].y, g.v[t].z, g.v[t].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        w1[j][0] = __fmaf_rn(gv[j], xa.x, w1[j][0]); w1[j][1] = __fmaf_rn(gv[j], xa.y, w1[j][1]);
                        w1[j][2] = __fmaf_rn(gv[j], xa.z, w1[j][2]); w1[j][3] = __fmaf_rn(gv[j], xa.w, w1[j][3]);
                        w1[j][4] = __fmaf_rn(gv[j], xb.x, w1[j][4]); w1[j][5] = __fmaf_rn(gv[j], xb.y, w1[j][5]);
                        w1[j][6] = __fmaf_rn(gv[j], xb.z, w1[j][6]); w1[j][7] = __fmaf_rn(gv[j], xb.w, w1[j][7]);
                        s1[j] += gv[j];
                    }
                }
            }
        };
        G1 g = g1_load(r0);
        __syncthreads();                                           // batch 0 is in buffer 0
        for (unsigned t = 0; t < nb; ++t) {
            const G1 gn = g1_load(r0 + (t + 1) * DW2_ROWS);
            compute(lds + (t & 1) * DW2_BUF * 4, g);
            g = gn;
            __syncthreads();
        }
        const int n = lane & 31, h = lane >> 5;
        // accumulator (ia, jb), register r, lane (n, h): operand blocks (2 iq + ia, 2 jq + jb), slots ((r & 3) + 8 (r >> 2) + 4 h, n)
        // = dW[4 slot_a + 2 iq + ia][4 n + 2 jq + jb]: the two jb of a register are neighbours in memory
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + 2 * iq + ia;
                *reinterpret_cast<float2*>(P + (size_t)orow * EH + 4 * n + 2 * jq) =
                    make_float2(c[ia][0][r] + sm[ia][0][r], c[ia][1][r] + sm[ia][1][r]);
            }
        __syncthreads();                                           // the batch buffers are dead
        if (L == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* red = lds + ((size_t)(wave * 2 + rsub) * 128 + 4 * fi + j) * 9;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) red[cc] = w1[j][cc];
                red[8] = s1[j];
            }
        }
    } else {
        // ================================================ producers ================================================
        const unsigned pw = wave - 4, sa = pw >> 1, hh = pw & 1;         // array (0: G, 1: H), row half of a k-block
        const bool g3 = L == 0 && sa == 0;                              // this wave builds G3
        const bool pooled0 = POOL && g3;
        const bool h1r = H1R && L == 1 && sa == 1;                      // this wave recomputes H1
        const unsigned pbytes = (R / K) * EH * 4;
        const float* slab_base = L == 0 ? (sa == 0 ? J.g_msgs : J.h2) : (sa == 0 ? J.g2 : J.h1);
        const __amdgpu_buffer_rsrc_t rsa = pooled0 ? rsrc(J.g_pooled, pbytes) : rsrc(slab_base + (size_t)r0 * EH, h1r ? 0u : sbytes);
        const __amdgpu_buffer_rsrc_t rsm = rsrc(J.g_msgs + (size_t)r0 * EH, (POOL && MSGS && g3) ? sbytes : 0u);
        const unsigned kbytes = srows * 16;
        const __amdgpu_buffer_rsrc_t rsk = rsrc(DROP ? J.keep_bits + (size_t)r0 * 4 : nullptr, (DROP && g3) ? kbytes : 0u);
        const unsigned keep_all = (DROP && g3) ? 0u : 0xffffffffu;
        const float sc = g3 ? J.scale : 1.f;
        const unsigned kvoff = rsub * 64 + (fi >> 3) * 4;
        // the batch's x rows: for the consumers' dW1 (L = 1; one float per producer thread) and as the A operand of H1R
        const __amdgpu_buffer_rsrc_t rsx = rsrc(J.x + (size_t)r0 * IN, L == 1 ? srows * IN * 4 : 0u);
        const unsigned ptid = tid - 256;
        const unsigned xvoff = ((ptid & 7) < IN) ? ((ptid >> 3) * IN + (ptid & 7)) * 4 : 0x7fff0000u;
        float w1b[2][4], b1n[2] = {0.f, 0.f};
        unsigned xav[4] = {0x7fff0000u, 0x7fff0000u, 0x7fff0000u, 0x7fff0000u};
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) w1b[v][s_] = 0.f;
        if (h1r) {       // virtual block vb = 2 hh + v: features 4 n + vb; B operand lane (n, g): W1[feature][2 s + g]
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const unsigned cx = 2u * s_ + rsub;
                if (cx < IN) {
                    xav[s_] = (fi * IN + cx) * 4;
#pragma unroll
                    for (int v = 0; v < 2; ++v) w1b[v][s_] = J.w1[(size_t)(4 * fi + 2 * hh + v) * IN + cx];
                }
            }
#pragma unroll
            for (int v = 0; v < 2; ++v) b1n[v] = J.b1[4 * fi + 2 * hh + v];
        }
        float sb[4] = {0.f, 0.f, 0.f, 0.f};          // column sums of the G array's four features (db3 / db2)

        struct Stage { float4 a[8], m[8]; float xa[4], x; unsigned kw[8]; };
        auto stage_load = [&](unsigned rb_) -> Stage {
            Stage S;
            const unsigned rb = __builtin_amdgcn_readfirstlane(rb_);
            const unsigned xso = rb < r1 ? (rb - r0) * IN * 4 : srows * IN * 4;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                S.xa[s_] = 0.f;
                if (H1R) S.xa[s_] = ld(rsx, h1r ? xav[s_] : 0x7fff0000u, xso);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {                                // e = 4 kb + t
                const unsigned row = rb + 16 * (e >> 2) + 8 * hh + (e & 3);     // scalar: the row of lane half 0 (half 1: + 4)
                const unsigned rel = row < r1 ? (row - r0) * (EH * 4) : sbytes;
                unsigned vo = avoff, so = rel;
                if (POOL) {
                    const unsigned mine = row + 4 * rsub;
                    const unsigned pv = mine < r1 ? __umulhi(mine, kmagic) * (EH * 4) + fi * 16 : pbytes;
                    vo = pooled0 ? pv : avoff;
                    so = pooled0 ? 0u : rel;
                }
                S.a[e] = ld4(rsa, vo, so);
                S.m[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (POOL && MSGS) S.m[e] = ld4(rsm, avoff, rel);
                S.kw[e] = 0u;
                if (DROP) S.kw[e] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsk, (int)kvoff, (int)(row < r1 ? (row - r0) * 16 : kbytes), 0);
            }
            S.x = ld(rsx, xvoff, xso);
            return S;
        };
        auto stage_write = [&](const Stage& S, float* buf) {
            uint2* base = reinterpret_cast<uint2*>(buf);
            if (h1r) {
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = b1n[v];
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) acc = mfma32(S.xa[s_], w1b[v][s_], acc);
                    // lane (n = fi, h = rsub): registers 4 q .. 4 q + 3 = half h of entry (block vb, slot n + 32 (q & 1)) of k-block q >> 1
                    uint2* d0 = base + ((size_t)DW2_ARR + (2 * hh + v) * 64 + fi) * 2 + rsub;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned h0, m0, l0, h1, m1, l1;
                        split3(relu1(acc[4 * q]), relu1(acc[4 * q + 1]), h0, m0, l0);
                        split3(relu1(acc[4 * q + 2]), relu1(acc[4 * q + 3]), h1, m1, l1);
                        uint2* d = d0 + ((q >> 1) * 256 + 32 * (q & 1)) * 2;
                        d[0] = make_uint2(h0, h1);
                        d[2 * 512] = make_uint2(m0, m1);
                        d[2 * 1024] = make_uint2(l0, l1);
                    }
                }
            } else {
                // entry (array sa, piece, k-block, block j, slot fi + 32 hh), half rsub (rows 4 rsub .. + 3 of the unit), as uint2
                uint2* d0 = base + ((size_t)sa * DW2_ARR + fi + 32 * hh) * 2 + rsub;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float u[4][4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 a = S.a[4 * kb + t], m = S.m[4 * kb + t];
                        u[t][0] = (a.x + m.x) * sc; u[t][1] = (a.y + m.y) * sc; u[t][2] = (a.z + m.z) * sc; u[t][3] = (a.w + m.w) * sc;
                        if (DROP) {
                            const unsigned kw = (S.kw[4 * kb + t] | keep_all) >> ((4 * fi) & 31);
#pragma unroll
                            for (int j = 0; j < 4; ++j) u[t][j] = keep_if(u[t][j], kw, j);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sb[j] += (u[0][j] + u[1][j]) + (u[2][j] + u[3][j]);
                        unsigned h0, m0, l0, h1, m1, l1;
                        split3(u[0][j], u[1][j], h0, m0, l0);
                        split3(u[2][j], u[3][j], h1, m1, l1);
                        uint2* d = d0 + (kb * 256 + j * 64) * 2;
                        d[0] = make_uint2(h0, h1);
                        d[2 * 512] = make_uint2(m0, m1);
                        d[2 * 1024] = make_uint2(l0, l1);
                    }
                }
            }
            buf[2 * DW2_ARR * 4 + ptid] = S.x;
        };
        // DEPTH batches ahead.  Measured at the 4096-agent scene (us, same box): depth 2: 39.6, 3: 41.2, 4: 41.5, 5: 44.8 -- the
        // bytes in flight are not what holds this kernel back (a deeper ring costs registers and moves instead).
        constexpr int DEPTH = (POOL && MSGS) ? (DROP && H1R ? 1 : 2) : (DROP ? (PIML_DW2_DEPTH > 3 ? 3 : PIML_DW2_DEPTH) : PIML_DW2_DEPTH);      // (the g_msgs variant holds two arrays per batch; with keep bits and the recomputation on top, two batches in flight spilled 25 registers)
        constexpr int UNROLL = (DEPTH % 2) ? 2 * DEPTH : DEPTH;
        Stage S[DEPTH];
        S[0] = stage_load(r0);
        stage_write(S[0], lds);
#pragma unroll
        for (int i = 0; i < DEPTH - 1; ++i) S[i] = stage_load(r0 + (i + 1) * DW2_ROWS);      // S[i]: batch i + 1
        __syncthreads();                                           // batch 0 is in buffer 0
        for (unsigned t0 = 0; t0 < nb; t0 += UNROLL) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                const unsigned t = t0 + i;                             // the consumers multiply batch t; t % DEPTH == i % DEPTH
                if (t < nb) {
                    S[(i + DEPTH - 1) % DEPTH] = stage_load(r0 + (t + DEPTH) * DW2_ROWS);
                    stage_write(S[i % DEPTH], lds + ((i + 1) & 1) * DW2_BUF * 4);          // batch t + 1
                    __syncthreads();
                }
            }
        }
        __syncthreads();                                           // the batch buffers are dead
        if (sa == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lds[DW2_RED + (hh * 2 + rsub) * 128 + 4 * fi + j] = sb[j];
        }
    }
    // ---- column sums (and dW1): the partial sums of the row groups meet in LDS ----
    __syncthreads();
    if (tid < 128) {
        const float* q = lds + DW2_RED + tid;
        const float dbG = (q[0] + q[128]) + (q[256] + q[384]);
        if (L == 0) {
            P[EH * EH + tid] = dbG;
        } else {
            float acc[9];
#pragma unroll
            for (int cc = 0; cc < 9; ++cc) {
                float v[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) v[g] = lds[((size_t)g * 128 + tid) * 9 + cc];
                acc[cc] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
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
// against h2) but their H side is recomputed, not loaded, and building G3 is the dearest staging: measured best at an even
// split (PIML_DW2_L0_SHARE, per mille: 300: 51.7 us, 420: 42.0, 500: 40.2, 550: 43.9, 600: 46.4)
static int dw2_l0_share() {
    static int v = getenv("PIML_DW2_L0_SHARE") ? atoi(getenv("PIML_DW2_L0_SHARE")) : 500;
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
void enc_dw2_launch(const EncArgs& A, int total, hipStream_t s, bool l0_only) {
    Dw2Args D;
    D.A = A;
    const int w0 = A.nbr > 1 ? A.wg_split : total;
    enc_dw2_split(w0, &D.n0[0], &D.n1[0]);
    D.n0[1] = D.n1[1] = 0;
    if (A.nbr > 1) enc_dw2_split(total - w0, &D.n0[1], &D.n1[1]);
    if (l0_only) {
        D.n0[0] = w0; D.n1[0] = 0;
        D.n0[1] = A.nbr > 1 ? total - w0 : 0; D.n1[1] = 0;
    }
    const bool pool = A.br[0].g_pooled != nullptr, msgs = A.br[0].g_msgs != nullptr;
    const bool drop = A.br[0].keep_bits != nullptr, h1r = A.br[0].h1 == nullptr;
    const dim3 g(total);
    if (pool && msgs) dw2_go<true, true>(D, g, drop, h1r, s);
    else if (pool) dw2_go<true, false>(D, g, drop, h1r, s);
    else dw2_go<false, true>(D, g, drop, h1r, s);
}

}  // namespace piml
