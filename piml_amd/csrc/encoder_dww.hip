// Weight gradients of the PINNSF encoder on split bf16 products: the SLAB kernel (few rows; wide staging loads, round 3).
//
// Reference arithmetic: the autograd of MLP(in, [128, 128, 128]) (src/models/model.py:40-65) under the processor
// Dropout_p(2 x) and the neighbour-axis sum (:82-119, :1279-1283):
//     dW3 = G3^T H2, db3 = colsum G3        G3 = keep * scale * (g_pooled[row / k] + g_msgs[row])
//     dW2 = G2^T H1, db2 = colsum G2
//     dW1 = G1^T X,  db1 = colsum G1
// A workgroup owns a row slab and both layers: split-K over slabs, one partial slot per workgroup in enc_bwd_dw_kernel's layout
// (encoder.hip; the same slot sum afterwards).  Both operands of a product are data, so both are split into their three
// bf16 pieces on the way into LDS, per buffer (u32x4): [array 4: G3 G2 H2 H1][piece 3][block 4][lane 64] = 48 KB + the
// batch's x rows; two buffers; a batch is 16 rows = one k-block; wave (L, iq, jq) owns the output blocks {2 iq, 2 iq + 1} x
// {2 jq, 2 jq + 1} of layer L (main + small accumulators: 128 registers), 24 matrix instructions a batch; dW1 / db1 on the
// vector pipe by feature.  Every load goes through a buffer resource over the slab with the (wave-uniform) row in the
// SCALAR offset: no address arithmetic on the vector pipe, rows past the slab answered with zeros by the range check.
// History (round 2 form, removed in round 3): a staging thread owned one feature and loaded eight rows of it with eight
// dword loads (a fragment entry = 8 rows of ONE feature) -- 21 load instructions per wave and batch, 256 bytes each, and
// the CU's address unit, not the matrix pipe, set the pace (s_memtime, wave 0, per batch: load issue 1 950 cycles as a
// burst, products 1 980, split + LDS writes 1 250, barrier 320; the loads were then issued BETWEEN the products, as they
// still are here).  Measured on that form and dropped, all within +-3 us: loads two batches ahead in two register sets,
// the two waves of a SIMD taking the phases of a barrier interval in opposite order, one scalar offset per 8-row unit.
// Here a thread loads FOUR consecutive features of four rows with four 16-byte loads (a wave instruction = two whole
// rows, 1 KB) and writes, per feature, the three pieces of its four rows as half a fragment entry (ds_write_b64): 9 load
// instructions per wave and batch, 46 -> 42 us at the 4096-agent scene.  For
// those writes to be conflict-free the features of a thread must land in four DIFFERENT blocks at the SAME lane slot, so
// the operand features are dealt round-robin:
//     feature f  <->  block f & 3, slot f >> 2             (instead of block f >> 5, slot f & 31)
// on both operand sides; the contraction (rows) is untouched, every output element is the same sum of the same products
// in the same order, and only the address an accumulator register is stored to changes (epilogue).
#include "common.hpp"
#include "encoder.hpp"
#include "x3.hpp"

namespace piml {

template <bool POOL, bool MSGS, bool DROP>
__global__ __launch_bounds__(ENC_THREADS) void enc_bwd_dw_x3w_kernel(EncArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (A.nbr > 1 && (int)blockIdx.x >= A.wg_split) ? 1 : 0;
    const piml_encoder_branch J = b ? A.br[1] : A.br[0];
    const int wg0 = b ? A.wg_split : 0;
    const int nwg = b ? (int)gridDim.x - A.wg_split : (A.nbr > 1 ? A.wg_split : (int)gridDim.x);
    const unsigned p = (unsigned)((int)blockIdx.x - wg0);
    const unsigned R = (unsigned)J.rows;                   // rows < 2^24 (checked on the host): 32-bit indexing
    const unsigned IN = __builtin_amdgcn_readfirstlane((unsigned)J.in_dim), K = __builtin_amdgcn_readfirstlane((unsigned)J.k);
    const unsigned kmagic = __builtin_amdgcn_readfirstlane((unsigned)((0x100000000ull + K - 1) / K));      // row / K == umulhi(row, kmagic) for row * K < 2^32
    unsigned slab = (R + nwg - 1) / nwg;
    slab = (slab + 1) & ~1u;
    const unsigned r0 = __builtin_amdgcn_readfirstlane(p * slab < R ? p * slab : R);
    const unsigned r1 = __builtin_amdgcn_readfirstlane(r0 + slab < R ? r0 + slab : R);
    const int L = wave >> 2, iq = (wave >> 1) & 1, jq = wave & 1;

    f32x16 c[2][2], sm[2][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c[u >> 1][u & 1][r] = 0.f; sm[u >> 1][u & 1][r] = 0.f; }
    // staging role: array sa (0: G3, 1: G2, 2: H2, 3: H1), row half hh of the batch; lane (fi, rsub): features 4 fi .. 4 fi + 3
    // of rows 8 hh + 4 rsub .. + 3.  (sa, hh: wave-uniform.)
    const unsigned sa = wave >> 1, hh = wave & 1, fi = lane & 31, rsub = lane >> 5;
    // dW1 / db1 role: feature sf, rows 4 rg .. 4 rg + 3 of the batch
    const unsigned sf = tid & 127, rg = wave >> 1;
    float sb[4] = {0.f, 0.f, 0.f, 0.f};          // bias sums of this thread's four features (db3 on array 0, db2 on array 1)
    float s1 = 0.f;
    float w1[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) w1[cc] = 0.f;
    // Buffer resources whose range is this workgroup's slab; the row goes into the SCALAR offset (clamped to the range for
    // rows past the slab, which the hardware range check then answers with zeros), the lane's place inside two rows into the
    // vector offset.  A wave that has no business with an array gets a resource of zero bytes: its loads return zeros without
    // touching memory, and the instruction stream stays free of branches (the loads are interleaved with the products).
    const unsigned srows = r1 - r0, sbytes = srows * EH * 4;
    auto rsrc = [&](const void* base, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    };
    const bool pooled0 = POOL && sa == 0;                        // G3 from the per-agent gradient: row / k into the whole array
    const unsigned pbytes = (R / K) * EH * 4;
    const float* slab_base = (sa == 0 ? J.g_msgs : (sa == 1 ? J.g2 : (sa == 2 ? J.h2 : J.h1)));
    const __amdgpu_buffer_rsrc_t rsa = pooled0 ? rsrc(J.g_pooled, pbytes) : rsrc(slab_base + (size_t)r0 * EH, sbytes);
    const __amdgpu_buffer_rsrc_t rsm = rsrc(J.g_msgs + (size_t)r0 * EH, (POOL && MSGS && sa == 0) ? sbytes : 0u);
    const __amdgpu_buffer_rsrc_t rsg = rsrc(J.g1 + (size_t)r0 * EH, sbytes);
    const __amdgpu_buffer_rsrc_t rsx = rsrc(J.x + (size_t)r0 * IN, srows * IN * 4);
    // dropout: the keep word of (row, feature block f >> 5), 16 bytes per row (array 0 only: g3 = keep * scale * (...))
    const unsigned kbytes = srows * 16;
    const __amdgpu_buffer_rsrc_t rsk = rsrc(DROP ? J.keep_bits + (size_t)r0 * 4 : nullptr, (DROP && sa == 0) ? kbytes : 0u);
    const unsigned keep_all = (DROP && sa == 0) ? 0u : 0xffffffffu;
    const float sc = sa == 0 ? J.scale : 1.f;
    const unsigned xvoff = (tid < 128 && (unsigned)(tid & 7) < IN) ? ((tid >> 3) * IN + (tid & 7)) * 4 : 0x7fff0000u;
    const unsigned avoff = rsub * (4 * EH * 4) + fi * 16, kvoff = rsub * 64 + (fi >> 3) * 4;
    auto ld = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
    };
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
    };

    struct Stage { float4 a[4], m[4]; float g1[4], x; unsigned kw[4]; };
    auto stage_load = [&](unsigned rb_) -> Stage {           // issue the loads of the batch starting at row rb
        Stage S;
        const unsigned rb = __builtin_amdgcn_readfirstlane(rb_);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned row = rb + 8 * hh + t;                       // scalar: the row of lane half 0 (half 1: + 4)
            const unsigned rel = row < r1 ? (row - r0) * (EH * 4) : sbytes;
            unsigned vo = avoff, so = rel;
            if (POOL) {
                const unsigned mine = row + 4 * rsub;
                const unsigned pv = mine < r1 ? __umulhi(mine, kmagic) * (EH * 4) + fi * 16 : pbytes;
                vo = pooled0 ? pv : avoff;
                so = pooled0 ? 0u : rel;
            }
            S.a[t] = ld4(rsa, vo, so);
            S.m[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (POOL && MSGS) S.m[t] = ld4(rsm, avoff, rel);
            S.kw[t] = 0u;
            if (DROP) S.kw[t] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsk, (int)kvoff, (int)(row < r1 ? (row - r0) * 16 : kbytes), 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned row = rb + 4 * rg + t;
            S.g1[t] = ld(rsg, sf * 4, row < r1 ? (row - r0) * (EH * 4) : sbytes);
        }
        S.x = ld(rsx, xvoff, rb < r1 ? (rb - r0) * IN * 4 : srows * IN * 4);
        return S;
    };
    float gq[4];                                             // g1 values of the batch in the compute phase
    auto stage_write = [&](const Stage& S, float* buf) {     // registers -> split -> LDS
        // entry (array sa, piece, block j, slot fi + 32 hh), half rsub (rows 4 rsub .. 4 rsub + 3 of the unit), as uint2
        uint2* d0 = reinterpret_cast<uint2*>(buf) + ((size_t)sa * DWX_ARR + fi + 32 * hh) * 2 + rsub;
        float u[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u[t][0] = (S.a[t].x + S.m[t].x) * sc; u[t][1] = (S.a[t].y + S.m[t].y) * sc;
            u[t][2] = (S.a[t].z + S.m[t].z) * sc; u[t][3] = (S.a[t].w + S.m[t].w) * sc;
            if (DROP) {
                const unsigned kw = (S.kw[t] | keep_all) >> ((4 * fi) & 31);
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
            uint2* d = d0 + j * 128;
            d[0] = make_uint2(h0, h1);
            d[512] = make_uint2(m0, m1);
            d[1024] = make_uint2(l0, l1);
        }
        if (tid < 128) buf[4 * DWX_ARR * 4 + tid] = S.x;
    };
    auto take_g1 = [&](const Stage& S) {
#pragma unroll
        for (int t = 0; t < 4; ++t) gq[t] = S.g1[t];
    };
    auto compute = [&](const float* buf) {
        const u32x4* B = reinterpret_cast<const u32x4*>(buf);
        const u32x4* Ap = B + (L ? 1 : 0) * DWX_ARR + (2 * iq) * 64 + lane;
        const u32x4* Bp = B + (L ? 3 : 2) * DWX_ARR + (2 * jq) * 64 + lane;
        u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ah[u] = Ap[u * 64]; am[u] = Ap[256 + u * 64]; al[u] = Ap[512 + u * 64];
            bh[u] = Bp[u * 64]; bm[u] = Bp[256 + u * 64]; bl[u] = Bp[512 + u * 64];
        }
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) kblock_x3(c[ia][jb], sm[ia][jb], ah[ia], am[ia], al[ia], bh[jb], bm[jb], bl[jb]);
        // dW1 / db1: rows 4 rg .. 4 rg + 3 of the batch
        const float4* xr = reinterpret_cast<const float4*>(buf + 4 * DWX_ARR * 4 + rg * 32);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 xa = xr[2 * t], xb = xr[2 * t + 1];
            const float g = gq[t];
            w1[0] = __fmaf_rn(g, xa.x, w1[0]); w1[1] = __fmaf_rn(g, xa.y, w1[1]);
            w1[2] = __fmaf_rn(g, xa.z, w1[2]); w1[3] = __fmaf_rn(g, xa.w, w1[3]);
            w1[4] = __fmaf_rn(g, xb.x, w1[4]); w1[5] = __fmaf_rn(g, xb.y, w1[5]);
            w1[6] = __fmaf_rn(g, xb.z, w1[6]); w1[7] = __fmaf_rn(g, xb.w, w1[7]);
            s1 += g;
        }
    };
    constexpr int NLOADS = 4 + ((POOL && MSGS) ? 4 : 0) + (DROP ? 4 : 0) + 4 + 1;
    if (r0 < r1) {
        const unsigned nb = (r1 - r0 + DW_X3_ROWS - 1) / DW_X3_ROWS;
        {
            const Stage S = stage_load(r0);
            stage_write(S, lds);
            take_g1(S);
        }
        __syncthreads();
        for (unsigned t = 0; t < nb; ++t) {
            float* cur = lds + (t & 1) * DWX_BUF * 4;
            float* nxt = lds + ((t + 1) & 1) * DWX_BUF * 4;
            // the next batch's loads are issued between this batch's products; nothing of them is touched before the fence
            const Stage S2 = stage_load(r0 + (t + 1) * DW_X3_ROWS);
            compute(cur);
            __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);          // fragment + x reads
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one product
                if (i < NLOADS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // one load
            }
            __builtin_amdgcn_sched_barrier(0);
            stage_write(S2, nxt);
            take_g1(S2);
            __syncthreads();
        }
    }
    float* P = J.partials + (size_t)p * ENC_PART;
    const int n = lane & 31, h = lane >> 5;
    // accumulator (ia, jb), register r, lane (n, h): operand blocks (2 iq + ia, 2 jq + jb), slots ((r & 3) + 8 (r >> 2) + 4 h, n)
    // = dW[4 slot_a + 2 iq + ia][4 n + 2 jq + jb]: the two jb of a register are neighbours in memory
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + 2 * iq + ia;
            *reinterpret_cast<float2*>(P + L * 16384 + (size_t)orow * EH + 4 * n + 2 * jq) =
                make_float2(c[ia][0][r] + sm[ia][0][r], c[ia][1][r] + sm[ia][1][r]);
        }
    // dW1 and the bias gradients: partial sums of the row groups meet in LDS (the batch buffers are dead)
    __syncthreads();
    {
        float* red = lds + (rg * 128 + sf) * 9;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) red[cc] = w1[cc];
        red[8] = s1;
        if (sa < 2) {
            float* red2 = lds + DWX_RED + ((hh * 2 + rsub) * 128 + 4 * fi) * 2 + sa;
#pragma unroll
            for (int j = 0; j < 4; ++j) red2[2 * j] = sb[j];
        }
    }
    __syncthreads();
    if (tid < 128) {
        float acc[9];
#pragma unroll
        for (int cc = 0; cc < 9; ++cc)
            acc[cc] = (lds[(0 * 128 + tid) * 9 + cc] + lds[(1 * 128 + tid) * 9 + cc]) + (lds[(2 * 128 + tid) * 9 + cc] + lds[(3 * 128 + tid) * 9 + cc]);
        float* o = P + 32768 + tid * IN;                  // dW1 row-major (128, in_dim) at the head of its 1024 floats
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
            if ((unsigned)cc < IN) o[cc] = acc[cc];
        P[32768 + 1024 + 256 + tid] = acc[8];
        const float* q = lds + DWX_RED + tid * 2;
        P[32768 + 1024 + tid] = (q[0] + q[256]) + (q[512] + q[768]);
        P[32768 + 1024 + 128 + tid] = (q[1] + q[257]) + (q[513] + q[769]);
    }
}

int enc_dww_set_attributes() {
    const void* dw[6] = {reinterpret_cast<const void*>(enc_bwd_dw_x3w_kernel<true, true, false>),
                         reinterpret_cast<const void*>(enc_bwd_dw_x3w_kernel<true, false, false>),
                         reinterpret_cast<const void*>(enc_bwd_dw_x3w_kernel<false, true, false>),
                         reinterpret_cast<const void*>(enc_bwd_dw_x3w_kernel<true, true, true>),
                         reinterpret_cast<const void*>(enc_bwd_dw_x3w_kernel<true, false, true>),
                         reinterpret_cast<const void*>(enc_bwd_dw_x3w_kernel<false, true, true>)};
    for (const void* f : dw)
        if (int e = (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, DWX_LDS_BYTES)) return e;
    return 0;
}

void enc_dww_launch(const EncArgs& B, int grid, bool drop, hipStream_t s) {
    const bool pool = B.br[0].g_pooled != nullptr, msgs = B.br[0].g_msgs != nullptr;
    const dim3 g(grid), b(ENC_THREADS);
    if (drop) {
        if (pool && msgs) hipLaunchKernelGGL((enc_bwd_dw_x3w_kernel<true, true, true>), g, b, DWX_LDS_BYTES, s, B);
        else if (pool) hipLaunchKernelGGL((enc_bwd_dw_x3w_kernel<true, false, true>), g, b, DWX_LDS_BYTES, s, B);
        else hipLaunchKernelGGL((enc_bwd_dw_x3w_kernel<false, true, true>), g, b, DWX_LDS_BYTES, s, B);
    } else {
        if (pool && msgs) hipLaunchKernelGGL((enc_bwd_dw_x3w_kernel<true, true, false>), g, b, DWX_LDS_BYTES, s, B);
        else if (pool) hipLaunchKernelGGL((enc_bwd_dw_x3w_kernel<true, false, false>), g, b, DWX_LDS_BYTES, s, B);
        else hipLaunchKernelGGL((enc_bwd_dw_x3w_kernel<false, true, false>), g, b, DWX_LDS_BYTES, s, B);
    }
}

}  // namespace piml
