// The fused PINNSF network as one call per direction: the launch stages of encoder.hip / decoder.hip in as few launches
// as their data dependences allow.
//
// Replaces the body of PINNSF.forward / its autograd backward (src/models/model.py:1271-1305: ped_encoder /
// obs_encoder -> sum over neighbours -> decoders -> predictors -> desired force, and the `pinnsf_m` collision head
// :1296-1300).  Most stages of one step are small next to the chip (the decoder tails and the head work on 4096 agents,
// the packs and reductions on < 1 MB), and every dependent launch on gfx950 pays ~4.5 us for the end-of-kernel
// write-back + start-of-kernel invalidate of the eight per-XCD L2s.  Default (serial) order, one stream:
//
//   pack      ONE launch for every weight image (pinnsf_pack_kernel)                 -- skipped with PIML_PACKED_VALID
//   forward   encoders (both branches) -> [neighbour-axis sums + decoder tails + desired force + collision head]
//   backward  [decoder dX chain + dW partials] -> encoder dX -> encoder dW -> ONE slot sum for all four partial sets
//
// PIML_FORK (opt-in, measured SLOWER inside captured graphs on ROCm 7.2: every cross-stream edge of a replayed graph
// costs more than the ~5 us stage it hides) puts the independent stages on library-owned side streams instead:
//
//   forward    main:  [enc pack] -> encoders ----------------> pooling -> decoder tails -> join
//              side0: dec pack, head pack ----\                           ^
//              side1:                          \-> (after encoders) head -/
//   backward   main:  decoder dX -> encoder dX -> encoder dW ------------> join -> encoder reduction
//              side0:            \-> decoder dW -> decoder reduction ----/
//
// The side streams belong to the library (one pair per device, created on first use outside any capture).
#include <cstdio>
#include <mutex>

#include "common.hpp"
#include "pack.hpp"
#include "reduce.hpp"
#include "stages.hpp"
#include "trace.hpp"

namespace piml {

namespace {

struct Side {
    hipStream_t s[2] = {nullptr, nullptr};
    hipEvent_t ev[6] = {};
    bool ok = false;
};

constexpr int kMaxDevices = 64;
Side g_side[kMaxDevices];
std::mutex g_mu;

int side_streams(Side** out) {
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_mu);
    Side& S = g_side[dev];
    if (!S.ok) {
        // creating streams / events is not a capturable operation: fail clearly instead of invalidating the capture
        for (auto& s : S.s)
            if (hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) return e;
        for (auto& ev : S.ev)
            if (hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) return e;
        S.ok = true;
    }
    *out = &S;
    return hipSuccess;
}

// `to` continues after everything enqueued on `from` so far
int edge(hipStream_t from, hipStream_t to, hipEvent_t ev) {
    if (hipError_t e = hipEventRecord(ev, from)) return e;
    return hipStreamWaitEvent(to, ev, 0);
}

// a failing stage names itself on stderr: the hipError_t alone does not say which of the ~10 launches refused
#define PIML_TRY(x)                                                                        \
    do {                                                                                   \
        if (int e_ = (x)) {                                                                \
            fprintf(stderr, "libpiml_hip: %s -> %d (%s:%d)\n", #x, e_, __FILE__, __LINE__); \
            return e_;                                                                     \
        }                                                                                  \
    } while (0)

}  // namespace

// ---- one launch for every weight image of the network: blockIdx.y = encoder branches, decoder branches, head (reduce.hpp) ----
__global__ __launch_bounds__(256) void pinnsf_pack_kernel(PackAll A) {
    __shared__ double red[256];
    pack_block(A, (int)blockIdx.x, 256, red);
}

int launch_pack(const PackAll& A, hipStream_t s) {
    hipLaunchKernelGGL(pinnsf_pack_kernel, dim3((unsigned)pack_blocks_total(A, 256)), dim3(256), 0, s, A);
    trace_mark("pinnsf_pack", s);
    return hipGetLastError();
}

// ---- one launch for every slot sum of the backward pass (reduce.hpp): a set of slots per gx workgroups (encoder branches --
// one set each, or two with the layer-split slots of encoder_dw2.hip -- then the decoder branches) ----
__global__ __launch_bounds__(256) void pinnsf_reduce_kernel(ReduceAll A) { reduce_block(A, (int)blockIdx.x); }

int launch_slot_sums(const ReduceAll& R, hipStream_t s) {
    hipLaunchKernelGGL(pinnsf_reduce_kernel, dim3((unsigned)(R.gx * R.nsets)), dim3(256), 0, s, R);
    return hipGetLastError();
}

// ---- PIML_POOL_TRAIN: the folded first layers' gradients -> the gradients of their factors (reduce.hpp: UnfoldSet) ----
// float64 accumulation (the kernel is 4 M multiply-adds: its time is the launch); one thread per output element.
__global__ __launch_bounds__(256) void pinnsf_unfold_kernel(ReduceAll R) {
    const UnfoldSet U = R.unf[blockIdx.y];
    const int x = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ G = U.dgrads;
    const float* __restrict__ gb = U.dgrads + DD * DH + DD * DD + 2 * DD;
    const double sc = (double)U.scale;
    if (x < 64) {                     // row x of dW1: s (G[x][:] W3^T + k g_b[x] b3)
        __shared__ float g[DH];
        __shared__ double part[256];
        if (tid < DH) g[tid] = G[(size_t)x * DH + tid];
        __syncthreads();
        const int m = tid & 127, half = tid >> 7;
        const float4* wr = reinterpret_cast<const float4*>(U.w3 + (size_t)m * EH + 64 * half);
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 wv = wr[q];
            const float* gq = g + 64 * half + 4 * q;
            a += (double)gq[0] * wv.x + (double)gq[1] * wv.y + (double)gq[2] * wv.z + (double)gq[3] * wv.w;
        }
        part[tid] = a;
        __syncthreads();
        if (tid < DH)
            U.dw1_out[(size_t)x * DH + tid] = (float)(sc * ((part[tid] + part[tid + 128]) + (double)U.k * (double)gb[x] * (double)U.b3[tid]));
    } else if (x < 128) {             // rows 2 (x - 64), + 1 of dW3 = s W1^T G
        const int m = 2 * (x - 64) + (tid >> 7), j = tid & 127;
        double a = 0.0;
#pragma unroll 8
        for (int i = 0; i < DD; ++i) a += (double)U.w1[(size_t)i * DH + m] * (double)G[(size_t)i * DH + j];
        U.egrads[(size_t)m * EH + j] = (float)(sc * a);
    } else if (tid < EH) {            // db3 = s k W1^T g_b
        double a = 0.0;
        for (int i = 0; i < DD; ++i) a += (double)U.w1[(size_t)i * DH + tid] * (double)gb[i];
        U.egrads[2 * EH * EH + 8 * EH + tid] = (float)(sc * (double)U.k * a);
    }
}

// (Round 5, built, parity green, measured and removed: the unfold as workgroups of the launch that sums the slots, gated by a
// device-side ticket the decoder sets' sum workgroups raise -- sums written with device-scope stores, unfold reads through LDS in
// one round of device-scope loads.  At the tail of the launch the chain [sums, ticket, uncached loads, products] ran 19 - 21 us
// against 9.4 + 4.7 as two launches; dispatched early -- right behind the decoder sets -- the 258 polling workgroups cost the
// launch more still, 33 us; an ACQUIRE in the poll invalidates the XCD's L2 on every round: 49 us.  Two launches it stays.)
static int launch_unfold_now(const ReduceAll& R, hipStream_t s) {
    if (R.nunf <= 0) return hipSuccess;
    hipLaunchKernelGGL(pinnsf_unfold_kernel, dim3(kUnfoldBlocks, (unsigned)R.nunf), dim3(256), 0, s, R);
    trace_mark("pinnsf_unfold", s);
    return hipGetLastError();
}

// ---- deferred unfold (piml_pinnsf_unfold_defer): one waiting entry per device ----
// The unfold READS the folded layers' summed gradients and OVERWRITES the unfolded ones, so of the backward passes of one optimiser
// step that accumulate into the same buffers (PIML_ACCUMULATE: the frames of a training rollout) only the LAST pass's unfold
// matters -- the earlier ones compute from partial sums what the last one computes again.  While deferring, launch_unfold records
// the sets instead of launching; another network's sets first launch what is waiting.
namespace {
struct PendingUnfold {
    ReduceAll R;
    hipStream_t stream = nullptr;
    bool valid = false, deferring = false;
};
PendingUnfold g_pending_unfold[kMaxDevices];
PendingUnfold* pending_unfold_entry() {
    int dev = 0;
    if (hipGetDevice(&dev) || dev < 0 || dev >= kMaxDevices) return nullptr;
    return &g_pending_unfold[dev];
}
bool same_unfold(const ReduceAll& a, const ReduceAll& b) {
    if (a.nunf != b.nunf) return false;
    for (int i = 0; i < a.nunf; ++i)
        if (a.unf[i].dgrads != b.unf[i].dgrads || a.unf[i].dw1_out != b.unf[i].dw1_out || a.unf[i].egrads != b.unf[i].egrads ||
            a.unf[i].w1 != b.unf[i].w1 || a.unf[i].w3 != b.unf[i].w3 || a.unf[i].b3 != b.unf[i].b3 || a.unf[i].k != b.unf[i].k ||
            a.unf[i].scale != b.unf[i].scale)
            return false;
    return true;
}
}  // namespace

int launch_unfold(const ReduceAll& R, hipStream_t s) {
    if (R.nunf <= 0) return hipSuccess;
    PendingUnfold* P = pending_unfold_entry();
    ReduceAll old;
    hipStream_t olds = nullptr;
    bool flush_old = false, deferred = false;
    if (P) {
        std::lock_guard<std::mutex> lock(g_mu);
        if (P->deferring) {
            if (P->valid && !same_unfold(P->R, R)) { old = P->R; olds = P->stream; flush_old = true; }
            P->R = R; P->stream = s; P->valid = true;
            deferred = true;
        }
    }
    if (!deferred) return launch_unfold_now(R, s);
    return flush_old ? launch_unfold_now(old, olds) : (int)hipSuccess;
}

PIML_API int piml_pinnsf_unfold_defer(int on, void* stream) {
    PendingUnfold* P = pending_unfold_entry();
    if (!P) return hipErrorInvalidDevice;
    ReduceAll R;
    hipStream_t s = nullptr;
    bool go = false;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (P->valid) { R = P->R; s = stream ? as_stream(stream) : P->stream; go = true; P->valid = false; }
        P->deferring = on != 0;
    }
    return go ? launch_unfold_now(R, s) : (int)hipSuccess;
}

// ---- deferred slot sums (PIML_DEFER_SLOT_SUMS): one waiting entry per device ----
namespace {
struct PendingSums {
    ReduceAll R;
    hipStream_t stream = nullptr;
    bool valid = false;
};
PendingSums g_pending[kMaxDevices];

PendingSums* pending_entry() {
    int dev = 0;
    if (hipGetDevice(&dev) || dev < 0 || dev >= kMaxDevices) return nullptr;
    return &g_pending[dev];
}
}  // namespace

namespace {
struct PendingPack {
    PackAll A;
    hipStream_t stream = nullptr;
    bool valid = false;
};
PendingPack g_pending_pack[kMaxDevices];
PendingPack* pending_pack_entry() {
    int dev = 0;
    if (hipGetDevice(&dev) || dev < 0 || dev >= kMaxDevices) return nullptr;
    return &g_pending_pack[dev];
}
}  // namespace

// consumer: the stream of the launch that is about to read the images (nullptr: unknown -- the stream the pack was left on).  A
// pack left on another stream than its consumer's runs on the CONSUMER's stream: there it is ordered in front of the reader
// (and inside the same capture), which the stream it was left on does not promise.
int pending_pack_flush(hipStream_t consumer) {
    PendingPack* P = pending_pack_entry();
    if (!P) return hipErrorInvalidDevice;
    PackAll A;
    hipStream_t s;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!P->valid) return hipSuccess;
        A = P->A; s = P->stream; P->valid = false;
    }
    return launch_pack(A, consumer ? consumer : s);
}

int pending_pack_leave(const PackAll& A, hipStream_t s) {
    if (int e = pending_pack_flush(nullptr)) return e;
    PendingPack* P = pending_pack_entry();
    if (!P) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_mu);
    P->A = A; P->stream = s; P->valid = true;
    return hipSuccess;
}

bool pending_pack_take(hipStream_t s, PackAll* out) {
    PendingPack* P = pending_pack_entry();
    if (!P) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!P->valid || P->stream != s) return false;
    *out = P->A;
    P->valid = false;
    return true;
}

int pending_slot_sums_flush() {
    PendingSums* P = pending_entry();
    if (!P) return hipErrorInvalidDevice;
    ReduceAll R;
    hipStream_t s;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!P->valid) return hipSuccess;
        R = P->R; s = P->stream; P->valid = false;
    }
    if (int e = launch_slot_sums(R, s)) return e;
    trace_mark("pinnsf_reduce", s);
    return launch_unfold(R, s);
}

int pending_slot_sums_leave(const ReduceAll& R, hipStream_t s) {
    PendingSums* P = pending_entry();
    if (!P) return hipErrorInvalidDevice;
    {
        // Sums of ANOTHER operator of the same backward pass waiting on this stream (the bottleneck variants: row decoder, then
        // encoders, piml_rowdecoder_bwd_acc / piml_encoder_bwd_acc with PIML_DEFER_SLOT_SUMS): the two descriptions become one
        // launch -- when they fit, accumulate alike, carry no unfold, and write DIFFERENT gradient buffers (the same buffer twice
        // is a second pass over the same parameters: its predecessor has to run first).
        std::lock_guard<std::mutex> lock(g_mu);
        ReduceAll& Q = P->R;
        bool merge = P->valid && P->stream == s && Q.accumulate == R.accumulate && Q.nunf == 0 && R.nunf == 0 &&
                     Q.nsets + R.nsets <= (int)(sizeof(Q.set) / sizeof(Q.set[0]));
        for (int i = 0; merge && i < Q.nsets; ++i)
            for (int j = 0; j < R.nsets; ++j)
                if (Q.set[i].grads == R.set[j].grads) { merge = false; break; }
        if (merge) {
            for (int j = 0; j < R.nsets; ++j) Q.set[Q.nsets++] = R.set[j];
            if (R.gx > Q.gx) Q.gx = R.gx;
            return hipSuccess;
        }
    }
    if (int e = pending_slot_sums_flush()) return e;          // sums already waiting: they run now, on the stream they were left on
    std::lock_guard<std::mutex> lock(g_mu);
    P->R = R; P->stream = s; P->valid = true;
    return hipSuccess;
}

bool pending_slot_sums_take(hipStream_t s, ReduceAll* out) {
    PendingSums* P = pending_entry();
    if (!P) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!P->valid || P->stream != s) return false;
    *out = P->R;
    P->valid = false;
    return true;
}

}  // namespace piml

using namespace piml;

PIML_API int piml_pinnsf_streams_init(void) {
    Side* S;
    return side_streams(&S);
}

// One launch on `stream`.  (Not forked internally: a fork nested inside a caller's forked stream crashed
// hipStreamEndCapture on ROCm 7.2, tools/probe_capture_fork.py.)
PIML_API int piml_pinnsf_pack(const piml_encoder_branch* enc, const piml_decoder_branch* dec, int nbr,
                              const piml_collision_head* head, int flags, void* stream) {
    if (!enc || !dec || nbr < 1 || nbr > 2) return hipErrorInvalidValue;
    PackAll A = {};
    A.nbr = nbr;
    A.has_head = head != nullptr;
    A.skip_f32 = enc_f32_images_needed() ? 0 : 1;
    for (int i = 0; i < nbr; ++i) {
        const piml_encoder_branch& e = enc[i];
        const piml_decoder_branch& d = dec[i];
        if (e.in_dim < 1 || e.in_dim > 8 || !e.w1 || !e.b1 || !e.w2 || !e.b2 || !e.w3 || !e.b3 || !e.packed || !d.w1 || !d.b1 ||
            !d.w2 || !d.b2 || !d.w3 || !d.b3 || !d.packed)
            return hipErrorInvalidValue;
        A.enc[i] = e;
        A.dec[i] = d;
    }
    if (head) {
        if (!head->w1 || !head->b1 || !head->w2 || !head->b2 || !head->packed) return hipErrorInvalidValue;
        A.head = *head;
        if (head->fold_w3) {
            if (!head->fold_b3) return hipErrorInvalidValue;
            A.has_fold = 1;
        }
    }
    for (int i = 0; i < nbr; ++i)
        if (dec[i].fold_w3) {
            if (!dec[i].fold_b3) return hipErrorInvalidValue;
            A.has_fold = 1;
        }
    if (flags & PIML_DEFER_PACK) return pending_pack_leave(A, as_stream(stream));      // the next relfeat forward on `stream` runs it
    return launch_pack(A, as_stream(stream));
}

PIML_API int piml_pinnsf_pack_flush(void) { return pending_pack_flush(nullptr); }

// every slot sum of the backward pass (encoder + decoder partials) in one launch on `s`
static int reduce_all(const piml_encoder_branch* enc, const piml_decoder_branch* dec, int nbr, hipStream_t s, bool accumulate, bool defer = false,
                      bool sums = false) {
    ReduceAll R = {};
    R.accumulate = accumulate ? 1 : 0;
    int w0 = 0, n = 0, maxl = 0;
    const int total = piml_encoder_workgroups(enc, nbr, &w0);
    const int dslots = piml_decoder_workgroups(dec[0].agents);
    int n0[2] = {0, 0}, n1[2] = {0, 0};
    const bool dw2 = !sums && enc_dw2_used(enc, nbr, n0, n1);
    // diagnostic builds (RESULTS WRONG ON PURPOSE): PIML_REDUCE_ENC_DIV / PIML_REDUCE_DEC_DIV = read only every n-th part of the
    // encoder / decoder slots -- what the reduce launch would cost behind a reduction of the slots inside the producing kernels
#ifndef PIML_REDUCE_ENC_DIV
#define PIML_REDUCE_ENC_DIV 1
#endif
#ifndef PIML_REDUCE_DEC_DIV
#define PIML_REDUCE_DEC_DIV 1
#endif
    int div = PIML_REDUCE_ENC_DIV;
    auto add = [&](const float* parts, float* grads, int slots, int lanes, int split, int off0, int off1) {
        slots = (slots + div - 1) / div;
        R.set[n++] = ReduceSet{parts, grads, slots, lanes, split, off0, off1};
        if (lanes > maxl) maxl = lanes;
    };
    for (int i = 0; i < nbr; ++i) {
        if (sums) {          // PIML_POOL_TRAIN: one layer-1 slot (dW2 | dW1 | db2 | db1) per workgroup; dW3 / db3 come from the unfold
            add(enc[i].partials, enc[i].grads, nbr == 1 ? total : (i == 0 ? w0 : total - w0), DW2_L1_LANES, DW2_L1_SPLIT, DW2_L1_OFF0, DW2_L1_OFF1);
            R.unf[i] = UnfoldSet{dec[i].grads, dec[i].w1, enc[i].w3, enc[i].b3, dec[i].fold_scale, enc[i].k, dec[i].dw1_out, enc[i].grads};
            R.nunf = nbr;
        } else if (dw2) {
            add(enc[i].partials, enc[i].grads, n0[i], DW2_L0_LANES, DW2_L0_SPLIT, 0, DW2_L0_OFF1);
            add(enc[i].partials + (size_t)n0[i] * (DW2_L0_LANES * 4), enc[i].grads, n1[i], DW2_L1_LANES, DW2_L1_SPLIT, DW2_L1_OFF0, DW2_L1_OFF1);
        } else {
            add(enc[i].partials, enc[i].grads, nbr == 1 ? total : (i == 0 ? w0 : total - w0), ENC_PART / 4, 0x7fffffff, 0, 0);
        }
    }
    div = PIML_REDUCE_DEC_DIV;
    for (int i = 0; i < nbr; ++i) add(dec[i].partials, dec[i].grads, dslots, DEC_PART / 4, 0x7fffffff, 0, 0);
    R.nsets = n;
    R.gx = (maxl + 15) / 16;
    if (defer) return pending_slot_sums_leave(R, s);          // the next piml_relfeat_self_bwd on `s` (or a flush) runs them
    if (int e = launch_slot_sums(R, s)) return e;
    trace_mark("pinnsf_reduce", s);
    return launch_unfold(R, s);
}

PIML_API int piml_pinnsf_slot_sums_flush(void) { return pending_slot_sums_flush(); }

PIML_API int piml_pinnsf_pool_h2_ok(const piml_encoder_branch* enc, int nbr) { return enc_pool_h2_ok(enc, nbr) ? 1 : 0; }
PIML_API int piml_pinnsf_pool_train_ok(const piml_encoder_branch* enc, int nbr) { return enc_pool_train_ok(enc, nbr) ? 1 : 0; }
PIML_API int piml_pinnsf_pool_msgs_ok(const piml_encoder_branch* enc, int nbr) { return enc_pool_msgs_ok(enc, nbr) ? 1 : 0; }

PIML_API int piml_pinnsf_fwd(const piml_encoder_branch* enc, const piml_decoder_branch* dec, int nbr,
                             const piml_collision_head* head, const float* self_features, float tau, float* acc,
                             int flags, void* stream) {
    hipStream_t m = as_stream(stream);
    const bool pack = !(flags & PIML_PACKED_VALID);
    PIML_TRY(pending_pack_flush(m));         // a deferred pack nobody took: now, in front of this call's launches (no-op otherwise)
    if (flags & PIML_POOL_TRAIN) {            // training on the agents' sums of h2 (see the header)
        if ((flags & (PIML_FORK | PIML_POOL_H2)) || !enc_pool_train_ok(enc, nbr)) return hipErrorInvalidValue;
        for (int i = 0; i < nbr; ++i)
            if (dec[i].pooled != enc[i].sum_a || dec[i].msgs != enc[i].sum_b) return hipErrorInvalidValue;
        if (pack) PIML_TRY(piml_pinnsf_pack(enc, dec, nbr, head, 0, stream));
        PIML_TRY(enc_stage_fwd_sum(enc, nbr, m, nbr > 1 ? acc : nullptr, nbr > 1 ? dec[0].agents * 2 : 0));
        trace_mark("enc_fwd_sum", m);
        PIML_TRY(dec_stage_fwd_sum(dec, nbr, head, self_features, tau, acc, m));
        trace_mark("dec_fwd_head_sum", m);
        return hipSuccess;
    }
    if (flags & PIML_POOL_MSGS) {             // the agents' sums of the messages from the encoder forward's registers (see the header)
        if ((flags & (PIML_FORK | PIML_POOL_H2)) || !enc_pool_msgs_ok(enc, nbr)) return hipErrorInvalidValue;
        for (int i = 0; i < nbr; ++i)
            if (dec[i].pooled != enc[i].sum_a || dec[i].msgs != enc[i].sum_b) return hipErrorInvalidValue;
        if (head && head->rows > 0 && head->msgs != enc[0].msgs) return hipErrorInvalidValue;      // the head reads the rows the forward stores
        if (pack) PIML_TRY(piml_pinnsf_pack(enc, dec, nbr, head, 0, stream));
        PIML_TRY(enc_stage_fwd(enc, nbr, m, nbr > 1 ? acc : nullptr, nbr > 1 ? dec[0].agents * 2 : 0, true));
        trace_mark("enc_fwd", m);
        PIML_TRY(dec_stage_fwd_sum(dec, nbr, head, self_features, tau, acc, m, false));
        trace_mark("dec_fwd_head_sum", m);
        return hipSuccess;
    }
    if (flags & PIML_POOL_H2) {               // inference: the agents' sums of h2 instead of the messages (see the header)
        if ((flags & PIML_FORK) || head || !enc_pool_h2_ok(enc, nbr)) return hipErrorInvalidValue;
        if (pack) PIML_TRY(piml_pinnsf_pack(enc, dec, nbr, nullptr, 0, stream));
        PIML_TRY(enc_stage_fwd_pool(enc, nbr, m, nbr > 1 ? acc : nullptr, nbr > 1 ? dec[0].agents * 2 : 0));
        trace_mark("enc_fwd", m);
        PIML_TRY(dec_stage_fwd_ph2(dec, nbr, self_features, tau, acc, m));
        trace_mark("dec_fwd_head", m);
        return hipSuccess;
    }
    if (!(flags & PIML_FORK)) {
        if (pack) PIML_TRY(piml_pinnsf_pack(enc, dec, nbr, head, 0, stream));
        // the split decoder tiles accumulate their two branches into `acc`: cleared by the encoder launch
        PIML_TRY(enc_stage_fwd(enc, nbr, m, nbr > 1 ? acc : nullptr, nbr > 1 ? dec[0].agents * 2 : 0));
        trace_mark("enc_fwd", m);
        PIML_TRY(dec_stage_fwd_fused(dec, nbr, head, self_features, tau, acc, m));
        trace_mark("dec_fwd_head", m);
        return hipSuccess;
    }
    Side* S;
    PIML_TRY(side_streams(&S));
    hipStream_t s0 = S->s[0], s1 = S->s[1];
    if (pack) {
        PIML_TRY(edge(m, s0, S->ev[0]));
        PIML_TRY(dec_stage_pack(dec, nbr, s0));
        if (head) PIML_TRY(head_stage_pack(head, s0));
        PIML_TRY(hipEventRecord(S->ev[1], s0));
        PIML_TRY(enc_stage_pack(enc, nbr, m));
    }
    PIML_TRY(enc_stage_fwd(enc, nbr, m));
    if (head) {
        PIML_TRY(edge(m, s1, S->ev[2]));
        if (pack) PIML_TRY(hipStreamWaitEvent(s1, S->ev[1], 0));
        PIML_TRY(head_stage_fwd(head, s1));
        PIML_TRY(hipEventRecord(S->ev[3], s1));
    }
    PIML_TRY(dec_stage_pool(dec, nbr, m));
    if (pack) PIML_TRY(hipStreamWaitEvent(m, S->ev[1], 0));
    PIML_TRY(dec_stage_fwd(dec, nbr, self_features, tau, acc, m));
    if (head) PIML_TRY(hipStreamWaitEvent(m, S->ev[3], 0));
    return hipSuccess;
}

PIML_API int piml_pinnsf_bwd(const piml_encoder_branch* enc, const piml_decoder_branch* dec, int nbr, const float* g_pred,
                             const float* self_features, float tau, float* g_self, int flags, void* stream) {
    hipStream_t m = as_stream(stream);
    if (flags & PIML_POOL_TRAIN) {            // backward of a PIML_POOL_TRAIN forward
        if (flags & PIML_FORK) return hipErrorInvalidValue;
        for (int i = 0; i < nbr; ++i)
            if (!dec[i].dw1_out || !dec[i].fold_w3 || dec[i].g_pooled != enc[i].g_pooled) return hipErrorInvalidValue;
        PIML_TRY(dec_stage_bwd_fused(dec, nbr, g_pred, self_features, tau, g_self, m, true));
        trace_mark("dec_bwd", m);
        PIML_TRY(enc_stage_bwd_sum(enc, nbr, m));
        trace_mark("enc_bwd_dx", m);
        return reduce_all(enc, dec, nbr, m, (flags & PIML_ACCUMULATE) != 0, (flags & PIML_DEFER_SLOT_SUMS) != 0, true);
    }
    if (!(flags & PIML_FORK)) {
        PIML_TRY(dec_stage_bwd_fused(dec, nbr, g_pred, self_features, tau, g_self, m));
        trace_mark("dec_bwd", m);
        PIML_TRY(enc_stage_bwd_dx(enc, nbr, m));
        trace_mark("enc_bwd_dx", m);
        PIML_TRY(enc_stage_bwd_dw(enc, nbr, m));
        trace_mark("enc_bwd_dw", m);
        PIML_TRY(reduce_all(enc, dec, nbr, m, (flags & PIML_ACCUMULATE) != 0, (flags & PIML_DEFER_SLOT_SUMS) != 0));
        return hipSuccess;
    }
    if (flags & (PIML_ACCUMULATE | PIML_DEFER_SLOT_SUMS)) return hipErrorInvalidValue;      // (the forked form sums its slots per stage: not offered there)
    PIML_TRY(dec_stage_bwd_dx(dec, nbr, g_pred, self_features, tau, g_self, m));
    Side* S;
    PIML_TRY(side_streams(&S));
    hipStream_t s0 = S->s[0];
    PIML_TRY(edge(m, s0, S->ev[4]));
    PIML_TRY(dec_stage_bwd_dw(dec, nbr, g_pred, true, s0));
    PIML_TRY(hipEventRecord(S->ev[5], s0));
    PIML_TRY(enc_stage_bwd_dx(enc, nbr, m));
    PIML_TRY(enc_stage_bwd_dw(enc, nbr, m));
    PIML_TRY(hipStreamWaitEvent(m, S->ev[5], 0));
    return enc_stage_reduce(enc, nbr, m);
}
