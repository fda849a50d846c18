// The slot sums of a backward pass (one launch for every partial set) and their DEFERRED form: a piml_pinnsf_bwd called with
// PIML_DEFER_SLOT_SUMS leaves the description of its sums here instead of launching them, and the next piml_relfeat_self_bwd
// on the same stream runs them as the leading workgroups of ITS launch (relfeat.hip): the two kernels are independent (the
// sums read the weight-gradient slots, the relfeat backward the feature gradients), each is small next to the chip, and a
// launch boundary on gfx950 costs ~4.5 us.  Private to libpiml_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "pack.hpp"

namespace piml {

struct ReduceSet {
    const float* parts;
    float* grads;
    int slots, lanes, split, off0, off1;       // float4 geometry: sum_slots_16x16 (pack.hpp)
};
// PIML_POOL_TRAIN: behind the slot sums, the gradient of a decoder's FOLDED first layer W1' = s W1 W3, b1' = b1 + k s W1 b3
// (G = d/d(W1'), g_b = d/d(b1'), both in the decoder's summed `grads`) is unfolded into the gradients of its factors:
//     dW1 = s (G W3^T + k g_b b3^T)   -> dw1_out (64, 128)
//     dW3 = s W1^T G, db3 = s k W1^T g_b   -> the dW3 / db3 fields of the encoder's `grads`
struct UnfoldSet {
    const float* dgrads;   // decoder `grads` (DEC_PART floats)
    const float *w1, *w3, *b3;
    float scale;
    int k;
    float* dw1_out;
    float* egrads;         // encoder `grads` (ENC_PART layout)
};
struct ReduceAll {
#ifndef PIML_REDUCE_SETS
#define PIML_REDUCE_SETS 8       // (the bottleneck variants leave 4 encoder + 2 row decoder + 1 head sets in one backward pass)
#endif
    ReduceSet set[PIML_REDUCE_SETS];
    int nsets;
    int accumulate;       // PIML_ACCUMULATE: grads += the sums
    int gx;               // workgroups per set (the widest set's (lanes + 15) / 16)
    UnfoldSet unf[2];
    int nunf;             // > 0: launch_unfold behind the sums
};
constexpr int kUnfoldBlocks = 64 + 64 + 1;       // per set: rows of dW1 | row pairs of dW3 | db3
int launch_unfold(const ReduceAll& R, hipStream_t s);

// workgroup `bid` of gx * nsets: set bid / gx, column block bid % gx
__device__ __forceinline__ void reduce_block(const ReduceAll& A, int bid) {
    const int y = bid / A.gx, x = bid - y * A.gx;
    const ReduceSet S = A.set[y];
    if (x * 16 < S.lanes) sum_slots_16x16_at(x, S.parts, S.grads, S.slots, S.lanes, S.split, S.off0, S.off1, A.accumulate != 0);
}

// ---- the weight pack of the network (every operand image in one launch, network.hip) and its deferred form: a
// piml_pinnsf_pack called with PIML_DEFER_PACK leaves its description here, and the next relfeat FORWARD launch on the same
// stream runs it as its trailing workgroups (the pack depends on the weights only; alone it is a ~4 us launch in front of
// the step's chain).  Every consumer of packed images (piml_pinnsf_fwd with PIML_PACKED_VALID, piml_encoder_fwd_packed,
// piml_rowdecoder_fwd_packed) first launches a pack that is still waiting. ----
struct PackAll {
    piml_encoder_branch enc[2];
    piml_decoder_branch dec[2];
    piml_collision_head head;
    int nbr, has_head;
    int has_fold;         // a decoder branch or the head carries fold_w3: nbr + has_head fold sets ride behind the image sets
    int skip_f32;         // the encoders' f32-instruction fragment images (4 x 16384 floats per branch) are not written: every
                          // kernel the library would launch reads the split-product images (piml_encoder_products(1), the default)
};
// live elements of an encoder set: with skip_f32 the two f32 fragment blocks [0, 32768) and [PACK_FWD, PACK_FWD + 32768) are left out
constexpr int ENC_LIVE_X3 = PACK_FLOATS - 2 * 32768;
__host__ __device__ inline int enc_set_elems(const PackAll& A) { return A.skip_f32 ? ENC_LIVE_X3 : PACK_FLOATS; }
__host__ __device__ inline int pack_elems_total(const PackAll& A) {
    return A.nbr * (enc_set_elems(A) + DEC_PACK_PLAIN) + (A.has_head ? HEAD_PACK_PLAIN : 0);
}

// element t of the concatenated image sets (encoder branches, decoder branches, head)
__device__ __forceinline__ void pack_flat(const PackAll& A, int t) {
    const int ne = enc_set_elems(A);
    for (int y = 0; y < A.nbr; ++y) {
        if (t < ne) {
            int e = t;
            if (A.skip_f32) e = t < PACK_FWD - 32768 ? t + 32768 : t + 2 * 32768;      // [32768, PACK_FWD) | [PACK_FWD + 32768, PACK_FLOATS)
            const piml_encoder_branch& J = A.enc[y];
            J.packed[e] = pack_value(J, e);
            return;
        }
        t -= ne;
    }
    for (int y = 0; y < A.nbr; ++y) {
        if (t < DEC_PACK_PLAIN) {
            const piml_decoder_branch& J = A.dec[y];
            J.packed[t] = dec_pack_value(J, t);
            return;
        }
        t -= DEC_PACK_PLAIN;
    }
    if (A.has_head && t < HEAD_PACK_PLAIN) A.head.packed[t] = head_pack_value(A.head, t);
}
__host__ __device__ inline int fold_sets(const PackAll& A) { return A.has_fold ? A.nbr + (A.has_head ? 1 : 0) : 0; }
// the folded images (pack.hpp): a workgroup of `threads` threads takes (threads / 64) / CH groups, CH = 8 waves per group (4 for
// 256 threads)
__host__ __device__ inline int fold_groups_per_block(int threads) { return threads >= 512 ? threads / 512 : 1; }
__host__ __device__ inline int fold_blocks(const PackAll& A, int threads) {
    const int g = fold_groups_per_block(threads);
    return (fold_sets(A) * FOLD_GROUPS + g - 1) / g;
}
// workgroup fb of fold_blocks(A, threads); red: threads doubles of LDS.  EVERY thread of the workgroup must call it (barrier).
template <int CH>
__device__ __forceinline__ void fold_block(const PackAll& A, int fb, int threads, double* red) {
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int grp = fb * ((threads >> 6) / CH) + wave / CH, chunk = wave % CH;
    const int f = grp / FOLD_GROUPS, r = grp - f * FOLD_GROUPS, i = r / 3, hc = r - 3 * i;
    const bool live = f < fold_sets(A);
    const bool is_head = f >= A.nbr;
    const piml_decoder_branch& J = A.dec[is_head ? 0 : f];
    const float* w1 = is_head ? A.head.w1 : J.w1;
    const float* w3 = is_head ? A.head.fold_w3 : J.fold_w3;
    const float* b3 = is_head ? A.head.fold_b3 : J.fold_b3;
    const bool have = live && w3 != nullptr;
    double part = 0.0;
    if (have && (hc < 2 || lane == 0))
        part = fold_partial<CH>(w1 + (size_t)i * DH, hc < 2 ? w3 + 64 * hc + lane : b3, hc < 2 ? EH : 1, chunk);
    red[threadIdx.x] = part;
    __syncthreads();
    if (have && chunk == 0) {
        double sum = 0.0;
#pragma unroll
        for (int c = 0; c < CH; ++c) sum += red[threadIdx.x + 64 * c];
        if (is_head) head_fold_store(A.head, i, hc, lane, sum);
        else dec_fold_store(J, i, hc, lane, sum);
    }
}
// the pack as trailing workgroups of another launch: `threads` threads per workgroup, kPackPerThread elements per thread (a
// thread's elements are `threads` apart: coalesced stores, independent gathers in flight together; fewer, fatter workgroups
// -- 1 100 workgroups of one element per thread cost the relfeat forward 4 us of tail), then the fold workgroups
constexpr int kPackPerThread = 4;
struct PackWork {
    PackAll A;
    int first_block;          // blockIdx.x of the first pack workgroup; < 0: no pack rides in this launch
};
__host__ __device__ inline int pack_plain_blocks(const PackAll& A, int threads) {
    return (pack_elems_total(A) + threads * kPackPerThread - 1) / (threads * kPackPerThread);
}
__host__ __device__ inline int pack_blocks_total(const PackAll& A, int threads) { return pack_plain_blocks(A, threads) + fold_blocks(A, threads); }
// red: `threads` doubles of LDS (the folded images' partial sums); the whole workgroup calls this
__device__ __forceinline__ void pack_block(const PackAll& A, int bid, int threads, double* red) {
    const int plain = pack_plain_blocks(A, threads);
    if (bid < plain) {
        const int total = pack_elems_total(A);
#pragma unroll
        for (int j = 0; j < kPackPerThread; ++j) {
            const int t = (bid * kPackPerThread + j) * threads + (int)threadIdx.x;
            if (t < total) pack_flat(A, t);
        }
    } else if (threads >= 512) {
        fold_block<8>(A, bid - plain, threads, red);
    } else {
        fold_block<4>(A, bid - plain, threads, red);
    }
}
int launch_pack(const PackAll& A, hipStream_t s);
int pending_pack_leave(const PackAll& A, hipStream_t s);
bool pending_pack_take(hipStream_t s, PackAll* out);
int pending_pack_flush(hipStream_t consumer);

int launch_slot_sums(const ReduceAll& R, hipStream_t s);          // the stand-alone launch (pinnsf_reduce_kernel)
// deferred sums of the current device: leave (a second deferral first launches the one already waiting, on ITS stream),
// take (true: *out holds sums deferred on stream s, the entry is cleared), flush (launch what is waiting, if anything)
int pending_slot_sums_leave(const ReduceAll& R, hipStream_t s);
bool pending_slot_sums_take(hipStream_t s, ReduceAll* out);
int pending_slot_sums_flush();

}  // namespace piml
