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
    ReduceSet set[6];
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
};
constexpr int kPackMax = PACK_FLOATS > DEC_PACK ? (PACK_FLOATS > HEAD_PACK ? PACK_FLOATS : HEAD_PACK) : (DEC_PACK > HEAD_PACK ? DEC_PACK : HEAD_PACK);

// element e of image set y (encoder branches, decoder branches, head)
__device__ __forceinline__ void pack_element(const PackAll& A, int y, int e) {
    if (y < A.nbr) {
        const piml_encoder_branch& J = A.enc[y];
        if (e < PACK_FLOATS) J.packed[e] = pack_value(J, e);
    } else if (y < 2 * A.nbr) {
        const piml_decoder_branch& J = A.dec[y - A.nbr];
        if (e < DEC_PACK_PLAIN) J.packed[e] = dec_pack_value(J, e);
    } else if (e < HEAD_PACK_PLAIN) {
        A.head.packed[e] = head_pack_value(A.head, e);
    }
}
__host__ __device__ inline int pack_sets(const PackAll& A) { return 2 * A.nbr + (A.has_head ? 1 : 0); }
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
// the pack as trailing workgroups of another launch: `threads` threads per workgroup, workgroup `bid` of pack_blocks(...)
struct PackWork {
    PackAll A;
    int first_block;          // blockIdx.x of the first pack workgroup; < 0: no pack rides in this launch
};
__host__ __device__ inline int pack_blocks_per_set(int threads) { return (kPackMax + threads - 1) / threads; }
__host__ __device__ inline int pack_blocks_total(const PackAll& A, int threads) {
    return pack_blocks_per_set(threads) * pack_sets(A) + fold_blocks(A, threads);
}
// red: `threads` doubles of LDS (the folded images' partial sums); the whole workgroup calls this
__device__ __forceinline__ void pack_block(const PackAll& A, int bid, int threads, double* red) {
    const int per = pack_blocks_per_set(threads), plain = per * pack_sets(A);
    if (bid < plain) {
        const int y = bid / per, x = bid - y * per;
        pack_element(A, y, x * threads + (int)threadIdx.x);
    } else if (threads >= 512) {
        fold_block<8>(A, bid - plain, threads, red);
    } else {
        fold_block<4>(A, bid - plain, threads, red);
    }
}
int launch_pack(const PackAll& A, hipStream_t s);
int pending_pack_leave(const PackAll& A, hipStream_t s);
bool pending_pack_take(hipStream_t s, PackAll* out);
int pending_pack_flush();

int launch_slot_sums(const ReduceAll& R, hipStream_t s);          // the stand-alone launch (pinnsf_reduce_kernel)
// deferred sums of the current device: leave (a second deferral first launches the one already waiting, on ITS stream),
// take (true: *out holds sums deferred on stream s, the entry is cleared), flush (launch what is waiting, if anything)
int pending_slot_sums_leave(const ReduceAll& R, hipStream_t s);
bool pending_slot_sums_take(hipStream_t s, ReduceAll* out);
int pending_slot_sums_flush();

}  // namespace piml
