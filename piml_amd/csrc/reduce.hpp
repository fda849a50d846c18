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
struct ReduceAll {
    ReduceSet set[6];
    int nsets;
    int accumulate;       // PIML_ACCUMULATE: grads += the sums
    int gx;               // workgroups per set (the widest set's (lanes + 15) / 16)
};

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
};
constexpr int kPackMax = PACK_FLOATS > DEC_PACK ? (PACK_FLOATS > HEAD_PACK ? PACK_FLOATS : HEAD_PACK) : (DEC_PACK > HEAD_PACK ? DEC_PACK : HEAD_PACK);

// element e of image set y (encoder branches, decoder branches, head)
__device__ __forceinline__ void pack_element(const PackAll& A, int y, int e) {
    if (y < A.nbr) {
        const piml_encoder_branch& J = A.enc[y];
        if (e < PACK_FLOATS) J.packed[e] = pack_value(J, e);
    } else if (y < 2 * A.nbr) {
        const piml_decoder_branch& J = A.dec[y - A.nbr];
        if (e < DEC_PACK) J.packed[e] = dec_pack_value(J, e);
    } else if (e < HEAD_PACK) {
        A.head.packed[e] = head_pack_value(A.head.w1, A.head.b1, A.head.w2, A.head.b2, e);
    }
}
// the pack as trailing workgroups of another launch: `threads` threads per workgroup, workgroup `bid` of pack_blocks(...)
struct PackWork {
    PackAll A;
    int first_block;          // blockIdx.x of the first pack workgroup; < 0: no pack rides in this launch
};
__host__ __device__ inline int pack_blocks_per_set(int threads) { return (kPackMax + threads - 1) / threads; }
__device__ __forceinline__ void pack_block(const PackAll& A, int bid, int threads) {
    const int per = pack_blocks_per_set(threads);
    const int y = bid / per, x = bid - y * per;
    pack_element(A, y, x * threads + (int)threadIdx.x);
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
