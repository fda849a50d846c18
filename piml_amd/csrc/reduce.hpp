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

int launch_slot_sums(const ReduceAll& R, hipStream_t s);          // the stand-alone launch (pinnsf_reduce_kernel)
// deferred sums of the current device: leave (a second deferral first launches the one already waiting, on ITS stream),
// take (true: *out holds sums deferred on stream s, the entry is cleared), flush (launch what is waiting, if anything)
int pending_slot_sums_leave(const ReduceAll& R, hipStream_t s);
bool pending_slot_sums_take(hipStream_t s, ReduceAll* out);
int pending_slot_sums_flush();

}  // namespace piml
