// Keep-mask bits of the PINNSF processor's train-mode dropout (reference: ResDNN.forward = Dropout_p(2 x),
// src/models/model.py:82-119 with SURVEY quirk Q3; model.train() at src/models/simulators.py:311, --dropout 0.5 at
// src/main.py:45).  The fused encoder kernels apply the mask in their epilogue / at the head of their backward chain
// (encoder_x3.hip, encoder.hip, mlpglue.hip: scale_ksum); the split-product forward kernels draw the p = 0.5 mask
// themselves (one Philox call per row); this file is the stand-alone generator for every other case.  Stream: philox.hpp.
#include "common.hpp"
#include "philox.hpp"
#include "stages.hpp"
#include "../../include/piml_hip.h"

namespace piml {

struct DropJob {
    unsigned* bits;
    long long rows;
    unsigned stream;
};
struct DropArgs {
    DropJob job[2];
    int njobs;
    long long n0;            // threads of job 0
    u64* state;
    int words, cols;
    unsigned thresh16;       // keep iff 16 random bits >= thresh16 (0 .. 65536)
    int fair;                // p == 0.5: one call per 128 features
};

// one thread per (row, word of 32 features)
__global__ __launch_bounds__(256) void dropout_keep_bits_kernel(DropArgs A) {
    const u64 seed = A.state[0], off = A.state[1];
    long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const int j = (A.njobs > 1 && id >= A.n0) ? 1 : 0;
    if (j) id -= A.n0;
    const DropJob J = A.job[j];
    if (id < J.rows * A.words) {
        const unsigned row = (unsigned)(id / A.words), w = (unsigned)(id % A.words);
        unsigned m = 0;
        if (A.fair) {
            const PhiloxOut r = philox4x32_10((unsigned)off, (unsigned)(off >> 32), row, (J.stream << 16) | (0xFFFFu - (w >> 2)),
                                              (unsigned)seed, (unsigned)(seed >> 32));
            const unsigned q = w & 3;
            m = q == 0 ? r.x : (q == 1 ? r.y : (q == 2 ? r.z : r.w));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {                        // call i: features 32 w + 8 i .. + 7
                const PhiloxOut r = philox4x32_10((unsigned)off, (unsigned)(off >> 32), row, (J.stream << 16) | (w * 4 + i),
                                                  (unsigned)seed, (unsigned)(seed >> 32));
                const unsigned v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    m |= ((v[u] & 0xFFFFu) >= A.thresh16 ? 1u : 0u) << (8 * i + 2 * u);
                    m |= ((v[u] >> 16) >= A.thresh16 ? 1u : 0u) << (8 * i + 2 * u + 1);
                }
            }
        }
        const int left = A.cols - 32 * (int)w;                  // features past `cols` are never kept
        if (left < 32) m &= (1u << left) - 1u;
        J.bits[id] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) dropout_advance(A.state, off, gridDim.x);
}

// keep-masks of up to two branches in ONE launch (streams 0 and 1 of the same draw)
int dropout_stage(u64* state, const long long* rows, unsigned* const* bits, const unsigned* streams, int njobs, int cols, float p,
                  hipStream_t s) {
    if (!state || njobs < 1 || njobs > 2 || cols <= 0 || !(p >= 0.f && p <= 1.f)) return hipErrorInvalidValue;
    DropArgs A = {};
    A.njobs = njobs;
    A.state = state;
    A.cols = cols;
    A.words = (cols + 31) / 32;
    A.fair = p == kFairP;
    const double t = (double)p * 65536.0;
    A.thresh16 = (unsigned)(t + 0.5);
    long long total = 0;
    for (int i = 0; i < njobs; ++i) {
        if (rows[i] <= 0 || rows[i] >= (1ll << 32) || !bits[i]) return hipErrorInvalidValue;
        A.job[i] = DropJob{bits[i], rows[i], streams[i]};
        if (i == 0) A.n0 = rows[i] * A.words;
        total += rows[i] * A.words;
    }
    if (njobs == 1) A.job[1] = A.job[0];
    hipLaunchKernelGGL(dropout_keep_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A);
    return hipGetLastError();
}

}  // namespace piml

using namespace piml;

PIML_API int piml_dropout_keep_bits(unsigned long long* state, long long rows, int cols, float p, int stream_id, unsigned* keep_bits,
                                    void* stream) {
    if (rows < 0 || cols <= 0 || !(p >= 0.f && p <= 1.f) || rows >= (1ll << 32) || stream_id < 0 || stream_id > 0xFFFF)
        return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!state || !keep_bits) return hipErrorInvalidValue;
    const unsigned sid = (unsigned)stream_id;
    return dropout_stage(reinterpret_cast<u64*>(state), &rows, &keep_bits, &sid, 1, cols, p, as_stream(stream));
}
