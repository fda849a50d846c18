// One Adam step of every parameter of a group in ONE launch, the step counters included (round 6).
//
// Reference: `torch.optim.Adam(self.model.parameters(), lr, weight_decay=...)` of BaseSimulator.set_optimizer / finetune
// (src/models/simulators.py:69-71, :104-129) stepped at :358-359 and :319 -- on the device that is `_foreach_add_(steps, 1)` + the
// multi-tensor fused kernel (5 + 13 us per step at PINNSF's 22 parameters of 130 k elements: launch-bound, and a tenth of a
// pointwise pre-training step).  Here a tensor is cut into 1024-element pieces, one workgroup each; every workgroup reads the
// tensor's step counter, works with counter + 1, and the LAST one of the tensor to finish (a ticket per tensor) writes the new
// counter back -- by then every workgroup of the tensor has read the old one.
// The arithmetic is that of PyTorch's fused kernel, operation for operation and precision for precision (hyper-parameters are
// doubles there, so the moment updates and the weight decay are evaluated in double -- contracted to fma by PyTorch's build, written
// out as fma here -- and rounded to float once; the bias corrections come from pow(double, double)): tests/test_losses_gpu.py holds
// it BITWISE against torch.optim.Adam(fused=True, capturable=True).
#include "common.hpp"
#include "../../include/piml_hip.h"

#include <cmath>

namespace piml {

constexpr int ADAM_MAX = 40;        // tensors per launch (kernel arguments by value)
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_PIECE = 1024;    // elements per workgroup

struct AdamArgs {
    float* param[ADAM_MAX];
    const float* grad[ADAM_MAX];
    float* exp_avg[ADAM_MAX];
    float* exp_avg_sq[ADAM_MAX];
    float* step[ADAM_MAX];
    int n[ADAM_MAX];
    int first[ADAM_MAX + 1];        // first workgroup of tensor t (first[cnt] = the grid)
    unsigned* tickets;              // one per tensor of this launch, zero between launches
    int cnt;
    double lr, beta1, beta2, weight_decay, eps;
};

__global__ __launch_bounds__(ADAM_THREADS) void adam_step_kernel(const AdamArgs A) {
    // the tables are read where they lie, in the kernel argument segment, with a run-time index (scalar loads; indexing the by-value
    // copy `A` at run time would make the compiler move all 2.6 KB of it to scratch first)
    const AdamArgs* K = (const AdamArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    int t = 0;
#pragma unroll
    for (int q = 1; q < ADAM_MAX; ++q) t = (q < A.cnt && (int)blockIdx.x >= A.first[q]) ? q : t;
    t = __builtin_amdgcn_readfirstlane(t);
    float* __restrict__ p = K->param[t];
    const float* __restrict__ g = K->grad[t];
    float* __restrict__ m = K->exp_avg[t];
    float* __restrict__ v = K->exp_avg_sq[t];
    float* sp = K->step[t];
    const int n = K->n[t], first = K->first[t], next = K->first[t + 1];
    // the piece's elements are requested first: the counter's round trip and the two pow() run under them
    constexpr int U = ADAM_PIECE / ADAM_THREADS;
    const int e0 = ((int)blockIdx.x - first) * ADAM_PIECE + (int)threadIdx.x;
    float pv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = e0 + u * ADAM_THREADS, ec = e < n ? e : 0;
        const bool ok = e < n;
        pv[u] = ok ? p[ec] : 0.f; gv[u] = ok ? g[ec] : 0.f; mv[u] = ok ? m[ec] : 0.f; vv[u] = ok ? v[ec] : 0.f;
    }
    const float step = *sp + 1.f;                                            // _foreach_add_(state_steps, 1)
    const double bc1d = 1.0 - pow(A.beta1, (double)step);
    const double bc2d = 1.0 - pow(A.beta2, (double)step);
    const float bias_correction1 = (float)bc1d, bias_correction2_sqrt = (float)sqrt(bc2d);
    const double lr = A.lr, beta1 = A.beta1, beta2 = A.beta2, wd = A.weight_decay, eps = A.eps;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = e0 + u * ADAM_THREADS;
        if (e < n) {
            float param = pv[u], grad = gv[u], exp_avg = mv[u], exp_avg_sq = vv[u];
            // (PyTorch's build contracts a * b + c * d into fma(a, b, c * d): the double results differ in the last place, which
            // decides the float rounding of exact ties -- 0.1 - 0.5 % of the elements; written out here, this build contracts nothing)
            if (wd != 0) grad = (float)fma((double)param, wd, (double)grad);
            exp_avg = (float)fma(beta1, (double)exp_avg, (1 - beta1) * grad);
            exp_avg_sq = (float)fma(beta2, (double)exp_avg_sq, (1 - beta2) * grad * grad);
            const float step_size = (float)(lr / bias_correction1);
            const float denom = (float)((sqrtf(exp_avg_sq) / bias_correction2_sqrt) + eps);
            param -= step_size * exp_avg / denom;
            p[e] = param; m[e] = exp_avg; v[e] = exp_avg_sq;
        }
    }
    // the tensor's last workgroup out writes the counter: every workgroup has read it by the time it takes its ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        // (no fence: nothing a workgroup WROTE is read by another -- the ticket only orders every workgroup's READ of the counter,
        // which has returned by now, in front of the last one's write; an agent-scope release would write the L2 back, ~8 us here)
        const unsigned done = atomicAdd(A.tickets + t, 1u) + 1u;
        if (done == (unsigned)(next - first)) {
            *sp = step;
            A.tickets[t] = 0u;
        }
    }
}

}  // namespace piml

using namespace piml;

PIML_API int piml_adam_tickets(void) { return ADAM_MAX; }

PIML_API int piml_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                            float* const* steps, const long long* sizes, int n, double lr, double beta1, double beta2, double weight_decay,
                            double eps, unsigned* tickets, void* stream) {
    if (n < 0 || (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !sizes || !tickets))) return hipErrorInvalidValue;
    for (int t0 = 0; t0 < n; t0 += ADAM_MAX) {
        const int cnt = n - t0 < ADAM_MAX ? n - t0 : ADAM_MAX;
        AdamArgs A = {};
        int blocks = 0;
        for (int t = 0; t < cnt; ++t) {
            if (!params[t0 + t] || !grads[t0 + t] || !exp_avg[t0 + t] || !exp_avg_sq[t0 + t] || !steps[t0 + t] || sizes[t0 + t] < 0 ||
                sizes[t0 + t] >= (1ll << 30))
                return hipErrorInvalidValue;
            A.param[t] = params[t0 + t]; A.grad[t] = grads[t0 + t]; A.exp_avg[t] = exp_avg[t0 + t]; A.exp_avg_sq[t] = exp_avg_sq[t0 + t];
            A.step[t] = steps[t0 + t]; A.n[t] = (int)sizes[t0 + t];
            A.first[t] = blocks;
            const int b = (int)((sizes[t0 + t] + ADAM_PIECE - 1) / ADAM_PIECE);
            blocks += b < 1 ? 1 : b;                              // (an empty tensor still steps its counter)
        }
        A.first[cnt] = blocks;
        A.tickets = tickets; A.cnt = cnt;
        A.lr = lr; A.beta1 = beta1; A.beta2 = beta2; A.weight_decay = weight_decay; A.eps = eps;
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, as_stream(stream), A);
    }
    return hipGetLastError();
}
