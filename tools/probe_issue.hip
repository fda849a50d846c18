// Calibration: VALU issue rate (cycles per wave64 instruction per SIMD) for v_fma_f32,
// v_pk_fma_f32, v_cmp and ds_read_b64/b128, and the shader clock under sustained load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %d line %d\n", e, __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* stamps, int iters) {
    __shared__ float4 lds[4096];
    for (int t = threadIdx.x; t < 4096; t += blockDim.x) lds[t] = make_float4(t, 1, 2, 3);
    __syncthreads();
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    const float b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long cnt = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent v_fma_f32
            a0 = __fmaf_rn(a0, b, c); a1 = __fmaf_rn(a1, b, c); a2 = __fmaf_rn(a2, b, c); a3 = __fmaf_rn(a3, b, c);
            a4 = __fmaf_rn(a4, b, c); a5 = __fmaf_rn(a5, b, c); a6 = __fmaf_rn(a6, b, c); a7 = __fmaf_rn(a7, b, c);
        } else if (MODE == 1) {   // 4 v_pk_fma_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 x0 = {a0, a1}, x1 = {a2, a3}, x2 = {a4, a5}, x3 = {a6, a7}, bb = {b, b}, cc = {c, c};
            x0 = __builtin_elementwise_fma(x0, bb, cc); x1 = __builtin_elementwise_fma(x1, bb, cc);
            x2 = __builtin_elementwise_fma(x2, bb, cc); x3 = __builtin_elementwise_fma(x3, bb, cc);
            a0 = x0.x; a1 = x0.y; a2 = x1.x; a3 = x1.y; a4 = x2.x; a5 = x2.y; a6 = x3.x; a7 = x3.y;
        } else if (MODE == 2) {   // 4 ds_read_b64 (lane-consecutive)
            const float2* l2 = reinterpret_cast<const float2*>(lds);
            int base = (i * 256 + (threadIdx.x & 63)) & 8191;
            float2 q0 = l2[base & 8191], q1 = l2[(base + 64) & 8191], q2 = l2[(base + 128) & 8191], q3 = l2[(base + 192) & 8191];
            a0 += q0.x; a1 += q0.y; a2 += q1.x; a3 += q1.y; a4 += q2.x; a5 += q2.y; a6 += q3.x; a7 += q3.y;
        } else if (MODE == 3) {   // 2 ds_read_b128
            int base = (i * 128 + (threadIdx.x & 63)) & 4095;
            float4 q0 = lds[base & 4095], q1 = lds[(base + 64) & 4095];
            a0 += q0.x; a1 += q0.y; a2 += q0.z; a3 += q0.w; a4 += q1.x; a5 += q1.y; a6 += q1.z; a7 += q1.w;
        } else {   // 8 v_cmp + ballot accumulate
            cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(a0 + i < a1)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(a2 + i < a3));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)cnt;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
int run(const char* name, int ops_per_iter, int block) {
    const int blocks = 256 * (2048 / block), iters = 20000;
    float* out; unsigned long long* st;
    CK(hipMalloc(&out, (size_t)blocks * block * 4)); CK(hipMalloc(&st, blocks * 16));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), 0, 0, out, st, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), 0, 0, out, st, iters);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= blocks; rt /= blocks;
    double ghz = cyc / (rt * 10.0);   // realtime ticks are 10 ns
    int waves_per_simd = (2048 / 64) / 4;   // full occupancy: 32 waves/CU
    double cyc_per_instr_per_simd = cyc / ((double)iters * ops_per_iter * waves_per_simd);
    printf("%-14s block=%4d: %.3f ms, clock %.2f GHz, %.2f cycles per wave-instruction per SIMD (8 waves/SIMD)\n", name, block, ms, ghz, cyc_per_instr_per_simd);
    hipFree(out); hipFree(st); return 0;
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("v_fma_f32", 8, 1024); run<1>("v_pk_fma_f32", 4, 1024); run<2>("ds_read_b64", 4, 1024);
        run<3>("ds_read_b128", 2, 1024); run<4>("v_cmp+ballot", 2, 1024);
    }
    return 0;
}
