// Calibration for encoder_bwd3.hip: cycles per v_mfma_f32_32x32x16_bf16 issued by ONE wave per SIMD, as a function of what sits
// between two products: nothing, s_nop 1, n plain vector instructions, packed-f32 instructions, LDS reads; results in VGPRs
// (asm, weight operand in an AGPR) or AGPRs (builtin).  256 workgroups x 4 waves; prints cycles per product (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %d line %d\n", e, __LINE__); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define VF(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(kb), "v"(kc))
#define VPK(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(kb2), "v"(kc2))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters) {
    __shared__ u32x4 lds[1024];
    for (int t = threadIdx.x; t < 1024; t += 256) lds[t] = (u32x4){(unsigned)t, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 a = lds[lane], b = lds[64 + lane], w = lds[128 + lane];
    f32x16 acc, sm;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; sm[r] = 0.f; }
    float v0 = lane, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f;
    f2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f};
    u32x4 pend = lds[lane];
    u32x4 ring[4] = {pend, pend, pend, pend};
    float kb = 1.0001f, kc = 0.5f;
    f2 kb2 = {kb, kb}, kc2 = {kc, kc};
    asm volatile("" : "+v"(kb), "+v"(kc), "+v"(kb2), "+v"(kc2));
    asm volatile("" : "+a"(w));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (MODE == 9) {        // builtin, accumulators wherever hipcc puts them, 5 + 1 pattern
                if (j % 6 == 5) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
                else sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), sm, 0, 0, 0);
            } else if (MODE == 10 || MODE == 11) {   // asm, results in AGPRs
                if (j % 6 == 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(w));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(sm) : "v"(a), "a"(w));
            } else if (MODE == 1 || MODE == 8) {   // asm + s_nop 1
                if (j % 6 == 5) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w));
                else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sm) : "v"(a), "a"(w));
            } else {                // asm, VGPR results, AGPR weight operand
                if (j % 6 == 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sm) : "v"(a), "a"(w));
            }
            SB();
            if (MODE == 2 || MODE == 8 || MODE == 9 || MODE == 10) {   // 5 plain vector instructions
                VF(v0); VF(v1); VF(v2); VF(v3); VF(v4);
            } else if (MODE == 12) {  // 3 plain
                VF(v0); VF(v1); VF(v2);
            } else if (MODE == 13) {  // 6 plain
                VF(v0); VF(v1); VF(v2); VF(v3); VF(v4); VF(v5);
            } else if (MODE == 3 || MODE == 11) {  // 8 plain
                VF(v0); VF(v1); VF(v2); VF(v3); VF(v4); VF(v5); VF(v0); VF(v1);
            } else if (MODE == 4) {  // 2 packed f32 + 2 plain
                VPK(p0); VPK(p1); VF(v0); VF(v1);
            } else if (MODE == 5) {  // one ds_read_b128 + 3 plain
                v5 += __uint_as_float(pend[0] & 0x3f800000u);
                const u32x4 q = lds[(256 + ((i + j) & 7) * 64 + lane) & 1023];
                pend = q;                    // consumed one slot later
                VF(v0); VF(v1); VF(v2);
            } else if (MODE == 6) {  // cvt_pk + shl + and + 2 sub + cvt_pk (half a split3)
                const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){v0, v1}, __attribute__((ext_vector_type(2))) __bf16));
                v2 = v0 - __uint_as_float(hi << 16); v3 = v1 - __uint_as_float(hi & 0xffff0000u);
                v0 = v2 * kb; v1 = v3 * kb;
            } else if (MODE == 14) {  // a dependent chain of 6 (half a split3, opaque)
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v5) : "v"(v0), "v"(v1));
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(v2) : "v"(v5));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(v3) : "v"(v5));
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v0) : "v"(v2));
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v1) : "v"(v3));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v4) : "v"(v0), "v"(v1));
            } else if (MODE == 15 || MODE == 16) {  // LDS reads whose results are used six slots later + 4 plain
                if (j % 6 == 0) {
                    v5 += __uint_as_float(ring[0][0] & 0x3f800000u) + __uint_as_float(ring[1][1] & 0x3f800000u);
#pragma unroll
                    for (int q = 0; q < (MODE == 16 ? 4 : 2); ++q) ring[q] = lds[(256 + ((i + q) & 3) * 64 + lane) & 1023];
                }
                VF(v0); VF(v1); VF(v2); VF(v3);
            } else if (MODE == 7) {  // 12 plain
                VF(v0); VF(v1); VF(v2); VF(v3); VF(v4); VF(v5); VF(v0); VF(v1); VF(v2); VF(v3); VF(v4); VF(v5);
            }
            SB();
        }
    }
    if (MODE == 10 || MODE == 11) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc), "+a"(sm));
    else asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc), "+v"(sm));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float tot = v0 + v1 + v2 + v3 + v4 + v5 + p0.x + p0.y + p1.x + p1.y;
    for (int r = 0; r < 16; ++r) tot += acc[r] + sm[r];
    out[blockIdx.x * 256 + threadIdx.x] = tot;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name, float* out, unsigned long long* st) {
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, st, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), st, 256 * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto x : h) s += (double)x;
    printf("%-52s %7.1f cycles per product\n", name, s / 256 / (iters * 12.0));
    return 0;
}

int main() {
    float* out; unsigned long long* st;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&st, 256 * 8));
    run<0>("asm, nothing between", out, st);
    run<1>("asm, s_nop 1 in front", out, st);
    run<12>("asm, 3 v_fma between", out, st);
    run<2>("asm, 5 v_fma between", out, st);
    run<13>("asm, 6 v_fma between", out, st);
    run<3>("asm, 8 v_fma between", out, st);
    run<7>("asm, 12 v_fma between", out, st);
    run<4>("asm, 2 v_pk_fma + 2 v_fma between", out, st);
    run<5>("asm, ds_read_b128 (used a slot later) + 4 vector between", out, st);
    run<6>("asm, half a split3 between", out, st);
    run<14>("asm, dependent chain of 6 (half a split3) between", out, st);
    run<15>("asm, 4 v_fma; 2 ds_read_b128 per six, used six later", out, st);
    run<16>("asm, 4 v_fma; 4 ds_read_b128 per six, used six later", out, st);
    run<8>("asm, s_nop 1 + 5 v_fma between", out, st);
    run<9>("builtin, 5 v_fma between", out, st);
    run<10>("asm, results in AGPRs, 5 v_fma between", out, st);
    run<11>("asm, results in AGPRs, 8 v_fma between", out, st);
    return 0;
}
