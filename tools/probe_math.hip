// Checks on the GPU that the float32 primitives the kernels rely on are correctly rounded.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* x, const float* y, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o[i] = __fsqrt_rn(x[i]);
    o[n + i] = sqrtf(x[i]);
    o[2 * n + i] = __fdiv_rn(x[i], y[i]);
    o[3 * n + i] = x[i] / y[i];
    o[4 * n + i] = __fmaf_rn(y[i], y[i], __fmul_rn(x[i], x[i]));
    o[5 * n + i] = __fsqrt_rn(__fmaf_rn(y[i], y[i], __fmul_rn(x[i], x[i])));
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), y(n), o(6 * n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        x[i] = (float)rand() / RAND_MAX * (i % 3 == 0 ? 1e-3f : 30.f) + (i % 7 == 0 ? 1e-30f : 0.f);
        y[i] = ((float)rand() / RAND_MAX - 0.5f) * 20.f;
        if (i % 1000 == 0) x[i] = 1e-41f * (i % 97);   // denormals
    }
    float *dx, *dy, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, 6 * n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dy, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 4, hipMemcpyDeviceToHost);
    long bad[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        float r[6] = {sqrtf(x[i]), sqrtf(x[i]), x[i] / y[i], x[i] / y[i], fmaf(y[i], y[i], x[i] * x[i]),
                      sqrtf(fmaf(y[i], y[i], x[i] * x[i]))};
        for (int q = 0; q < 6; ++q) {
            float g = o[q * n + i];
            if (!(g == r[q] || (g != g && r[q] != r[q]))) {
                if (bad[q]++ < 3) printf("q=%d x=%a y=%a gpu=%a cpu=%a\n", q, x[i], y[i], g, r[q]);
            }
        }
    }
    printf("mismatches: __fsqrt_rn %ld sqrtf %ld __fdiv_rn %ld div %ld fma %ld norm2 %ld of %d\n", bad[0], bad[1],
           bad[2], bad[3], bad[4], bad[5], n);
    return 0;
}
