// Which XCD does workgroup b of a launch run on?  (round 6, review item 8: an XCD-local slot sum needs every workgroup of a
// partial-sum group on ONE XCD.)  Each workgroup records HW_REG_XCC_ID and its start clock; the host prints the map for a grid of
// 256 and 512 workgroups, with and without 160 KB of LDS per workgroup (one workgroup per CU), and repeats it under a HIP graph.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O2 tools/probe_xcc.hip -o /tmp/probe_xcc && /tmp/probe_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void where(int* xcc, unsigned long long* t0) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
        t0[blockIdx.x] = __builtin_amdgcn_s_memtime();
        lds[0] = 1.f;
    }
    // some work so that the workgroups overlap in time
    float a = threadIdx.x;
    for (int i = 0; i < 4000; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.f) xcc[0] = -1;
}

static void run(int grid, int ldsb, bool graph) {
    int* d; unsigned long long* t;
    hipMalloc(&d, grid * sizeof(int)); hipMalloc(&t, grid * sizeof(unsigned long long));
    hipFuncSetAttribute(reinterpret_cast<const void*>(where), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t s; hipStreamCreate(&s);
    if (graph) {
        hipGraph_t g; hipGraphExec_t e;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        hipLaunchKernelGGL(where, dim3(grid), dim3(256), ldsb, s, d, t);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
        for (int i = 0; i < 3; ++i) hipGraphLaunch(e, s);
    } else {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(where, dim3(grid), dim3(256), ldsb, s, d, t);
    }
    hipStreamSynchronize(s);
    std::vector<int> h(grid);
    hipMemcpy(h.data(), d, grid * sizeof(int), hipMemcpyDeviceToHost);
    int rr = 0, cnt[16] = {0};
    for (int b = 0; b < grid; ++b) { rr += h[b] == (b % 8); cnt[h[b] & 15]++; }
    printf("grid %4d, LDS %6d B, %s: workgroup b on XCD b %% 8 for %d of %d; per XCD:", grid, ldsb, graph ? "graph " : "stream", rr, grid);
    for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
    printf("; first 16:");
    for (int b = 0; b < 16 && b < grid; ++b) printf(" %d", h[b]);
    printf("\n");
    hipFree(d); hipFree(t); hipStreamDestroy(s);
}

int main() {
    for (int graph = 0; graph < 2; ++graph)
        for (int grid : {256, 512, 250})
            for (int ldsb : {0, 160 * 1024}) run(grid, ldsb, graph != 0);
    return 0;
}
