// Back-to-back v_mfma_i32_16x16x64_i8 with different accumulator reuse patterns (one wave per SIMD): does a FIR step
// written with three accumulators (hh, md, ll, md) stall on the md dependency?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_dep.hip -o tools/ubench_mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int mode> __global__ __launch_bounds__(256) void k(int iters, int *out)
{
    v4i a = { (int)threadIdx.x, 2, 3, 4 }, b = { 5, 6, 7, (int)blockIdx.x };
    v4i c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        if (mode == 0) { // 4 independent accumulators
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        } else if (mode == 1) { // hh, md, ll, md
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        } else if (mode == 2) { // fully dependent chain
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        } else { // two accumulators alternating
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
static void launch(int mode, int iters, int *out)
{
    switch (mode) {
    case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, iters, out); break;
    case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, iters, out); break;
    case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, iters, out); break;
    default: hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, iters, out); break;
    }
}

int main()
{
    int *out; (void)hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 4; mode++) {
        launch(mode, iters, out); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        launch(mode, iters, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.3f ms -> %.2f ns per MFMA (one wave per SIMD)\n", mode, ms, ms * 1e6 / iters / 4);
    }
    return 0;
}
