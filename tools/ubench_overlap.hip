// Do v_mfma_i32_16x16x64_i8 and ordinary VALU instructions from DIFFERENT waves of one SIMD overlap on gfx950?
// 512-thread workgroups = 2 waves per SIMD.  mode 1: waves 0-3 run an MFMA loop, waves 4-7 idle; mode 2: waves 4-7 run
// a VALU loop, waves 0-3 idle; mode 3: both.  If mode 3 takes max(mode 1, mode 2) the two pipes overlap; if it takes the
// sum they share the issue port.   hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap.hip -o tools/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mode, int iters, int *out)
{
    const int wave = threadIdx.x >> 6;
    v4i a = { (int)threadIdx.x, 2, 3, 4 }, b = { 5, 6, 7, (int)blockIdx.x };
    v4i c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
    float f0 = threadIdx.x, f1 = 1.5f, f2 = 2.5f, f3 = 3.5f;
    if (wave < 4) {
        if (mode & 1) {
            for (int i = 0; i < iters; i++) {
                c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
            }
        }
    } else {
        if (mode & 2) {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 4; u++) { // 16 independent-ish FMAs = 64 issue cycles, like 4 MFMAs of 16 cycles
                    f0 = __builtin_fmaf(f0, 1.0001f, 0.5f);
                    f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
                    f2 = __builtin_fmaf(f2, 1.0001f, 0.5f);
                    f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
                }
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + (int)(f0 + f1 + f2 + f3);
}

int main()
{
    int *out;
    hipMalloc(&out, 2048 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 1; mode <= 3; mode++) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out); // one workgroup per CU
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.3f ms  -> %.1f ns per loop trip\n", mode, mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both", ms,
               ms * 1e6 / iters);
    }
    return 0;
}
