// Same question as ubench_overlap.hip for the 32x32x32 int8 MFMA (twice the passes per instruction), and for VALU work
// issued by the SAME wave between MFMAs.  mode 1: waves 0-3 MFMA only; 2: waves 4-7 VALU only; 3: both;
// 5: waves 0-3 run MFMA and independent VALU interleaved in one instruction stream (waves 4-7 idle).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap32.hip -o tools/ubench_overlap32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define VALU16()                                                                                             \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                          \
        f0 = __builtin_fmaf(f0, 1.0001f, 0.5f);                                                              \
        f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);                                                              \
        f2 = __builtin_fmaf(f2, 1.0001f, 0.5f);                                                              \
        f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);                                                              \
    }

__global__ __launch_bounds__(512) void k(int mode, int iters, int *out)
{
    const int wave = threadIdx.x >> 6;
    v4i a = { (int)threadIdx.x, 2, 3, 4 }, b = { 5, 6, 7, (int)blockIdx.x };
    v16i c0 = {}, c1 = {};
    float f0 = threadIdx.x, f1 = 1.5f, f2 = 2.5f, f3 = 3.5f;
    if (wave < 4) {
        if (mode == 5) {
            for (int i = 0; i < iters; i++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                VALU16();
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                VALU16();
            }
        } else if (mode & 1) {
            for (int i = 0; i < iters; i++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            }
        }
    } else {
        if (mode & 2) {
            for (int i = 0; i < iters; i++) {
                VALU16();
                VALU16();
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 16; i++) {
        s += c0[i] + c1[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s + (int)(f0 + f1 + f2 + f3);
}

int main()
{
    int *out;
    hipMalloc(&out, 2048 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    const int modes[4] = { 1, 2, 3, 5 };
    for (int mi = 0; mi < 4; mi++) {
        const int mode = modes[mi];
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out); // one workgroup per CU
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.3f ms  -> %.1f ns per loop trip (2 x 32x32x32 i8 MFMA and/or 32 v_fma_f32)\n", mode,
               mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : mode == 3 ? "both, different waves" : "both, one wave", ms,
               ms * 1e6 / iters);
    }
    return 0;
}
