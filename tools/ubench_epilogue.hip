// ubench_epilogue.hip - VALU cost of the per-(channel, output) epilogue of the multifm kernels in isolation
// (registers + LDS LUT only, no HBM traffic): recombine -> r14 -> derotate -> r14 -> q*conj(p) -> discriminate.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../tsl-sdr_amd/csrc -o ubench_epilogue ubench_epilogue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "mfm_numerics.h"
typedef short s2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
static __device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c){ return __builtin_amdgcn_sdot2(__builtin_bit_cast(s2,a), __builtin_bit_cast(s2,b), c, false); }
static __device__ __forceinline__ uint32_t rp(uint32_t re_b, uint32_t im_b){ return ((re_b >> 14) & 0xffffu) | ((im_b << 2) & 0xffff0000u); }

template<int MODE> // 0 = full epilogue, 1 = without discriminate, 2 = discriminate only
__global__ __launch_bounds__(256) void k_epi(const float2 *lut_g, int *out, int iters)
{
    __shared__ float2 lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    const uint32_t rvx = 0x40003fffu ^ (threadIdx.x << 3), rvy = 0x3fff4000u ^ threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t hh = x >> 20, md = (x >> 7) & 0xffffu, ll = x & 0xfffffu;
            uint32_t q, s_re, s_im;
            if (MODE != 2) {
                const uint32_t a_re = (((hh << 8) + md) << 8) + ll, a_im = (((md << 8) + hh) << 8) + (ll ^ 0x5555u);
                const uint32_t f = rp(a_re, a_im);
                q = rp((uint32_t)dot2(f, rvx, 8192), (uint32_t)dot2(f, rvy, 8192));
                const uint32_t p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x111, 0xf, 0xf, false);
                s_re = (uint32_t)dot2(q, p, 0);
                int a, b;
                asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"(a) : "v"(q), "v"(p));
                asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[0,1,0,0]" : "=v"(b) : "v"(q), "v"(p));
                s_im = (uint32_t)a - (uint32_t)b;
            } else {
                s_re = x; s_im = x * 747796405u;
            }
            if (MODE != 1) acc += (uint32_t)mfm_discriminate((int)s_re, (int)s_im, lut);
            else acc += s_re ^ s_im;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)acc;
}

template<int MODE> void run(const float2 *lut, int *out, const char *name, int bpc)
{
    const int iters = 2000, grid = 256 * bpc;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_epi<MODE>, dim3(grid), dim3(256), 0, 0, lut, out, 10);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_epi<MODE>, dim3(grid), dim3(256), 0, 0, lut, out, iters);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double pairs = (double)grid * 256 * iters * 4;           // lane-pairs
    double simd_cycles = ms * 1e-3 * 2.1e9 * 1024;            // at ~2.1 GHz
    printf("%-28s blocks/CU=%d: %.1f G pairs/s, %.0f SIMD-cycles per wave-pair (64 lanes) @2.1GHz\n", name, bpc,
           pairs / (ms * 1e-3) / 1e9, simd_cycles / (pairs / 64));
}

int main()
{
    float2 h[256];
    for (int i = 0; i < 256; i++) { h[i].x = atanf(i / 255.f); h[i].y = atanf((i + 1 > 255 ? 255 : i + 1) / 255.f) - h[i].x; }
    float2 *lut; int *out;
    CHECK(hipMalloc(&lut, sizeof(h))); CHECK(hipMemcpy(lut, h, sizeof(h), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int bpc : {4, 8}) {
        run<0>(lut, out, "full epilogue", bpc);
        run<1>(lut, out, "without discriminate", bpc);
        run<2>(lut, out, "discriminate only", bpc);
    }
    return 0;
}
