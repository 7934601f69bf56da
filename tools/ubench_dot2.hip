// ubench_dot2.hip - instruction-rate micro-benchmarks that set the VALU roofline for the multifm kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_dot2 ubench_dot2.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short s2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template<int MODE>
__global__ __launch_bounds__(256) void k_rate(const unsigned *coef, int *out, int iters)
{
    int acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = i + threadIdx.x;
    unsigned x0 = threadIdx.x * 2654435761u, x1 = x0 ^ 0x9e3779b9u;
    const __attribute__((address_space(4))) unsigned *cp = (const __attribute__((address_space(4))) unsigned *)coef;
    unsigned c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = cp[i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (MODE == 0) {        // v_dot2c_i32_i16 with SGPR coefficient
                    acc[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, c[i & 7]), __builtin_bit_cast(s2, (i & 1) ? x1 : x0), acc[i], false);
                } else if (MODE == 1) { // v_dot2c_i32_i16 VGPR only
                    acc[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, x0), __builtin_bit_cast(s2, x1), acc[i], false);
                } else if (MODE == 2) { // v_mad_i32_i24
                    acc[i] = __mul24((int)x0, (int)x1) + acc[i];
                } else if (MODE == 3) { // v_dot4_i32_i8
                    acc[i] = __builtin_amdgcn_sdot4((int)x0, (int)x1, acc[i], false);
                } else if (MODE == 4) { // v_fma_f32
                    acc[i] = __float_as_int(__builtin_fmaf(__int_as_float(x0), __int_as_float(x1), __int_as_float(acc[i])));
                } else if (MODE == 5) { // v_add_u32
                    acc[i] = acc[i] + (int)x0;
                }
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// scalar-load bandwidth: every wave streams `bytes` of a table through s_load_dwordx16
__global__ __launch_bounds__(256) void k_smem(const unsigned *tbl, int *out, int nchunks, int stride_chunks)
{
    const __attribute__((address_space(4))) unsigned *cp = (const __attribute__((address_space(4))) unsigned *)tbl;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned s = 0;
    unsigned base = ((blockIdx.x * 4 + wave) * 7919u) % (unsigned)stride_chunks;
    for (int ch = 0; ch < nchunks; ch++) {
        const unsigned o = ((base + ch) % (unsigned)stride_chunks) * 32;
#pragma unroll
        for (int i = 0; i < 32; i++) s += cp[o + i];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (int)s;
}

template<int MODE> double run_rate(int *d_out, unsigned *d_coef, const char *name, int blocks_per_cu)
{
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, d_coef, d_out, 10);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, d_coef, d_out, iters);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)grid * 256 * iters * 128;
    double rate = ops / (ms * 1e-3) / 1e12;
    printf("%-34s blocks/CU=%d  %.2f T lane-ops/s  (%.1f%% of 78.6)\n", name, blocks_per_cu, rate, rate / 78.6432 * 100);
    return rate;
}

int main()
{
    int *d_out; unsigned *d_coef;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    CHECK(hipMalloc(&d_coef, 64 << 20));
    CHECK(hipMemset(d_coef, 1, 64 << 20));
    for (int bpc : {1, 2, 4, 8}) {
        run_rate<0>(d_out, d_coef, "v_dot2c_i32_i16 (sgpr coef)", bpc);
        run_rate<1>(d_out, d_coef, "v_dot2c_i32_i16 (vgpr)", bpc);
    }
    run_rate<2>(d_out, d_coef, "v_mul_i32_i24+add", 4);
    run_rate<3>(d_out, d_coef, "v_dot4_i32_i8", 4);
    run_rate<4>(d_out, d_coef, "v_fma_f32", 4);
    run_rate<5>(d_out, d_coef, "v_add_u32", 4);
    // SMEM streaming: table sizes 8 KB (K$ resident), 64 KB, 1 MB
    for (int kb : {8, 64, 1024}) {
        const int stride_chunks = kb * 1024 / 128, nchunks = 4096, grid = 256 * 6;
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_smem, dim3(grid), dim3(256), 0, 0, d_coef, d_out, 16, stride_chunks);
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_smem, dim3(grid), dim3(256), 0, 0, d_coef, d_out, nchunks, stride_chunks);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        double bytes = (double)grid * 4 * nchunks * 128;
        printf("s_load stream, table %4d KB: %.1f GB/s chip, %.2f B/clk/CU @2.4GHz\n", kb, bytes / (ms * 1e-3) / 1e9,
               bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
