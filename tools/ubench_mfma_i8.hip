// ubench_mfma_i8.hip - semantics + rate probe of v_mfma_i32_32x32x32_i8 on gfx950, as used by the
// exact int16 FIR-as-GEMM kernel: (1) A/B k-slot pairing, (2) C/D layout, (3) int32 wrap-around
// (no saturation), (4) issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void k_sem(const int8_t *A, const int8_t *B, const int *Cin, int *Dout)
{
    const int l = threadIdx.x, g = l >> 5, i = l & 31;
    v4i a, b;
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; j++) { ab[j] = A[i * 32 + 16 * g + j]; bb[j] = B[(16 * g + j) * 32 + i]; }
    __builtin_memcpy(&a, ab, 16); __builtin_memcpy(&b, bb, 16);
    v16i c;
    for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * g; c[r] = Cin[row * 32 + i]; }
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * g; Dout[row * 32 + i] = c[r]; }
}

__global__ __launch_bounds__(256) void k_rate(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
    v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; it++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    int s = 0;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    std::vector<int8_t> A(1024), B(1024); std::vector<int> C(1024), D(1024), R(1024);
    srand(1);
    for (int t = 0; t < 1024; t++) { A[t] = (int8_t)(rand() & 255); B[t] = (int8_t)(rand() & 255); C[t] = (t & 1) ? 0x7fffff00 : -0x7fffff00 + t; }
    A[0] = -128; B[0] = -128; A[33] = -128; B[1] = 127;
    for (int i = 0; i < 32; i++) for (int n = 0; n < 32; n++) { uint32_t s = (uint32_t)C[i * 32 + n]; for (int k = 0; k < 32; k++) s += (uint32_t)((int)A[i * 32 + k] * (int)B[k * 32 + n]); R[i * 32 + n] = (int)s; }
    int8_t *dA, *dB; int *dC, *dD;
    CHECK(hipMalloc(&dA, 1024)); CHECK(hipMalloc(&dB, 1024)); CHECK(hipMalloc(&dC, 4096)); CHECK(hipMalloc(&dD, 256 * 8 * 256 * 4));
    CHECK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    CHECK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int t = 0; t < 1024; t++) bad += D[t] != R[t];
    printf("mfma_i32_32x32x32_i8 semantics (k-slot pairing, C/D layout, int32 wrap): %d mismatches of 1024\n", bad);
    for (int bpc : {1, 2}) {
        const int iters = 4000, grid = 256 * bpc;
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_rate, dim3(grid), dim3(256), 0, 0, dD, 10);
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_rate, dim3(grid), dim3(256), 0, 0, dD, iters);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        double macs = (double)grid * 4 * iters * 4 * 32768.0;
        printf("rate blocks/CU=%d: %.1f T int8-MAC/s = %.1f TOPS  (= %.1f T int16-MAC/s via 4 products; dot2 VALU measured ~73)\n",
               bpc, macs / (ms * 1e-3) / 1e12, 2 * macs / (ms * 1e-3) / 1e12, macs / 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
