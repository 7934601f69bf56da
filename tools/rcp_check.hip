/*
 * rcp_check.hip - what v_rcp_f32 returns on this GPU, for every binary32 significand, and the discriminator's division on it.
 *
 *   1. table: for B = 2^23 .. 2^24 - 1, by how many ulps v_rcp_f32((float)B) differs from the correctly rounded 1 / B
 *      (written as 2^23 signed bytes; tools/div_proof.c reads it and runs the hard cases of the division on the
 *      reciprocals the device really returns);
 *   2. the division itself on the device, both forms (residual steps behind q0: 1 and 2), against the IEEE quotient
 *      (__fdiv_rn): every A <= B for the N largest and N smallest significands B (the classical hard divisors are the
 *      ones next to a power of two), and 2^34 pseudo-random pairs.
 *
 *   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o rcp_check rcp_check.hip && ./rcp_check out_table.bin
 */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int STEPS>
__device__ __forceinline__ float quot(float a, float b)
{
#pragma clang fp contract(off)
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    float q = a * r1;
#pragma unroll
    for (int i = 0; i < STEPS; i++) {
        const float e = __builtin_fmaf(-b, q, a);
        q = __builtin_fmaf(e, r1, q);
    }
    return q;
}

__global__ void rcp_table(int8_t *dev)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float b = (float)((1u << 23) + i);
    const float r = __builtin_amdgcn_rcpf(b), rn = __fdiv_rn(1.0f, b);
    dev[i] = (int8_t)((int32_t)__float_as_uint(r) - (int32_t)__float_as_uint(rn));
}

/* every A in [2^23, 2^24) against divisor B (both orders of magnitude: A >= B as A / (2B)... the quotient's case is set by
 * the significands alone: a = A or A / 2 so that a <= b) */
__global__ void all_a(uint32_t B, unsigned long long *bad1, unsigned long long *bad2)
{
    const uint32_t A = (1u << 23) + blockIdx.x * blockDim.x + threadIdx.x;
    const float b = (float)B;
    const float a = A <= B ? (float)A : 0.5f * (float)A;
    const float want = __fdiv_rn(a, b);
    if (quot<1>(a, b) != want) {
        atomicAdd(bad1, 1ull);
    }
    if (quot<2>(a, b) != want) {
        atomicAdd(bad2, 1ull);
    }
}

__global__ void random_pairs(uint64_t seed, uint32_t per_thread, unsigned long long *bad1, unsigned long long *bad2)
{
    uint64_t x = seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x + 1) * 0x9E3779B97F4A7C15ull);
    unsigned long long b1 = 0, b2 = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        /* operands as the kernel sees them: |int32| converted to float, min over max */
        const float u = (float)(uint32_t)(x & 0x7fffffffu), v = (float)(uint32_t)((x >> 32) & 0x7fffffffu);
        const float a = fminf(u, v), b = fmaxf(u, v);
        if (b == 0.0f) {
            continue;
        }
        const float want = __fdiv_rn(a, b);
        b1 += quot<1>(a, b) != want;
        b2 += quot<2>(a, b) != want;
    }
    if (b1) {
        atomicAdd(bad1, b1);
    }
    if (b2) {
        atomicAdd(bad2, b2);
    }
}

int main(int argc, char **argv)
{
    int8_t *d_dev;
    unsigned long long *d_bad;
    CK(hipMalloc(&d_dev, 1u << 23));
    CK(hipMalloc(&d_bad, 16));
    rcp_table<<<(1u << 23) / 256, 256>>>(d_dev);
    std::vector<int8_t> dev(1u << 23);
    CK(hipMemcpy(dev.data(), d_dev, 1u << 23, hipMemcpyDeviceToHost));
    long hist[9] = { 0 };
    for (int8_t c : dev) {
        hist[c < -4 ? 0 : c > 4 ? 8 : c + 4]++;
    }
    printf("v_rcp_f32 against RN(1/B), all 2^23 significands: ");
    for (int c = -4; c <= 4; c++) {
        if (hist[c + 4]) {
            printf("%+d ulp: %ld  ", c, hist[c + 4]);
        }
    }
    printf("\nthe divisors next to 2^24: ");
    for (uint32_t B = (1u << 24) - 8; B < (1u << 24); B++) {
        printf("%u:%+d ", B, dev[B - (1u << 23)]);
    }
    printf("\n");
    if (argc > 1) {
        FILE *f = fopen(argv[1], "wb");
        fwrite(dev.data(), 1, dev.size(), f);
        fclose(f);
    }
    const uint32_t N = 2048;
    unsigned long long bad[2] = { 0, 0 }, tot[2] = { 0, 0 };
    for (int side = 0; side < 2; side++) {
        CK(hipMemset(d_bad, 0, 16));
        for (uint32_t k = 0; k < N; k++) {
            const uint32_t B = side ? (1u << 24) - 1 - k : (1u << 23) + k;
            all_a<<<(1u << 23) / 256, 256>>>(B, d_bad, d_bad + 1);
        }
        CK(hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost));
        printf("every significand A against the %u %s divisors: one residual step behind q0 wrong %llu times, two %llu times\n", N,
               side ? "largest" : "smallest", bad[0], bad[1]);
        tot[0] += bad[0], tot[1] += bad[1];
    }
    CK(hipMemset(d_bad, 0, 16));
    random_pairs<<<4096, 256>>>(20261002ull, 1u << 14, d_bad, d_bad + 1);
    CK(hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost));
    printf("2^34 random pairs of |int32| converted to float: one step wrong %llu times, two %llu times\n", bad[0], bad[1]);
    tot[0] += bad[0], tot[1] += bad[1];
    printf("total: one step %llu, two steps %llu\n", tot[0], tot[1]);
    return 0;
}
