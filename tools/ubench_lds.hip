// ds_read_b128 throughput for the B-fragment address pattern of mfm_kernel_mfma.hip (lane = 16*kg + n reads 16 bytes at
// n * rs + 16 * kg) against a contiguous pattern, for several row strides rs.  One workgroup of 512 threads per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o tools/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int rs, int mode, int iters, int *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lane = threadIdx.x & 63, kg = lane >> 4, n = lane & 15;
    for (unsigned i = threadIdx.x; i < 16384; i += 512) {
        reinterpret_cast<int *>(smem)[i] = i;
    }
    __syncthreads();
    unsigned addr = mode == 0 ? lane * 16u : (mode == 1 ? n * (unsigned)rs + 16u * kg : (n * (unsigned)rs + 64u * kg));
    v4i acc = { 0, 0, 0, 0 };
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const v4i v = *reinterpret_cast<const v4i *>(smem + ((addr + u * 256u) & 0xfff0u));
            acc += v;
        }
        addr += 16;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main()
{
    int *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000;
    const int cfgs[][2] = { { 0, 0 }, { 208, 1 }, { 224, 1 }, { 192, 1 }, { 80, 1 }, { 272, 1 }, { 208, 2 }, { 144, 1 }, { 176, 1 } };
    for (auto &c : cfgs) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, c[0], c[1], iters, out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, c[0], c[1], iters, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * 8 * iters * 8 * 1024; // CUs x waves x iters x reads x bytes per wave-read
        printf("mode %d rs %3d: %.3f ms -> %.1f B/clk/CU at 2.4 GHz (%.2f TB/s aggregate)\n", c[1], c[0], ms,
               bytes / 256 / (ms * 1e-3) / 2.4e9, bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
