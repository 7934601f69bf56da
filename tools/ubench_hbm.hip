// What HBM bandwidth does a plain streaming kernel get on this box for the channel kernel's traffic mix
// (268 MB read, 90 MB written per 2^26-sample block)?   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o tools/ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
__global__ __launch_bounds__(256) void rd(const uint4 *x, size_t n, uint4 *sink)
{
    uint4 acc = { 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = x[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void wr(uint4 *y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        y[i] = make_uint4((unsigned)i, 1, 2, 3);
    }
}
__global__ __launch_bounds__(256) void rw(const uint4 *x, size_t n, uint4 *y) // 3 reads : 1 write
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 3; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
        y[i] = make_uint4(a.x ^ b.x ^ c.x, a.y ^ b.y ^ c.y, a.z ^ b.z ^ c.z, a.w ^ b.w ^ c.w);
    }
}
/* ---- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE (`ubench_hbm calib` under --pmc): kernels that move a KNOWN number
 * of bytes in the channel kernel's own patterns.  pcm8: 64 channel rows of `n_out` int16, written as the second-generation
 * kernel writes them - a lane stores 8 bytes (4 consecutive outputs), 16 lanes cover 128 contiguous bytes of one row, a tile is
 * 64 outputs of 64 channels - plain and with the non-temporal hint.  rd16: 16 bytes per lane, streaming.  Every kernel
 * prints its byte count; the counters of the same kernels are in the profile. ---- */
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(512) void pcm8(uint8_t *pcm, uint32_t out_stride, uint32_t ntiles)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, kg = lane >> 4, n = lane & 15u;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t ch = wave * 8u + 2u * kg + c;
            v2u w = { t, lane };
            v2u *dst = reinterpret_cast<v2u *>(pcm + ((size_t)ch * out_stride + 64u * t + 4u * n) * 2u);
            if (NT) {
                __builtin_nontemporal_store(w, dst);
            } else {
                *dst = w;
            }
        }
    }
}
static int calib()
{
    const uint32_t ntiles = 10923, out_stride = 699072; /* the headline's launch: 64 channels x 699 050 outputs */
    const size_t wbytes = (size_t)64 * out_stride * 2, rbytes = 268435456;
    uint8_t *pcm; uint4 *x, *sink;
    (void)hipMalloc(&pcm, wbytes); (void)hipMalloc(&x, rbytes); (void)hipMalloc(&sink, 64);
    (void)hipMemset(x, 1, rbytes);
    for (int rep = 0; rep < 6; rep++) {
        hipLaunchKernelGGL(pcm8<false>, dim3(512), dim3(512), 0, 0, pcm, out_stride, ntiles);
        hipLaunchKernelGGL(pcm8<true>, dim3(512), dim3(512), 0, 0, pcm, out_stride, ntiles);
        hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, x, rbytes / 16, sink);
        hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint4 *>(pcm), wbytes / 16);
    }
    (void)hipDeviceSynchronize();
    printf("calib: pcm8<plain> and pcm8<nt> write %zu bytes each (%u tiles x 64 channels x 128 B), rd reads %zu bytes, wr writes %zu bytes\n",
           (size_t)ntiles * 64 * 128, ntiles, rbytes, wbytes / 16 * 16);
    return 0;
}
int main(int argc, char **argv)
{
    if (argc > 1 && 0 == strcmp(argv[1], "calib")) {
        return calib();
    }
    const size_t rbytes = 268435456, wbytes = 89478485 / 16 * 16;
    uint4 *x, *y;
    (void)hipMalloc(&x, rbytes); (void)hipMalloc(&y, rbytes);
    (void)hipMemset(x, 1, rbytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 30; rep++) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, x, rbytes / 16, y);
            if (mode == 1) hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, y, wbytes / 16);
            if (mode == 2) hipLaunchKernelGGL(rw, dim3(4096), dim3(256), 0, 0, x, rbytes / 16, y);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 10 && ms < best) best = ms;
        }
        const double bytes = mode == 0 ? (double)rbytes : mode == 1 ? (double)wbytes : (double)rbytes * 4 / 3;
        printf("%s: %.1f us -> %.2f TB/s\n", mode == 0 ? "read 268 MB" : mode == 1 ? "write 89 MB" : "read 268 MB + write 89 MB", best * 1e3,
               bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
