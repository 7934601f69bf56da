// Where do the workgroups of a 512 x 512-thread launch with ~79 KB of LDS each land (two per CU), and which SIMD does each
// of their waves run on?  hipcc --offload-arch=gfx950 -O2 -o tools/exp/hwid_probe tools/exp/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <map>

__global__ __launch_bounds__(512, 4) void probe(uint32_t *out, int spin)
{
    extern __shared__ uint8_t smem[];
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    smem[threadIdx.x] = (uint8_t)hw;
    // stay resident long enough for every workgroup of the grid to have been placed
    uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin) {
        __builtin_amdgcn_s_sleep(8);
    }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + wave) * 2 + 0] = hw;
        out[(blockIdx.x * 8 + wave) * 2 + 1] = xcc;
    }
}

int main(int argc, char **argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 512;
    const int lds = argc > 2 ? atoi(argv[2]) : 80691;
    uint32_t *d;
    hipMalloc(&d, grid * 8 * 2 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(d, 0, grid * 8 * 2 * 4);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(512), lds, 0, d, 20000 /* 200 us */);
        hipDeviceSynchronize();
        std::vector<uint32_t> h(grid * 8 * 2);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        // per CU: the workgroups on it, their TG_ID, and per wave SIMD / wave slot
        std::map<uint32_t, std::vector<int>> cu;
        for (int b = 0; b < grid; b++) {
            const uint32_t hw = h[b * 16], xcc = h[b * 16 + 1] & 15u;
            const uint32_t key = (xcc << 16) | (((hw >> 13) & 7u) << 8) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
            cu[key].push_back(b);
        }
        printf("rep %d: %zu CUs used\n", rep, cu.size());
        int shown = 0, same_simd_pattern = 0, tg_parity_differs = 0, two = 0;
        for (auto &kv : cu) {
            if (kv.second.size() == 2) {
                two++;
                const int a = kv.second[0], b = kv.second[1];
                bool same = true;
                for (int w = 0; w < 8; w++) {
                    same = same && (((h[(a * 8 + w) * 2] >> 4) & 3u) == ((h[(b * 8 + w) * 2] >> 4) & 3u));
                }
                same_simd_pattern += same;
                tg_parity_differs += (((h[a * 16] >> 16) & 1u) != ((h[b * 16] >> 16) & 1u));
            }
            if (shown < 6) {
                shown++;
                printf(" cu %05x:", kv.first);
                for (int b : kv.second) {
                    printf("  wg %3d tg %u simd", b, (h[b * 16] >> 16) & 15u);
                    for (int w = 0; w < 8; w++) {
                        printf(" %u", (h[(b * 8 + w) * 2] >> 4) & 3u);
                    }
                    printf(" slot");
                    for (int w = 0; w < 8; w++) {
                        printf(" %u", h[(b * 8 + w) * 2] & 15u);
                    }
                }
                printf("\n");
            }
        }
        printf(" CUs with two workgroups: %d; same wave->SIMD pattern in both: %d; TG_ID parity differs: %d\n", two, same_simd_pattern,
               tg_parity_differs);
        // wave -> SIMD pattern histogram
        std::map<uint32_t, int> pat;
        for (int b = 0; b < grid; b++) {
            uint32_t p = 0;
            for (int w = 0; w < 8; w++) {
                p = p * 4 + ((h[(b * 8 + w) * 2] >> 4) & 3u);
            }
            pat[p]++;
        }
        for (auto &kv : pat) {
            printf(" pattern");
            for (int w = 7; w >= 0; w--) {
                printf(" %u", (kv.first >> (2 * w)) & 3u);
            }
            printf(": %d workgroups\n", kv.second);
        }
    }
    return 0;
}
