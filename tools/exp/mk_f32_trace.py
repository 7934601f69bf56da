"""tools/exp/f32_trace.hip = csrc/mfm_f32.hip + time stamps in the persistent kernel (two workgroups in detail, start / end of all)"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "tsl-sdr_amd/csrc/mfm_f32.hip")).read()
def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)
rep('#include "../../include/multifm_hip.h"\n#include "mfm_taps.h"', '#include "../../include/multifm_hip.h"\n#include "mfm_taps.h"\n__device__ unsigned long long g_f32_trace[2 * 8 * 64 * 8];\n__device__ unsigned long long g_f32_span[1024 * 2];\n#define TR(slot) do { if ((blockIdx.x == 100u || blockIdx.x == 101u) && lane == 0 && it < 64) g_f32_trace[(((blockIdx.x - 100u) * 8 + wave) * 64 + it) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)')
rep("    uint32_t buf = 0;\n    /* one tile: NG column groups", "    uint32_t buf = 0, it = 0;\n    if (tid == 0 && blockIdx.x < 1024) g_f32_span[2 * blockIdx.x] = __builtin_amdgcn_s_memtime();\n    /* one tile: NG column groups")
rep("            const bool more = ntile < t_end;\n", "            const bool more = ntile < t_end;\n            TR(0);\n")
rep("            stage_load(more ? ntile : tile, more ? nckk : ck);\n", "            stage_load(more ? ntile : tile, more ? nckk : ck);\n            TR(1);\n")
rep("            if (more) {\n                stage_store(buf ^ 1u);\n            }\n            __syncthreads();\n            buf ^= 1u;\n            if (ck + 1u < nck) {\n                continue;\n            }",
    "            TR(2);\n            if (more) {\n                stage_store(buf ^ 1u);\n            }\n            TR(3);\n            __syncthreads();\n            TR(4);\n            buf ^= 1u;\n            if (ck + 1u < nck) {\n                it++;\n                continue;\n            }")
marker = "                        if (rel == (int)L.n_new - 1) {\n                            L.prev_out[ch] = make_float2(o_re[g][j], o_im[g][j]);\n                        }\n                    }\n                }\n            }\n"
rep(marker, marker + "            TR(5);\n            it++;\n")
old2 = "        default:\n            do_tile(std::integral_constant<uint32_t, 4>{}, t_whole);\n            break;\n        }\n    }\n}\n"
rep(old2, old2[:-2] + "    if (tid == 0 && blockIdx.x < 1024) g_f32_span[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();\n}\n")
s += '''
extern "C" int mfm_f32_debug_trace(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f32_trace), sizeof(g_f32_trace)) == hipSuccess ? 0 : -1;
}
extern "C" int mfm_f32_debug_span(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f32_span), sizeof(g_f32_span)) == hipSuccess ? 0 : -1;
}
'''
open(os.path.join(ROOT, "tools/exp/f32_trace.hip"), "w").write(s)
