"""tools/exp/f32_knock.hip = csrc/mfm_f32.hip with knock-outs in the persistent kernel selected by -DX=<bits>:
1 no global input loads (constants), 2 no PCM stores, 4 no epilogue arithmetic (stores of raw accumulators),
8 no matrix instructions, 16 no A-fragment reloads, 32 no B reads (constant fragments)"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "tsl-sdr_amd/csrc/mfm_f32.hip")).read()
def rep(old, new, count=1):
    global s
    assert old in s, old[:70]
    s = s.replace(old, new, count)
s = "#ifndef X\n#define X 0\n#endif\n" + s
# 1: input loads
rep("            for (uint32_t j = 0; j < P_NLD; j++) {\n                r[j] = p[j * step];\n            }",
    "            for (uint32_t j = 0; j < P_NLD; j++) {\n                if (X & 1) { r[j] = make_float2((float)(tid + j), 1.0f); } else\n                r[j] = p[j * step];\n            }")
# 2: stores (plain path)
rep("                        if (g > 0 || n != 0u) {\n                            pf[16 * (int)g] = pcm;\n                            pi[16 * (int)g] = (int16_t)pcm; /* truncation, as multifm/fm_demod.c:72 */\n                        }",
    "                        if (X & 2) { if (pcm == 12345.678f) pf[0] = pcm; } else\n                        if (g > 0 || n != 0u) {\n                            pf[16 * (int)g] = pcm;\n                            pi[16 * (int)g] = (int16_t)pcm; /* truncation, as multifm/fm_demod.c:72 */\n                        }")
# 4: epilogue arithmetic
rep("                    const float pcm = p_fast_atan2f(s_im, s_re, lut_s) * (16384.0f / 3.14159265358979f);",
    "                    const float pcm = (X & 4) ? s_re + s_im : p_fast_atan2f(s_im, s_re, lut_s) * (16384.0f / 3.14159265358979f);")
# 8: MFMAs
rep("    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b.x, macc[g], 0, 0, 0);\n    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b.y, macc[g], 0, 0, 0);\n    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b.z, macc[g], 0, 0, 0);\n    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b.w, macc[g], 0, 0, 0);",
    "    if (X & 8) { macc[g][0] += a[q].x + b.x; macc[g][1] += a[q].y + b.y; } else {\n    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b.x, macc[g], 0, 0, 0);\n    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b.y, macc[g], 0, 0, 0);\n    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b.z, macc[g], 0, 0, 0);\n    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b.w, macc[g], 0, 0, 0); }")
# 16: A reloads
rep("    if constexpr (g == NG - 1u) {\n        a[q] = anext[q * 64u];\n    }", "    if constexpr (g == NG - 1u && !(X & 16)) {\n        a[q] = anext[q * 64u];\n    }")
open(os.path.join(ROOT, "tools/exp/f32_knock.hip"), "w").write(s)
