/*
 * mfm_numerics.h - exact scalar arithmetic of the multifm channel epilogue, written once and
 * compiled twice: for gfx950 inside the fused kernel, and for the host as the "host twin" the
 * test-suite uses to prove (exhaustively where the domain allows) that these formulas give the
 * same bits as the reference's expressions.
 *
 * Reference expressions (pvachon/tsl-sdr):
 *   round_q30_q15 ............ filter/complex.h:30-34
 *   derotation ............... filter/direct_fir.c:151-172, :406-413
 *   discriminator ............ multifm/fm_demod.c:55-72
 *   fast_atan2f .............. multifm/fast_atan2f.c:101-174
 *
 * No FP contraction anywhere in this file: the reference's default build never fuses
 * fast_atan2f.c:131 (DESIGN.md "Float reproducibility").
 */
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define MFM_HD __host__ __device__ __forceinline__
#else
#define MFM_HD static inline
#endif

/* (float)pi, (float)(pi/2): fast_atan2f.c:141,150,157 assign the double literals to a float */
#define MFM_PI_F 3.14159265358979323846f
#define MFM_HALF_PI_F 1.57079632679489661923f

/* fast_atan2f.c:10,121: `z < 0.003921569` promotes z to double.  0.003921569 is not a float, so
 * (double)z < c  <=>  z < nextafterf-up of c, i.e. the smallest float >= c: 0x1.010104p-8f. */
#define MFM_TAN_MAP_RES_F 0x1.010104p-8f

/* 16384/pi split in two floats: hi = fl32(16384/pi), lo = fl32(16384/pi - hi).
 * pcm = (int)fmaf(m, hi, m*lo) equals (int16)(float)(((double)m / M_PI) * 16384.0)
 * (fm_demod.c:71-72) for EVERY float m in [0, (float)pi]: tests/test_numerics_host.py checks all
 * 1 078 530 012 of them against the oracle. */
#define MFM_Q14_OVER_PI_HI 0x1.45f306p+12f
#define MFM_Q14_OVER_PI_LO 0x1.b9391p-13f

/* filter/complex.h:30-34 without the int16 truncation (callers pack the low 16 bits). */
MFM_HD int32_t mfm_r14_wide(int32_t a)
{
    return (a >> 14) + ((a >> 13) & 1);
}

MFM_HD uint32_t mfm_pack16(int32_t lo, int32_t hi)
{
    return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
}

MFM_HD int32_t mfm_lo16(uint32_t p)
{
    return (int32_t)(int16_t)(p & 0xffffu);
}

MFM_HD int32_t mfm_hi16(uint32_t p)
{
    return (int32_t)p >> 16;
}

/* fm_demod.c:71-72 for a non-negative angle (see MFM_Q14_OVER_PI_*). */
MFM_HD int32_t mfm_mag_to_pcm(float mag)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float lo = mag * MFM_Q14_OVER_PI_LO;
    const float sc = __builtin_fmaf(mag, MFM_Q14_OVER_PI_HI, lo);
    return (int32_t)sc;
}

/*
 * Phase of s = q * conj(prev) as int16 PCM.  s_re/s_im are the wrapped int32 products of
 * fm_demod.c:63-64; lut[i] = { T[i], T[i+1]-T[i] } with T the 257-entry table of
 * fast_atan2f.c:14-81 (the difference is the same float subtraction line :131 performs).
 */
#if defined(__HIPCC__)
#define MFM_LUT_T const float2 *
#define MFM_LUT_X(e) (e).x
#define MFM_LUT_Y(e) (e).y
#else
struct mfm_lut_ent {
    float x, y;
};
#define MFM_LUT_T const struct mfm_lut_ent *
#define MFM_LUT_X(e) (e).x
#define MFM_LUT_Y(e) (e).y
#endif

MFM_HD int32_t mfm_discriminate(int32_t s_re, int32_t s_im, MFM_LUT_T lut)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float y = (float)s_im, x = (float)s_re; /* fm_demod.c:68 */
    const float ya = __builtin_fabsf(y), xa = __builtin_fabsf(x);
    const float mx = ya > xa ? ya : xa, mn = ya > xa ? xa : ya;

    /* fast_atan2f.c:114-117: z = min/max (equal magnitudes give x_abs/y_abs = 1) */
    const float z = mn / mx;

    float base = z; /* :121-122 */
    if (!(z < MFM_TAN_MAP_RES_F)) {
        float alpha = z * 255.0f;               /* :125 */
        const int idx = ((int)alpha) & 0xff;    /* :126 */
        alpha = alpha - (float)idx;             /* :127 */
        const float t0 = MFM_LUT_X(lut[idx]), dt = MFM_LUT_Y(lut[idx]);
        const float prod = dt * alpha;
        base = t0 + prod;                       /* :130-131, unfused */
    }

    /* :134-163 folded: every branch is sign(y) * (K + u) with K in {0, pi, pi/2}, u = +-base */
    const bool x_nonneg = x >= 0.0f, wide = xa > ya;
    const float k = wide ? (x_nonneg ? 0.0f : MFM_PI_F) : MFM_HALF_PI_F;
    const float u = (x_nonneg == wide) ? base : -base;
    const float mag = k + u;

    /* fm_demod.c:71-72 on |angle|, sign restored afterwards (division, rounding and the
     * truncating cast are all odd-symmetric) */
    int32_t pcm = mfm_mag_to_pcm(mag);
    pcm = (y >= 0.0f) ? pcm : -pcm;
    /* fast_atan2f.c:111-112: (0,0) -> 0 (mx == 0 made z NaN above; it is discarded here) */
    return (mx > 0.0f) ? pcm : 0;
}
