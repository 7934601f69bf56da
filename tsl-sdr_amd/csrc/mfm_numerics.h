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

/* Correctly rounded mn / mx.  On the device this is the core of the compiler's own IEEE fdiv
 * expansion (v_rcp_f32 + three Newton/residual steps) without the v_div_scale / v_div_fixup
 * wrapping that only matters for subnormal, infinite or zero operands: here both operands are
 * int32 values converted to float, mn <= mx, so the quotient is in [2^-31, 1] (or 0/0 -> NaN, which
 * the caller discards). */
MFM_HD float mfm_div_unit(float mn, float mx)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(mx);
    const float e0 = __builtin_fmaf(-mx, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = mn * r1;
    const float e1 = __builtin_fmaf(-mx, q0, mn);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-mx, q1, mn);
    return __builtin_fmaf(e2, r1, q1);
#else
    return mn / mx;
#endif
}

MFM_HD int32_t mfm_discriminate(int32_t s_re, int32_t s_im, MFM_LUT_T lut)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float y = (float)s_im, x = (float)s_re; /* fm_demod.c:68 */
    const float ya = __builtin_fabsf(y), xa = __builtin_fabsf(x);
    const bool tall = ya > xa;
    const float mx = tall ? ya : xa, mn = tall ? xa : ya;

    /* fast_atan2f.c:114-117: z = min/max (equal magnitudes give x_abs/y_abs = 1) */
    const float z = mfm_div_unit(mn, mx);

    /* :121-132, both arms evaluated, no branch (idx is 0 when z is below the threshold or NaN) */
    float alpha = z * 255.0f;               /* :125 */
    const int idx = ((int)alpha) & 0xff;    /* :126 */
    alpha = alpha - (float)idx;             /* :127 */
    const float t0 = MFM_LUT_X(lut[idx]), dt = MFM_LUT_Y(lut[idx]);
    const float prod = dt * alpha;
    const float interp = t0 + prod;         /* :130-131, unfused */
    const float base = (z < MFM_TAN_MAP_RES_F) ? z : interp;

    /* :134-163 folded: every branch is sign(y) * (K + u) with K in {0, pi, pi/2}, u = +-base */
    const bool x_nonneg = s_re >= 0, wide = xa > ya;
    const float k = wide ? (x_nonneg ? 0.0f : MFM_PI_F) : MFM_HALF_PI_F;
    const float u = (x_nonneg == wide) ? base : -base;
    const float mag = k + u;

    /* fm_demod.c:71-72 on |angle|, sign restored afterwards (division, rounding and the
     * truncating cast are all odd-symmetric) */
    const float lo = mag * MFM_Q14_OVER_PI_LO;
    const float sc = __builtin_fmaf(mag, MFM_Q14_OVER_PI_HI, lo);
#if defined(__HIP_DEVICE_COMPILE__)
    /* sign of y = sign of s_im; (0,0) made z NaN and NaN converts to 0 (v_cvt_i32_f32), which is
     * fast_atan2f.c:111-112's answer */
    const float signed_sc = __int_as_float((__float_as_int(sc) & 0x7fffffff) | (s_im & (int32_t)0x80000000));
    return (int32_t)signed_sc;
#else
    int32_t pcm = (int32_t)sc;
    pcm = (s_im >= 0) ? pcm : -pcm;
    /* fast_atan2f.c:111-112: (0,0) -> 0 (mx == 0 made z NaN above; it is discarded here) */
    return (mx > 0.0f) ? pcm : 0;
#endif
}
