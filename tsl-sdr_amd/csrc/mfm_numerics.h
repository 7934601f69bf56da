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

/* Correctly rounded mn / mx.  On the device: v_rcp_f32, one Newton step on the reciprocal, the quotient estimate and ONE
 * residual step (Markstein's form) - without the v_div_scale / v_div_fixup wrapping that only matters for subnormal,
 * infinite or zero operands: here both operands are int32 values converted to float, mn <= mx, so the quotient is in
 * [2^-31, 1] (or 0/0 -> NaN, which the caller discards).
 * Why one residual step is enough (rounds 1-3 ran two, because Markstein's theorem wants a correctly rounded reciprocal
 * and v_rcp_f32 is specified to 1 ulp): q1 = RN(v), v = q0 + r1 (mn - mx q0) exactly, and |v - mn/mx| < 2^-23 ulp; RN(v)
 * can differ from RN(mn/mx) only when mn/mx lies that close to the midpoint of two floats, which for 24-bit significands
 * A, B means |A 2^k - t B| <= 4 with t odd - at most a handful of A per B.  tools/div_proof.c enumerates all 46 517 418
 * such pairs over all 2^23 divisors and runs this sequence on each, with the reciprocal gfx950 really returns for B
 * (tools/rcp_check.hip tabulates v_rcp_f32 for every significand: 89 % correctly rounded, 9 % one ulp low, 2 % one ulp
 * high): 0 wrong (profiles/r04_division_proof.txt).  The form is NOT right for every 1-ulp reciprocal: B = 2^24 - 1 needs
 * the correctly rounded one, B = 2^24 - 3 must not come out one ulp high - so the engine checks those quotients on the
 * device at commit (mfm_engine.hip, division self-test) and refuses a device whose v_rcp_f32 answers differently. */
MFM_HD float mfm_div_unit(float mn, float mx)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(mx);
    const float e0 = __builtin_fmaf(-mx, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = mn * r1;
    const float e1 = __builtin_fmaf(-mx, q0, mn);
    return __builtin_fmaf(e1, r1, q0);
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
#if defined(__HIP_DEVICE_COMPILE__)
    /* one instruction each (operands are converted integers, never NaN; fmaxf/fminf would add canonicalising
     * self-maxes) */
    float mx, mn;
    asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(mx) : "v"(x), "v"(y));
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
#else
    const bool tall = ya > xa;
    const float mx = tall ? ya : xa, mn = tall ? xa : ya;
#endif

    /* fast_atan2f.c:114-117: z = min/max (equal magnitudes give x_abs/y_abs = 1) */
    const float z = mfm_div_unit(mn, mx);

    /* :121-132, both arms evaluated, no branch (idx is 0 when z is below the threshold or NaN) */
    float alpha = z * 255.0f;               /* :125 */
#if defined(__HIP_DEVICE_COMPILE__)
    /* z is in [0, 1] or NaN, so (int)alpha is in [0, 255] (NaN converts to 0) and needs no mask; alpha - (float)idx
     * is alpha - floor(alpha), exact, which is what v_fract_f32 returns */
    const int idx = (int)alpha;             /* :126 */
    alpha = __builtin_amdgcn_fractf(alpha); /* :127 */
#else
    const int idx = ((int)alpha) & 0xff;    /* :126 */
    alpha = alpha - (float)idx;             /* :127 */
#endif
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
     * fast_atan2f.c:111-112's answer.  One v_bfi_b32 (the compiler splits the and/and/or form). */
    int32_t signed_sc;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(signed_sc) : "s"(0x7fffffff), "v"(sc), "v"(s_im));
    return (int32_t)__int_as_float(signed_sc);
#else
    int32_t pcm = (int32_t)sc;
    pcm = (s_im >= 0) ? pcm : -pcm;
    /* fast_atan2f.c:111-112: (0,0) -> 0 (mx == 0 made z NaN above; it is discarded here) */
    return (mx > 0.0f) ? pcm : 0;
#endif
}

#if defined(__HIPCC__)
/*
 * Two discriminators at once, for the MFMA kernel: the float chain (division refinement, table interpolation,
 * octant offset, Q14 scaling) is written on 2-vectors so that it compiles to packed FP32 instructions
 * (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction, same results as the scalar
 * form above).  t_lut / d_lut are the table and its first difference as two separate arrays, so the four
 * look-ups land directly in register pairs.
 */
typedef float mfm_v2f __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ void mfm_discriminate2(const int32_t s_re[2], const int32_t s_im[2], const float *t_lut,
                                                         const float *d_lut, int32_t pcm[2])
{
#if defined(__HIP_DEVICE_COMPILE__) /* the host pass only needs the declaration */
#pragma clang fp contract(off)
    mfm_v2f mx, mn, r0;
    float x[2], y[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        x[i] = (float)s_re[i];
        y[i] = (float)s_im[i];
        float a, b;
        asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(a) : "v"(x[i]), "v"(y[i]));
        asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(b) : "v"(x[i]), "v"(y[i]));
        mx[i] = a;
        mn[i] = b;
        r0[i] = __builtin_amdgcn_rcpf(a);
    }
    /* mfm_div_unit */
    const mfm_v2f one = { 1.0f, 1.0f };
    const mfm_v2f e0 = __builtin_elementwise_fma(-mx, r0, one);
    const mfm_v2f r1 = __builtin_elementwise_fma(e0, r0, r0);
    const mfm_v2f q0 = mn * r1;
    const mfm_v2f e1 = __builtin_elementwise_fma(-mx, q0, mn);
    const mfm_v2f z = __builtin_elementwise_fma(e1, r1, q0);

    const mfm_v2f alpha = z * 255.0f;
    mfm_v2f frac, t0, dt;
    {
        /* The four look-ups as four ds_read_b32 straight into the halves of the two register pairs.  Left to the
         * compiler, T[idx] and dT[idx] (256 floats apart) become one ds_read2st64_b32 per pair - which lands (T, dT) of
         * pair 0 in one register pair and needs three moves to regroup into (T0, T1) and (dT0, dT1).  d_lut must be
         * t_lut + 256 (1024 bytes); nothing else of this wave is in flight in LDS here, and the block drains itself. */
        const int i0 = (int)alpha[0], i1 = (int)alpha[1];
        frac[0] = __builtin_amdgcn_fractf(alpha[0]);
        frac[1] = __builtin_amdgcn_fractf(alpha[1]);
        const uint32_t a0 = (uint32_t)(uintptr_t)(t_lut + i0), a1 = (uint32_t)(uintptr_t)(t_lut + i1);
        float ta, tb, da, db;
        asm volatile("ds_read_b32 %0, %4\n\t"
                     "ds_read_b32 %1, %5\n\t"
                     "ds_read_b32 %2, %4 offset:1024\n\t"
                     "ds_read_b32 %3, %5 offset:1024\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(ta), "=&v"(tb), "=&v"(da), "=&v"(db)
                     : "v"(a0), "v"(a1)
                     : "memory");
        t0[0] = ta;
        t0[1] = tb;
        dt[0] = da;
        dt[1] = db;
        (void)d_lut;
    }
    const mfm_v2f prod = dt * frac;
    const mfm_v2f interp = t0 + prod;
    mfm_v2f k, u;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float base = (z[i] < MFM_TAN_MAP_RES_F) ? z[i] : interp[i];
        const bool x_nonneg = s_re[i] >= 0, wide = __builtin_fabsf(x[i]) > __builtin_fabsf(y[i]);
        k[i] = wide ? (x_nonneg ? 0.0f : MFM_PI_F) : MFM_HALF_PI_F;
        u[i] = (x_nonneg == wide) ? base : -base;
    }
    const mfm_v2f mag = k + u;
    const mfm_v2f lo = mag * MFM_Q14_OVER_PI_LO;
    const mfm_v2f hi = { MFM_Q14_OVER_PI_HI, MFM_Q14_OVER_PI_HI };
    const mfm_v2f sc = __builtin_elementwise_fma(mag, hi, lo);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int32_t signed_sc;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(signed_sc) : "s"(0x7fffffff), "v"(sc[i]), "v"(s_im[i]));
        pcm[i] = (int32_t)__int_as_float(signed_sc);
    }
#else
    (void)s_re, (void)s_im, (void)t_lut, (void)d_lut, (void)pcm;
#endif
}
#endif
