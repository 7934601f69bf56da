/*
 * mfm_taps.c - per-channel set-up arithmetic, kept in C so the double-precision expressions are
 * evaluated exactly as the reference's C evaluates them (same libm cexp/pow, same operand order,
 * no contraction).  The results (int16 taps, int16 rotator increment) are the kernel's inputs.
 */
#include "mfm_taps.h"

#include <complex.h>
#include <math.h>

#define MFM_Q14 ((double)(1ll << 14)) /* Q_15_SHIFT is 14: filter/filter.h:16 */

void mfm_taps_rotate_q14(const double *lpf_taps, size_t nr_taps, int32_t offset_hz, uint32_t sample_rate,
                         double gain, int16_t *coeff_re, int16_t *coeff_im)
{
    /* multifm/demod.c:210 */
    const double f_offs = -2.0 * M_PI * (double)offset_hz / (double)sample_rate;

    for (size_t i = 0; i < nr_taps; i++) {
        /* multifm/demod.c:234: gain and the real tap scale the unit phasor, left to right */
        const double complex tap = gain * cexp(CMPLX(0, f_offs * (double)i)) * lpf_taps[i];
        /* multifm/demod.c:242-243: truncation toward zero */
        coeff_re[i] = (int16_t)(creal(tap) * MFM_Q14);
        coeff_im[i] = (int16_t)(cimag(tap) * MFM_Q14);
    }
}

void mfm_taps_rotate_f64(const double *lpf_taps, size_t nr_taps, int32_t offset_hz, uint32_t sample_rate, double gain,
                         double *coeff_re, double *coeff_im)
{
    /* the same expression without the int16 casts of multifm/demod.c:242-243: the floating-point path's taps */
    const double f_offs = -2.0 * M_PI * (double)offset_hz / (double)sample_rate;

    for (size_t i = 0; i < nr_taps; i++) {
        const double complex tap = gain * cexp(CMPLX(0, f_offs * (double)i)) * lpf_taps[i];
        coeff_re[i] = creal(tap);
        coeff_im[i] = cimag(tap);
    }
}

void mfm_taps_rot_increment(int32_t offset_hz, uint32_t sample_rate, uint32_t decimation, int16_t *incr_re,
                            int16_t *incr_im)
{
    /* filter/direct_fir.c:72-77 */
    const double fwt0 = 2.0 * M_PI * (double)offset_hz / (double)sample_rate;
    const double complex w = cexp(CMPLX(0, -fwt0 * (double)decimation));
    *incr_re = (int16_t)(int32_t)(creal(w) * MFM_Q14);
    *incr_im = (int16_t)(int32_t)(cimag(w) * MFM_Q14);
}

double mfm_taps_gain_from_db(double gain_db)
{
    return pow(10.0, gain_db / 10.0); /* multifm/receiver.c:220 */
}
