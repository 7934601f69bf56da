/* mfm_taps.h - channel set-up arithmetic (see mfm_taps.c). Internal. */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void mfm_taps_rotate_q14(const double *lpf_taps, size_t nr_taps, int32_t offset_hz, uint32_t sample_rate,
                         double gain, int16_t *coeff_re, int16_t *coeff_im);
void mfm_taps_rotate_f64(const double *lpf_taps, size_t nr_taps, int32_t offset_hz, uint32_t sample_rate, double gain,
                         double *coeff_re, double *coeff_im);
void mfm_taps_rot_increment(int32_t offset_hz, uint32_t sample_rate, uint32_t decimation, int16_t *incr_re,
                            int16_t *incr_im);
double mfm_taps_gain_from_db(double gain_db);

#ifdef __cplusplus
}
#endif
