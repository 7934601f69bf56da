/*
 * f32_oracle.h - fp64 restatement of the engine's floating-point IQ path (TEST INFRASTRUCTURE ONLY).
 *
 * The reference has no floating-point channel path: BASELINE.json's configs[4] asks for
 * "fp32 vs int16 IQ path" and north_star sets the tolerance at 1e-5 relative "for the float FIR/atan2 stage".  The
 * float path is therefore the build's own; what it must agree with is this file: the same algorithm as the reference's
 * integer path with the Q14 quantisation steps removed, evaluated in double precision.
 *
 *   taps      c[i] = (gain * cexp(j * f_offs * i)) * h[i]           multifm/demod.c:210,232-243 before the int16 cast
 *   FIR       a[n] = sum_i c[i] * x[n*D + i]                        filter/direct_fir.c:363-384
 *   derotate  o[n] = a[n] * w^n,  w = cexp(-j*2*pi*off*D/fs)        filter/direct_fir.c:72-79,151-172 without the
 *                                                                   Q14 rounding of the recursion (w^n evaluated from
 *                                                                   the exactly reduced phase (off*D*n mod fs)/fs)
 *   discrim.  s = o[n] * conj(o[n-1]); phi = fast_atan2(s_im, s_re) multifm/fm_demod.c:55-72, fast_atan2f.c:101-174
 *             pcm = phi / pi * 16384                                (the table is the reference's float table,
 *                                                                   interpolated in double)
 *
 * PARITY UNPINNED: there is no reference implementation of this path to pin against.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct mfmo_f32_chan;

/* one channel; lpf_taps as in demod_thread_new (multifm/demod.h:104-110) */
struct mfmo_f32_chan *mfmo_f32_chan_new(int32_t offset_hz, uint32_t sample_rate, uint32_t decimation,
                                        const double *lpf_taps, size_t nr_taps, double gain);
void mfmo_f32_chan_free(struct mfmo_f32_chan *c);

/* the double-precision taps (for tests of the tap builder) */
void mfmo_f32_chan_taps(const struct mfmo_f32_chan *c, double *re, double *im);

/*
 * Push nr_samples interleaved float IQ samples; writes up to out_cap outputs: pcm (double, phi/pi*16384) and, if
 * iq_out is not NULL, the derotated filtered samples (re, im interleaved).  Returns the number of outputs.  State
 * (unconsumed samples, output index, previous filtered sample) carries over between calls.
 */
size_t mfmo_f32_chan_push(struct mfmo_f32_chan *c, const float *iq, size_t nr_samples, double *pcm, double *iq_out,
                          size_t out_cap);

#ifdef __cplusplus
}
#endif
