/* f32_oracle.c - see f32_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "f32_oracle.h"
#include "mfm_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct mfmo_f32_chan {
    uint32_t fs, decim;
    size_t nt;
    double *cr, *ci;
    int64_t step_mod; /* (off * D) mod fs, in [0, fs) */
    uint64_t n_out;   /* absolute index of the next output */
    double prev_re, prev_im;
    float *hist;      /* unconsumed samples, interleaved */
    size_t nh, cap;
    double lut[257];
};

struct mfmo_f32_chan *mfmo_f32_chan_new(int32_t offset_hz, uint32_t sample_rate, uint32_t decimation,
                                        const double *lpf_taps, size_t nr_taps, double gain)
{
    struct mfmo_f32_chan *c = calloc(1, sizeof(*c));
    c->fs = sample_rate;
    c->decim = decimation;
    c->nt = nr_taps;
    c->cr = malloc(nr_taps * sizeof(double));
    c->ci = malloc(nr_taps * sizeof(double));
    /* multifm/demod.c:210 */
    const double f_offs = -2.0 * M_PI * (double)offset_hz / (double)sample_rate;
    for (size_t i = 0; i < nr_taps; i++) {
        /* multifm/demod.c:234: (gain * cexp(j f_offs i)) * h[i], without the casts of :242-243 */
        const double complex t = (gain * cexp(CMPLX(0, f_offs * (double)i))) * lpf_taps[i];
        c->cr[i] = creal(t);
        c->ci[i] = cimag(t);
    }
    int64_t m = ((int64_t)offset_hz * (int64_t)decimation) % (int64_t)sample_rate;
    if (m < 0) {
        m += sample_rate;
    }
    c->step_mod = m;
    float tf[257];
    mfmo_atan_table(tf);
    for (int i = 0; i < 257; i++) {
        c->lut[i] = (double)tf[i];
    }
    return c;
}

void mfmo_f32_chan_free(struct mfmo_f32_chan *c)
{
    if (c) {
        free(c->cr);
        free(c->ci);
        free(c->hist);
        free(c);
    }
}

void mfmo_f32_chan_taps(const struct mfmo_f32_chan *c, double *re, double *im)
{
    memcpy(re, c->cr, c->nt * sizeof(double));
    memcpy(im, c->ci, c->nt * sizeof(double));
}

/* multifm/fast_atan2f.c:101-174 in double, on the reference's (float) table */
static double fast_atan2_d(const double *lut, double y, double x)
{
    const double xa = fabs(x), ya = fabs(y);
    if (!(xa > 0.0 || ya > 0.0)) {
        return 0.0; /* :111-112 */
    }
    const double z = (ya > xa) ? xa / ya : ya / xa; /* :114-117 */
    double base;
    if (z < 0.003921569) { /* :121 */
        base = z;
    } else {
        double alpha = z * 255.0;
        int idx = ((int)alpha) & 0xff;
        alpha -= (double)idx;
        base = lut[idx] + (lut[idx + 1] - lut[idx]) * alpha; /* :125-131 */
    }
    double ang;
    if (xa > ya) { /* :134-163 */
        ang = (x >= 0.0) ? base : M_PI - base;
    } else {
        ang = (x >= 0.0) ? M_PI_2 - base : M_PI_2 + base;
    }
    return (y < 0.0) ? -ang : ang;
}

size_t mfmo_f32_chan_push(struct mfmo_f32_chan *c, const float *iq, size_t nr_samples, double *pcm, double *iq_out,
                          size_t out_cap)
{
    if (c->nh + nr_samples > c->cap) {
        c->cap = (c->nh + nr_samples) * 2 + 16;
        c->hist = realloc(c->hist, c->cap * 2 * sizeof(float));
    }
    memcpy(c->hist + 2 * c->nh, iq, nr_samples * 2 * sizeof(float));
    c->nh += nr_samples;
    size_t pos = 0, n = 0;
    while (c->nh - pos >= c->nt && n < out_cap) {
        double ar = 0.0, ai = 0.0;
        const float *x = c->hist + 2 * pos;
        for (size_t i = 0; i < c->nt; i++) {
            const double xr = x[2 * i], xi = x[2 * i + 1];
            ar += c->cr[i] * xr - c->ci[i] * xi;
            ai += c->cr[i] * xi + c->ci[i] * xr;
        }
        /* w^n with the phase reduced exactly: (off*D*n) mod fs */
        const uint64_t ph = (uint64_t)(((unsigned __int128)(uint64_t)c->step_mod * (c->n_out % c->fs)) % c->fs);
        const double ang = -2.0 * M_PI * ((double)ph / (double)c->fs);
        const double wr = cos(ang), wi = sin(ang);
        const double or_ = ar * wr - ai * wi, oi = ar * wi + ai * wr;
        /* multifm/fm_demod.c:63-64 */
        const double sr = or_ * c->prev_re + oi * c->prev_im;
        const double si = oi * c->prev_re - or_ * c->prev_im;
        pcm[n] = fast_atan2_d(c->lut, si, sr) / M_PI * 16384.0;
        if (iq_out) {
            iq_out[2 * n] = or_;
            iq_out[2 * n + 1] = oi;
        }
        c->prev_re = or_;
        c->prev_im = oi;
        c->n_out++;
        n++;
        pos += c->decim;
    }
    memmove(c->hist, c->hist + 2 * pos, (c->nh - pos) * 2 * sizeof(float));
    c->nh -= pos;
    return n;
}
