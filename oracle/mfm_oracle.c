/*
 * mfm_oracle.c - CPU restatement of the multifm channel hot path (TEST INFRASTRUCTURE ONLY).
 *
 * See mfm_oracle.h for scope and pinning status.  Every function cites the reference lines it
 * follows (paths relative to the pvachon/tsl-sdr tree).  Build: oracle/Makefile, with
 * -ffp-contract=off -fwrapv (int32 wrap-around is what the reference relies on at
 * filter/complex.h:44-45 and multifm/fm_demod.c:63-64).
 */
#include "mfm_oracle.h"

#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------- */
/* Scalar pieces                                                                         */
/* ------------------------------------------------------------------------------------- */

int16_t mfmo_r14(int32_t a)
{
    /* filter/complex.h:30-34 */
    return (int16_t)((a >> MFMO_Q_SHIFT) + ((a >> (MFMO_Q_SHIFT - 1)) & 1));
}

static float g_atan_tbl[257];
static pthread_once_t g_atan_once = PTHREAD_ONCE_INIT;

static void atan_tbl_init(void)
{
    /* multifm/fast_atan2f.c:14-81.  The reference's literals are atan(i/255) written with seven
     * significant digits; the 257th entry repeats the 256th (index + 1 look-up at :131). */
    for (int i = 0; i < 257; i++) {
        char txt[32];
        int k = i < 255 ? i : 255;
        snprintf(txt, sizeof(txt), "%.6e", atan((double)k / 255.0));
        g_atan_tbl[i] = strtof(txt, NULL);
    }
}

void mfmo_atan_table(float tbl[257])
{
    pthread_once(&g_atan_once, atan_tbl_init);
    memcpy(tbl, g_atan_tbl, sizeof(g_atan_tbl));
}

static inline float atan2_core(float y, float x, int fused)
{
    /* multifm/fast_atan2f.c:101-174 */
    const float ya = fabsf(y), xa = fabsf(x);
    float z, base;

    if (!((ya > 0.0f) || (xa > 0.0f))) { /* :111-112 */
        return 0.0f;
    }
    z = (ya < xa) ? ya / xa : xa / ya; /* :114-117 */

    if ((double)z < 0.003921569) { /* :121-122, TAN_MAP_RES is a double literal (:10) */
        base = z;
    } else {
        float alpha = z * 255.0f;           /* :125 */
        int idx = ((int)alpha) & 0xff;      /* :126 */
        alpha -= (float)idx;                /* :127 */
        base = g_atan_tbl[idx];             /* :130 */
        if (fused) {
            base = fmaf(g_atan_tbl[idx + 1] - g_atan_tbl[idx], alpha, base);
        } else {
            float d = g_atan_tbl[idx + 1] - g_atan_tbl[idx];
            float p = d * alpha;
            base = base + p;                /* :131 */
        }
    }

    if (xa > ya) {                          /* :134-147 */
        if (x >= 0.0f) {
            return (y >= 0.0f) ? base : -base;
        } else {
            const float pi_f = 3.14159265358979323846f;
            return (y >= 0.0f) ? pi_f - base : base - pi_f;
        }
    } else {                                /* :148-163 */
        const float hp_f = 1.57079632679489661923f;
        if (y >= 0.0f) {
            return (x >= 0.0f) ? hp_f - base : hp_f + base;
        } else {
            return (x >= 0.0f) ? -hp_f + base : -hp_f - base;
        }
    }
}

float mfmo_fast_atan2f(float y, float x)
{
    pthread_once(&g_atan_once, atan_tbl_init);
    return atan2_core(y, x, 0);
}

float mfmo_fast_atan2f_fma(float y, float x)
{
    pthread_once(&g_atan_once, atan_tbl_init);
    return atan2_core(y, x, 1);
}

int16_t mfmo_phi_to_pcm(float phi)
{
    /* multifm/fm_demod.c:42,71-72: float phi_scaled = (phi/M_PI) * to_q15; (int16_t)phi_scaled */
    const float to_q15 = (float)(1 << MFMO_Q_SHIFT);
    float phi_scaled = (float)(((double)phi / M_PI) * (double)to_q15);
    return (int16_t)phi_scaled;
}

void mfmo_phi_to_pcm_range(uint32_t first_bits, uint32_t count, int16_t *out)
{
    for (uint32_t i = 0; i < count; i++) {
        uint32_t b = first_bits + i;
        float phi;
        memcpy(&phi, &b, sizeof(phi));
        out[i] = mfmo_phi_to_pcm(phi);
    }
}

int16_t mfmo_fm_step(int16_t a_re16, int16_t a_im16, int32_t last_re, int32_t last_im)
{
    /* multifm/fm_demod.c:55-72 */
    int32_t b_re = last_re, b_im = -last_im, a_re = a_re16, a_im = a_im16;
    int32_t s_re = (int32_t)((uint32_t)(a_re * b_re) - (uint32_t)(a_im * b_im));
    int32_t s_im = (int32_t)((uint32_t)(a_re * b_im) + (uint32_t)(a_im * b_re));
    float phi = mfmo_fast_atan2f((float)s_im, (float)s_re);
    return mfmo_phi_to_pcm(phi);
}

void mfmo_discriminate_batch(const int32_t *s_re, const int32_t *s_im, size_t n, int16_t *out, int fused)
{
    /* multifm/fm_demod.c:68-72 on precomputed s = a * conj(prev) */
    pthread_once(&g_atan_once, atan_tbl_init);
    for (size_t i = 0; i < n; i++) {
        float phi = atan2_core((float)s_im[i], (float)s_re[i], fused);
        out[i] = mfmo_phi_to_pcm(phi);
    }
}

/* ------------------------------------------------------------------------------------- */
/* Channel set-up                                                                        */
/* ------------------------------------------------------------------------------------- */

void mfmo_make_taps(const double *lpf_taps, size_t nr_taps, int32_t offset_hz, uint32_t sample_rate,
                    double gain, int16_t *coeff_re, int16_t *coeff_im)
{
    /* multifm/demod.c:210 */
    double f_offs = -2.0 * M_PI * (double)offset_hz / (double)sample_rate;
    const double q15 = (double)(1ll << MFMO_Q_SHIFT);
    for (size_t i = 0; i < nr_taps; i++) {
        /* multifm/demod.c:234, operand order as written there: (gain * cexp(..)) * lpf_taps[i] */
        const double complex lpf_tap = gain * cexp(CMPLX(0, f_offs * (double)i)) * lpf_taps[i];
        coeff_re[i] = (int16_t)(creal(lpf_tap) * q15); /* :242 */
        coeff_im[i] = (int16_t)(cimag(lpf_tap) * q15); /* :243 */
    }
}

void mfmo_rot_incr(int32_t offset_hz, uint32_t sample_rate, unsigned decimation, int16_t *incr_re,
                   int16_t *incr_im)
{
    /* filter/direct_fir.c:72-77 */
    double fwt0 = 2.0 * M_PI * (double)offset_hz / (double)sample_rate;
    double q15 = (double)(1ll << MFMO_Q_SHIFT);
    double complex w = cexp(CMPLX(0, -fwt0 * (double)decimation));
    *incr_re = (int16_t)(int32_t)(creal(w) * q15);
    *incr_im = (int16_t)(int32_t)(cimag(w) * q15);
}

double mfmo_gain_from_db(double db)
{
    return pow(10.0, db / 10.0); /* multifm/receiver.c:220 */
}

void mfmo_rot_step(int16_t *rot_re, int16_t *rot_im, int16_t incr_re, int16_t incr_im)
{
    /* filter/direct_fir.c:166-167 -> filter/complex.h:51-62 (cmul_q15_q15) */
    int32_t a_re = *rot_re, a_im = *rot_im, b_re = incr_re, b_im = incr_im;
    *rot_re = mfmo_r14(a_re * b_re - a_im * b_im);
    *rot_im = mfmo_r14(a_re * b_im + a_im * b_re);
}

/* ------------------------------------------------------------------------------------- */
/* Streaming channel (closed-form stream semantics)                                      */
/* ------------------------------------------------------------------------------------- */

struct mfmo_chan {
    int16_t *cre, *cim;
    size_t nr_taps;
    unsigned decim;
    int16_t incr_re, incr_im;
    int16_t rot_re, rot_im;
    int derotate;
    int32_t last_re, last_im; /* multifm/fm_demod.c:16-17 */
    /* unconsumed tail of the stream, interleaved IQ */
    int16_t *pend;
    size_t pend_samples, pend_cap;
};

struct mfmo_chan *mfmo_chan_new(const int16_t *coeff_re, const int16_t *coeff_im, size_t nr_taps,
                                unsigned decimation, int16_t incr_re, int16_t incr_im)
{
    struct mfmo_chan *ch = calloc(1, sizeof(*ch));
    /* decimation > taps makes the reference dereference a NULL sb_active (direct_fir.c:394-398) */
    if (!ch || !nr_taps || !decimation || decimation > nr_taps) {
        free(ch);
        return NULL;
    }
    pthread_once(&g_atan_once, atan_tbl_init);
    ch->cre = malloc(nr_taps * sizeof(int16_t));
    ch->cim = malloc(nr_taps * sizeof(int16_t));
    memcpy(ch->cre, coeff_re, nr_taps * sizeof(int16_t));
    memcpy(ch->cim, coeff_im, nr_taps * sizeof(int16_t));
    ch->nr_taps = nr_taps;
    ch->decim = decimation;
    ch->incr_re = incr_re;
    ch->incr_im = incr_im;
    /* filter/direct_fir.c:78-79 and the test at :406 */
    ch->rot_re = 1 << MFMO_Q_SHIFT;
    ch->rot_im = 0;
    ch->derotate = !(0 == incr_re && 0 == incr_im);
    return ch;
}

void mfmo_chan_free(struct mfmo_chan *ch)
{
    if (ch) {
        free(ch->cre);
        free(ch->cim);
        free(ch->pend);
        free(ch);
    }
}

void mfmo_chan_rot(const struct mfmo_chan *ch, int16_t *rot_re, int16_t *rot_im)
{
    *rot_re = ch->rot_re;
    *rot_im = ch->rot_im;
}

/* Window checks (bench.py's `verified`, the full-size GPU tests): put a fresh channel where a stream stands after
 * nr_outputs outputs as far as the rotator is concerned - filter/direct_fir.c:166-167 applied nr_outputs times, which is
 * all the state direct_fir carries besides the samples themselves.  The discriminator's last sample (fm_demod.c:16-17)
 * stays zero: callers feed the window one output early and drop that output's PCM. */
void mfmo_chan_skip_outputs(struct mfmo_chan *ch, uint64_t nr_outputs)
{
    if (!ch->derotate) {
        return;
    }
    int16_t re = ch->rot_re, im = ch->rot_im;
    if (nr_outputs > (1ull << 24)) {
        /* The recurrence is a map of a finite set (two int16) into itself and does not depend on the input, so from any
         * state its orbit is a pre-period of mu steps followed by a cycle of lam steps, and n steps end where
         * mu + (n - mu) mod lam steps do.  Brent's cycle search with the same step function; nothing else changes.
         * (Bench-rate streams pass 2^32 outputs within a second: stepping one by one would take the oracle minutes.) */
        int16_t tr = re, ti = im, hr = re, hi = im;
        uint64_t power = 1, lam = 1, mu = 0;
        mfmo_rot_step(&hr, &hi, ch->incr_re, ch->incr_im);
        while (!(tr == hr && ti == hi)) {
            if (power == lam) {
                tr = hr;
                ti = hi;
                power *= 2;
                lam = 0;
            }
            mfmo_rot_step(&hr, &hi, ch->incr_re, ch->incr_im);
            lam++;
        }
        tr = re, ti = im, hr = re, hi = im;
        for (uint64_t i = 0; i < lam; i++) {
            mfmo_rot_step(&hr, &hi, ch->incr_re, ch->incr_im);
        }
        while (!(tr == hr && ti == hi)) {
            mfmo_rot_step(&tr, &ti, ch->incr_re, ch->incr_im);
            mfmo_rot_step(&hr, &hi, ch->incr_re, ch->incr_im);
            mu++;
        }
        if (nr_outputs > mu) {
            nr_outputs = mu + (nr_outputs - mu) % lam;
        }
    }
    for (uint64_t i = 0; i < nr_outputs; i++) {
        mfmo_rot_step(&re, &im, ch->incr_re, ch->incr_im);
    }
    ch->rot_re = re;
    ch->rot_im = im;
}

/* One output from a window of nr_taps samples starting at w (interleaved IQ). */
static inline void chan_one_output(struct mfmo_chan *ch, const int16_t *w, int16_t *q_re, int16_t *q_im,
                                   int16_t *pcm)
{
    uint32_t acc_re = 0, acc_im = 0;
    const size_t T = ch->nr_taps;
    const int16_t *cre = ch->cre, *cim = ch->cim;

    /* filter/direct_fir.c:363-384 + filter/complex.h:40-46 (cmul_q15_q30(c, s)) */
    for (size_t i = 0; i < T; i++) {
        int32_t s_re = w[2 * i], s_im = w[2 * i + 1], c_re = cre[i], c_im = cim[i];
        acc_re += (uint32_t)(c_re * s_re) - (uint32_t)(c_im * s_im);
        acc_im += (uint32_t)(c_re * s_im) + (uint32_t)(c_im * s_re);
    }

    int32_t o_re = (int32_t)acc_re, o_im = (int32_t)acc_im;
    if (ch->derotate) {
        /* filter/direct_fir.c:406-409 -> :151-172 */
        int32_t f_re = mfmo_r14(o_re), f_im = mfmo_r14(o_im);
        int32_t r_re = ch->rot_re, r_im = ch->rot_im;
        o_re = (int32_t)((uint32_t)(f_re * r_re) - (uint32_t)(f_im * r_im));
        o_im = (int32_t)((uint32_t)(f_re * r_im) + (uint32_t)(f_im * r_re));
        mfmo_rot_step(&ch->rot_re, &ch->rot_im, ch->incr_re, ch->incr_im);
    }
    /* filter/direct_fir.c:412-413 */
    *q_re = mfmo_r14(o_re);
    *q_im = mfmo_r14(o_im);

    /* multifm/fm_demod.c:53-79 */
    *pcm = mfmo_fm_step(*q_re, *q_im, ch->last_re, ch->last_im);
    ch->last_re = *q_re;
    ch->last_im = *q_im;
}

size_t mfmo_chan_feed(struct mfmo_chan *ch, const int16_t *iq, size_t nr_samples, int16_t *pcm_out,
                      int16_t *iq_out, size_t max_out)
{
    const size_t T = ch->nr_taps, D = ch->decim;
    size_t n_out = 0;

    /* 1. outputs whose window starts inside the pending tail */
    if (ch->pend_samples) {
        size_t need = ch->pend_samples + nr_samples;
        if (need > ch->pend_cap) {
            ch->pend_cap = need + T;
            ch->pend = realloc(ch->pend, ch->pend_cap * 2 * sizeof(int16_t));
        }
        /* only as much of the new data as the tail windows can reach is needed, but keep it simple */
        memcpy(ch->pend + 2 * ch->pend_samples, iq, nr_samples * 2 * sizeof(int16_t));
        size_t tot = ch->pend_samples + nr_samples, pos = 0;
        while (pos + T <= tot && n_out < max_out) {
            int16_t qr, qi, p;
            chan_one_output(ch, ch->pend + 2 * pos, &qr, &qi, &p);
            pcm_out[n_out] = p;
            if (iq_out) {
                iq_out[2 * n_out] = qr;
                iq_out[2 * n_out + 1] = qi;
            }
            n_out++;
            pos += D;
        }
        size_t rem = tot - pos; /* pos <= tot because D <= T (checked in mfmo_chan_new) */
        memmove(ch->pend, ch->pend + 2 * pos, rem * 2 * sizeof(int16_t));
        ch->pend_samples = rem;
        return n_out;
    }

    /* 2. fresh data only */
    size_t pos = 0;
    while (pos + T <= nr_samples && n_out < max_out) {
        int16_t qr, qi, p;
        chan_one_output(ch, iq + 2 * pos, &qr, &qi, &p);
        pcm_out[n_out] = p;
        if (iq_out) {
            iq_out[2 * n_out] = qr;
            iq_out[2 * n_out + 1] = qi;
        }
        n_out++;
        pos += D;
    }
    size_t rem = nr_samples - pos;
    if (rem > ch->pend_cap) {
        ch->pend_cap = rem + T;
        ch->pend = realloc(ch->pend, ch->pend_cap * 2 * sizeof(int16_t));
    }
    if (rem) {
        memcpy(ch->pend, iq + 2 * pos, rem * 2 * sizeof(int16_t));
    }
    ch->pend_samples = rem;
    return n_out;
}

/* ------------------------------------------------------------------------------------- */
/* Thread-per-channel batch runner (cpu_baseline)                                        */
/* ------------------------------------------------------------------------------------- */

struct run_job {
    const int16_t *iq;
    size_t nr_samples, nr_chan, nr_taps, out_stride;
    const int16_t *cre, *cim, *incr;
    unsigned decim, tid, nthreads;
    int16_t *pcm_out, *iq_out;
    size_t n_out;
};

static void *run_worker(void *arg)
{
    struct run_job *j = arg;
    for (size_t c = j->tid; c < j->nr_chan; c += j->nthreads) {
        struct mfmo_chan *ch = mfmo_chan_new(j->cre + c * j->nr_taps, j->cim + c * j->nr_taps, j->nr_taps,
                                             j->decim, j->incr[2 * c], j->incr[2 * c + 1]);
        j->n_out = mfmo_chan_feed(ch, j->iq, j->nr_samples, j->pcm_out + c * j->out_stride,
                                  j->iq_out ? j->iq_out + 2 * c * j->out_stride : NULL, j->out_stride);
        mfmo_chan_free(ch);
    }
    return NULL;
}

size_t mfmo_run_channels(const int16_t *iq, size_t nr_samples, size_t nr_chan, const int16_t *coeff_re,
                         const int16_t *coeff_im, size_t nr_taps, unsigned decimation,
                         const int16_t *incr, int16_t *pcm_out, int16_t *iq_out, size_t out_stride,
                         unsigned nr_threads)
{
    if (nr_threads < 1) {
        nr_threads = 1;
    }
    if (nr_threads > nr_chan) {
        nr_threads = (unsigned)nr_chan;
    }
    pthread_t *thr = calloc(nr_threads, sizeof(*thr));
    struct run_job *jobs = calloc(nr_threads, sizeof(*jobs));
    for (unsigned t = 0; t < nr_threads; t++) {
        jobs[t] = (struct run_job){ .iq = iq, .nr_samples = nr_samples, .nr_chan = nr_chan,
                                    .nr_taps = nr_taps, .out_stride = out_stride, .cre = coeff_re,
                                    .cim = coeff_im, .incr = incr, .decim = decimation, .tid = t,
                                    .nthreads = nr_threads, .pcm_out = pcm_out, .iq_out = iq_out };
        if (nr_threads == 1) {
            run_worker(&jobs[t]);
        } else {
            pthread_create(&thr[t], NULL, run_worker, &jobs[t]);
        }
    }
    if (nr_threads > 1) {
        for (unsigned t = 0; t < nr_threads; t++) {
            pthread_join(thr[t], NULL);
        }
    }
    size_t n_out = jobs[0].n_out;
    free(jobs);
    free(thr);
    return n_out;
}

/* ------------------------------------------------------------------------------------- */
/* Structure-following two-slot walk                                                     */
/* ------------------------------------------------------------------------------------- */

struct slot_buf {
    const int16_t *data;
    size_t nr_samples;
};

struct twoslot {
    struct slot_buf *active, *next; /* filter/direct_fir.h:40-48 */
    size_t sample_offset;           /* :31-33 */
    size_t nr_samples;              /* :36-38 */
};

size_t mfmo_twoslot_run(const int16_t *iq, size_t buf_samples, size_t nr_bufs, const int16_t *coeff_re,
                        const int16_t *coeff_im, size_t nr_taps, unsigned decimation, int16_t incr_re,
                        int16_t incr_im, int16_t *pcm_out, int16_t *iq_out, size_t max_out)
{
    struct slot_buf *bufs = calloc(nr_bufs, sizeof(*bufs));
    struct twoslot fir = { 0 };
    int16_t rot_re = 1 << MFMO_Q_SHIFT, rot_im = 0;
    int32_t last_re = 0, last_im = 0;
    const int derotate = !(0 == incr_re && 0 == incr_im);
    size_t n_out = 0;

    pthread_once(&g_atan_once, atan_tbl_init);

    for (size_t b = 0; b < nr_bufs; b++) {
        bufs[b].data = iq + 2 * b * buf_samples;
        bufs[b].nr_samples = buf_samples;

        /* multifm/demod.c:58 -> filter/direct_fir.c:118-146 */
        if (!fir.active) {
            fir.active = &bufs[b];
        } else if (!fir.next) {
            fir.next = &bufs[b];
        } else {
            fprintf(stderr, "mfm_oracle: twoslot queue full (A_E_BUSY, direct_fir.c:136)\n");
            abort();
        }
        fir.nr_samples += bufs[b].nr_samples;

        /* multifm/demod.c:63-115: while (nr_samples >= nr_coeffs) { up to 1024 outputs } */
        while (fir.nr_samples >= nr_taps) {
            size_t produced = 0;
            for (size_t i = 0; i < 1024; i++) { /* LPF_OUTPUT_LEN, multifm/demod.h:12 */
                /* filter/direct_fir.c:346-352 */
                if (fir.sample_offset + nr_taps > fir.active->nr_samples && !fir.next) {
                    break;
                }
                /* :355-391 walk active then next */
                uint32_t acc_re = 0, acc_im = 0;
                size_t remain = nr_taps, off = fir.sample_offset;
                struct slot_buf *cur = fir.active;
                do {
                    size_t avail = cur->nr_samples - off, start = nr_taps - remain;
                    size_t take = avail < remain ? avail : remain;
                    for (size_t k = 0; k < take; k++) {
                        const int16_t *s = &cur->data[2 * (off + k)];
                        int32_t s_re = s[0], s_im = s[1], c_re = coeff_re[k + start],
                                c_im = coeff_im[k + start];
                        acc_re += (uint32_t)(c_re * s_re) - (uint32_t)(c_im * s_im);
                        acc_im += (uint32_t)(c_re * s_im) + (uint32_t)(c_im * s_re);
                    }
                    off = 0;
                    cur = fir.next;
                    remain -= take;
                } while (remain != 0);

                /* :394-401 advance (strict '>', and the *new* active's count is used) */
                if (fir.sample_offset + decimation > fir.active->nr_samples) {
                    if (!fir.next) {
                        fprintf(stderr, "mfm_oracle: reference would dereference NULL here "
                                        "(direct_fir.c:396-398)\n");
                        abort();
                    }
                    fir.active = fir.next;
                    fir.next = NULL;
                    fir.sample_offset = (fir.sample_offset + decimation) - fir.active->nr_samples;
                } else {
                    fir.sample_offset += decimation;
                }
                fir.nr_samples -= decimation;

                int32_t o_re = (int32_t)acc_re, o_im = (int32_t)acc_im;
                if (derotate) { /* :406-409 */
                    int32_t f_re = mfmo_r14(o_re), f_im = mfmo_r14(o_im), r_re = rot_re, r_im = rot_im;
                    o_re = (int32_t)((uint32_t)(f_re * r_re) - (uint32_t)(f_im * r_im));
                    o_im = (int32_t)((uint32_t)(f_re * r_im) + (uint32_t)(f_im * r_re));
                    mfmo_rot_step(&rot_re, &rot_im, incr_re, incr_im);
                }
                int16_t q_re = mfmo_r14(o_re), q_im = mfmo_r14(o_im); /* :412-413 */

                if (n_out < max_out) {
                    pcm_out[n_out] = mfmo_fm_step(q_re, q_im, last_re, last_im);
                    if (iq_out) {
                        iq_out[2 * n_out] = q_re;
                        iq_out[2 * n_out + 1] = q_im;
                    }
                    n_out++;
                }
                last_re = q_re;
                last_im = q_im;
                produced++;
            }
            (void)produced;
        }
    }
    free(bufs);
    return n_out;
}

/* ------------------------------------------------------------------------------------- */
/* PCM stage: rational resampler + DC blocker                                            */
/* ------------------------------------------------------------------------------------- */

struct mfmo_resampler {
    int16_t *phase; /* [interp][plen] */
    size_t plen;
    unsigned interp, decim;
    size_t phase_id;  /* filter/polyphase_fir_priv.h: last_phase */
    int16_t *pend;    /* unconsumed samples */
    size_t pend_n, pend_cap;
};

struct mfmo_resampler *mfmo_resampler_new(const int16_t *coeffs, size_t nr_coeffs, unsigned interpolate,
                                          unsigned decimate)
{
    if (!coeffs || !nr_coeffs || !interpolate || !decimate) {
        return NULL;
    }
    struct mfmo_resampler *r = calloc(1, sizeof(*r));
    /* filter/polyphase_fir.c:70-76 */
    size_t plen = (nr_coeffs + interpolate - 1) / interpolate;
    plen = (plen + 3) & ~(size_t)3;
    r->plen = plen;
    r->interp = interpolate;
    r->decim = decimate;
    r->phase = calloc((size_t)interpolate * plen, sizeof(int16_t));
    for (size_t i = 0; i < nr_coeffs; i++) {
        r->phase[(i % interpolate) * plen + (i / interpolate)] = coeffs[i]; /* :81-83 */
    }
    return r;
}

void mfmo_resampler_free(struct mfmo_resampler *r)
{
    if (r) {
        free(r->phase);
        free(r->pend);
        free(r);
    }
}

size_t mfmo_resampler_phase_len(const struct mfmo_resampler *r)
{
    return r->plen;
}

size_t mfmo_resampler_feed(struct mfmo_resampler *r, const int16_t *pcm, size_t nr_samples, int16_t *out,
                           size_t max_out)
{
    size_t need = r->pend_n + nr_samples;
    if (need > r->pend_cap) {
        r->pend_cap = need + r->plen + 16;
        r->pend = realloc(r->pend, r->pend_cap * sizeof(int16_t));
    }
    memcpy(r->pend + r->pend_n, pcm, nr_samples * sizeof(int16_t));
    const size_t tot = need;
    size_t pos = 0, n_out = 0;
    /* filter/polyphase_fir.c:184: strictly more than one phase length of unconsumed samples */
    while (tot - pos > r->plen && n_out < max_out) {
        const int16_t *c = r->phase + r->phase_id * r->plen;
        uint32_t acc = 0;
        for (size_t k = 0; k < r->plen; k++) { /* filter/utils.c:94-103 */
            acc += (uint32_t)((int32_t)r->pend[pos + k] * (int32_t)c[k]);
        }
        out[n_out++] = mfmo_r14((int32_t)acc); /* utils.c:112 */
        r->phase_id += r->decim;               /* polyphase_fir.c:206-211 */
        pos += r->phase_id / r->interp;
        r->phase_id %= r->interp;
        if (pos > tot) {
            fprintf(stderr, "mfm_oracle: resampler stepped past the end (decimate/interpolate > phase length)\n");
            abort();
        }
    }
    memmove(r->pend, r->pend + pos, (tot - pos) * sizeof(int16_t));
    r->pend_n = tot - pos;
    return n_out;
}

void mfmo_dc_blocker_init(struct mfmo_dc_blocker *b, double pole)
{
    memset(b, 0, sizeof(*b));
    b->p = (int16_t)((1.0 - pole) * (double)(1 << MFMO_Q_SHIFT)); /* filter/dc_blocker.h:56 */
}

void mfmo_dc_blocker_apply(struct mfmo_dc_blocker *b, int16_t *samples, size_t nr_samples)
{
    for (size_t i = 0; i < nr_samples; i++) { /* filter/dc_blocker.h:80-90, int32 wrap-around */
        b->acc = (int32_t)((uint32_t)b->acc - (uint32_t)b->x_n_1);
        b->x_n_1 = (int32_t)((uint32_t)(int32_t)samples[i] << MFMO_Q_SHIFT);
        b->acc = (int32_t)((uint32_t)b->acc + (uint32_t)b->x_n_1 - (uint32_t)((int32_t)b->p * b->y_n_1));
        b->y_n_1 = b->acc >> MFMO_Q_SHIFT;
        samples[i] = (int16_t)b->y_n_1;
    }
}

void mfmo_resampler_quantize_taps(const double *taps, size_t n, int16_t *out)
{
    for (size_t i = 0; i < n; i++) {
        out[i] = (int16_t)(taps[i] * (double)(1 << MFMO_Q_SHIFT)); /* decoder/decoder.c:530-533 */
    }
}

/* ---- 8-bit ingest ---------------------------------------------------------------------------------------- */

void mfmo_unpack_bytes(const uint8_t *in, size_t nr_bytes, int format, int16_t *out)
{
    if (format == 3) {
        /* rtl_sdr_if.c:156-158 (the generic branch; the NEON branch computes the same values) */
        for (size_t i = 0; i < nr_bytes; i++) {
            out[i] = (int16_t)(((int16_t)in[i] - 127) << 7);
        }
        return;
    }
    const int8_t *sin = (const int8_t *)in; /* file_if.c:76,122: the bounce buffer is read through an int8_t pointer */
    const size_t rem = nr_bytes % 4, body = nr_bytes - rem;
    for (size_t i = 0; i < body; i++) {
        out[i] = (format == 2) ? (int16_t)((int16_t)sin[i] - 127) : (int16_t)sin[i]; /* :91-96, :139-144 */
    }
    /* :98-102 / :146-150: the remainder is stored as a bare cast in BOTH formats.  (The 4-wide loop of the
     * reference has already run over these elements and past the end of the read; its results there are
     * overwritten here.) */
    for (size_t i = body; i < nr_bytes; i++) {
        out[i] = (int16_t)sin[i];
    }
}
