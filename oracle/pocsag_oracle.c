/*
 * pocsag_oracle.c - see pocsag_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * A restatement, not a copy: the reference's control flow is kept (so that every quirk survives) but the code is
 * written against this file's own flat state structs.
 */
#include "pocsag_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdbool.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------
 * BCH(31,21)
 * ---------------------------------------------------------------------------------------------------------- */

#define BCH_N 31

static int g_alpha_to[32], g_index_of[32];
static pthread_once_t g_bch_once = PTHREAD_ONCE_INIT;

/* bch_code.c:41-72 with p[] = {1,0,1,0,0,1}: x^5 = x^2 + 1. */
void mfmo_bch_tables(int alpha_to[32], int index_of[32])
{
    static const int p[6] = { 1, 0, 1, 0, 0, 1 };
    const int m = 5;
    int top = 0;

    memset(alpha_to, 0, 32 * sizeof(int));
    memset(index_of, 0, 32 * sizeof(int));
    for (int i = 0; i < m; i++) {
        alpha_to[i] = 1 << i;
        index_of[1 << i] = i;
        if (p[i]) {
            top ^= 1 << i;
        }
    }
    alpha_to[m] = top;
    index_of[top] = m;
    const int msb = 1 << (m - 1);
    for (int i = m + 1; i < BCH_N; i++) {
        int prev = alpha_to[i - 1];
        alpha_to[i] = (prev >= msb) ? (top ^ ((prev ^ msb) << 1)) : (prev << 1);
        index_of[alpha_to[i]] = i;
    }
    index_of[0] = -1;
}

static void bch_init_once(void)
{
    mfmo_bch_tables(g_alpha_to, g_index_of);
}

/* bch_code.c:307-398 */
int mfmo_bch3121_decode(uint32_t *word)
{
    pthread_once(&g_bch_once, bch_init_once);
    const int *a = g_alpha_to, *lg = g_index_of;
    uint32_t r = *word;
    int s[5], any = 0, rc = 0;

    /* :322-339 - four syndromes, then to index form */
    for (int i = 1; i <= 4; i++) {
        int acc = 0;
        for (int j = 0; j < BCH_N; j++) {
            if ((r >> (BCH_N - 1 - j)) & 1) {
                acc ^= a[(i * j) % BCH_N];
            }
        }
        any |= (acc != 0);
        s[i] = lg[acc];
    }

    if (any) {
        if (s[1] != -1) {
            int s1cubed = (s[1] * 3) % BCH_N;
            if (s[3] == s1cubed) {
                /* :344-346 - one error */
                r ^= 1u << (BCH_N - 1 - s[1]);
            } else {
                /* :347-389 - assume two: sigma(x) = 1 + s1 x + ((s1^3 + s3) / s1) x^2, in the scaled form the
                 * reference uses, then a Chien search over all 31 positions */
                int aux = a[s1cubed] ^ ((s[3] != -1) ? a[s[3]] : 0);
                int reg1 = (s[2] - lg[aux] + BCH_N) % BCH_N;
                int reg2 = (s[1] - lg[aux] + BCH_N) % BCH_N;
                int loc[3], count = 0;
                for (int i = 1; i <= BCH_N; i++) {
                    reg1 = (reg1 + 1) % BCH_N;
                    reg2 = (reg2 + 2) % BCH_N;
                    if ((1 ^ a[reg1] ^ a[reg2]) == 0) {
                        if (count < 3) {
                            loc[count] = i % BCH_N;
                        }
                        count++;
                    }
                }
                if (count == 2) {
                    r ^= 1u << (BCH_N - 1 - loc[0]);
                    r ^= 1u << (BCH_N - 1 - loc[1]);
                } else {
                    rc = 1;
                }
            }
        } else if (s[2] != -1) {
            /* :391-393 - unreachable in practice (s2 = s1^2), kept because the reference has it */
            rc = 1;
        }
        /* s1 == 0 but s3 != 0: falls through with rc 0 and the word untouched (reference behaviour) */
    }
    *word = r;
    return rc;
}

struct bch_job {
    uint32_t *w;
    uint8_t *rc;
    size_t n;
};

static void *bch_job_main(void *arg)
{
    struct bch_job *j = arg;
    for (size_t i = 0; i < j->n; i++) {
        j->rc[i] = (uint8_t)mfmo_bch3121_decode(&j->w[i]);
    }
    return NULL;
}

void mfmo_bch3121_decode_batch(uint32_t *words, uint8_t *rc, size_t n, unsigned threads)
{
    if (threads < 1) {
        threads = 1;
    }
    if (threads > 256) {
        threads = 256;
    }
    pthread_t tid[256];
    struct bch_job jobs[256];
    size_t per = (n + threads - 1) / threads, pos = 0;
    unsigned started = 0;
    for (unsigned t = 0; t < threads && pos < n; t++) {
        size_t m = (n - pos < per) ? n - pos : per;
        jobs[t] = (struct bch_job){ words + pos, rc + pos, m };
        pos += m;
        if (threads == 1) {
            bch_job_main(&jobs[t]);
        } else if (pthread_create(&tid[t], NULL, bch_job_main, &jobs[t])) {
            bch_job_main(&jobs[t]);
            tid[t] = 0;
        }
        started++;
    }
    if (threads > 1) {
        for (unsigned t = 0; t < started; t++) {
            if (tid[t]) {
                pthread_join(tid[t], NULL);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * POCSAG message layer
 * ---------------------------------------------------------------------------------------------------------- */

#define SYNC_CODEWORD 0x7cd215d8u /* pager_pocsag_priv.h:40 */
#define IDLE_CODEWORD 0x6983915eu /* pager_pocsag_priv.h:46 */

enum { MSG_NONE = 0, MSG_UNKNOWN = 1, MSG_ALPHA = 2, MSG_NUMERIC = 3 };

struct mfmo_pocsag_msgdec {
    char alpha[512];
    size_t n_alpha;
    int score_alpha;
    bool seen_nonprint;
    char numeric[512];
    size_t n_numeric;
    uint32_t cap_code;
    uint32_t reg_alpha;
    size_t bits_alpha;
    uint32_t reg_numeric;
    size_t bits_numeric;
    uint8_t function;
    bool early_termination;
    int msg_type;
};

struct sink {
    struct mfmo_pocsag_msg *msgs;
    size_t max_msgs, *nr_msgs;
    struct mfmo_pocsag_event *ev;
    size_t max_ev, *nr_ev;
};

/* pager_pocsag.c:46-61 */
static void msgdec_reset(struct mfmo_pocsag_msgdec *d)
{
    d->reg_numeric = 0;
    d->bits_numeric = 0;
    d->n_numeric = 0;
    d->reg_alpha = 0;
    d->bits_alpha = 0;
    d->n_alpha = 0;
    d->seen_nonprint = false;
    d->score_alpha = 0;
    d->early_termination = false;
    d->msg_type = MSG_NONE;
    d->function = 0;
}

static void emit_msg(struct sink *s, int type, uint32_t baud, uint32_t cap, uint32_t function, const char *text,
                     size_t len, uint64_t sample)
{
    if (s->msgs && *s->nr_msgs < s->max_msgs) {
        struct mfmo_pocsag_msg *m = &s->msgs[*s->nr_msgs];
        memset(m, 0, sizeof(*m));
        m->type = (uint32_t)type;
        m->baud = baud;
        m->capcode = cap;
        m->function = function;
        m->len = (uint32_t)len;
        m->sample = sample;
        memcpy(m->text, text, len < 511 ? len : 511);
    }
    (*s->nr_msgs)++;
}

/* pager_pocsag.c:242-297 */
static void msgdec_deliver(struct mfmo_pocsag_msgdec *d, struct sink *s, uint32_t baud, uint64_t sample)
{
    if (d->msg_type == MSG_NONE) {
        return;
    }
    if (d->n_alpha != 0) {
        char last = d->alpha[d->n_alpha - 1];
        if (last == 0x4 || last == 0x3 || last == 0x0 || last == 0x17) {
            d->score_alpha = 1;
        }
    }
    if (d->n_numeric > 40) {
        d->score_alpha = 1;
    }
    d->msg_type = (d->score_alpha > 0) ? MSG_ALPHA : MSG_NUMERIC;
    if (d->msg_type == MSG_ALPHA) {
        emit_msg(s, MSG_ALPHA, baud, d->cap_code, d->function, d->alpha, d->n_alpha, sample);
    } else {
        emit_msg(s, MSG_NUMERIC, baud, d->cap_code, d->function, d->numeric, d->n_numeric, sample);
    }
    msgdec_reset(d);
}

static const char numeric_charmap[16] = { '0', '1', '2', '3', '4', '5', '6', '7', '8', '9', 'X', 'U', ' ', '-', '[', ']' };

/* pager_pocsag.c:319-432; returns the number of words accepted (16 = all) */
static unsigned msgdec_process_batch(struct mfmo_pocsag_msgdec *d, const uint32_t *batch, struct sink *s,
                                     uint32_t baud, uint64_t sample)
{
    for (unsigned z = 0; z < 16; z++) {
        uint32_t w = batch[z] & 0x7fffffffu;
        if (mfmo_bch3121_decode(&w)) {
            /* :334-346 - the rest of the batch is dropped */
            if (d->msg_type != MSG_NONE) {
                d->early_termination = true;
                msgdec_deliver(d, s, baud, sample);
            }
            return z;
        }
        if (w == IDLE_CODEWORD) {
            if (d->msg_type != MSG_NONE) {
                msgdec_deliver(d, s, baud, sample);
            }
            continue;
        }
        if ((w & 1) == 0) {
            /* :358-365 - address word; the 18 address bits are used as they sit in the word (LSB first) */
            msgdec_deliver(d, s, baud, sample);
            d->msg_type = MSG_UNKNOWN;
            d->function = (w >> 19) & 0x3;
            d->cap_code = (((w >> 1) & ((1u << 18) - 1)) << 3) + ((z >> 1) & 0x7);
        } else if (d->msg_type == MSG_UNKNOWN) {
            uint32_t val = (w >> 1) & 0xfffffu;
            /* :374-399 - 7-bit characters, LSB first.  The reference writes message_alpha[next_byte_alpha++]
             * with no bound (undefined past 511); this restatement stops storing at 511 characters. */
            d->reg_alpha |= val << d->bits_alpha;
            d->bits_alpha += 20;
            while (d->bits_alpha >= 7) {
                char c = (char)(d->reg_alpha & 0x7f);
                if (d->n_alpha < 511) {
                    d->alpha[d->n_alpha++] = c;
                }
                if (isprint((unsigned char)c) || c == 0xa || c == 0xd) {
                    if (!d->seen_nonprint) {
                        d->score_alpha++;
                    }
                } else {
                    d->seen_nonprint = true;
                    if (c != 0x03 && c != 0x04 && c != 0x17 && c != 0x0) {
                        d->score_alpha -= 10;
                    }
                }
                d->reg_alpha >>= 7;
                d->bits_alpha -= 7;
            }
            /* :401-415 - the same 20 bits as BCD digits */
            if (d->n_numeric < 511) {
                d->reg_numeric |= val << d->bits_numeric;
                d->bits_numeric += 20;
                while (d->bits_numeric >= 4 && d->n_numeric < 511) {
                    d->numeric[d->n_numeric++] = numeric_charmap[d->reg_numeric & 0xf];
                    d->reg_numeric >>= 4;
                    d->bits_numeric -= 4;
                }
            }
        }
    }
    return 16;
}

struct mfmo_pocsag_msgdec *mfmo_pocsag_msgdec_new(void)
{
    struct mfmo_pocsag_msgdec *d = calloc(1, sizeof(*d));
    if (d) {
        msgdec_reset(d);
    }
    return d;
}

void mfmo_pocsag_msgdec_free(struct mfmo_pocsag_msgdec *d)
{
    free(d);
}

int mfmo_pocsag_msgdec_batch(struct mfmo_pocsag_msgdec *d, const uint32_t *words, int flush, uint32_t baud,
                             uint64_t sample, struct mfmo_pocsag_msg *msgs, size_t max_msgs, size_t *nr_msgs)
{
    struct sink s = { msgs, max_msgs, nr_msgs, NULL, 0, NULL };
    if (words) {
        return (int)msgdec_process_batch(d, words, &s, baud, sample);
    }
    if (flush) {
        msgdec_deliver(d, &s, baud, sample);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * POCSAG slicer / sync state machine
 * ---------------------------------------------------------------------------------------------------------- */

enum { ST_SEARCH = 0, ST_SYNCHRONIZED = 1, ST_BATCH_RECEIVE = 2, ST_SEARCH_SYNCWORD = 3 };

struct baud_detect {
    uint32_t samples_per_bit;
    uint16_t baud_rate;
    uint32_t cur_word;
    uint32_t nr_eye_matches;
    uint32_t eye_detect[75];
};

struct mfmo_pocsag {
    uint16_t sample_skip;
    uint16_t baud_rate;
    /* batch */
    uint16_t b_cur_sample_skip;
    uint32_t b_words[16];
    uint16_t b_word;
    uint16_t b_word_bit;
    uint16_t b_bit_count;
    /* sync search */
    uint16_t s_cur_sample_skip;
    size_t s_nr_bits;
    uint32_t s_word;
    struct baud_detect det[3];
    struct mfmo_pocsag_msgdec dec;
    int state;
    uint64_t pos; /* absolute sample counter (not in the reference; for event stamps) */
};

static bool sync_ok(uint32_t w)
{
    return __builtin_popcount(w ^ SYNC_CODEWORD) <= 4; /* :39-43 */
}

static void batch_reset(struct mfmo_pocsag *p)
{
    memset(p->b_words, 0, sizeof(p->b_words)); /* :63-71 */
    p->b_word = 0;
    p->b_word_bit = 0;
    p->b_cur_sample_skip = 0;
    p->b_bit_count = 0;
}

static void baud_search_reset(struct mfmo_pocsag *p)
{
    static const uint32_t spb[3] = { 75, 32, 16 }; /* 38400 / {512, 1200, 2400}, :128-139 */
    static const uint16_t baud[3] = { 512, 1200, 2400 };
    for (int i = 0; i < 3; i++) {
        memset(&p->det[i], 0, sizeof(p->det[i]));
        p->det[i].samples_per_bit = spb[i];
        p->det[i].baud_rate = baud[i];
    }
}

static void emit_event(struct sink *s, const struct mfmo_pocsag_event *e)
{
    if (s->ev && *s->nr_ev < s->max_ev) {
        s->ev[*s->nr_ev] = *e;
    }
    (*s->nr_ev)++;
}

/* :81-117 */
static void baud_on_sample(struct mfmo_pocsag *p, struct baud_detect *d, int16_t sample, struct sink *s)
{
    uint32_t bit = sample < 0 ? 1 : 0;
    uint32_t *w = &d->eye_detect[d->cur_word];
    *w = (*w << 1) | bit;
    if (sync_ok(*w)) {
        d->nr_eye_matches++;
    } else if (d->nr_eye_matches > d->samples_per_bit / 2) {
        p->sample_skip = (uint16_t)d->samples_per_bit;
        p->baud_rate = d->baud_rate;
        batch_reset(p);
        p->b_cur_sample_skip = (uint16_t)(d->nr_eye_matches / 2);
        p->state = ST_SYNCHRONIZED;
        struct mfmo_pocsag_event e;
        memset(&e, 0, sizeof(e));
        e.type = MFMO_POCSAG_EV_SYNC_FOUND;
        e.baud = d->baud_rate;
        e.sample = p->pos;
        e.aux = d->nr_eye_matches;
        emit_event(s, &e);
    } else {
        d->nr_eye_matches = 0;
    }
    d->cur_word = (d->cur_word + 1) % d->samples_per_bit;
}

struct mfmo_pocsag *mfmo_pocsag_new(void)
{
    struct mfmo_pocsag *p = calloc(1, sizeof(*p));
    if (!p) {
        return NULL;
    }
    baud_search_reset(p); /* :182-183 */
    msgdec_reset(&p->dec);
    p->state = ST_SEARCH;
    return p;
}

void mfmo_pocsag_free(struct mfmo_pocsag *p)
{
    free(p);
}

/* :434-543 */
int mfmo_pocsag_on_pcm(struct mfmo_pocsag *p, const int16_t *pcm, size_t nr_samples,
                       struct mfmo_pocsag_event *ev, size_t max_ev, size_t *nr_ev,
                       struct mfmo_pocsag_msg *msgs, size_t max_msgs, size_t *nr_msgs)
{
    struct sink s = { msgs, max_msgs, nr_msgs, ev, max_ev, nr_ev };
    size_t next = 0;

    while (next < nr_samples) {
        switch (p->state) {
        case ST_SEARCH:
            while (next < nr_samples) {
                /* all three detectors see the sample, in this order, even when an earlier one already fired */
                for (int i = 0; i < 3; i++) {
                    baud_on_sample(p, &p->det[i], pcm[next], &s);
                }
                next++;
                p->pos++;
                if (p->state == ST_SYNCHRONIZED) {
                    break;
                }
            }
            break;
        case ST_SYNCHRONIZED:
            p->state = ST_BATCH_RECEIVE;
            /* fall through */
        case ST_BATCH_RECEIVE:
            while (next < nr_samples) {
                bool done = false;
                if (++p->b_cur_sample_skip == p->sample_skip) {
                    uint32_t bit = pcm[next] < 0 ? 1 : 0;
                    /* :477 - `bit << bit_count` with bit_count up to 511; x86 masks the count to 5 bits */
                    p->b_words[p->b_word] |= bit << (p->b_bit_count & 31);
                    p->b_word_bit++;
                    p->b_bit_count++;
                    p->b_cur_sample_skip = 0;
                    if (p->b_word_bit == 32) {
                        p->b_word_bit = 0;
                        p->b_word++;
                        if (p->b_word == 16) {
                            struct mfmo_pocsag_event e;
                            memset(&e, 0, sizeof(e));
                            e.type = MFMO_POCSAG_EV_BATCH;
                            e.baud = p->baud_rate;
                            e.sample = p->pos;
                            for (int z = 0; z < 16; z++) {
                                uint32_t w = p->b_words[z] & 0x7fffffffu;
                                e.raw[z] = p->b_words[z];
                                if (mfmo_bch3121_decode(&w)) {
                                    e.fail_mask |= 1u << z;
                                }
                                e.corrected[z] = w;
                            }
                            e.nr_ok = msgdec_process_batch(&p->dec, p->b_words, &s, p->baud_rate, p->pos);
                            emit_event(&s, &e);
                            p->state = ST_SEARCH_SYNCWORD;
                            p->b_word_bit = 0;
                            p->b_word = 0;
                            p->s_cur_sample_skip = 0; /* :73-79 */
                            p->s_nr_bits = 0;
                            p->s_word = 0;
                            done = true;
                        }
                    }
                }
                next++;
                p->pos++;
                if (done) {
                    break;
                }
            }
            break;
        case ST_SEARCH_SYNCWORD:
            while (next < nr_samples) {
                bool done = false;
                if (++p->s_cur_sample_skip == p->sample_skip) {
                    p->s_cur_sample_skip = 0;
                    p->s_word = (p->s_word << 1) | (pcm[next] < 0 ? 1u : 0u);
                    p->s_nr_bits++;
                    if (p->s_nr_bits == 32) {
                        struct mfmo_pocsag_event e;
                        memset(&e, 0, sizeof(e));
                        e.baud = p->baud_rate;
                        e.sample = p->pos;
                        e.aux = p->s_word;
                        if (!sync_ok(p->s_word)) {
                            e.type = MFMO_POCSAG_EV_SYNC_LOST;
                            emit_event(&s, &e);
                            p->state = ST_SEARCH;
                            p->sample_skip = 0;
                            baud_search_reset(p);
                            msgdec_deliver(&p->dec, &s, p->baud_rate, p->pos);
                        } else {
                            e.type = MFMO_POCSAG_EV_SYNC_KEPT;
                            emit_event(&s, &e);
                            p->state = ST_BATCH_RECEIVE;
                            batch_reset(p);
                        }
                        done = true;
                    }
                }
                next++;
                p->pos++;
                if (done) {
                    break;
                }
            }
            break;
        }
    }
    return 0;
}

/* ---- Mueller-Muller clock recovery: pager/mueller_muller.c ---- */

void mfmo_mm_init(struct mfmo_mm *mm, float kw, float km, float samples_per_bit, float error_min, float error_max)
{
    /* mueller_muller.c:17-26 */
    memset(mm, 0, sizeof(*mm));
    mm->next_offset = 0.0f;
    mm->m = mm->w = mm->ideal_step_size = samples_per_bit;
    mm->kw = kw;
    mm->km = km;
    mm->error_min = error_min;
    mm->error_max = error_max;
    mm->samples_per_bit = samples_per_bit;
}

static float mm_sign(float v)
{
    return (float)(v > 0) - (float)(v < 0); /* :36-39 */
}

size_t mfmo_mm_process(struct mfmo_mm *mm, const int16_t *samples, size_t nr_samples, int16_t *decisions,
                       size_t max_decisions)
{
    float cur_sample = mm->next_offset, nr_samples_f = (float)nr_samples, w = mm->w, m = mm->m; /* :57-60 */
    size_t cur = 0;
    while (cur_sample < nr_samples_f && cur < max_decisions) { /* :66 */
        const float sample = samples[(size_t)(cur_sample + 0.5f)]; /* :67 */
        decisions[cur++] = (int16_t)sample;                         /* :71 */
        /* :77 */
        const float w_error = mm_sign(mm->last_sample) * sample - mm_sign(sample) * mm->last_sample;
        w += w_error * mm->kw; /* :80 */
        if (mm->error_min > w) { /* :87-91 */
            w = mm->error_min;
        } else if (mm->error_max < w) {
            w = mm->error_max;
        }
        m += w + mm->km * sample;   /* :93 */
        cur_sample += floorf(m);     /* :96 */
        m -= floorf(m);              /* :98 */
        mm->last_sample = sample;    /* :101 */
    }
    mm->next_offset = cur_sample - nr_samples_f; /* :109-111 */
    mm->w = w;
    mm->m = m;
    return cur;
}
