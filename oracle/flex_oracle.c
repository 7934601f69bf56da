/*
 * flex_oracle.c - CPU restatement of pager/pager_flex.c (TEST INFRASTRUCTURE ONLY, see flex_oracle.h for the
 * pinning status and the two places where undefined behaviour of the reference is made definite).
 *
 * The sample walk keeps the reference's structure - one call per PCM sample, the same counters with the same
 * widths - so that every quirk (uint8_t run counter, int16_t swing arithmetic, sample counter that is set to
 * half the run length) falls out of the types rather than out of a re-derivation.  Citations are to
 * pager/pager_flex.c unless another file is named.
 */
#include "flex_oracle.h"

#include <stdbool.h>
#include <stdlib.h>
#include <string.h>

#include "pocsag_oracle.h" /* mfmo_bch3121_decode: the same BCH(31,21) object the reference builds at :1364 */

/* :46-96 */
static const struct mfmo_flex_coding flex_codings[4] = {
    { .seq_a = 0x78f3, .baud = 1600, .fsk_levels = 2, .sample_skip = 9, .sync_2_samples = 4, .sym_bits = 1,
      .sample_fudge = 0, .nr_phases = 1, .symbols_per_block = 2816 },
    { .seq_a = 0x84e7, .baud = 3200, .fsk_levels = 2, .sample_skip = 4, .sync_2_samples = 24, .sym_bits = 1,
      .sample_fudge = 2, .nr_phases = 2, .symbols_per_block = 5632 },
    { .seq_a = 0x4f97, .baud = 3200, .fsk_levels = 4, .sample_skip = 9, .sync_2_samples = 12, .sym_bits = 2,
      .sample_fudge = 0, .nr_phases = 2, .symbols_per_block = 2816 },
    { .seq_a = 0x215f, .baud = 6400, .fsk_levels = 4, .sample_skip = 4, .sync_2_samples = 32, .sym_bits = 2,
      .sample_fudge = 2, .nr_phases = 4, .symbols_per_block = 5632 },
};

const struct mfmo_flex_coding *mfmo_flex_coding(unsigned idx)
{
    return idx < 4 ? &flex_codings[idx] : NULL;
}

/* :107-119 - sum of the six nibbles of the low 21 bits, modulo 16 */
static unsigned word_checksum(uint32_t w)
{
    unsigned sum = 0;
    w &= 0x1fffff;
    for (int n = 0; n < 6; n++) {
        sum += (w >> (4 * n)) & 0xf;
    }
    return sum & 0xf;
}

/* ------------------------------------------------------------------------------------------------------------
 * message layer (:527-1198)
 * ---------------------------------------------------------------------------------------------------------- */

struct msg_sink {
    struct mfmo_flex_msg *msgs;
    size_t max, *nr;
};

struct msg_ctx {
    struct msg_sink *sink;
    uint32_t baud, phase, cycle, frame;
    uint64_t sample;
    char buf[256]; /* flex->msg_buf (pager_flex_priv.h:317) */
    size_t len;    /* flex->msg_len */
};

static void put_msg(struct msg_ctx *c, uint32_t kind, uint64_t capcode, uint32_t a0, uint32_t a1, uint32_t a2,
                    bool with_text)
{
    struct msg_sink *s = c->sink;
    if (s->msgs && *s->nr < s->max) {
        struct mfmo_flex_msg *m = &s->msgs[*s->nr];
        memset(m, 0, sizeof(*m));
        m->kind = kind;
        m->baud = c->baud;
        m->phase = c->phase;
        m->cycle = c->cycle;
        m->frame = c->frame;
        m->aux0 = a0;
        m->aux1 = a1;
        m->aux2 = a2;
        m->capcode = capcode;
        m->sample = c->sample;
        if (with_text) {
            m->len = (uint32_t)c->len;
            memcpy(m->text, c->buf, c->len);
        }
    }
    (*s->nr)++;
}

/* a copy of word `idx` of the phase run through bch_code_decode; -1 when uncorrectable or outside the phase */
static int fetch_fixed(const uint32_t *words, size_t idx, uint32_t *out)
{
    if (idx >= MFMO_FLEX_PHASE_WORDS) {
        return -1;
    }
    uint32_t w = words[idx];
    if (mfmo_bch3121_decode(&w)) {
        return -1;
    }
    *out = w;
    return 0;
}

static const char num_lut[16] = { '0', '1', '2', '3', '4', '5', '6', '7', '8', '9', 'X', 'U', ' ', '-', ']', '[' }; /* :686-704 */

/* :597-681.  words = base + word_start of the reference, i.e. word i of the message is words[start + i]. */
static int decode_alphanumeric(struct msg_ctx *c, uint64_t capcode, uint32_t long_word, const uint32_t *words, size_t start,
                               size_t nr_words)
{
    size_t first = 1;
    int skip = 0;
    uint32_t status;
    bool fragment, maildrop = false;
    unsigned seq;

    if (long_word != 0xffffffffu) {
        first = 0;
        status = long_word;
    } else if (fetch_fixed(words, start, &status)) {
        return -1;
    }
    fragment = (status >> 10) & 1;
    seq = (status >> 11) & 3;
    if (seq == 3) {
        skip = 1;
        maildrop = (status >> 20) & 1;
    }
    for (size_t i = first; i < nr_words; i++) {
        uint32_t cw;
        if (fetch_fixed(words, start + i, &cw)) {
            return -1;
        }
        if (skip) {
            cw >>= 7;
        }
        for (int j = skip; j < 3; j++) {
            char ch = (char)(cw & 0x7f);
            if (ch == 0x3) {
                break;
            }
            c->buf[c->len++] = ch;
            if (c->len == 255) {
                break;
            }
            cw >>= 7;
        }
        skip = 0;
        if (c->len == 255) {
            break;
        }
    }
    put_msg(c, MFMO_FLEX_MSG_ALNUM, capcode, (uint32_t)fragment | (uint32_t)maildrop << 1 | seq << 2, 0, 0, true);
    return 0;
}

/* :709-824 - 4-bit digits packed across 21-bit words, the first word giving 19 bits */
static int decode_numeric(struct msg_ctx *c, uint64_t capcode, uint32_t long_word, const uint32_t *words, size_t start,
                          size_t nr_words)
{
    uint32_t cur = 0, next = 0;
    size_t nr_bits = nr_words * 21, cur_bits = 19, next_offs = 0, next_bits = 21;

    if (long_word != 0xffffffffu) {
        cur = (long_word & 0x1fffff) >> 2;
        nr_bits += 19;
    } else {
        if (fetch_fixed(words, start, &cur)) {
            return -1;
        }
        cur = (cur & 0x1fffff) >> 2;
        nr_bits -= 2;
        next_offs = 1;
    }
    if (next_offs < nr_words) {
        if (fetch_fixed(words, start + next_offs, &next)) {
            return -1;
        }
        next &= 0x1fffff;
    }
    nr_bits &= ~(size_t)3;

    do {
        const size_t whole = cur_bits & ~(size_t)3;
        for (size_t i = 0; i < whole; i += 4) {
            c->buf[c->len++] = num_lut[cur & 0xf];
            if (c->len == 255) {
                break;
            }
            cur >>= 4;
            cur_bits -= 4;
            nr_bits -= 4;
        }
        if (c->len == 255) {
            break;
        }
        if (cur_bits != 0 && nr_bits != 0) {
            /* top up the 1..3 left-over bits to a digit with the low bits of the next word */
            switch (cur_bits) {
            case 1:
                cur |= (next & 0x7) << 1;
                next >>= 3;
                next_bits -= 3;
                break;
            case 2:
                cur |= (next & 0x3) << 2;
                next >>= 2;
                next_bits -= 2;
                break;
            case 3:
                cur |= (next & 0x1) << 3;
                next >>= 1;
                next_bits -= 1;
                break;
            }
            cur_bits = 4;
        } else if (cur_bits == 0 && nr_bits != 0) {
            cur = next;
            cur_bits = next_bits;
            next_bits = 21;
            next_offs++;
            if (next_offs < nr_words) {
                if (fetch_fixed(words, start + next_offs, &next)) {
                    return -1;
                }
                next &= 0x1fffff;
            }
        }
    } while (nr_bits != 0);

    put_msg(c, MFMO_FLEX_MSG_NUM, capcode, 0, 0, 0, true);
    return 0;
}

/* :829-883 */
static int decode_tone(struct msg_ctx *c, uint64_t capcode, uint32_t first, uint32_t second)
{
    first &= 0x1fffff;
    const unsigned type = (first >> 7) & 3;
    switch (type) {
    case 0: /* three digits in the vector word, five more in the second one when the address is long */
        first >>= 9;
        for (int i = 0; i < 3; i++) {
            c->buf[c->len++] = num_lut[first & 0xf];
            first >>= 4;
        }
        if (second != 0xffffffffu) {
            second &= 0x1fffff;
            for (int i = 0; i < 5; i++) {
                c->buf[c->len++] = num_lut[second & 0xf];
                second >>= 4;
            }
        }
        put_msg(c, MFMO_FLEX_MSG_NUM, capcode, 0, 0, 0, true);
        return 0;
    case 1:
    case 2:
        put_msg(c, MFMO_FLEX_NOTE_TONE, capcode, type, first, second, false);
        return 0;
    default:
        return -1;
    }
}

/* :885-933 */
static int decode_short_instruction(struct msg_ctx *c, uint64_t capcode, uint32_t v)
{
    v &= 0x7fffff;
    if (word_checksum(v) != 0xf) {
        return -1;
    }
    put_msg(c, MFMO_FLEX_MSG_SIV, capcode, (v >> 7) & 0x7, (v >> 10) & 0x7ff, 0, false);
    return 0;
}

/* :938-1033 - vec_offs indexes the phase's words (the reference passes &phase_words[vec_offs] and phase_words) */
static int decode_vector(struct msg_ctx *c, uint64_t capcode, uint32_t *words, size_t vec_offs, size_t nr_vec)
{
    c->len = 0;
    for (size_t i = 0; i < nr_vec; i++) {
        if (vec_offs + i >= MFMO_FLEX_PHASE_WORDS) {
            return -1;
        }
        if (mfmo_bch3121_decode(&words[vec_offs + i])) { /* corrected in place, :959 */
            return -1;
        }
    }
    const uint32_t v = words[vec_offs];
    if (word_checksum(v) != 0xf) {
        return -1;
    }
    const unsigned type = (v >> 4) & 0x7;
    const size_t start = (v >> 7) & 0x7f;
    const uint32_t long_word = (nr_vec == 2) ? words[vec_offs + 1] : 0xffffffffu;
    size_t length;

    switch (type) {
    case 0x2: /* tone */
        return decode_tone(c, capcode, v, long_word);
    case 0x3: /* standard numeric */
        length = ((v >> 14) & 0x7) + 1;
        if (nr_vec == 2) {
            length -= 1;
        }
        return decode_numeric(c, capcode, long_word, words, start, length);
    case 0x5: /* alphanumeric */
        length = (v >> 14) & 0x7f;
        if (nr_vec == 2) {
            length -= 1; /* wraps for a zero length field, as in the reference (:1005) */
        }
        return decode_alphanumeric(c, capcode, long_word, words, start, length);
    case 0x1: /* short instruction */
        return decode_short_instruction(c, capcode, v);
    default: /* secure, special numeric, hex, numbered numeric: logged, not decoded (:1019-1024) */
        put_msg(c, MFMO_FLEX_NOTE_UNSUPPORTED, capcode, type, 0, 0, false);
        return 0;
    }
}

/* :527-573 - corrects (and masks) the address word(s) in place */
static int decode_address(uint32_t *addr, uint64_t *capcode, size_t *extra)
{
    *capcode = 0;
    *extra = 0;
    if (mfmo_bch3121_decode(&addr[0])) {
        return -1;
    }
    addr[0] &= 0x1fffff;
    const uint32_t first = addr[0];
    if ((first > 0x8000 && first <= 0x1e0000) || (first > 0x1f0000 && first < 0x1f7fff)) {
        *capcode = first - 32768;
        return 0;
    }
    if (mfmo_bch3121_decode(&addr[1])) {
        return -1;
    }
    addr[1] &= 0x1fffff;
    const uint32_t second = addr[1];
    *extra = 1;
    /* uint32_t arithmetic as in the reference (:567), widened afterwards */
    *capcode = (uint32_t)(0x1f9001u + (((0x1fffffu - second) * 32768u) + first - 1u));
    return 0;
}

/* :1041-1086 */
static void extra_biw(struct msg_ctx *c, uint32_t w)
{
    w &= 0x7fffffffu;
    if (mfmo_bch3121_decode(&w)) {
        put_msg(c, MFMO_FLEX_NOTE_EXTRA_BIW, 0, 0, 0, 0, false);
        return;
    }
    w &= 0x1fffff;
    put_msg(c, MFMO_FLEX_NOTE_EXTRA_BIW, 0, word_checksum(w) == 0xf ? 2 : 1, w, 0, false);
}

/* :1088-1198 */
static void phase_process(struct msg_ctx *c, uint32_t *words)
{
    uint32_t biw = words[0] & 0x7fffffffu;
    if (mfmo_bch3121_decode(&biw)) {
        put_msg(c, MFMO_FLEX_NOTE_BIW_BCH, 0, biw, 0, 0, false);
        return;
    }
    if (word_checksum(biw) != 0xf) {
        put_msg(c, MFMO_FLEX_NOTE_BIW_CKSUM, 0, biw, 0, 0, false);
        return;
    }
    const unsigned vsw = (biw >> 10) & 0x3f, eob = (biw >> 8) & 0x3;
    if (eob > vsw) {
        put_msg(c, MFMO_FLEX_NOTE_BIW_COUNT, 0, vsw, eob, 0, false);
        return;
    }
    if (eob != 0) {
        put_msg(c, MFMO_FLEX_NOTE_BIW_EOB, 0, eob, 0, 0, false);
        for (size_t i = 1; i < eob; i++) {
            extra_biw(c, words[i]);
        }
    }
    const size_t addr_start = 1 + eob;
    for (size_t i = addr_start; i < vsw; i++) {
        const size_t vec_offs = i + vsw - addr_start;
        uint64_t capcode;
        size_t extra;
        if (decode_address(&words[i], &capcode, &extra)) {
            put_msg(c, MFMO_FLEX_NOTE_ADDR_ERROR, 0, 0, 0, 0, false);
            return;
        }
        if (decode_vector(c, capcode, words, vec_offs, extra + 1)) {
            put_msg(c, MFMO_FLEX_NOTE_VEC_ERROR, capcode, 0, 0, 0, false);
        }
        i += extra;
    }
}

int mfmo_flex_phase_process(uint32_t words[MFMO_FLEX_PHASE_WORDS], unsigned coding_idx, unsigned phase, unsigned cycle,
                            unsigned frame, uint64_t sample, struct mfmo_flex_msg *msgs, size_t max_msgs,
                            size_t *nr_msgs)
{
    if (coding_idx >= 4 || !nr_msgs) {
        return -1;
    }
    struct msg_sink sink = { msgs, max_msgs, nr_msgs };
    struct msg_ctx c = { .sink = &sink, .baud = flex_codings[coding_idx].baud, .phase = phase, .cycle = cycle,
        .frame = frame, .sample = sample, .len = 0 };
    phase_process(&c, words);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * sample walk (:129-525, :1200-1455)
 * ---------------------------------------------------------------------------------------------------------- */

enum { ST_SYNC_1, ST_SYNC_2, ST_BLOCK };                                       /* pager_flex_priv.h:13-34 */
enum { S1_SEARCH_BS1, S1_BS1, S1_A, S1_B, S1_INV_A, S1_FIW, S1_SYNCED };       /* :36-73 */
enum { S2_COMMA, S2_C, S2_INV_COMMA, S2_INV_C, S2_SYNCED };                    /* :113-138 */

struct phase_buf {
    uint32_t words[MFMO_FLEX_PHASE_WORDS];
    uint8_t cur_bit, cur_word, base_word;
};

struct mfmo_flex {
    /* struct pager_flex */
    int16_t sample_range, sample_delta;
    int state;
    int16_t skip, skip_count;
    uint8_t cycle_id, frame_id;
    /* struct pager_flex_sync */
    uint32_t sync_words[10];
    int s1;
    uint8_t sample_counter, bit_counter;
    uint32_t a;
    uint16_t b;
    uint32_t inv_a, fiw;
    int coding; /* index, -1 = none */
    int32_t sum_high, sum_low;
    unsigned n_high, n_low;
    /* struct pager_flex_sync_2 */
    int s2;
    uint16_t nr_dots, c, inv_c;
    uint8_t nr_c;
    /* struct pager_flex_block */
    struct phase_buf ph[4];
    int32_t nr_symbols;
    bool phase_ff;
    /* bookkeeping of this restatement */
    uint64_t pos;           /* index of the sample being processed */
    uint32_t eye;
    uint64_t sync_sample;
    uint32_t fiw_fixed;
};

struct ev_sink {
    struct mfmo_flex_event *ev;
    size_t max, *nr;
};

/* :173-262 */
static void reset_sync1(struct mfmo_flex *f)
{
    memset(f->sync_words, 0, sizeof(f->sync_words));
    f->s1 = S1_BS1;
    f->sample_counter = 0;
    f->bit_counter = 0;
    f->a = 0;
    f->b = 0;
    f->inv_a = 0;
    f->fiw = 0;
    f->coding = -1;
    f->sum_high = f->sum_low = 0;
    f->n_high = f->n_low = 0;
}

static void reset_all(struct mfmo_flex *f)
{
    f->state = ST_SYNC_1;
    f->skip = 0;
    f->skip_count = 0;
    f->sample_range = 0;
    f->sample_delta = 0;
    f->frame_id = 0;
    f->cycle_id = 0;
    reset_sync1(f);
    f->s2 = S2_COMMA;
    f->nr_dots = 0;
    f->c = 0;
    f->inv_c = 0;
    f->nr_c = 0;
    f->nr_symbols = 0;
    f->phase_ff = false;
    for (int p = 0; p < 4; p++) {
        f->ph[p].cur_bit = 0;
        f->ph[p].cur_word = 0;
        f->ph[p].base_word = 0;
    }
}

static struct mfmo_flex_event *new_event(struct mfmo_flex *f, struct ev_sink *es, uint32_t type)
{
    static struct mfmo_flex_event scratch;
    struct mfmo_flex_event *e = &scratch;
    if (es->ev && *es->nr < es->max) {
        e = &es->ev[*es->nr];
    }
    (*es->nr)++;
    memset(e, 0, sizeof(*e));
    e->type = type;
    e->coding = f->coding < 0 ? 0xffffffffu : (uint32_t)f->coding;
    e->sample = f->pos;
    e->eye = f->eye;
    e->a = f->a;
    e->b = f->b;
    e->inv_a = f->inv_a;
    e->fiw_raw = f->fiw;
    return e;
}

/* :129-138 */
static int slice_2fsk(int16_t sample)
{
    return !((uint16_t)sample >> 15);
}

/* :148-171 */
static int slice_4fsk(const struct mfmo_flex *f, int16_t sample)
{
    sample = (int16_t)(sample - f->sample_delta);
    if (sample < 0) {
        return (-sample > f->sample_range / 4) ? 0 : 1;
    }
    return (sample > f->sample_range / 4) ? 2 : 3;
}

static int slice(const struct mfmo_flex *f, int16_t sample)
{
    return flex_codings[f->coding].fsk_levels == 2 ? slice_2fsk(sample) : slice_4fsk(f, sample);
}

/* :264-287 */
static bool check_baud(struct mfmo_flex *f)
{
    const uint16_t code = (uint16_t)(f->a >> 16), inv_code = (uint16_t)(f->inv_a >> 16);
    for (int i = 0; i < 4; i++) {
        const uint32_t seq = flex_codings[i].seq_a;
        /* the reference computes both XORs in int, so the complement of seq_a has its upper 16 bits set (:278) */
        if (__builtin_popcount(seq ^ code) < 4 || __builtin_popcount(~(int)seq ^ (int)inv_code) < 4) {
            f->coding = i;
            return true;
        }
    }
    return false;
}

static void swing_add(struct mfmo_flex *f, int16_t sample)
{
    if (sample > 0) {
        f->sum_high += sample;
        f->n_high++;
    } else {
        f->sum_low += sample;
        f->n_low++;
    }
}

/* :295-458; returns false when the A words matched no coding (the caller reports it) */
static bool sync1_update(struct mfmo_flex *f, int16_t sample)
{
    bool ok = true;
    f->sample_counter = (uint8_t)((f->sample_counter + 1) % 10);
    const uint32_t bit = (uint32_t)slice_2fsk(sample);

    switch (f->s1) {
    case S1_SEARCH_BS1:
    case S1_BS1: {
        uint32_t *reg = &f->sync_words[f->sample_counter];
        *reg = (*reg << 1) | bit;
        const bool match = (*reg == 0xaaaaaaaau);
        if (f->s1 == S1_SEARCH_BS1) {
            if (match) {
                f->bit_counter = 1;
                f->s1 = S1_BS1;
            }
        } else if (match) {
            f->bit_counter++;
        } else {
            if (f->bit_counter >= 3) {
                f->s1 = S1_A;
                f->eye = f->bit_counter;
                f->sample_counter = f->bit_counter / 2; /* becomes the sampling clock (:339) */
            } else {
                f->s1 = S1_SEARCH_BS1;
            }
            f->bit_counter = 0;
        }
        break;
    }
    case S1_A:
        if (f->sample_counter == 0) {
            f->a = (f->a << 1) | bit;
            swing_add(f, sample);
            if (++f->bit_counter == 32) {
                f->s1 = S1_B;
                f->bit_counter = 0;
            }
        }
        break;
    case S1_B:
        if (f->sample_counter == 0) {
            f->b = (uint16_t)((f->b << 1) | bit);
            swing_add(f, sample);
            if (++f->bit_counter == 16) {
                f->s1 = S1_INV_A;
                f->bit_counter = 0;
            }
        }
        break;
    case S1_INV_A:
        if (f->sample_counter == 0) {
            f->inv_a = (f->inv_a << 1) | bit;
            swing_add(f, sample);
            if (++f->bit_counter == 32) {
                if (check_baud(f)) {
                    f->s1 = S1_FIW;
                } else {
                    ok = false;
                }
                f->bit_counter = 0;
            }
        }
        break;
    case S1_FIW:
        if (f->sample_counter == 0) {
            f->fiw = (f->fiw >> 1) | (bit << 31);
            swing_add(f, sample);
            if (++f->bit_counter == 32) {
                if (f->n_high != 0 && f->n_low != 0) {
                    const int16_t high = (int16_t)(f->sum_high / (int)f->n_high), low = (int16_t)(f->sum_low / (int)f->n_low);
                    f->sample_range = (int16_t)(high - low);
                    f->sample_delta = (int16_t)(high - (int)f->sample_range / 2);
                }
                f->s1 = S1_SYNCED;
            }
        }
        break;
    }
    return ok;
}

/* :1312-1345; 0 = accepted */
static unsigned handle_fiw(struct mfmo_flex *f)
{
    if (f->n_high == 0 || f->n_low == 0) {
        f->fiw_fixed = 0;
        return 3;
    }
    uint32_t w = f->fiw & 0x7fffffffu;
    if (mfmo_bch3121_decode(&w)) {
        f->fiw_fixed = w;
        return 1;
    }
    f->fiw_fixed = w;
    f->cycle_id = (w >> 4) & 0xf;
    f->frame_id = (w >> 8) & 0x7f;
    return word_checksum(w) == 0xf ? 0 : 2;
}

/* :460-525 */
static void sync2_update(struct mfmo_flex *f, int16_t sample)
{
    const struct mfmo_flex_coding *cd = &flex_codings[f->coding];
    switch (f->s2) {
    case S2_COMMA:
        if (++f->nr_dots == cd->sync_2_samples) {
            f->s2 = S2_C;
        }
        break;
    case S2_C:
        f->c = (uint16_t)((f->c << cd->sym_bits) | slice(f, sample));
        f->nr_c += cd->sym_bits;
        if (f->nr_c == 16) {
            f->s2 = S2_INV_COMMA;
            f->nr_dots = 0;
        }
        break;
    case S2_INV_COMMA:
        if (++f->nr_dots == cd->sync_2_samples) {
            f->s2 = S2_INV_C;
            f->nr_c = 0;
        }
        break;
    case S2_INV_C:
        f->inv_c = (uint16_t)((f->inv_c << cd->sym_bits) | slice(f, sample));
        f->nr_c += cd->sym_bits;
        if (f->nr_c == 16) {
            f->s2 = S2_SYNCED;
        }
        break;
    }
}

/* :1200-1222 - words fill LSB first, eight words of a block bit-interleaved */
static void phase_append(struct phase_buf *p, bool bit)
{
    uint32_t *w = &p->words[p->base_word + p->cur_word];
    *w = (*w >> 1) | ((uint32_t)bit << 31);
    p->cur_word = (p->cur_word + 1) % 8;
    if (p->cur_word == 0) {
        p->cur_bit++;
    }
    if (p->cur_bit == 32) {
        p->base_word += 8;
        p->cur_bit = 0;
        p->cur_word = 0;
    }
}

/* :1224-1310 */
static void block_update(struct mfmo_flex *f, int16_t sample, struct ev_sink *es, struct msg_sink *ms)
{
    const struct mfmo_flex_coding *cd = &flex_codings[f->coding];
    const int sym = slice(f, sample);

    switch (cd->nr_phases) {
    case 1:
        phase_append(&f->ph[0], sym == 1);
        break;
    case 2:
        if (cd->fsk_levels == 2) {
            phase_append(&f->ph[f->phase_ff ? 2 : 0], sym == 1);
            f->phase_ff = !f->phase_ff;
        } else {
            phase_append(&f->ph[0], (sym & 2) != 0);
            phase_append(&f->ph[2], (sym & 1) != 0);
        }
        break;
    default:
        phase_append(&f->ph[f->phase_ff ? 2 : 0], (sym & 2) != 0);
        phase_append(&f->ph[f->phase_ff ? 3 : 1], (sym & 1) != 0);
        f->phase_ff = !f->phase_ff;
        break;
    }

    if (++f->nr_symbols != cd->symbols_per_block) {
        return;
    }
    struct mfmo_flex_event *e = new_event(f, es, MFMO_FLEX_EV_FRAME);
    e->sync_sample = f->sync_sample;
    e->fiw = f->fiw_fixed;
    e->sample_range = f->sample_range;
    e->sample_delta = f->sample_delta;
    e->cycle = f->cycle_id;
    e->frame = f->frame_id;
    static const uint8_t order[3][4] = { { 0 }, { 0, 2 }, { 0, 1, 2, 3 } };
    const uint8_t *seq = order[cd->nr_phases == 1 ? 0 : cd->nr_phases == 2 ? 1 : 2];
    for (unsigned k = 0; k < cd->nr_phases; k++) { /* phases this coding does not carry stay zero in the event */
        memcpy(e->words[seq[k]], f->ph[seq[k]].words, sizeof(e->words[0]));
    }
    for (unsigned k = 0; k < cd->nr_phases; k++) {
        struct msg_ctx c = { .sink = ms, .baud = cd->baud, .phase = seq[k], .cycle = f->cycle_id, .frame = f->frame_id,
            .sample = f->pos, .len = 0 };
        phase_process(&c, f->ph[seq[k]].words);
    }
    reset_all(f);
}

struct mfmo_flex *mfmo_flex_new(void)
{
    struct mfmo_flex *f = calloc(1, sizeof(*f));
    if (f) {
        reset_all(f);
    }
    return f;
}

void mfmo_flex_free(struct mfmo_flex *f)
{
    free(f);
}

/* :1401-1455 */
int mfmo_flex_on_pcm(struct mfmo_flex *f, const int16_t *pcm, size_t nr_samples,
                     struct mfmo_flex_event *ev, size_t max_ev, size_t *nr_ev,
                     struct mfmo_flex_msg *msgs, size_t max_msgs, size_t *nr_msgs)
{
    size_t dummy_ev = 0, dummy_msgs = 0;
    struct ev_sink es = { ev, max_ev, nr_ev ? nr_ev : &dummy_ev };
    struct msg_sink ms = { msgs, max_msgs, nr_msgs ? nr_msgs : &dummy_msgs };

    for (size_t i = 0; i < nr_samples; i++, f->pos++) {
        if (f->skip_count != 0) {
            f->skip_count--;
            continue;
        }
        f->skip_count = f->skip;
        switch (f->state) {
        case ST_SYNC_1:
            if (!sync1_update(f, pcm[i])) {
                new_event(f, &es, MFMO_FLEX_EV_BAD_BAUD);
                reset_sync1(f);
            } else if (f->s1 == S1_SYNCED) {
                const unsigned rc = handle_fiw(f);
                if (rc == 0) {
                    f->state = ST_SYNC_2;
                    f->skip = flex_codings[f->coding].sample_skip;
                    f->skip_count = (int16_t)(f->skip + flex_codings[f->coding].sample_fudge);
                    f->sync_sample = f->pos;
                } else {
                    struct mfmo_flex_event *e = new_event(f, &es, MFMO_FLEX_EV_BAD_FIW);
                    e->fiw = f->fiw_fixed;
                    e->fiw_rc = rc;
                    e->sample_range = f->sample_range;
                    e->sample_delta = f->sample_delta;
                    reset_all(f);
                }
            }
            break;
        case ST_SYNC_2:
            sync2_update(f, pcm[i]);
            if (f->s2 == S2_SYNCED) {
                f->state = ST_BLOCK;
            }
            break;
        case ST_BLOCK:
            block_update(f, pcm[i], &es, &ms);
            break;
        }
    }
    return 0;
}
