/*
 * mfm_pager_flex.c - see mfm_pager_flex.h.  The GPU stage delivers the phase words as collected (no correction:
 * which words get corrected, in which order and whether in place depends on what the words before them say), so
 * BCH(31,21) runs here, word by word, on the table form the device uses (mfm_hosttwin_bch3121_decode).
 */
#include "mfm_pager_flex.h"

#include <inttypes.h>

#define PAG_MSG(sev, sys, msg, ...) MESSAGE("PAGER", sev, sys, msg, ##__VA_ARGS__)

#define NO_LONG_WORD 0xfffffffful /* "the vector has no second word" marker of the reference (:943) */
#define INFO_MASK    0x1ffffful   /* the 21 information bits of a word */

struct pager_flex {
    pager_flex_on_alnum_msg_func_t on_alnum_msg;
    pager_flex_on_num_msg_func_t on_num_msg;
    pager_flex_on_siv_msg_func_t on_siv_msg;
    pager_flex_note_func_t note;
    void *user;
    uint32_t freq_hz;
    /* the frame and phase being walked */
    uint16_t baud;
    uint8_t phase, cycle_id, frame_id;
    uint32_t *words; /* MFM_FLEX_PHASE_WORDS of them */
    /* flex->msg_buf / msg_len (pager_flex_priv.h:317-322) */
    char msg_buf[256];
    size_t msg_len;
};

/* phases a coding carries, in the order the reference processes them (:1291-1305) */
static const uint8_t phase_order[5][4] = { { 0 }, { 0 }, { 0, 2 }, { 0 }, { 0, 1, 2, 3 } };

static const char type_code[8][4] = { "SEC", "SIV", "TON", "NUM", "SNM", "ALN", "HEX", "NNM" }; /* :578-588 */
static const char digit_of[16] = { '0', '1', '2', '3', '4', '5', '6', '7', '8', '9', 'X', 'U', ' ', '-', ']', '[' }; /* :686-704 */

static void note(struct pager_flex *f, int kind, uint64_t cap, uint32_t a0, uint32_t a1, uint32_t a2)
{
    if (f->note) {
        f->note(f, kind, f->phase, cap, a0, a1, a2);
    }
}

/* :107-119 */
static unsigned nibble_sum(uint32_t w)
{
    unsigned s = 0;
    for (w &= INFO_MASK; w; w >>= 4) {
        s += w & 0xf;
    }
    return s & 0xf;
}

static bool bch_fix(uint32_t *w)
{
    return 0 == mfm_hosttwin_bch3121_decode(w);
}

/* word idx of the phase, corrected, without touching the phase */
static bool corrected_copy(const struct pager_flex *f, size_t idx, uint32_t *out)
{
    if (idx >= MFM_FLEX_PHASE_WORDS) {
        return false;
    }
    *out = f->words[idx];
    return bch_fix(out);
}

static void put_char(struct pager_flex *f, char ch)
{
    f->msg_buf[f->msg_len++] = ch;
}

/* :597-681 - body = index of the first message word, count = words the vector announced */
static aresult_t body_alphanumeric(struct pager_flex *f, uint64_t cap, uint32_t long_word, size_t body, size_t count)
{
    uint32_t head;
    size_t i = 0;
    if (NO_LONG_WORD == long_word) {
        if (!corrected_copy(f, body, &head)) {
            return A_E_INVAL;
        }
        i = 1;
    } else {
        head = long_word; /* a long address moves the header into the second vector word */
    }
    const bool fragment = 0 != (head & (1u << 10));
    const uint8_t seq = (head >> 11) & 0x3;
    bool maildrop = false;
    unsigned first_slot = 0;
    if (3 == seq) {
        /* an initial fragment: the first character slot holds the signature */
        first_slot = 1;
        maildrop = 0 != (head & (1u << 20));
    }
    for (; i < count && f->msg_len != 255; i++) {
        uint32_t cw;
        if (!corrected_copy(f, body + i, &cw)) {
            return A_E_INVAL;
        }
        cw >>= 7 * first_slot;
        for (unsigned slot = first_slot; slot < 3; slot++, cw >>= 7) {
            const char ch = (char)(cw & 0x7f);
            if (0x3 == ch) {
                break;
            }
            put_char(f, ch);
            if (255 == f->msg_len) {
                break;
            }
        }
        first_slot = 0;
    }
    f->msg_buf[f->msg_len] = '\0';
    return f->on_alnum_msg(f, f->baud, f->phase, f->cycle_id, f->frame_id, cap, fragment, maildrop, seq, f->msg_buf, f->msg_len);
}

/*
 * :709-824 - digits of 4 bits run across words of 21 bits; the first word gives 19 (two check bits are dropped).
 * The bookkeeping (bits left in the current word, bits announced, look-ahead word) is the reference's, with its
 * unsigned wrap-around, so that short or inconsistent vectors end the same way.
 */
static aresult_t body_numeric(struct pager_flex *f, uint64_t cap, uint32_t long_word, size_t body, size_t count)
{
    uint32_t cur, ahead = 0;
    size_t announced = count * 21, have = 19, ahead_idx, ahead_have = 21;

    if (NO_LONG_WORD != long_word) {
        cur = ((uint32_t)long_word & INFO_MASK) >> 2;
        announced += 19;
        ahead_idx = 0;
    } else {
        if (!corrected_copy(f, body, &cur)) {
            return A_E_INVAL;
        }
        cur = (cur & INFO_MASK) >> 2;
        announced -= 2;
        ahead_idx = 1;
    }
    if (ahead_idx < count) {
        if (!corrected_copy(f, body + ahead_idx, &ahead)) {
            return A_E_INVAL;
        }
        ahead &= INFO_MASK;
    }
    announced &= ~(size_t)0x3;

    do {
        for (size_t digits = have / 4; digits != 0; digits--) {
            put_char(f, digit_of[cur & 0xf]);
            if (255 == f->msg_len) {
                goto deliver;
            }
            cur >>= 4;
            have -= 4;
            announced -= 4;
        }
        if (0 == announced) {
            break;
        }
        if (0 != have) {
            /* 1..3 bits left: complete the digit with the low bits of the look-ahead word */
            const unsigned need = 4 - (unsigned)have;
            cur |= (ahead & ((1u << need) - 1)) << have;
            ahead >>= need;
            ahead_have -= need;
            have = 4;
        } else {
            cur = ahead;
            have = ahead_have;
            ahead_have = 21;
            if (++ahead_idx < count) {
                if (!corrected_copy(f, body + ahead_idx, &ahead)) {
                    return A_E_INVAL;
                }
                ahead &= INFO_MASK;
            }
        }
    } while (0 != announced);

deliver:
    f->msg_buf[f->msg_len] = '\0';
    return f->on_num_msg(f, f->baud, f->phase, f->cycle_id, f->frame_id, cap, f->msg_buf, f->msg_len);
}

/* :829-883 */
static aresult_t vector_tone(struct pager_flex *f, uint64_t cap, uint32_t vec, uint32_t second)
{
    vec &= INFO_MASK;
    const unsigned short_type = (vec >> 7) & 0x3;
    switch (short_type) {
    case 0: /* PAGER_FLEX_SHORT_TYPE_3_OR_8 */
        for (unsigned d = 0; d < 3; d++) {
            put_char(f, digit_of[(vec >> (9 + 4 * d)) & 0xf]);
        }
        if (NO_LONG_WORD != second) {
            second &= INFO_MASK;
            for (unsigned d = 0; d < 5; d++) {
                put_char(f, digit_of[(second >> (4 * d)) & 0xf]);
            }
        }
        f->msg_buf[f->msg_len] = '\0';
        return f->on_num_msg(f, f->baud, f->phase, f->cycle_id, f->frame_id, cap, f->msg_buf, f->msg_len);
    case 1: /* PAGER_FLEX_SHORT_TYPE_8_SOURCES */
        PAG_MSG(SEV_INFO, "TONE", "%02u/%03u/%c [ %9" PRIu64 "] Sourced Tone: [%08x, %08x]", f->cycle_id, f->frame_id,
                f->phase + 'A', cap, vec, second);
        note(f, PAGER_FLEX_NOTE_TONE, cap, short_type, vec, second);
        return A_OK;
    case 2: /* PAGER_FLEX_SHORT_TYPE_SOURCES_AND_NUM */
        PAG_MSG(SEV_INFO, "TONE", "%02u/%03u/%c [ %9" PRIu64 "] Sequenced Tone: [%08x, %08x]", f->cycle_id, f->frame_id,
                f->phase + 'A', cap, vec, second);
        note(f, PAGER_FLEX_NOTE_TONE, cap, short_type, vec, second);
        return A_OK;
    default:
        return A_E_INVAL;
    }
}

/* :885-933 */
static aresult_t vector_short_instruction(struct pager_flex *f, uint64_t cap, uint32_t vec)
{
    vec &= 0x7fffff;
    if (0xf != nibble_sum(vec)) {
        return A_E_INVAL;
    }
    const unsigned siv_type = (vec >> 7) & 0x7, siv_data = (vec >> 10) & 0x7ff;
    switch (siv_type) {
    case PAGER_FLEX_SIV_TEMP_ADDRESS_ACTIVATION:
        break;
    case PAGER_FLEX_SIV_SYSTEM_EVENT:
        PAG_MSG(SEV_INFO, "SIV", "%02u/%03u/%c - [%9" PRIu64 "] System Event (data = %08x)", f->cycle_id, f->frame_id,
                f->phase + 'A', cap, siv_data);
        break;
    case PAGER_FLEX_SIV_RESERVED_TEST:
        PAG_MSG(SEV_INFO, "SIV", "%02u/%03u/%c - [%9" PRIu64 "] Reserved Test (data = %08x)", f->cycle_id, f->frame_id,
                f->phase + 'A', cap, siv_data);
        break;
    default:
        PAG_MSG(SEV_INFO, "SIV", "%02u/%03u/%c - [%9" PRIu64 "] Unknown SIV %u (data = %08x)", f->cycle_id, f->frame_id,
                f->phase + 'A', cap, siv_type, siv_data);
    }
    if (NULL != f->on_siv_msg) {
        f->on_siv_msg(f, f->baud, f->phase, f->cycle_id, f->frame_id, cap, (uint8_t)siv_type, siv_data);
    }
    return A_OK;
}

/* :938-1033 - at = index of the (first) vector word, span = 1, or 2 behind a long address */
static aresult_t vector_decode(struct pager_flex *f, uint64_t cap, size_t at, size_t span)
{
    f->msg_len = 0;
    for (size_t i = 0; i < span; i++) {
        /* the vector words are corrected where they lie (:959) */
        if (at + i >= MFM_FLEX_PHASE_WORDS || !bch_fix(&f->words[at + i])) {
            return A_E_INVAL;
        }
    }
    const uint32_t vec = f->words[at];
    if (0xf != nibble_sum(vec)) {
        return A_E_INVAL;
    }
    const uint32_t second = (2 == span) ? f->words[at + 1] : NO_LONG_WORD;
    const unsigned vec_type = (vec >> 4) & 0x7;
    const size_t body = (vec >> 7) & 0x7f;
    size_t count;

    switch (vec_type) {
    case 0x2: /* PAGER_FLEX_MESSAGE_TONE */
        return FAILED(vector_tone(f, cap, vec, second)) ? A_E_INVAL : A_OK;
    case 0x3: /* PAGER_FLEX_MESSAGE_STANDARD_NUMERIC: three bits of length */
        count = ((vec >> 14) & 0x7) + 1 - (2 == span ? 1 : 0);
        return FAILED(body_numeric(f, cap, second, body, count)) ? A_E_INVAL : A_OK;
    case 0x5: /* PAGER_FLEX_MESSAGE_ALPHANUMERIC: seven; the subtraction wraps for a zero field, as in :1005 */
        count = ((vec >> 14) & 0x7f) - (size_t)(2 == span ? 1 : 0);
        return FAILED(body_alphanumeric(f, cap, second, body, count)) ? A_E_INVAL : A_OK;
    case 0x1: /* PAGER_FLEX_MESSAGE_SPECIAL_INSTRUCTION */
        return FAILED(vector_short_instruction(f, cap, vec)) ? A_E_INVAL : A_OK;
    default: /* secure, special numeric, hex, numbered numeric */
        PAG_MSG(SEV_INFO, "UNSUPP-MSG", "%02u/%03u/%c [%9" PRIu64 "] Unsupported Message: %s", f->cycle_id, f->frame_id,
                f->phase + 'A', cap, type_code[vec_type]);
        note(f, PAGER_FLEX_NOTE_UNSUPPORTED, cap, vec_type, 0, 0);
        return A_OK;
    }
}

/* :527-573 - the address word(s) at `at`, corrected and cut to 21 bits where they lie */
static aresult_t address_decode(struct pager_flex *f, size_t at, uint64_t *cap, size_t *extra)
{
    uint32_t *a = &f->words[at];
    *cap = 0;
    *extra = 0;
    if (!bch_fix(&a[0])) {
        return A_E_INVAL;
    }
    a[0] &= INFO_MASK;
    const uint32_t lo = a[0];
    const bool is_short = (lo > 0x8000 && lo <= 0x1e0000) || (lo > 0x1f0000 && lo < 0x1f7fff);
    if (is_short) {
        *cap = lo - 32768;
        return A_OK;
    }
    if (!bch_fix(&a[1])) {
        return A_E_INVAL;
    }
    a[1] &= INFO_MASK;
    *extra = 1;
    /* 32-bit arithmetic, as written in the reference (:567); only "1-2" long addresses are covered there */
    *cap = (uint32_t)(0x1f9001u + ((0x1fffffu - a[1]) * 32768u + lo - 1u));
    return A_OK;
}

/* :1041-1086 */
static void extra_block_info(struct pager_flex *f, uint32_t w)
{
    w &= 0x7fffffffu;
    if (!bch_fix(&w)) {
        PAG_MSG(SEV_INFO, "BLOCK", "Additional BIW could not be corrected.");
        note(f, PAGER_FLEX_NOTE_EXTRA_BIW, 0, 0, 0, 0);
        return;
    }
    w &= INFO_MASK;
    if (0xf != nibble_sum(w)) {
        PAG_MSG(SEV_INFO, "BLOCK", "Additional BIW failed checksumming.");
        note(f, PAGER_FLEX_NOTE_EXTRA_BIW, 0, 1, w, 0);
        return;
    }
    note(f, PAGER_FLEX_NOTE_EXTRA_BIW, 0, 2, w, 0);
    const unsigned function = (w >> 4) & 0x7, field = w >> 7;
    switch (function) {
    case 0:
        PAG_MSG(SEV_INFO, "BLOCK-LOCAL-IDS", "SSID word");
        break;
    case 1:
        PAG_MSG(SEV_INFO, "BLOCK-DATE", "%02u-%02u-%u", ((field >> 9) & 0x1f) + 1994, ((field >> 4) & 0x1f) + 1, field & 0xf);
        break;
    case 2:
        PAG_MSG(SEV_INFO, "BLOCK-TIME", "%02u:%02u:%02u", (field >> 9) & 0x1f, (field >> 3) & 0x3f, (field & 0x7) << 3);
        break;
    case 5:
        PAG_MSG(SEV_INFO, "BLOCK-SYS-INFO", "System Information Field");
        break;
    case 7:
        PAG_MSG(SEV_INFO, "BLOCK-SYS-COUNTRY", "Country Information");
        break;
    default:
        PAG_MSG(SEV_INFO, "BLOCK", "Unknown function %u.", function);
    }
}

/* :1088-1198 */
aresult_t pager_flex_process_phase(struct pager_flex *f, uint32_t *words, uint16_t baud, uint8_t phase, uint8_t cycle_no,
                                   uint8_t frame_no)
{
    TSL_ASSERT_ARG(NULL != f);
    TSL_ASSERT_ARG(NULL != words);
    TSL_ASSERT_ARG(phase < 4);
    f->words = words;
    f->baud = baud;
    f->phase = phase;
    f->cycle_id = cycle_no;
    f->frame_id = frame_no;

    uint32_t biw = words[0] & 0x7fffffffu;
    if (!bch_fix(&biw)) {
        PAG_MSG(SEV_INFO, "BAD-BIW", "%02u/%03u/%c: Skipping (could not correct BIW %08x)", cycle_no, frame_no, phase + 'A', biw);
        note(f, PAGER_FLEX_NOTE_BIW_BCH, 0, biw, 0, 0);
        return A_OK;
    }
    if (0xf != nibble_sum(biw)) {
        PAG_MSG(SEV_INFO, "BAD-BIW", "%02u/%03u/%c: Skipping - bad checksum (for BIW %08x)", cycle_no, frame_no, phase + 'A', biw);
        note(f, PAGER_FLEX_NOTE_BIW_CKSUM, 0, biw, 0, 0);
        return A_OK;
    }
    const unsigned vectors_at = (biw >> 10) & 0x3f, end_of_block = (biw >> 8) & 0x3;
    if (end_of_block > vectors_at) {
        PAG_MSG(SEV_INFO, "BAD-BIW", "%02u/%03u/%c: Skipping BIW - bad vector count count of %u (EoB = %u)", cycle_no, frame_no,
                phase + 'A', vectors_at, end_of_block);
        note(f, PAGER_FLEX_NOTE_BIW_COUNT, 0, vectors_at, end_of_block, 0);
        return A_OK;
    }
    if (0 != end_of_block) {
        PAG_MSG(SEV_INFO, "BLOCK", "%02u/%02u/%c BIW end of block = %u", cycle_no, frame_no, phase + 'A', end_of_block);
        note(f, PAGER_FLEX_NOTE_BIW_EOB, 0, end_of_block, 0, 0);
        for (unsigned i = 1; i < end_of_block; i++) {
            extra_block_info(f, words[i]);
        }
    }

    /* addresses fill [1 + end_of_block, vectors_at); the k-th address WORD owns the k-th vector word */
    const size_t addresses_at = 1 + end_of_block;
    size_t at = addresses_at;
    while (at < vectors_at) {
        uint64_t cap;
        size_t extra;
        if (FAILED_UNLIKELY(address_decode(f, at, &cap, &extra))) {
            PAG_MSG(SEV_WARNING, "BCH-ERROR", "%02u/%03u/%c Address could not be corrected", cycle_no, frame_no, phase + 'A');
            note(f, PAGER_FLEX_NOTE_ADDR_ERROR, 0, 0, 0, 0);
            return A_OK;
        }
        if (FAILED_UNLIKELY(vector_decode(f, cap, vectors_at + (at - addresses_at), extra + 1))) {
            PAG_MSG(SEV_WARNING, "BCH-ERROR", "%02u/%03u/%c [%9" PRIu64 "] Uncorrectable Error", cycle_no, frame_no, phase + 'A', cap);
            note(f, PAGER_FLEX_NOTE_VEC_ERROR, cap, 0, 0, 0);
        }
        at += 1 + extra;
    }
    return A_OK;
}

aresult_t pager_flex_on_events(struct pager_flex *flex, const struct mfm_flex_event *events, size_t nr_events,
                               const struct mfm_flex_frame_words *frames)
{
    TSL_ASSERT_ARG(NULL != flex);
    TSL_ASSERT_ARG(NULL != events || 0 == nr_events);
    for (size_t i = 0; i < nr_events; i++) {
        const struct mfm_flex_event *e = &events[i];
        switch (e->type) {
        case MFM_FLEX_EV_BAD_BAUD: /* :411 */
            PAG_MSG(SEV_WARNING, "UNKNOWN-BAUD", "Unknown baud identifier code: %04x/%04x", e->a, e->inv_a);
            break;
        case MFM_FLEX_EV_BAD_FIW:
            if (1 == e->fiw_rc) { /* :1325 */
                PAG_MSG(SEV_INFO, "BAD-FIW", "FIW %08x could not be corrected with BCH(31, 23).", e->fiw);
            }
            break;
        case MFM_FLEX_EV_FRAME: {
            TSL_ASSERT_ARG(NULL != frames);
            TSL_ASSERT_ARG(1 == e->nr_phases || 2 == e->nr_phases || 4 == e->nr_phases);
            if (0xffffffffu == e->frame_index) {
                break; /* the stage had no room for this frame's words (caller-chosen max_events) */
            }
            struct mfm_flex_frame_words work = frames[e->frame_index];
            for (unsigned k = 0; k < e->nr_phases; k++) {
                const uint8_t ph = phase_order[e->nr_phases][k];
                TSL_BUG_IF_FAILED(pager_flex_process_phase(flex, work.words[ph], (uint16_t)e->baud, ph, (uint8_t)e->cycle,
                                                           (uint8_t)e->frame));
            }
            break;
        }
        default:
            return A_E_INVAL;
        }
    }
    return A_OK;
}

aresult_t pager_flex_new(struct pager_flex **pflex, uint32_t freq_hz, pager_flex_on_alnum_msg_func_t on_aln_msg,
                         pager_flex_on_num_msg_func_t on_num_msg, pager_flex_on_siv_msg_func_t on_siv_msg)
{
    TSL_ASSERT_ARG(NULL != pflex);
    TSL_ASSERT_ARG(NULL != on_aln_msg);
    TSL_ASSERT_ARG(NULL != on_num_msg);
    struct pager_flex *f = calloc(1, sizeof(*f));
    if (NULL == f) {
        return A_E_NOMEM;
    }
    f->freq_hz = freq_hz;
    f->on_alnum_msg = on_aln_msg;
    f->on_num_msg = on_num_msg;
    f->on_siv_msg = on_siv_msg;
    *pflex = f;
    return A_OK;
}

aresult_t pager_flex_delete(struct pager_flex **pflex)
{
    TSL_ASSERT_ARG(NULL != pflex);
    TSL_ASSERT_ARG(NULL != *pflex);
    free(*pflex);
    *pflex = NULL;
    return A_OK;
}

void pager_flex_set_user(struct pager_flex *flex, void *user)
{
    flex->user = user;
}

void *pager_flex_get_user(struct pager_flex *flex)
{
    return flex->user;
}

void pager_flex_set_note_hook(struct pager_flex *flex, pager_flex_note_func_t hook)
{
    flex->note = hook;
}
