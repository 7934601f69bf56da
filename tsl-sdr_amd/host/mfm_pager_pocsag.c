/*
 * mfm_pager_pocsag.c - see mfm_pager_pocsag.h.  Words arrive already BCH-corrected from the GPU stage
 * (corrected[], fail_mask), so nothing here touches the code itself.
 */
#include "mfm_pager_pocsag.h"

#include <ctype.h>

#define POCSAG_IDLE_CODEWORD 0x6983915eu /* pager/pager_pocsag_priv.h:46 */

enum msg_type { MSG_NONE = 0, MSG_UNKNOWN = 1, MSG_ALPHA = 2, MSG_NUMERIC = 3 };

struct pager_pocsag {
    pager_pocsag_on_numeric_msg_func_t on_numeric;
    pager_pocsag_on_alpha_msg_func_t on_alpha;
    uint32_t freq_hz;
    bool skip_bch;
    void *user;
    uint16_t baud_rate;
    /* the message being assembled (pager_pocsag_priv.h:66-136) */
    char message_alpha[512];
    size_t next_byte_alpha;
    int score_alpha;
    bool seen_nonprint;
    char message_numeric[512];
    size_t next_byte_numeric;
    uint32_t cap_code;
    uint32_t data_word_alpha;
    size_t data_word_alpha_valid_bits;
    uint32_t data_word_numeric;
    size_t data_word_numeric_valid_bits;
    uint8_t function;
    bool early_termination;
    enum msg_type msg_type;
};

static void message_reset(struct pager_pocsag *p)
{
    p->data_word_numeric = 0;
    p->data_word_numeric_valid_bits = 0;
    p->next_byte_numeric = 0;
    p->data_word_alpha = 0;
    p->data_word_alpha_valid_bits = 0;
    p->next_byte_alpha = 0;
    p->seen_nonprint = false;
    p->score_alpha = 0;
    p->early_termination = false;
    p->msg_type = MSG_NONE;
    p->function = 0;
}

/* pager_pocsag.c:242-297: pick alphanumeric or numeric by the printable-character score and hand the page over */
static aresult_t message_deliver(struct pager_pocsag *p)
{
    if (MSG_NONE == p->msg_type) {
        return A_OK;
    }
    if (0 != p->next_byte_alpha) {
        const char last = p->message_alpha[p->next_byte_alpha - 1];
        if (0x4 == last || 0x3 == last || 0x0 == last || 0x17 == last) {
            p->score_alpha = 1;
        }
    }
    if (p->next_byte_numeric > 40) {
        p->score_alpha = 1;
    }
    if (p->score_alpha > 0) {
        p->message_alpha[p->next_byte_alpha] = '\0';
        TSL_BUG_IF_FAILED(p->on_alpha(p, p->baud_rate, p->cap_code, p->message_alpha, p->next_byte_alpha, p->function));
    } else {
        p->message_numeric[p->next_byte_numeric] = '\0';
        TSL_BUG_IF_FAILED(p->on_numeric(p, p->baud_rate, p->cap_code, p->message_numeric, p->next_byte_numeric, p->function));
    }
    message_reset(p);
    return A_OK;
}

static const char numeric_charmap[16] = "0123456789XU -[]"; /* pager_pocsag.c:299-316 */

/* pager_pocsag.c:319-432 on words the GPU already ran through bch_code_decode */
static void process_batch(struct pager_pocsag *p, const struct mfm_pocsag_event *ev)
{
    for (unsigned z = 0; z < 16; z++) {
        if ((ev->fail_mask >> z) & 1u) {
            /* uncorrectable: the rest of the batch is dropped, what was collected so far is delivered */
            if (MSG_NONE != p->msg_type) {
                p->early_termination = true;
                (void)message_deliver(p);
            }
            return;
        }
        const uint32_t w = ev->corrected[z];
        if (POCSAG_IDLE_CODEWORD == w) {
            if (MSG_NONE != p->msg_type) {
                (void)message_deliver(p);
            }
            continue;
        }
        if (0 == (w & 1u)) {
            (void)message_deliver(p);
            p->msg_type = MSG_UNKNOWN;
            p->function = (uint8_t)((w >> 19) & 0x3);
            p->cap_code = (((w >> 1) & ((1u << 18) - 1)) << 3) + ((z >> 1) & 0x7);
        } else if (MSG_UNKNOWN == p->msg_type) {
            const uint32_t val = (w >> 1) & 0xfffffu;
            p->data_word_alpha |= val << p->data_word_alpha_valid_bits;
            p->data_word_alpha_valid_bits += 20;
            while (p->data_word_alpha_valid_bits >= 7) {
                const char c = (char)(p->data_word_alpha & 0x7f);
                if (p->next_byte_alpha < 511) { /* the reference has no bound here; 511 keeps the terminator in */
                    p->message_alpha[p->next_byte_alpha++] = c;
                }
                if (isprint((unsigned char)c) || 0xa == c || 0xd == c) {
                    if (!p->seen_nonprint) {
                        p->score_alpha++;
                    }
                } else {
                    p->seen_nonprint = true;
                    if (0x03 != c && 0x04 != c && 0x17 != c && 0x0 != c) {
                        p->score_alpha -= 10;
                    }
                }
                p->data_word_alpha >>= 7;
                p->data_word_alpha_valid_bits -= 7;
            }
            if (p->next_byte_numeric < 511) {
                p->data_word_numeric |= val << p->data_word_numeric_valid_bits;
                p->data_word_numeric_valid_bits += 20;
                while (p->data_word_numeric_valid_bits >= 4 && p->next_byte_numeric < 511) {
                    p->message_numeric[p->next_byte_numeric++] = numeric_charmap[p->data_word_numeric & 0xf];
                    p->data_word_numeric >>= 4;
                    p->data_word_numeric_valid_bits -= 4;
                }
            }
        }
    }
}

aresult_t pager_pocsag_new(struct pager_pocsag **ppocsag, uint32_t freq_hz, pager_pocsag_on_numeric_msg_func_t on_numeric,
                           pager_pocsag_on_alpha_msg_func_t on_alpha, bool skip_bch_decode)
{
    TSL_ASSERT_ARG(NULL != ppocsag);
    TSL_ASSERT_ARG(NULL != on_numeric);
    TSL_ASSERT_ARG(NULL != on_alpha);
    struct pager_pocsag *p = calloc(1, sizeof(*p));
    if (NULL == p) {
        return A_E_NOMEM;
    }
    p->on_numeric = on_numeric;
    p->on_alpha = on_alpha;
    p->freq_hz = freq_hz;
    p->skip_bch = skip_bch_decode;
    message_reset(p);
    *ppocsag = p;
    return A_OK;
}

aresult_t pager_pocsag_delete(struct pager_pocsag **ppocsag)
{
    TSL_ASSERT_ARG(NULL != ppocsag);
    TSL_ASSERT_ARG(NULL != *ppocsag);
    free(*ppocsag);
    *ppocsag = NULL;
    return A_OK;
}

aresult_t pager_pocsag_on_events(struct pager_pocsag *pocsag, const struct mfm_pocsag_event *events, size_t nr_events)
{
    TSL_ASSERT_ARG(NULL != pocsag);
    TSL_ASSERT_ARG(NULL != events || 0 == nr_events);
    for (size_t i = 0; i < nr_events; i++) {
        const struct mfm_pocsag_event *ev = &events[i];
        pocsag->baud_rate = (uint16_t)ev->baud;
        switch (ev->type) {
        case MFM_POCSAG_EV_BATCH:
            process_batch(pocsag, ev);
            break;
        case MFM_POCSAG_EV_SYNC_LOST:
            (void)message_deliver(pocsag); /* pager_pocsag.c:522 */
            break;
        default:
            break;
        }
    }
    return A_OK;
}

void pager_pocsag_set_user(struct pager_pocsag *pocsag, void *user)
{
    pocsag->user = user;
}

void *pager_pocsag_get_user(struct pager_pocsag *pocsag)
{
    return pocsag->user;
}
