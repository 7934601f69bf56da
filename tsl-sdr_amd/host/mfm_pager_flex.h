/*
 * mfm_pager_flex.h - the message half of the reference's FLEX decoder, fed by the GPU pager stage.
 *
 * pager/pager_flex.h:89-115 gives a front end three calls: pager_flex_new(&f, freq, on_alnum, on_num, on_siv),
 * pager_flex_on_pcm(f, pcm, n), pager_flex_delete(&f).  Here the sample-rate work of on_pcm (sync 1, frame
 * information word, sync 2, slicing, block de-interleave) runs on the GPU for all channels at once (mfm_flex_*,
 * include/multifm_hip.h) and hands back, per frame, the 88 words of every phase; what is left is
 * _pager_flex_phase_process and below (pager_flex.c:527-1198): block information word, address and vector
 * fields, alphanumeric / numeric / tone / short-instruction bodies - a walk over at most 88 words per phase that
 * corrects words in place as it goes.  Same object name, same constructor and callback signatures;
 * pager_flex_on_events() takes the place of pager_flex_on_pcm().
 *
 * Two deviations, both where the reference's behaviour is undefined: a word offset beyond the 88 words of the
 * phase (taken from a mis-corrected vector word; the reference reads past phase_words[]) fails that record, and
 * the message callbacks get a NUL-terminated buffer.
 */
#pragma once

#include <multifm_hip.h>

#include "mfm_tsl.h"

struct pager_flex;

/* pager/pager_flex.h:16-42 */
typedef aresult_t (*pager_flex_on_alnum_msg_func_t)(struct pager_flex *flex, uint16_t baud, uint8_t phase, uint8_t cycle_no,
                                                    uint8_t frame_no, uint64_t cap_code, bool fragmented, bool maildrop,
                                                    uint8_t seq_num, const char *message_bytes, size_t message_len);
typedef aresult_t (*pager_flex_on_num_msg_func_t)(struct pager_flex *flex, uint16_t baud, uint8_t phase, uint8_t cycle_no,
                                                  uint8_t frame_no, uint64_t cap_code, const char *message_bytes,
                                                  size_t message_len);

/* pager/pager_flex.h:44-61 */
#define PAGER_FLEX_SIV_TEMP_ADDRESS_ACTIVATION 0x0
#define PAGER_FLEX_SIV_SYSTEM_EVENT            0x1
#define PAGER_FLEX_SIV_RESERVED_TEST           0x3

/* pager/pager_flex.h:76-87 */
typedef aresult_t (*pager_flex_on_siv_msg_func_t)(struct pager_flex *flex, uint16_t baud, uint8_t phase, uint8_t cycle_no,
                                                  uint8_t frame_no, uint64_t cap_code, uint8_t siv_msg_type, uint32_t data);

/* pager/pager_flex.h:95-104; on_siv_msg may be NULL, the other two may not */
aresult_t pager_flex_new(struct pager_flex **pflex, uint32_t freq_hz, pager_flex_on_alnum_msg_func_t on_aln_msg,
                         pager_flex_on_num_msg_func_t on_num_msg, pager_flex_on_siv_msg_func_t on_siv_msg);
aresult_t pager_flex_delete(struct pager_flex **pflex);

/* the events of ONE channel, in stream order, and the frame-word array the same mfm_flex_fetch_events call filled
 * (FRAME events point into it with frame_index) */
aresult_t pager_flex_on_events(struct pager_flex *flex, const struct mfm_flex_event *events, size_t nr_events,
                               const struct mfm_flex_frame_words *frames);

/* one phase by itself (88 words, corrected in place as the reference does): what on_events runs per phase */
aresult_t pager_flex_process_phase(struct pager_flex *flex, uint32_t *words, uint16_t baud, uint8_t phase, uint8_t cycle_no,
                                   uint8_t frame_no);

/* opaque user pointer for the callbacks (the reference's callbacks reach their state through globals) */
void pager_flex_set_user(struct pager_flex *flex, void *user);
void *pager_flex_get_user(struct pager_flex *flex);

/*
 * Everything the reference only logs (PAG_MSG in pager_flex.c:868-1190) can also be observed: kind is one of
 * PAGER_FLEX_NOTE_*, the meaning of a0..a2 is listed with them.  The log lines are written either way.
 */
#define PAGER_FLEX_NOTE_BIW_BCH     16 /* :1124  a0 = BIW */
#define PAGER_FLEX_NOTE_BIW_CKSUM   17 /* :1130  a0 = BIW */
#define PAGER_FLEX_NOTE_BIW_COUNT   18 /* :1148  a0 = vector start word, a1 = end-of-block count */
#define PAGER_FLEX_NOTE_BIW_EOB     19 /* :1155  a0 = end-of-block count */
#define PAGER_FLEX_NOTE_EXTRA_BIW   20 /* :1048-1084  a0 = 0 uncorrectable / 1 checksum / 2 decoded, a1 = the 21 bits */
#define PAGER_FLEX_NOTE_ADDR_ERROR  21 /* :1180 */
#define PAGER_FLEX_NOTE_VEC_ERROR   22 /* :1188 */
#define PAGER_FLEX_NOTE_UNSUPPORTED 23 /* :1023  a0 = vector type */
#define PAGER_FLEX_NOTE_TONE        24 /* :868,:871  a0 = short type, a1 = first word, a2 = second word */
typedef void (*pager_flex_note_func_t)(struct pager_flex *flex, int kind, uint8_t phase, uint64_t cap_code, uint32_t a0,
                                       uint32_t a1, uint32_t a2);
void pager_flex_set_note_hook(struct pager_flex *flex, pager_flex_note_func_t hook);
