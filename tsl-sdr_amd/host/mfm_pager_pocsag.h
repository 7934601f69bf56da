/*
 * mfm_pager_pocsag.h - the page-assembly half of the reference's POCSAG decoder, fed by the GPU pager stage.
 *
 * pager/pager_pocsag.h:29-54 gives a front end three calls: pager_pocsag_new(&p, freq, on_numeric, on_alpha,
 * skip_bch), pager_pocsag_on_pcm(p, pcm, n), pager_pocsag_delete(&p).  Here the sample-rate work of on_pcm
 * (slicer, sync search, batch collection, BCH) runs on the GPU for all channels at once (mfm_pocsag_*,
 * include/multifm_hip.h) and hands back events; what is left is _pager_pocsag_process_batch /
 * _pager_pocsag_message_decode_deliver (pager_pocsag.c:242-432): a walk over at most 16 corrected words per
 * batch.  Same object name, same constructor and callback signatures; pager_pocsag_on_events() takes the place
 * of pager_pocsag_on_pcm().
 */
#pragma once

#include <multifm_hip.h>

#include "mfm_tsl.h"

struct pager_pocsag;

/* pager/pager_pocsag.h:29-46 */
typedef aresult_t (*pager_pocsag_on_numeric_msg_func_t)(struct pager_pocsag *pocsag, uint16_t baud_rate, uint32_t capcode,
                                                        const char *data, size_t data_len, uint8_t function);
typedef aresult_t (*pager_pocsag_on_alpha_msg_func_t)(struct pager_pocsag *pocsag, uint16_t baud_rate, uint32_t capcode,
                                                      const char *data, size_t data_len, uint8_t function);

/* pager/pager_pocsag.h:48-52; freq_hz and skip_bch_decode are stored and, as in the reference, not used */
aresult_t pager_pocsag_new(struct pager_pocsag **ppocsag, uint32_t freq_hz, pager_pocsag_on_numeric_msg_func_t on_numeric,
                           pager_pocsag_on_alpha_msg_func_t on_alpha, bool skip_bch_decode);
aresult_t pager_pocsag_delete(struct pager_pocsag **ppocsag);

/* the events of ONE channel, in stream order (mfm_pocsag_fetch_events returns them grouped by channel) */
aresult_t pager_pocsag_on_events(struct pager_pocsag *pocsag, const struct mfm_pocsag_event *events, size_t nr_events);

/* opaque user pointer for the callbacks (the reference's callbacks reach their state through globals) */
void pager_pocsag_set_user(struct pager_pocsag *pocsag, void *user);
void *pager_pocsag_get_user(struct pager_pocsag *pocsag);
