/*
 * pocsag_oracle.h - CPU restatement of the reference's BCH(31,21) decoder and POCSAG slicer / sync /
 * batch / message logic (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  The product path
 * (tsl-sdr_amd/) never links, imports or calls anything in oracle/.
 *
 * PINNING STATUS: PARITY UNPINNED by a compiled reference.  pager/bch_code.c and pager/pager_pocsag.c include
 * <tsl/...> headers the image lacks, so they are unbuildable here and no stand-in headers are written.  The
 * reference's own tests hold no vectors for them (pager/test/test_pager_pocsag.c needs capture files that are
 * not in the tree).  What the restatement IS checked against (tests/test_pocsag.py): the reference's protocol
 * constants (POCSAG_SYNC_CODEWORD / POCSAG_IDLE_CODEWORD, pager/pager_pocsag_priv.h:40,46, must be codewords of
 * the restated code), the exhaustive 1/2/3-bit error counts recorded from the reference in SURVEY.md section 8c
 * (31/31 and 465/465 corrected; 4495 triples -> 2480 "rc 1, word untouched" + 2015 "rc 0, wrong codeword"), and
 * the algebra of the code (g(x) = lcm(m1, m3) of x^5+x^2+1 divides every accepted word).
 *
 * All citations are relative to the reference tree (pvachon/tsl-sdr).
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BCH(31,21,t=2) over GF(2^5), p(x) = x^5 + x^2 + 1 (pager/pager_pocsag.c:150,177) -------------------- */

/* pager/bch_code.c:41-72: alpha_to[i] = x^i mod p(x) for i in 0..30, index_of[] its inverse, index_of[0] = -1. */
void mfmo_bch_tables(int alpha_to[32], int index_of[32]);

/* pager/bch_code.c:307-398.  Bit (30 - j) of the word is the coefficient of x^j; bit 31 is never looked at and
 * is carried through.  Returns 0 ("clean or corrected") or 1 ("detected, word untouched"). */
int mfmo_bch3121_decode(uint32_t *word);

/* n words in place, rc[i] in {0,1}; threads > 1 splits the range (for exhaustive checks on the GPU box). */
void mfmo_bch3121_decode_batch(uint32_t *words, uint8_t *rc, size_t n, unsigned threads);

/* ---- POCSAG (pager/pager_pocsag.c) ------------------------------------------------------------------------ */

#define MFMO_POCSAG_EV_SYNC_FOUND 1 /* SEARCH -> SYNCHRONIZED (pager_pocsag.c:100-108) */
#define MFMO_POCSAG_EV_BATCH      2 /* 16 words collected and run through _process_batch (:319-432, :480-497) */
#define MFMO_POCSAG_EV_SYNC_LOST  3 /* SEARCH_SYNCWORD -> SEARCH (:517-523) */
#define MFMO_POCSAG_EV_SYNC_KEPT  4 /* SEARCH_SYNCWORD -> BATCH_RECEIVE (:524-528) */

struct mfmo_pocsag_event {
    uint32_t type;
    uint32_t baud;          /* 512 / 1200 / 2400 (pocsag->baud_rate at the event) */
    uint64_t sample;        /* absolute index (since creation) of the PCM sample that caused the event */
    uint32_t aux;           /* SYNC_FOUND: nr_eye_matches; SYNC_LOST / SYNC_KEPT: the 32-bit sync word seen */
    uint32_t nr_ok;         /* BATCH: words accepted before _process_batch gave up (16 = whole batch) */
    uint32_t fail_mask;     /* BATCH: bit z set when bch_code_decode(word z) returns 1 (all 16 evaluated) */
    uint32_t pad;
    uint32_t raw[16];       /* BATCH: current_batch[] as collected */
    uint32_t corrected[16]; /* BATCH: (raw & 0x7fffffff) after bch_code_decode, all 16 evaluated */
};

struct mfmo_pocsag_msg {
    uint32_t type;          /* 2 = alphanumeric (on_alpha), 3 = numeric (on_numeric) */
    uint32_t baud;
    uint32_t capcode;
    uint32_t function;
    uint32_t len;
    uint32_t pad;
    uint64_t sample;        /* sample index at delivery */
    char text[512];
};

struct mfmo_pocsag;

struct mfmo_pocsag *mfmo_pocsag_new(void);
void mfmo_pocsag_free(struct mfmo_pocsag *p);

/* pager_pocsag_on_pcm (:434-543) on PCM at 38 400 Hz.  Events and messages are appended to the caller's arrays
 * (entries beyond the capacity are counted but not stored); returns 0. */
int mfmo_pocsag_on_pcm(struct mfmo_pocsag *p, const int16_t *pcm, size_t nr_samples,
                       struct mfmo_pocsag_event *ev, size_t max_ev, size_t *nr_ev,
                       struct mfmo_pocsag_msg *msgs, size_t max_msgs, size_t *nr_msgs);

/* The message layer alone (_process_batch :319-432 + _message_decode_deliver :242-297), driven by batches that
 * something else collected: words = 16 raw batch words.  flush != 0 with words == NULL delivers whatever is
 * pending, as the sync-lost transition does (:522). */
struct mfmo_pocsag_msgdec;
struct mfmo_pocsag_msgdec *mfmo_pocsag_msgdec_new(void);
void mfmo_pocsag_msgdec_free(struct mfmo_pocsag_msgdec *d);
int mfmo_pocsag_msgdec_batch(struct mfmo_pocsag_msgdec *d, const uint32_t *words, int flush, uint32_t baud,
                             uint64_t sample, struct mfmo_pocsag_msg *msgs, size_t max_msgs, size_t *nr_msgs);

/*
 * ---- Mueller-Muller clock recovery (pager/mueller_muller.c:10-115), BASELINE configs[3]'s "mueller_muller slicer" ----
 * The live decoder path does not use it (pager_pocsag.c has its own eye detectors); the
 * reference exercises it only from pager/test/test_mueller_muller.c on a capture file that is not in the tree.
 * PARITY UNPINNED (mueller_muller.c includes <tsl/...> headers the image lacks; its test holds no vectors).
 * Restated without floating-point contraction (the reference's default build has no -O flag).  As in the
 * reference (:66) the sample index can reach nr_samples: the caller provides one more readable sample (the
 * reference's test slices one long buffer, so that sample is the next slice's first).
 */
struct mfmo_mm {
    float samples_per_bit, kw, km, error_min, error_max;
    float w, m, next_offset, last_sample, ideal_step_size;
};
void mfmo_mm_init(struct mfmo_mm *mm, float kw, float km, float samples_per_bit, float error_min, float error_max);
/* returns the number of decisions written (never more than max_decisions; the reference aborts instead) */
size_t mfmo_mm_process(struct mfmo_mm *mm, const int16_t *samples, size_t nr_samples, int16_t *decisions,
                       size_t max_decisions);

#ifdef __cplusplus
}
#endif
