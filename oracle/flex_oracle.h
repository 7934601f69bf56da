/*
 * flex_oracle.h - CPU restatement of the reference's FLEX decoder, pager/pager_flex.c
 * (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  The product path
 * (tsl-sdr_amd/) never links, imports or calls anything in oracle/.
 *
 * PINNING STATUS: PARITY UNPINNED by a compiled reference.  pager/pager_flex.c includes <tsl/...> headers the
 * image lacks, so it is unbuildable here and no stand-in headers are written.  The reference's own test
 * (pager/test/test_pager_flex.c:49-57) is a new/delete smoke test with no vectors.  What the restatement IS
 * checked against (tests/test_flex.py): the protocol constants the reference holds (the four A sync codes and
 * their frame geometry, pager_flex.c:46-96; BS1 / A / B magic, pager_flex_priv.h:416-441), frames built by an
 * independent synthesiser written from the published FLEX frame layout (tsl-sdr_amd/synth.py: 1600 bit/s sync 1,
 * 25 ms sync 2, 11 blocks of 8 interleaved 32-bit words per phase, BCH(31,21) + parity, BIW / address / vector /
 * message words), and SURVEY.md section 8c's SURVEY-time probe that pager_flex.c compiles and decodes nothing on
 * noise.  BCH(31,21) is oracle/pocsag_oracle.c's (pinned as described there).
 *
 * The restatement walks one sample at a time exactly like pager_flex_on_pcm (:1401-1455) and keeps the
 * reference's word buffers, including its in-place corrections (:544-565, :958-963).  Two places where the
 * reference's behaviour is undefined are made definite, identically here and in the product:
 *   - a word index outside the 88 words of a phase (an address / vector / message offset taken from a
 *     mis-corrected word; the reference reads past phase_words[]) ends that record as "could not be decoded";
 *   - a sync-1 run with no positive or no non-positive sample (the reference divides by zero, :438-439) is
 *     treated as a failed frame information word.
 *
 * All citations are relative to the reference tree (pvachon/tsl-sdr).
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFMO_FLEX_PHASE_WORDS 88 /* pager_flex_priv.h:175 */

/* _pager_codings[] (pager_flex.c:46-96), by index */
struct mfmo_flex_coding {
    uint16_t seq_a, baud;
    uint8_t fsk_levels, sample_skip, sync_2_samples, sym_bits, sample_fudge, nr_phases;
    uint16_t symbols_per_block;
};
const struct mfmo_flex_coding *mfmo_flex_coding(unsigned idx); /* NULL beyond 3 */

#define MFMO_FLEX_EV_FRAME    1 /* all symbols of the frame collected, phases processed (:1289-1309) */
#define MFMO_FLEX_EV_BAD_BAUD 2 /* A / inverted A match no coding (:410-413) */
#define MFMO_FLEX_EV_BAD_FIW  3 /* _pager_flex_handle_fiw returned false (:1424-1427) */

struct mfmo_flex_event {
    uint32_t type;
    uint32_t coding;        /* index into the codings table; 0xffffffff for BAD_BAUD */
    uint64_t sample;        /* index (since creation) of the PCM sample that completed the event */
    uint64_t sync_sample;   /* FRAME: sample of the last FIW bit */
    uint32_t eye;           /* sync->bit_counter when the BS1 run ended (:339) */
    uint32_t a, b, inv_a;   /* :349-395 */
    uint32_t fiw_raw;       /* sync->fiw (:422-423) */
    uint32_t fiw;           /* after bch_code_decode of (fiw_raw & 0x7fffffff) (:1319-1323) */
    uint32_t fiw_rc;        /* 0 ok, 1 uncorrectable, 2 checksum (:1344), 3 no swing (see the header comment) */
    int32_t sample_range, sample_delta; /* :441-442 */
    uint32_t cycle, frame;  /* :1337-1338 */
    uint32_t pad;
    uint32_t words[4][MFMO_FLEX_PHASE_WORDS]; /* FRAME: phase_words[] of phases A..D as collected (before processing) */
};

/* what the message layer produced: the three callbacks (pager_flex.h:16-87) and, as "notes", every PAG_MSG of
 * _pager_flex_phase_process and below, so that the error paths are comparable too */
#define MFMO_FLEX_MSG_ALNUM        1  /* on_alnum_msg: aux0 = fragment | maildrop << 1 | seq_num << 2 */
#define MFMO_FLEX_MSG_NUM          2  /* on_num_msg */
#define MFMO_FLEX_MSG_SIV          3  /* on_siv_msg: aux0 = siv type, aux1 = data */
#define MFMO_FLEX_NOTE_BIW_BCH     16 /* :1124  aux0 = biw */
#define MFMO_FLEX_NOTE_BIW_CKSUM   17 /* :1130  aux0 = biw */
#define MFMO_FLEX_NOTE_BIW_COUNT   18 /* :1148  aux0 = vsw, aux1 = eob */
#define MFMO_FLEX_NOTE_BIW_EOB     19 /* :1155  aux0 = eob */
#define MFMO_FLEX_NOTE_EXTRA_BIW   20 /* :1042-1086  aux0 = 0 uncorrectable / 1 checksum / 2 ok, aux1 = word & 0x1fffff */
#define MFMO_FLEX_NOTE_ADDR_ERROR  21 /* :1180 */
#define MFMO_FLEX_NOTE_VEC_ERROR   22 /* :1188 */
#define MFMO_FLEX_NOTE_UNSUPPORTED 23 /* :1023  aux0 = vector type */
#define MFMO_FLEX_NOTE_TONE        24 /* :868,:871  aux0 = short type, aux1 = first word, aux2 = second word */

struct mfmo_flex_msg {
    uint32_t kind;
    uint32_t baud;
    uint32_t phase;         /* 0..3 = 'A'..'D' */
    uint32_t cycle, frame;
    uint32_t aux0, aux1, aux2;
    uint64_t capcode;
    uint64_t sample;        /* sample index of the frame's last symbol */
    uint32_t len;
    uint32_t pad;
    char text[256];
};

struct mfmo_flex;
struct mfmo_flex *mfmo_flex_new(void);
void mfmo_flex_free(struct mfmo_flex *f);

/* pager_flex_on_pcm (:1401-1455) on PCM at 16 000 Hz.  Events and messages are appended to the caller's arrays
 * (entries beyond the capacity are counted but not stored); returns 0. */
int mfmo_flex_on_pcm(struct mfmo_flex *f, const int16_t *pcm, size_t nr_samples,
                     struct mfmo_flex_event *ev, size_t max_ev, size_t *nr_ev,
                     struct mfmo_flex_msg *msgs, size_t max_msgs, size_t *nr_msgs);

/* _pager_flex_phase_process (:1088-1198) alone on the 88 words of one phase (modified in place, as the reference
 * does). */
int mfmo_flex_phase_process(uint32_t words[MFMO_FLEX_PHASE_WORDS], unsigned coding_idx, unsigned phase, unsigned cycle,
                            unsigned frame, uint64_t sample, struct mfmo_flex_msg *msgs, size_t max_msgs,
                            size_t *nr_msgs);

#ifdef __cplusplus
}
#endif
