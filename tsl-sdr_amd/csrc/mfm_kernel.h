/*
 * mfm_kernel.h - launch-time description of one pass of the fused multifm channel kernel.
 * Shared by the kernel (mfm_kernel.hip) and the engine (mfm_engine.hip).  Internal: the public
 * boundary is include/multifm_hip.h.
 */
#pragma once

#include <stdint.h>

#define MFM_WAVE 64
/* channels per register group (accumulator tile = OPL x CG complex sums per lane) */
#define MFM_CG 8
/* taps per coefficient chunk: one chunk = TG*CG*2 dwords = two s_load_dwordx16 */
#define MFM_TG 2
/* waves per workgroup */
#define MFM_NW 8
#define MFM_NT (MFM_NW * MFM_WAVE)

/* per-channel rotator table descriptor (filter/direct_fir.c:151-172 recurrence, tabulated) */
struct mfm_chan_info {
    uint64_t rot_base; /* index of R[0] in the rotator table (entries of uint2) */
    uint32_t mu;       /* pre-period of the rotator sequence */
    uint32_t lam;      /* period */
    uint32_t lam_magic; /* floor(2^32 / lam) */
    uint32_t pad0;      /* the first-generation matrix kernel keeps the channel's table position here in its LDS copy */
    uint32_t out_row;   /* row of the PCM / filtered-IQ output this channel's samples go to (the second-generation
                           kernel orders its rows by rotator class; the others keep the caller's order) */
    uint32_t rot_class; /* MFM_RC_* */
}; /* 8 dwords: the kernels read it as a dword array */

/* rotator classes (filter/direct_fir.c:151-172): what the recurrence rot <- r14(rot * incr) does for this increment.
 * mfm_chan_info::rot_class = class | (quarter turns per output) << 4.  Higher = cheaper; a set of channels runs at the
 * lowest class among them. */
#define MFM_RC_GENERAL 0u /* tabulated; derotation = two Q14 dot products + rounding */
#define MFM_RC_QUARTER 1u /* incr = (0, +-16384): rot walks the four axis points; r14(f * rot) = f * j^m exactly (with the
                             int16 cast's wrap): a swap of the halves and two signs */
#define MFM_RC_FLIP 2u    /* incr = (-16384, 0): rot alternates (16384, 0), (-16384, 0); r14(f * rot) = +-f */
#define MFM_RC_IDENT 3u   /* incr = (16384, 0) or zero: rot stays (16384, 0); r14(f * rot) = f */

/* per-channel state carried between launches (ping-pong) */
struct mfm_chan_state {
    uint32_t carry_q; /* packed (re,im) of the last filtered sample: fm_demod.c:16-17 */
    uint32_t kb;      /* folded rotator index of the next output */
};

struct mfm_launch {
    const uint32_t *x;   /* packed int16 IQ; x[0] is the first unconsumed stream sample */
    uint32_t n_avail;    /* samples readable at x */
    uint32_t n_new;      /* outputs this pass produces (per channel) */
    uint32_t decim;      /* D */
    uint32_t nchunks;    /* ceil(T / MFM_TG) */
    uint32_t nstage;     /* samples one tile stages: (OT-1)*D + T */
    uint32_t rs2;        /* LDS row stride in dwords */
    uint32_t lut_off;    /* dword offset of the atan LUT inside LDS */
    uint32_t ngroups;    /* channel groups of MFM_CG (channels padded with zero taps) */
    uint32_t gpw;        /* channel groups per workgroup slice */
    uint32_t nslices;    /* ceil(ngroups / gpw) */
    uint32_t ntiles;
    uint32_t nchan;      /* real channels */
    uint32_t out_stride; /* int16 elements between channels in pcm (and dwords in iq_dbg) */
    const uint32_t *coef;   /* [ngroups][nchunks][TG][CG][2] */
    const uint32_t *tapoff; /* [nchunks*TG] LDS byte offset of tap i */
    const struct mfm_chan_info *info;
    const uint2 *rot;       /* {(rr | -ri<<16), (ri | rr<<16)} */
    const struct mfm_chan_state *st_in;
    struct mfm_chan_state *st_out;
    const float2 *lut;      /* [256] {T[i], T[i+1]-T[i]} */
    int16_t *pcm;
    uint32_t *iq_dbg;       /* NULL unless a channel asked for the filtered-IQ stream */
};

/* ---------------------------------------------------------------------------------------------
 * FIR-as-GEMM variant (mfm_kernel_mfma.hip): exact int16 arithmetic through four int8 MFMA
 * products.  Used when decimation % 8 == 0, taps <= 32 * MFM_MFMA_KQ_STREAM_MAX (streamed beyond 32 * MFM_MFMA_KQ_MAX),
 * the tile fits LDS and every tap fits
 * +-32639; otherwise the v_dot2 kernel above runs.
 * ------------------------------------------------------------------------------------------- */
#define MFM_MFMA_NW 8            /* waves per workgroup; each owns 16 GEMM rows = 8 channels */
#define MFM_MFMA_KQ_MAX 4        /* k-steps of 64 int16 elements (= 32 complex taps) held in registers */
#define MFM_MFMA_KQ_STREAM_MAX 16 /* longer filters (up to 512 taps): all k-steps in registers at two waves per SIMD (the
                                     resident instances), or the A operand streamed from L2 in chunks of 4 k-steps */

#define MFM_M_PLANE_DIST 16384u
#define MFM_M_CH_MAX 8u /* at most this many 16-byte staging chunks per thread and tile */

struct mfm_launch_mfma {
    const uint32_t *x;
    uint32_t n_avail, n_new, decim;
    uint32_t x_last4;     /* last sample index at which a 16-byte load stays inside the input buffer (multiple of 4) */
    uint32_t kq;          /* k-steps of 64 elements: padded taps = 32 * kq */
    uint32_t kq_used;     /* k-steps that hold taps at all: ceil(elements / 64) <= kq (kq is rounded up to a power of two) */
    uint32_t ot;          /* NEW outputs per workgroup tile: 31 per iteration of 32 columns, two iterations (62) or, for
                             decimations whose 62-output tile does not fit LDS, one (31) */
    uint32_t nstage;      /* samples staged per tile: ot*D + 32*kq, rounded up to 4 */
    uint32_t rs;          /* LDS row stride in bytes (row = 2*D plane bytes), an odd multiple of 32 */
    uint32_t row_bytes;   /* plane bytes of a row as the GEMM sees it: 2*D rounded up to 16.  For decimations that are not
                             multiples of 8 the rows are padded and the taps carry zeros over the padding */
    uint32_t split_rows;  /* 1: a 4-sample staging chunk can straddle two rows (D % 4 != 0): stored sample by sample */
    uint32_t plane_bytes; /* bytes of one byte-plane in LDS (16-byte multiple) */
    uint32_t fixed_planes; /* 1: the four planes sit MFM_M_PLANE_DIST bytes apart (H0, L0, H1, L1) whatever their size,
                              so the distance is an instruction immediate; 0: packed, plane_bytes apart */
    uint32_t lut_off;     /* byte offset of the atan LUT in LDS */
    uint32_t sta_off;     /* byte offset of the staging-offset table in LDS: [MFM_M_CH_MAX][512 threads] dwords */
    uint32_t bof_off;     /* streaming variant (kq > 4): byte offset of the B-fragment offset table in LDS, [64 lanes][16] uint16 */
    uint32_t tbl_off;     /* byte offset of the per-channel rotator constants in LDS (8 dwords per channel:
                             mfm_chan_info with kb in pad[0]); 0 = too many channels, read them from global */
    uint32_t nslices;     /* ceil(row blocks / MFM_MFMA_NW) */
    uint32_t nrb;         /* row blocks of 16 rows (= 8 channels) */
    uint32_t ntiles, nitems;
    uint32_t nchan, out_stride;
    uint32_t ah_mask;     /* bit kq set: some tap of k-step kq (32 complex taps) lies outside [-128, 127], i.e. its high-byte
                             plane is not all zero.  Low-pass taps decay towards both ends, so for the outer k-steps the two
                             products with the high-byte plane are zero and are not computed. */
    uint32_t tail_src, tail_n; /* samples x[tail_src .. tail_src + tail_n) are the history the next block needs ... */
    uint32_t in8;         /* 0: x is packed int16 IQ.  7 / 14: x is 8-bit IQ off the wire (2 bytes per sample), the value is
                             the first rounding's shift and krow the matching row constants; x_last4 then is the last
                             sample index at which an 8-byte load stays inside the buffer */
    uint32_t in8_xor;     /* 0x80808080 when the bytes are unsigned (RTL-SDR), else 0 */
    uint32_t stream_taps; /* filters of 129..512 taps (kq 8 / 16): 1 = re-read the taps from L2 in every iteration (128-register
                             instances, two workgroups per CU where LDS allows); 0 = the resident instances (all taps in
                             registers, 256 registers, one workgroup per CU) where one is built for the geometry */
    uint32_t *tail_dst;   /* ... at the front of the other input buffer (workgroup 0 copies them) */
    const uint32_t *afrag;   /* [nrb][kq][plane hi,lo][lane][4 dwords] */
    const int32_t *krow;     /* [nrb][16]: 128 * sum_k W[row][k] + 8192 */
    const struct mfm_chan_info *info;
    const uint2 *rot;
    const struct mfm_chan_state *st_in;
    struct mfm_chan_state *st_out;
    const float2 *lut;
    int16_t *pcm;
    uint32_t *iq_dbg;
};

/* ---------------------------------------------------------------------------------------------
 * FIR-as-GEMM, second generation (mfm_kernel_v3.hip).  Same arithmetic and the same tap fragments as
 * mfm_kernel_mfma.hip; what changed is who owns which output:
 *   - a tile is 64 consecutive outputs, column n of column group g (g = 0..3) is output 64*tile + 4*n + g, so a
 *     lane ends up with FOUR CONSECUTIVE outputs of its two channels: the PCM leaves as one 8-byte store per
 *     channel and tile (the 2-byte stores of the first generation bound that kernel at ~105 us per 2^26-sample
 *     block all by themselves, profiles/r02_knockout.txt), and the rotator entries arrive as 16-byte loads;
 *   - a workgroup walks a CHUNK of consecutive tiles of one 64-channel slice, so the discriminator's one-sample
 *     history and the rotator-table position are carried in registers from tile to tile; only the first tile
 *     of a chunk recomputes the output in front of it (one extra column group per chunk instead of one
 *     recomputed column per 31);
 *   - the LDS image of a tile keeps the rows (one row = the D samples between two outputs) of equal index mod 4
 *     in four sub-planes, which makes "lane n reads row 4n + g" the same conflict-free stride pattern as
 *     "lane n reads row n" was.
 * Used when decimation % 8 == 0 (decimation % 32 == 0: sub-planes, a 64-byte k-step never straddles a row; otherwise the
 * chunk-row layout, mfm_launch_v3::layout), taps <= 128 and the image fits LDS.
 * ------------------------------------------------------------------------------------------- */
#define MFM_V3_OT 64u          /* outputs per tile */
#define MFM_V3_LEAD 4u         /* rows staged in front of a tile (the first tile of a chunk recomputes output -1) */
#define MFM_V3_CH_MAX 8u       /* 16-byte staging chunks per thread and tile, at most */
#define MFM_V3L_TP 72u         /* layout 3: dwords per channel row of a wave's transposition area (64 outputs + 8: the four
                                  lane groups of a wave then write different LDS banks) */
#define MFM_V3L_KQ_MAX 16u     /* layout 3: k-steps of 64 elements, at most (512 taps) */
/* layout 3: the k-step counts and held high-byte-plane counts instances are built for (a filter runs on the next count up:
 * the surplus k-steps hold zero taps, the surplus planes zeros) */
static inline uint32_t mfm_v3l_built_kq(uint32_t kq_used)
{
    return kq_used <= 4u ? 4u : kq_used <= 6u ? 6u : kq_used <= 8u ? 8u : kq_used <= 12u ? kq_used : kq_used <= 14u ? 14u : 16u;
}
static inline uint32_t mfm_v3l_built_nh(uint32_t kq, uint32_t nh)
{
    const uint32_t b = nh == 0u ? 0u : nh <= 2u ? 2u : nh <= 4u ? 4u : kq;
    return b < kq ? b : kq;
}
/* ... and the staging chunks per thread */
/* bytes between the two byte planes of an image of the long-filter kernel (not the shifted-copies form): a constant of the
 * row-block count, so that a fragment's low-plane read is the high-plane read's address + an instruction immediate (half
 * the address arithmetic of the matrix phase); the engine takes geometries whose planes fit and lays the buffers out at twice
 * this pitch */
static inline uint32_t mfm_v3l_plane_pitch(uint32_t rb)
{
    return rb == 2u ? 24576u : 31744u;
}
static inline uint32_t mfm_v3l_built_nch(uint32_t nch)
{
    return nch <= 4u ? 4u : 8u;
}

struct mfm_launch_v3 {
    const uint32_t *x;    /* the input buffer: [hist samples already consumed | n_avail samples from the first unconsumed one] */
    uint32_t n_avail, n_new, decim;
    uint32_t hist;        /* 0 at the start of a stream, else the decimation: the D samples in front of the first unconsumed
                             one, i.e. the first row of the window of the output in front of this launch.  That output is
                             RECOMPUTED (its filtered sample is the discriminator's history, multifm/fm_demod.c:16-17), so
                             a launch depends on the launch before it through nothing but input samples */
    uint32_t pcm_scope;   /* 1: the PCM stores go through to memory (system scope) instead of waiting in L2 for their line to fill:
                             launches of 512 channels and more, where the PCM stream is several times the input's bytes and
                             would push the rotator tables out of L2 between two uses (mfm3_store_pcm4) */
    uint64_t k_base;      /* outputs the stream produced before this launch: output n of the launch is rotated by the
                             (k_base + n)-th state of the recurrence (filter/direct_fir.c:151-172), found by folding that
                             index into the channel's table - so the rotator position needs no carried state either */
    uint32_t x_last4;
    uint32_t kq;          /* k-steps of 64 elements (1, 2 or 4) */
    uint32_t rs;          /* LDS row stride in bytes, an odd multiple of 32 */
    uint32_t sp_pitch;    /* bytes between the four sub-planes (rows = 0,1,2,3 mod 4) of a byte plane */
    uint32_t plane_pitch; /* bytes between the high-byte and the low-byte plane = 4 * sp_pitch */
    uint32_t buf_pitch;   /* bytes between the two staging buffers = 2 * plane_pitch */
    uint32_t nstage4;     /* 16-byte chunks (4 samples; 8 with in8) staged per tile: (LEAD + 64 + extra rows) * D / 4 */
    uint32_t lut_off, sta_off;
    uint32_t cross[4];    /* per k-step: rows between an output's first sample and the k-step's first element */
    uint32_t within[4];   /* per k-step: byte offset of that element inside its row (for kg = 0) */
    /* layout 1 - decimations that are multiples of 8 but not of 32 (a 64-byte k-step then straddles rows, and the
     * sub-plane layout's address would need the row of every lane's 16 bytes): the image is kept as 16-byte plane chunks,
     * chunk s of the tile at row s % t_per, slot s / t_per of a [t_per + 3][t_pitch] array of 16-byte cells, where
     * t_per = 4 * D / 8 is four outputs' worth of chunks.  The window of output 4n + g starts at chunk t_per * (n + 1) +
     * (D / 8) * g, so "lane n reads slot n (+ a constant) of row r" for every (g, k-step): 16 consecutive 16-byte cells,
     * conflict free, and the address is again lane register + wave-uniform constant.  Rows 0..2 of each slot are kept a
     * second time as rows t_per..t_per + 2 of the slot before, so that the four cells of a k-step (one per kg) never wrap. */
    /* layout 2 - decimation 25 (etc/pocsag_rtlsdr.json): sub-planes as in layout 0, but a row is the 25 samples between two
     * outputs (50 plane bytes) padded to 64, the taps carry zeros over the padding, a k-step is exactly one row and a window
     * spans six of them (kq = 6).  The image is staged sample by sample (a 16-byte chunk of the input straddles rows). */
    uint32_t layout;      /* 0: four sub-planes per byte plane (rs, sp_pitch, cross, within); 1: chunk rows (t_per, t_pitch);
                             2: padded rows, decimation 25; 3: plain rows, long filters (below) */
    uint32_t t_per, t_pitch;
    uint32_t nslices, nrb;
    uint32_t ntiles;      /* ceil(n_new / 64) */
    uint32_t cl;          /* tiles per chunk */
    uint32_t nchunks;     /* ceil(ntiles / cl) */
    uint32_t nitems;      /* chunks rounded up to a multiple of 8, times slices */
    uint32_t nchan, out_stride, ah_mask;
    uint32_t rc;          /* the lowest MFM_RC_* of the launch's channels (selects the kernel instance) */
    uint32_t tail_src, tail_n; /* samples x[tail_src .. tail_src + tail_n) - the next launch's hist + history tail - go to the
                                  front of tail_dst (workgroup 0 copies them); tail_n = 0: the engine carries them itself */
    uint32_t in8;         /* 0: x is packed int16 IQ.  7 / 14: x is 8-bit IQ off the wire (2 bytes per sample; n_avail,
                             x_last4, hist, tail_* count samples all the same), the value is the first rounding's shift and
                             krow the matching row constants (mfm_kernel_v3.hip) */
    uint32_t in8_xor;     /* 0x80808080 when the bytes are unsigned (RTL-SDR), else 0 */
    uint32_t stream_taps; /* filters of 129..512 taps (kq 8 / 16): 1 = re-read the taps from L2 in every iteration (128-register
                             instances, two workgroups per CU where LDS allows); 0 = the resident instances (all taps in
                             registers, 256 registers, one workgroup per CU) where one is built for the geometry */
    /* layout 3 - filters of 129..512 taps (and shorter ones at decimations layouts 0-2 do not take) on the second-generation
     * structure, mfm_kernel_v3l.hip: the image is kept as plain rows (one row = the D samples between two outputs, 2 * D
     * plane bytes rounded up to 16 with zero taps over the padding, row stride rs an odd multiple of 32 bytes - the first
     * generation's image, so any decimation), column n of column group g of an image is its output 16 * g + n, and the four
     * consecutive outputs per lane that the epilogue wants come out of a wave-private transposition area in LDS.  An image
     * holds ng column groups: a whole tile (4) where two of them fit LDS, half a tile (2) for large decimations. */
    uint32_t row_bytes;   /* plane bytes of a row as the GEMM sees it (2 * D rounded up to 16) */
    uint32_t split_rows;  /* 1: a 4-sample staging chunk can straddle two rows (D % 4 != 0): stored sample by sample */
    uint32_t kq_used;     /* k-steps that hold taps at all (<= kq) */
    uint32_t nh;          /* k-steps whose high-byte tap plane is not all zero.  The engine lays the k-steps of a row block out
                             in the order kperm: those nh first - so an instance holds "the first NH high-byte planes" and
                             needs no mask - then the other ones that hold taps, then (up to kq, the instance's count) steps
                             of zero taps.  Integer sums do not care about the order. */
    uint32_t kperm[4];    /* byte j: which 64-element step of the window is multiplied j-th */
    uint32_t ng;          /* column groups per staged image: 4, 2 or 1 */
    uint32_t rb;          /* row blocks (16 GEMM rows = 8 channels) per wave: 1 - slices of 64 channels - or 2 - slices of 128:
                             half the staging work and half the B-fragment traffic per (channel, output), where two row
                             blocks' taps fit 128 registers (no or few high-byte planes: configs[4]'s filter) */
    uint32_t nstage_p;    /* 4-sample chunks of the one-group image in front of a chunk's first tile (its column 0 is the
                             output in front of the chunk, recomputed) */
    uint32_t shift;       /* 1: decimation 1, 2 or 4 - the image is kept 8 / D times, copy c shifted by 2 D c plane bytes and sp_pitch
                             bytes behind copy c - 1, so that every column's window starts 16-byte aligned in one of the copies
                             (mfm_kernel_v3l.hip, SHIFT); row_bytes = rs = 2 D, no padding; a staging chunk is one sample
                             (nstage4 / nstage_p count samples); kq is 4, 8 or 16 */
    uint32_t tp_off;      /* LDS byte offset of the transposition areas: [8 waves][8 * rb channels][72] dwords, then the fold
                             constants (512 * rb bytes) and the exact-rotator tables (2048 * rb bytes) */
    uint32_t *tail_dst;
    /* MFM_F_TIMING: how long the launch took in the shader's own clocks.  Every workgroup that had work stamps itself at its
     * start and its end (s_memtime: shader-clock ticks; s_memrealtime: the constant 100 MHz reference) and folds its duration
     * into cyc[0] / cyc[1] with an atomic maximum - the workgroups of these kernels are persistent, so the longest one IS the
     * launch.  The values carry the launch's tag in their top 24 bits (cyc_tag << 40), which makes a slot reusable without
     * clearing it.  NULL: no stamps. */
    unsigned long long *cyc;
    unsigned long long *cyc_clear; /* the two words of the slot half a ring ahead, zeroed by this launch (NULL with cyc) */
    uint32_t cyc_tag, pad2;
    const uint32_t *afrag;
    const int32_t *krow;
    const struct mfm_chan_info *info;
    const void *rot;      /* rotator tables: 8-byte entries {(rr, -ri), (ri, rr)}, or 4-byte ones (rr | ri << 16) - whichever
                             mfm_rot_entry_bytes_v3() says this kernel file was built for */
    const float2 *lut;
    int16_t *pcm;
    uint32_t *iq_dbg;
};
