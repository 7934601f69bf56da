/*
 * mfm_v3l_plan.h - the instruction schedule of one matrix phase of the long-filter channel kernel (mfm_kernel_v3l.hip),
 * computed at compile time.
 *
 * Why a schedule at all: on gfx950 the matrix instructions and the other vector instructions of a SIMD do not overlap, except
 * that up to two instructions behind a v_mfma of the same wave issue in its shadow (profiles/r02_ubench_shadow.txt).  Round 5's
 * long-filter kernel left that to the compiler and got "MFMA time with a VALU instruction beside it" 9-11 % and SIMDs busy
 * 0.55-0.72 (profiles/r05_rocprofv3_pmc_summary.txt); knock-out builds (profiles/r06_knockout_v3l.txt) put the image's
 * staging stores alone at 21 % of configs[4]'s launch.  So the phase is emitted as a fixed sequence of single-instruction asm
 * statements - the compiler allocates registers, nothing else - and THIS file decides the sequence: a stream of matrix
 * instructions with, in each gap, at most two "fillers" out of three queues:
 *   1. the request of the B fragments PF k-steps ahead (v_add_u32 + one or two ds_read_b128);
 *   2. recombination, first Q14 rounding (filter/complex.h:30-34) and transposition-area stores of the PREVIOUS column group,
 *      whose accumulators are a second set of registers (DB) and at least three matrix instructions old;
 *   3. the NEXT image's staging (byte-plane split of the int16 samples, filter/direct_fir.c:363-384's operand, + ds_write_b64).
 * The s_waitcnt lgkmcnt(N) in front of each k-step is counted here too: LDS operations of a wave complete in order, so N is
 * the number of LGKM operations - fragment reads, transposition and staging stores alike - issued since the k-step's own
 * reads (tools/lgkm_check.py replays the counter over the disassembly of every instance).
 *
 * Plain C++17 constexpr, no device code: tests/test_v3l_plan.py compiles it on the host and checks the invariants.
 */
#pragma once

#include <stdint.h>

enum : uint8_t { MFM3L_P_HH = 0, MFM3L_P_MDH = 1, MFM3L_P_LL = 2, MFM3L_P_MD = 3 };
/* products: HH = high-byte tap plane x high-byte sample plane -> hh; MDH = high x low -> md; LL = low x low -> ll;
 * MD = low x high -> md.  One sample plane (8-bit input): HH = high taps x samples -> hh, LL = low taps x samples -> ll. */

enum : uint8_t {
    MFM3L_F_NONE = 0,
    MFM3L_F_ADD,   /* a = step: fragment address of that step (its k-step's lane offset + the column group's base) */
    MFM3L_F_RDH,   /* a = step: ds_read_b128 of the high-byte (or only) sample plane */
    MFM3L_F_ADDL,  /* a = step: shifted-copies form only - the low plane's address needs its own add */
    MFM3L_F_RDL,   /* a = step: the low-byte plane */
    MFM3L_F_LA,    /* a = row block, b = sum index 0..3, c = source group, d = level: t = (x << 8) + y */
    MFM3L_F_SH0,   /* a = row block, b = channel 0..1, c = source group: f.lo16 = t[2b] >> shift */
    MFM3L_F_SH1,   /* ... f.hi16 = t[2b + 1] >> shift */
    MFM3L_F_TPW,   /* a = row block, b = channel, c = source group: the packed sample to the transposition area */
    MFM3L_F_STG,   /* a = staging chunk, b = operation index */
    MFM3L_F_NOP16  /* 16 wait states: a matrix result read by a vector instruction right behind it */
};
#define MFM3L_G_PEND 0xffu /* source group: the last group of the phase before (its accumulators are the other parity's) */

struct mfm3l_mf {
    uint8_t prod, r, kq, g;
    uint8_t init;  /* first write of its accumulator in this column group: C = 0 (hh, md) or the row constant (ll) */
    uint8_t wait;  /* 0xff: none; else s_waitcnt lgkmcnt(wait) in front - this is the first matrix instruction of its k-step */
    uint8_t step;
    uint8_t pad;
};

struct mfm3l_fl {
    uint8_t kind, a, b, c, d;
};

template <int NMF_, int NFL_>
struct mfm3l_plan {
    static constexpr int NMF = NMF_, NFL = NFL_;
    mfm3l_mf mf[NMF_ > 0 ? NMF_ : 1];
    int gap_lo[NMF_ > 0 ? NMF_ : 1], gap_hi[NMF_ > 0 ? NMF_ : 1]; /* fillers behind matrix instruction m: fl[gap_lo[m] .. gap_hi[m]) */
    mfm3l_fl fl[NFL_];
    int tail_lo, tail_hi;  /* fillers behind the last gap */
    int tail_mid;          /* ... of which [tail_lo, tail_mid) end with the last staging operation: the chunks whose staging only
                              finishes there are reloaded behind it, in front of the rest of the tail */
    int stg_done[9];       /* per staging chunk: the matrix instruction in whose gap its last operation sits (NMF: the tail) */
    int max_gap;           /* most fillers in one gap outside flush points (2 by construction) */
    int shadowed, total;   /* fillers placed in gaps / all fillers (NOP16 excluded) */
};

/* How a staging chunk (4 samples) gets from its registers into the image - the list of single instructions per chunk:
 *   MFM3L_STG_I16    int16 input, rows whose length is a multiple of 4 samples: address add, 2 x perm + ds_write_b64 of the high
 *                    plane, 2 x perm + 2 x xor + ds_write_b64 of the low plane (9);
 *   MFM3L_STG_I8     8-bit input (one sample plane): address add, 2 x xor, ds_write_b64 (4);
 *   MFM3L_STG_I16_S  int16, decimations that are not multiples of 4 - the chunk can straddle two rows, every sample has its own
 *                    address: 4 x perm, 2 x xor, 4 address adds, 4 + 4 x ds_write_b16 (18);
 *   MFM3L_STG_I8_S   the same for 8-bit input: 2 x xor, 4 address adds, 4 x ds_write_b16 (10). */
enum : int { MFM3L_STG_I16 = 0, MFM3L_STG_I8 = 1, MFM3L_STG_I16_S = 2, MFM3L_STG_I8_S = 3 };
constexpr int mfm3l_stg_ops(int mode)
{
    return mode == MFM3L_STG_I16 ? 9 : mode == MFM3L_STG_I8 ? 4 : mode == MFM3L_STG_I16_S ? 18 : 10;
}
constexpr bool mfm3l_stg_is_lds(int mode, int op)
{
    return mode == MFM3L_STG_I16 ? (op == 3 || op == 8) : mode == MFM3L_STG_I8 ? op == 3 : mode == MFM3L_STG_I16_S ? op >= 10 : op >= 6;
}
/* levels of shift-adds that recombine the byte-plane products: (hh << 8 + md) << 8 + ll; no high tap plane: md << 8 + ll;
 * one sample plane: hh << 8 + ll, or nothing */
constexpr int mfm3l_la_levels(bool in8, int nh)
{
    return in8 ? (nh > 0 ? 1 : 0) : (nh > 0 ? 2 : 1);
}
constexpr int mfm3l_nmf(int kq, int nh, int ngc, int rb, bool in8)
{
    return ngc * rb * (in8 ? 1 : 2) * (kq + nh);
}
constexpr int mfm3l_rec_items(int rb, bool in8, int nh)
{
    return rb * (4 * mfm3l_la_levels(in8, nh) + 6);
}

/*
 * KQ k-steps, the first NH with a high-byte tap plane; NGC column groups; RB row blocks per wave; one sample plane (IN8);
 * fragments requested PF k-steps ahead into PF + 1 rotating buffers; SHIFTRD: the low plane's read needs its own address;
 * DB: two accumulator sets - a group's recombination runs in the gaps of the next group (else behind its last matrix
 * instruction, after 16 wait states); PEND_IN: the phase before left its last group to this one; CARRY_OUT: this phase
 * leaves its last group to the next (else it is flushed in the tail); NSTGC staging chunks to store in the gaps, each the
 * instruction list of STGM.
 */
template <int KQ, int NH, int NGC, int RB, bool IN8, int PF, bool SHIFTRD, bool DB, bool PEND_IN, bool CARRY_OUT, int NSTGC, int STGM = (IN8 ? MFM3L_STG_I8 : MFM3L_STG_I16)>
constexpr auto mfm3l_make_plan()
{
    constexpr int NMF = mfm3l_nmf(KQ, NH, NGC, RB, IN8);
    constexpr int NS = NGC * KQ;
    constexpr int RPK = IN8 ? 1 : 2;
    constexpr int NLA = mfm3l_la_levels(IN8, NH);
    constexpr int NREC = mfm3l_rec_items(RB, IN8, NH);
    constexpr int SPC = mfm3l_stg_ops(STGM);
    constexpr int NREQ = SHIFTRD && !IN8 ? 4 : 1 + RPK;
    constexpr int NFL = NS * NREQ + (NGC + 1) * (NREC + 1) + NSTGC * SPC + 4;
    static_assert(!CARRY_OUT || DB, "a group can only be left to the next phase when there are two accumulator sets");
    static_assert(!PEND_IN || DB, "");
    mfm3l_plan<NMF, NFL> p{};

    /* ---- the matrix instructions, in order ---- */
    int m = 0;
    for (int g = 0; g < NGC; g++) {
        for (int kq = 0; kq < KQ; kq++) {
            const int m0 = m;
            if (kq < NH) {
                for (int r = 0; r < RB; r++) {
                    p.mf[m++] = mfm3l_mf{ MFM3L_P_HH, (uint8_t)r, (uint8_t)kq, (uint8_t)g, (uint8_t)(kq == 0), 0xff, (uint8_t)(g * KQ + kq), 0 };
                }
                if (!IN8) {
                    for (int r = 0; r < RB; r++) {
                        p.mf[m++] = mfm3l_mf{ MFM3L_P_MDH, (uint8_t)r, (uint8_t)kq, (uint8_t)g, (uint8_t)(kq == 0), 0xff, (uint8_t)(g * KQ + kq), 0 };
                    }
                }
            }
            for (int r = 0; r < RB; r++) {
                p.mf[m++] = mfm3l_mf{ MFM3L_P_LL, (uint8_t)r, (uint8_t)kq, (uint8_t)g, (uint8_t)(kq == 0), 0xff, (uint8_t)(g * KQ + kq), 0 };
            }
            if (!IN8) {
                for (int r = 0; r < RB; r++) {
                    p.mf[m++] = mfm3l_mf{ MFM3L_P_MD, (uint8_t)r, (uint8_t)kq, (uint8_t)g, (uint8_t)(kq == 0 && NH == 0), 0xff, (uint8_t)(g * KQ + kq), 0 };
                }
            }
            p.mf[m0].wait = 0; /* marks the first of a k-step; the count is filled in below */
        }
    }

    /* ---- queues ---- */
    mfm3l_fl req[NS * NREQ + 1] = {};
    int req_h = 0, req_t = 0;
    mfm3l_fl rec[(NGC + 1) * NREC + 1] = {};
    int rec_ready[(NGC + 1) * NREC + 1] = {};
    int rec_h = 0, rec_t = 0;
    mfm3l_fl stg[NSTGC * SPC + 1] = {};
    int stg_h = 0, stg_t = 0;
    for (int j = 0; j < NSTGC; j++) {
        for (int o = 0; o < SPC; o++) {
            stg[stg_t++] = mfm3l_fl{ MFM3L_F_STG, (uint8_t)j, (uint8_t)o, 0, 0 };
        }
        p.stg_done[j] = NMF;
    }
    auto push_rec = [&](int src, int ready) {
        for (int lvl = 0; lvl < NLA; lvl++) {
            for (int r = 0; r < RB; r++) {
                for (int i = 0; i < 4; i++) {
                    rec_ready[rec_t] = ready;
                    rec[rec_t++] = mfm3l_fl{ MFM3L_F_LA, (uint8_t)r, (uint8_t)i, (uint8_t)src, (uint8_t)lvl };
                }
            }
        }
        for (int k = 0; k < 3; k++) {
            for (int r = 0; r < RB; r++) {
                for (int c = 0; c < 2; c++) {
                    rec_ready[rec_t] = ready;
                    rec[rec_t++] = mfm3l_fl{ k == 0 ? MFM3L_F_SH0 : k == 1 ? MFM3L_F_SH1 : MFM3L_F_TPW, (uint8_t)r, (uint8_t)c, (uint8_t)src, 0 };
                }
            }
        }
    };
    if (PEND_IN) {
        push_rec(MFM3L_G_PEND, 0);
    }

    /* ---- the walk: LGKM bookkeeping and the gaps ---- */
    int nl = 0;             /* LGKM operations issued so far */
    int rd_done[NS + 1] = {}; /* nl right behind the last read of a step's fragments */
    for (int st = 0; st < PF && st < NS; st++) { /* the prologue's requests (the kernel emits them in front of the first matrix instruction) */
        nl += RPK;
        rd_done[st] = nl;
    }
    int nf = 0;
    p.max_gap = 0;
    p.shadowed = p.total = 0;
    auto emit = [&](const mfm3l_fl &f, int at_m) {
        p.fl[nf++] = f;
        if (f.kind == MFM3L_F_RDH || f.kind == MFM3L_F_RDL || f.kind == MFM3L_F_TPW ||
            (f.kind == MFM3L_F_STG && mfm3l_stg_is_lds(STGM, f.b))) {
            nl++;
        }
        if ((f.kind == MFM3L_F_RDH && IN8) || f.kind == MFM3L_F_RDL) {
            rd_done[f.a] = nl;
        }
        if (f.kind == MFM3L_F_STG && f.b == SPC - 1) {
            p.stg_done[f.a] = at_m;
        }
        if (f.kind != MFM3L_F_NOP16) {
            p.total++;
        }
    };
    for (m = 0; m < NMF; m++) {
        const mfm3l_mf d = p.mf[m];
        if (d.wait != 0xff) {
            const int st = d.step;
            const int n = nl - rd_done[st];
            p.mf[m].wait = (uint8_t)(n > 15 ? 15 : n);
            if (st + PF < NS) {
                const uint8_t s2 = (uint8_t)(st + PF);
                req[req_t++] = mfm3l_fl{ MFM3L_F_ADD, s2, 0, 0, 0 };
                req[req_t++] = mfm3l_fl{ MFM3L_F_RDH, s2, 0, 0, 0 };
                if (!IN8) {
                    if (SHIFTRD) {
                        req[req_t++] = mfm3l_fl{ MFM3L_F_ADDL, s2, 0, 0, 0 };
                    }
                    req[req_t++] = mfm3l_fl{ MFM3L_F_RDL, s2, 0, 0, 0 };
                }
            }
        }
        p.gap_lo[m] = nf;
        int used = 0;
        while (used < 2) {
            if (req_h < req_t) {
                emit(req[req_h++], m);
            } else if (rec_h < rec_t && rec_ready[rec_h] <= m) {
                emit(rec[rec_h++], m);
            } else if (stg_h < stg_t) {
                emit(stg[stg_h++], m);
            } else {
                break;
            }
            used++;
            p.shadowed++;
        }
        if (used > p.max_gap) {
            p.max_gap = used;
        }
        const bool last_of_group = m + 1 == NMF || p.mf[m + 1].g != d.g;
        if (last_of_group) {
            if (DB) {
                /* what is left of the group BEFORE this one reads the accumulator set the next group starts to overwrite: out
                 * with it now, shadow or not (filters with few k-steps per group: more fillers than gaps) */
                while (rec_h < rec_t && rec[rec_h].c != d.g) {
                    emit(rec[rec_h++], m);
                }
                if (d.g + 1 < NGC || !CARRY_OUT) {
                    push_rec(d.g, m + 3); /* three matrix instructions between an accumulator's last write and its first reader */
                }
            } else {
                /* one accumulator set: the group is finished on the spot */
                emit(mfm3l_fl{ MFM3L_F_NOP16, 0, 0, 0, 0 }, m);
                rec_h = rec_t = 0;
                push_rec(d.g, 0);
                while (rec_h < rec_t) {
                    emit(rec[rec_h++], m);
                }
            }
        }
        p.gap_hi[m] = nf;
    }
    /* ---- the tail ---- */
    p.tail_lo = nf;
    while (req_h < req_t) {
        emit(req[req_h++], NMF);
    }
    bool nopped = false;
    while (rec_h < rec_t) {
        if (rec_ready[rec_h] > NMF - 1 && !nopped) {
            /* (what is left of the staging first: it covers part of the wait) */
            while (stg_h < stg_t) {
                emit(stg[stg_h++], NMF);
            }
            emit(mfm3l_fl{ MFM3L_F_NOP16, 0, 0, 0, 0 }, NMF);
            nopped = true;
        }
        emit(rec[rec_h++], NMF);
    }
    while (stg_h < stg_t) {
        emit(stg[stg_h++], NMF);
    }
    p.tail_hi = nf;
    p.tail_mid = p.tail_lo;
    for (int i = p.tail_lo; i < p.tail_hi; i++) {
        if (p.fl[i].kind == MFM3L_F_STG) {
            p.tail_mid = i + 1;
        }
    }
    return p;
}
