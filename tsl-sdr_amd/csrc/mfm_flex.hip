/*
 * mfm_flex.hip - the FLEX front half of the pager stage, batched over all channels on the GPU: sync 1 search and
 * sampling, frame information word, sync 2, 2- / 4-level slicing and the block de-interleave
 * (pager/pager_flex.c:129-171,264-525,1200-1345,1401-1455).  See include/multifm_hip.h for the boundary and the
 * event format.
 *
 * The reference walks one sample at a time through three nested state machines.  Followed through, almost all
 * of it is a fixed function of ONE sample index per frame:
 *
 *   m       "a BS1 register reads 0xaaaaaaaa at this sample": the ten registers of pager_flex_sync take every
 *           tenth sample, so register (n mod 10) at sample n holds bits n, n-10, ..., n-310.  One bit per sample,
 *           32 samples per word: an AND of 32 funnel-shifted views of the sign bits (fx_match_kernel).  A register
 *           is zero-filled at a reset and the pattern's oldest bit is a one, so nothing can match for 310 samples
 *           after a reset - that is the only way the reset shows in m.
 *   j       first sample after a run of >= 3 set bits of m (counted modulo 256, the reference's counter is a
 *           uint8_t).  It fixes the sampling clock: the 112 bits of A / B / inverted A / FIW sit at
 *           j + t + 10 k, t = 10 - (run / 2) mod 10.
 *   f       = j + t + 1110, the last FIW bit.  After it the reference skips `sample_skip` samples between
 *           symbols: block symbol k of the frame is sample f + (skip + fudge + 1) + (sync2 symbols + k)(skip + 1),
 *           and bit n of a phase is a fixed symbol of the block (de-interleave: word 8 b + i, bit q  <-  bit
 *           256 b + 8 q + i of the phase).
 *
 * So per channel one wave walks from event to event (fx_walk_kernel): scan m for the next run (a summary bit per word
 * lets it step over 131 072 idle samples at a time), gather the 112 sync samples with two ballots, reduce the swing,
 * check the A code and the FIW (BCH table of the POCSAG stage, copied to LDS on first use) and jump to the frame's end,
 * leaving a descriptor (first symbol's sample, coding, swing).  The words of all frames found in the call are then built
 * in parallel (fx_gather_kernel, one workgroup per frame): every symbol is sliced once into an LDS byte, the 88 x
 * phases words are assembled from LDS.  Samples come from the caller's block or, for the part of a frame that arrived
 * with earlier calls, from a 32 768-sample history ring per channel (fx_hist_kernel refreshes it last).
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/multifm_hip.h"
#include "mfm_bch.h"

extern "C" __attribute__((visibility("hidden"))) void mfm_internal_set_error(const char *msg);

namespace {

constexpr uint32_t FX_HIST = 32768; /* samples of history per channel; a frame end looks back 28 560 + 1 440 of them */
constexpr uint32_t FX_DEAD = 311;   /* first sample after a reset at r that can complete a BS1 register: r + 311 */
constexpr uint32_t FX_MATCH_WORDS = 256; /* m words per workgroup of the match kernel: one per thread */
constexpr uint32_t FX_HALO_WORDS = 10;   /* 310 samples of look-back, rounded up to words */
constexpr uint32_t FX_GATHER_THREADS = 256; /* one workgroup per frame builds its 88 x phases words */

enum : uint32_t { FX_SEARCH = 0, FX_SYNC1 = 1, FX_FRAME = 2 };

/* _pager_codings[] (pager_flex.c:46-96); sync2 = 2 * (sync_2_samples + 16 / sym_bits) symbols (:460-525) */
struct FxCoding {
    uint16_t seq_a, baud;
    uint8_t levels, skip, fudge, nr_phases;
    uint16_t sync2, symbols;
};
__device__ const FxCoding fx_codings[4] = {
    { 0x78f3, 1600, 2, 9, 0, 1, 40, 2816 },
    { 0x84e7, 3200, 2, 4, 2, 2, 80, 5632 },
    { 0x4f97, 3200, 4, 9, 0, 2, 40, 2816 },
    { 0x215f, 6400, 4, 4, 2, 4, 80, 5632 },
};

/* where PCM sample s of channel c lives: the caller's block for s >= base, the history ring below */
struct FxIn {
    const int16_t *x;
    size_t stride;
    const int16_t *hist;
    uint64_t base, end; /* the block holds samples [base, end) */
};

__device__ __forceinline__ int fx_sample(const FxIn &I, uint32_t c, uint64_t s)
{
    /* one load either way: pick the address, not the value */
    const int16_t *in_block = I.x + (size_t)c * I.stride + (size_t)(s - I.base);
    const int16_t *in_ring = I.hist + (size_t)c * FX_HIST + (size_t)(s & (FX_HIST - 1));
    return *(s >= I.base ? in_block : in_ring);
}

/* ---- m: one bit per sample, "the register this sample goes to now reads BS1" -------------------------------- */

struct __attribute__((packed, aligned(2))) FxPcm8 {
    int16_t v[8];
};

/* sign bits (sample >= 0, pager_flex.c:137) of the eight samples s .. s + 7, s a multiple of 8 */
__device__ __forceinline__ uint32_t fx_sign8(const FxIn &I, uint32_t c, int64_t s)
{
    uint32_t b = 0;
    if (s >= (int64_t)I.base && (uint64_t)s + 8 <= I.end) {
        /* the caller's block: any 2-byte alignment, one 16-byte load */
        const FxPcm8 p = *reinterpret_cast<const FxPcm8 *>(I.x + (size_t)c * I.stride + (size_t)((uint64_t)s - I.base));
#pragma unroll
        for (int i = 0; i < 8; i++) {
            b |= (uint32_t)(p.v[i] >= 0) << i;
        }
    } else if (s >= 0 && (uint64_t)s + 8 <= I.base) {
        /* the history ring: rows and s are multiples of 8, so the eight samples are contiguous and aligned */
        const uint4 q = *reinterpret_cast<const uint4 *>(I.hist + (size_t)c * FX_HIST + (size_t)((uint64_t)s & (FX_HIST - 1)));
        const uint32_t w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            b |= (uint32_t)((w[i] & 0x8000u) == 0) << (2 * i);
            b |= (uint32_t)((w[i] & 0x80000000u) == 0) << (2 * i + 1);
        }
    } else {
        /* a group that straddles the start of the stream, the ring / block seam or the end of the block */
        for (int i = 0; i < 8; i++) {
            const int64_t si = s + i;
            if (si >= 0 && (uint64_t)si < I.end) {
                b |= (uint32_t)(fx_sample(I, c, (uint64_t)si) >= 0) << i;
            }
        }
    }
    return b;
}

/*
 * M[c][i] covers samples 32 (w0 + i) .. + 31; bits at or beyond the end of the block are zero.  SUM[c][i / 64] bit
 * (i % 64) says M[c][i] is not zero, so that the walk can step over 4096 words at a time.
 */
__global__ __launch_bounds__(256) void fx_match_kernel(const FxIn I, uint32_t *M, uint32_t mstride, uint64_t *SUM, uint32_t sstride,
                                                       uint64_t w0, uint32_t nwords)
{
    __shared__ uint32_t bits[FX_MATCH_WORDS + FX_HALO_WORDS + 2];
    uint8_t *bytes = reinterpret_cast<uint8_t *>(bits);
    const uint32_t c = blockIdx.y, wb = blockIdx.x * FX_MATCH_WORDS, t = threadIdx.x;
    /* sign bits of samples [32 (w0 + wb - 10), 32 (w0 + wb + 256)), eight per byte */
    const int64_t s_first = ((int64_t)(w0 + wb) - (int64_t)FX_HALO_WORDS) * 32;
    for (uint32_t g = t; g < 4 * (FX_MATCH_WORDS + FX_HALO_WORDS); g += 256) {
        bytes[g] = (uint8_t)fx_sign8(I, c, s_first + 8 * (int64_t)g);
    }
    if (t == 0) {
        bits[FX_MATCH_WORDS + FX_HALO_WORDS] = 0;
    }
    __syncthreads();
    uint32_t m = 0;
    if (wb + t < nwords) {
        /* register bit k (k = 0 newest) is the sign bit 10 k samples back; BS1 wants a one at every odd k */
        m = 0xffffffffu;
#pragma unroll
        for (uint32_t k = 0; k < 32; k++) {
            const uint32_t off = 32 * FX_HALO_WORDS + 32 * t - 10 * k;
            const uint32_t v = __funnelshift_r(bits[off >> 5], bits[(off >> 5) + 1], off & 31);
            m &= (k & 1) ? v : ~v;
        }
        const uint64_t i0 = (w0 + wb + t) * 32;
        if (i0 + 32 > I.end) {
            m &= (I.end > i0) ? (0xffffffffu >> (32 - (uint32_t)(I.end - i0))) : 0u;
        }
        M[(size_t)c * mstride + wb + t] = m;
    }
    const uint64_t any = __ballot(m != 0);
    if ((t & 63) == 0 && wb + t < nwords) {
        SUM[(size_t)c * sstride + ((wb + t) >> 6)] = any;
    }
}

/* ---- the walk from event to event ------------------------------------------------------------------------- */

struct FxState {
    uint64_t p;        /* SEARCH: next sample to look at */
    uint64_t j;        /* SYNC1: the sample that ended the BS1 run; FRAME: the sample of the last FIW bit */
    uint32_t mode;
    uint32_t run;      /* SEARCH: matches counted in the run that is open at p */
    uint32_t eye;      /* the run length modulo 256 that opened the eye */
    uint32_t coding;
    uint32_t a, b, inv_a, fiw_raw, fiw;
    int32_t range, delta;
    uint32_t cycle, frame, pad;
};

/* a collected frame, as the walk leaves it for the gather kernel */
struct FxFrameDesc {
    uint64_t first;    /* sample of the block's first symbol */
    uint32_t coding;
    int32_t range, delta;
    uint32_t pad;
};

struct FxWalk {
    FxIn I;
    const uint32_t *M;
    const uint64_t *SUM;
    uint32_t mstride, sstride, nwords;
    uint64_t w0;
    FxState *st;
    mfm_flex_event *ev;
    mfm_flex_frame_words *fw;
    struct FxFrameDesc *fd; /* [channel][max_fw]: what fx_gather_kernel needs to build fw */
    uint32_t *counts;  /* [channel][2]: events, frames of this call */
    uint32_t max_ev, max_fw;
    const MfmBchTables *bch;
};

__device__ __forceinline__ int fx_wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += __shfl_xor(v, d);
    }
    return v;
}

/* pager_flex.c:107-119 */
__device__ __forceinline__ uint32_t fx_checksum(uint32_t w)
{
    w &= 0x1fffffu;
    uint32_t s = 0;
#pragma unroll
    for (int n = 0; n < 6; n++) {
        s += (w >> (4 * n)) & 0xfu;
    }
    return s & 0xfu;
}

/* _pager_flex_slice_4fsk, pager_flex.c:148-171 */
__device__ __forceinline__ uint32_t fx_slice4(int v, int delta, int range)
{
    const int s = (int16_t)(v - delta);
    if (s < 0) {
        return (-s > range / 4) ? 0u : 1u;
    }
    return (s > range / 4) ? 2u : 3u;
}

__device__ void fx_emit(const FxWalk &L, uint32_t c, uint32_t &nev, uint32_t type, uint64_t sample, const FxState &S, uint32_t fiw_rc,
                        uint32_t frame_index)
{
    if (nev < L.max_ev && threadIdx.x == 0) {
        mfm_flex_event e;
        e.type = type;
        e.channel = c;
        e.sample = sample;
        e.sync_sample = (type == MFM_FLEX_EV_FRAME) ? S.j : 0;
        e.coding = S.coding;
        e.baud = (S.coding < 4) ? fx_codings[S.coding].baud : 0;
        e.eye = S.eye;
        e.a = S.a;
        e.b = S.b;
        e.inv_a = S.inv_a;
        e.fiw_raw = S.fiw_raw;
        e.fiw = S.fiw;
        e.fiw_rc = fiw_rc;
        e.sample_range = S.range;
        e.sample_delta = S.delta;
        e.cycle = S.cycle;
        e.frame = S.frame;
        e.frame_index = frame_index;
        e.nr_phases = (S.coding < 4) ? fx_codings[S.coding].nr_phases : 0;
        e.reserved = 0;
        L.ev[(size_t)c * L.max_ev + nev] = e;
    }
    nev++;
}

__global__ __launch_bounds__(64) void fx_walk_kernel(const FxWalk L)
{
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    const FxIn &I = L.I;
    const uint64_t end = I.end;
    const uint32_t *M = L.M + (size_t)c * L.mstride;
    const uint64_t *SUM = L.SUM + (size_t)c * L.sstride;
    const uint32_t nsum = (L.nwords + 63) >> 6;
    /* the BCH tables, copied on the first frame information word of the call (an idle channel never pays for it) */
    __shared__ MfmBchTables bch_s;
    bool have_bch = false;
    FxState S = L.st[c];
    uint32_t nev = 0, nfw = 0;

    for (;;) {
        if (S.mode == FX_SEARCH) {
            bool opened = false;
            while (S.p < end) {
                if (S.run == 0) {
                    /* nothing open: go to the next set bit of m.  First the rest of the word p is in ... */
                    const uint32_t i = (uint32_t)((S.p >> 5) - L.w0);
                    const uint32_t head = M[i] & (0xffffffffu << (S.p & 31));
                    if (head != 0) {
                        S.p = (S.p & ~(uint64_t)31) + ((uint32_t)__ffs((int)head) - 1); /* < end: m is masked there */
                    } else {
                        /* ... then whole words through the summary, 64 x 64 words = 131 072 samples per step */
                        const uint32_t j = i + 1, sj = j >> 6;
                        uint64_t v = (sj + lane < nsum) ? SUM[sj + lane] : 0ull;
                        if (lane == 0) {
                            v &= ~0ull << (j & 63);
                        }
                        const uint64_t nz = __ballot(v != 0);
                        if (nz == 0) {
                            const uint64_t next = (L.w0 + (((uint64_t)sj + 64) << 6)) << 5;
                            S.p = next < end ? next : end; /* what lies beyond `end` belongs to the next call */
                            continue;
                        }
                        const uint32_t fl = (uint32_t)__ffsll((unsigned long long)nz) - 1;
                        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)fl), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)fl);
                        const uint64_t fv = (uint64_t)hi << 32 | lo;
                        const uint32_t wi = ((sj + fl) << 6) + ((uint32_t)__ffsll((unsigned long long)fv) - 1);
                        S.p = ((L.w0 + wi) << 5) + ((uint32_t)__ffs((int)M[wi]) - 1);
                    }
                }
                /* count the run on from p, one word at a time */
                const uint32_t sh = (uint32_t)(S.p & 31);
                const uint32_t inv = ~(M[(S.p >> 5) - L.w0] >> sh);
                uint32_t avail = 32 - sh;
                if (end - S.p < avail) {
                    avail = (uint32_t)(end - S.p);
                }
                uint32_t ones = (uint32_t)__ffs((int)inv) - 1; /* inv has a set bit: the shift brought in zeros, or sh = 0 ... */
                if (inv == 0) {
                    ones = 32;
                }
                if (ones > avail) {
                    ones = avail;
                }
                S.run += ones;
                S.p += ones;
                if (ones < avail) {
                    /* sample p does not match: the run is over (pager_flex.c:328-344) */
                    const uint32_t cnt = S.run & 255u;
                    S.run = 0;
                    if (cnt >= 3) {
                        S.mode = FX_SYNC1;
                        S.j = S.p;
                        S.eye = cnt;
                        opened = true;
                        break;
                    }
                    S.p += 1;
                }
            }
            if (!opened) {
                break; /* out of samples; a run that is still open goes on in the next call */
            }
        }

        if (S.mode == FX_SYNC1) {
            /* the sample counter was set to run / 2 at j and a bit is taken whenever it wraps to 0 (:339,:348) */
            const uint64_t s0 = S.j + (10 - ((S.eye / 2) % 10));
            if (s0 + 790 >= end) {
                break;
            }
            const bool have_fiw = s0 + 1110 < end;
            const int v0 = fx_sample(I, c, s0 + 10 * lane); /* bits 0..63 */
            const uint32_t k1 = 64 + lane;                  /* bits 64..111 */
            const bool use1 = k1 < 80 || (have_fiw && k1 < 112);
            const int v1 = use1 ? fx_sample(I, c, s0 + 10 * k1) : 0;
            const uint64_t bal0 = __ballot(v0 >= 0), bal1 = __ballot(use1 && v1 >= 0);
            S.a = __brev((uint32_t)bal0);                                   /* shifted in MSB first (:349) */
            S.b = __brev((uint32_t)(bal0 >> 32) & 0xffffu) >> 16;
            S.inv_a = __brev((uint32_t)(bal0 >> 48) | ((uint32_t)bal1 << 16));
            S.fiw_raw = 0;
            S.fiw = 0;
            S.range = 0;
            S.delta = 0;
            S.cycle = 0;
            S.frame = 0;
            S.coding = 0xffffffffu;
            for (uint32_t i = 0; i < 4; i++) { /* :264-287; the inverted word can never pass its test */
                if (S.coding == 0xffffffffu && __popc((uint32_t)fx_codings[i].seq_a ^ (S.a >> 16)) < 4) {
                    S.coding = i;
                }
            }
            if (S.coding == 0xffffffffu) {
                fx_emit(L, c, nev, MFM_FLEX_EV_BAD_BAUD, s0 + 790, S, 0, 0);
                S.mode = FX_SEARCH;
                S.run = 0;
                S.p = s0 + 790 + FX_DEAD;
                continue;
            }
            if (!have_fiw) {
                break;
            }
            S.fiw_raw = (uint32_t)(bal1 >> 16); /* shifted in LSB first (:422) */
            /* swing of the 112 sync samples (:352-358, :438-442) */
            const bool in1 = k1 < 112;
            const int sum_hi = fx_wave_sum((v0 > 0 ? v0 : 0) + (in1 && v1 > 0 ? v1 : 0));
            const int sum_lo = fx_wave_sum((v0 <= 0 ? v0 : 0) + (in1 && v1 <= 0 ? v1 : 0));
            const int n_hi = fx_wave_sum((v0 > 0) + (in1 && v1 > 0));
            const int n_lo = fx_wave_sum((v0 <= 0) + (in1 && v1 <= 0));
            const uint64_t f = s0 + 1110;
            uint32_t rc;
            if (n_hi == 0 || n_lo == 0) {
                rc = 3;
            } else {
                const int high = (int16_t)(sum_hi / n_hi), low = (int16_t)(sum_lo / n_lo);
                S.range = (int16_t)(high - low);
                S.delta = (int16_t)(high - S.range / 2);
                uint32_t bad;
                if (!have_bch) {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(L.bch);
                    uint32_t *dst = reinterpret_cast<uint32_t *>(&bch_s);
                    for (uint32_t i = lane; i < sizeof(MfmBchTables) / 4u; i += 64u) {
                        dst[i] = src[i];
                    }
                    have_bch = true; /* one wave: its LDS writes are visible to its later reads in program order */
                }
                S.fiw = mfm_bch_fix(&bch_s, S.fiw_raw & 0x7fffffffu, &bad); /* :1319-1327 */
                if (bad) {
                    rc = 1;
                } else if (fx_checksum(S.fiw) != 0xfu) {
                    rc = 2;
                } else {
                    rc = 0;
                    S.cycle = (S.fiw >> 4) & 0xfu;
                    S.frame = (S.fiw >> 8) & 0x7fu;
                }
            }
            if (rc != 0) {
                fx_emit(L, c, nev, MFM_FLEX_EV_BAD_FIW, f, S, rc, 0);
                S.mode = FX_SEARCH;
                S.run = 0;
                S.p = f + FX_DEAD;
                continue;
            }
            S.mode = FX_FRAME;
            S.j = f;
        }

        if (S.mode == FX_FRAME) {
            const FxCoding cd = fx_codings[S.coding];
            const uint32_t step = cd.skip + 1u;
            /* first processed sample after f is f + skip + fudge + 1 (:1421-1423, :1410-1451), then one per `step` */
            const uint64_t first = S.j + step + cd.fudge + (uint64_t)cd.sync2 * step;
            const uint64_t e = first + (uint64_t)(cd.symbols - 1u) * step;
            if (e >= end) {
                break;
            }
            if (nfw < L.max_fw && nev < L.max_ev && lane == 0) {
                /* the words are built by fx_gather_kernel, one workgroup per frame, behind this kernel */
                L.fd[(size_t)c * L.max_fw + nfw] = FxFrameDesc{ first, S.coding, S.range, S.delta, 0u };
            }
            fx_emit(L, c, nev, MFM_FLEX_EV_FRAME, e, S, 0, nfw);
            nfw++;
            S.mode = FX_SEARCH; /* _pager_flex_reset_sync (:1308) */
            S.run = 0;
            S.p = e + FX_DEAD;
        }
    }

    if (lane == 0) {
        L.st[c] = S;
        L.counts[2 * c] = nev;
        L.counts[2 * c + 1] = nfw;
    }
}

/*
 * The words of the frames the walk found: grid (max_fw, channels), one workgroup per frame.  Every symbol of the block
 * is sliced once (consecutive threads take consecutive symbols, 10 or 20 bytes apart) into one LDS byte, then the
 * 88 x phases words are built from LDS: bit jb of word 8 b + i of a phase is bit 256 b + 8 jb + i of that phase.
 */
__global__ __launch_bounds__(FX_GATHER_THREADS) void fx_gather_kernel(const FxWalk L)
{
    __shared__ uint8_t sym[5632];
    const uint32_t c = blockIdx.y, f = blockIdx.x;
    const uint32_t nfw = L.counts[2 * c + 1] < L.max_fw ? L.counts[2 * c + 1] : L.max_fw;
    if (f >= nfw) {
        return;
    }
    const FxFrameDesc d = L.fd[(size_t)c * L.max_fw + f];
    const FxCoding cd = fx_codings[d.coding];
    const uint32_t step = cd.skip + 1u;
    const bool four = cd.levels == 4;
    uint32_t *out = &L.fw[(size_t)c * L.max_fw + f].words[0][0];
#pragma unroll 11
    for (uint32_t k = threadIdx.x; k < cd.symbols; k += FX_GATHER_THREADS) {
        const int v = fx_sample(L.I, c, d.first + (uint64_t)k * step);
        sym[k] = (uint8_t)(four ? fx_slice4(v, d.delta, d.range) : (uint32_t)(v >= 0)); /* 2-level: 1 == symbol (:1246) */
    }
    __syncthreads();
    for (uint32_t item = threadIdx.x; item < 4 * MFM_FLEX_PHASE_WORDS; item += FX_GATHER_THREADS) {
        const uint32_t q = item / MFM_FLEX_PHASE_WORDS, w = item % MFM_FLEX_PHASE_WORDS;
        /* which symbols phase q rides on (:1242-1285): every one or every second, and which bit of a 4-level one */
        bool present;
        uint32_t mul = 1, add = 0, sel = 0;
        if (cd.nr_phases == 1) {
            present = q == 0;
        } else if (cd.nr_phases == 2) {
            present = q == 0 || q == 2;
            if (four) {
                sel = q == 0;
            } else {
                mul = 2;
                add = q >> 1;
            }
        } else {
            present = true;
            mul = 2;
            add = q >> 1;
            sel = (q & 1) == 0;
        }
        uint32_t word = 0;
        if (present) {
            const uint32_t n0 = (w >> 3) * 256 + (w & 7);
#pragma unroll
            for (uint32_t jb = 0; jb < 32; jb++) {
                word |= (((uint32_t)sym[(n0 + 8 * jb) * mul + add] >> sel) & 1u) << jb;
            }
        }
        out[item] = word;
    }
}

/* the last min(n, FX_HIST) samples of the block into the ring */
__global__ __launch_bounds__(256) void fx_hist_kernel(int16_t *hist, const int16_t *x, size_t stride, uint64_t base, uint32_t n)
{
    const uint32_t c = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    const uint32_t cnt = n < FX_HIST ? n : FX_HIST;
    if (k < cnt) {
        const uint32_t off = n - cnt + k;
        hist[(size_t)c * FX_HIST + (size_t)((base + off) & (FX_HIST - 1))] = x[(size_t)c * stride + off];
    }
}

thread_local char g_fx_error[256] = "";

} /* namespace */

#define FX_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t err_ = (expr);                                                                            \
        if (err_ != hipSuccess) {                                                                            \
            snprintf(g_fx_error, sizeof(g_fx_error), "%s failed: %s", #expr, hipGetErrorString(err_));       \
            mfm_internal_set_error(g_fx_error);                                                              \
            return MFM_E_DEVICE;                                                                             \
        }                                                                                                    \
    } while (0)

struct mfm_flex {
    mfm_flex_config cfg{};
    uint32_t max_ev = 0, max_fw = 0, mstride = 0, sstride = 0;
    int16_t *d_hist = nullptr;
    uint32_t *d_m = nullptr;
    uint64_t *d_sum = nullptr;
    FxState *d_st = nullptr;
    mfm_flex_event *d_ev = nullptr;
    mfm_flex_frame_words *d_fw = nullptr;
    FxFrameDesc *d_fd = nullptr;
    uint32_t *d_counts = nullptr;
    MfmBchTables *d_bch = nullptr;
    uint64_t total = 0;
    hipStream_t last_stream = nullptr;
    bool have_call = false;
};

extern "C" {

/* inside mfm_flex_create(): a device failure releases what exists and leaves *pf NULL */
#define FX_TRY_C(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t err_c_ = (expr);                                                                          \
        if (err_c_ != hipSuccess) {                                                                          \
            char msg_c_[256];                                                                                \
            snprintf(msg_c_, sizeof(msg_c_), "%s failed: %s", #expr, hipGetErrorString(err_c_));             \
            mfm_internal_set_error(msg_c_);                                                                  \
            mfm_flex_destroy(pf);                                                                            \
            return MFM_E_DEVICE;                                                                             \
        }                                                                                                    \
    } while (0)

int mfm_flex_create(struct mfm_flex **pf, const struct mfm_flex_config *cfg)
{
    if (!pf || !cfg) {
        return MFM_E_INVAL;
    }
    *pf = nullptr;
    if (cfg->abi_version != MFM_ABI_VERSION || 0 == cfg->nr_channels || 0 == cfg->max_in_samples ||
        cfg->max_in_samples > (1u << 26) || cfg->flags != 0) {
        return MFM_E_INVAL;
    }
    MfmBchTables *d_bch = nullptr;
    const int rc = mfm_internal_bch_device_tables(cfg->device, &d_bch); /* MFM_E_DEVICE without a GPU: no CPU path */
    if (rc != MFM_OK) {
        return rc;
    }
    mfm_flex *f = new (std::nothrow) mfm_flex();
    if (!f) {
        return MFM_E_NOMEM;
    }
    f->cfg = *cfg;
    f->d_bch = d_bch;
    /* an event needs at least 311 + 3 + 1 + 790 fresh samples, a frame 28 560 more */
    f->max_ev = cfg->max_events ? cfg->max_events : cfg->max_in_samples / 1024 + 8;
    f->max_fw = cfg->max_in_samples / 28672 + 2;
    if (f->max_fw > f->max_ev) {
        f->max_fw = f->max_ev;
    }
    f->mstride = cfg->max_in_samples / 32 + 4;
    f->sstride = f->mstride / 64 + 2;
    const size_t C = cfg->nr_channels;
    *pf = f;
    FX_TRY_C(hipSetDevice(cfg->device));
    FX_TRY_C(hipMalloc(&f->d_hist, C * FX_HIST * sizeof(int16_t)));
    FX_TRY_C(hipMemset(f->d_hist, 0, C * FX_HIST * sizeof(int16_t)));
    FX_TRY_C(hipMalloc(&f->d_m, C * f->mstride * sizeof(uint32_t)));
    FX_TRY_C(hipMalloc(&f->d_sum, C * f->sstride * sizeof(uint64_t)));
    FX_TRY_C(hipMalloc(&f->d_st, C * sizeof(FxState)));
    {
        /* pager_flex_new (:1371): registers zero-filled "before sample 0", so the search opens at sample 310 */
        std::vector<FxState> st(C);
        memset(st.data(), 0, C * sizeof(FxState));
        for (size_t c = 0; c < C; c++) {
            st[c].mode = FX_SEARCH;
            st[c].p = FX_DEAD - 1;
        }
        FX_TRY_C(hipMemcpy(f->d_st, st.data(), C * sizeof(FxState), hipMemcpyHostToDevice));
    }
    FX_TRY_C(hipMalloc(&f->d_ev, C * f->max_ev * sizeof(mfm_flex_event)));
    FX_TRY_C(hipMalloc(&f->d_fw, C * f->max_fw * sizeof(mfm_flex_frame_words)));
    FX_TRY_C(hipMalloc(&f->d_fd, C * f->max_fw * sizeof(FxFrameDesc)));
    FX_TRY_C(hipMalloc(&f->d_counts, C * 2 * sizeof(uint32_t)));
    FX_TRY_C(hipMemset(f->d_counts, 0, C * 2 * sizeof(uint32_t)));
    FX_TRY_C(hipDeviceSynchronize());
    return MFM_OK;
}

void mfm_flex_destroy(struct mfm_flex **pf)
{
    if (!pf || !*pf) {
        return;
    }
    mfm_flex *f = *pf;
    (void)hipSetDevice(f->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(f->d_hist);
    (void)hipFree(f->d_m);
    (void)hipFree(f->d_sum);
    (void)hipFree(f->d_st);
    (void)hipFree(f->d_ev);
    (void)hipFree(f->d_fw);
    (void)hipFree(f->d_fd);
    (void)hipFree(f->d_counts);
    delete f;
    *pf = nullptr;
}

int mfm_flex_process_device(struct mfm_flex *f, const int16_t *d_pcm, size_t in_stride, size_t nr_in, void *stream)
{
    if (!f || (!d_pcm && nr_in) || nr_in > f->cfg.max_in_samples || (nr_in && in_stride < nr_in)) {
        return MFM_E_INVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t C = f->cfg.nr_channels, n = (uint32_t)nr_in;
    FX_TRY(hipSetDevice(f->cfg.device));
    if (f->have_call && f->last_stream != s) {
        FX_TRY(hipStreamSynchronize(f->last_stream)); /* state lives on the device; keep calls ordered */
    }
    f->last_stream = s;
    f->have_call = true;
    if (0 == n) {
        FX_TRY(hipMemsetAsync(f->d_counts, 0, (size_t)C * 2 * sizeof(uint32_t), s));
        return MFM_OK;
    }
    const FxIn in{ d_pcm, in_stride, f->d_hist, f->total, f->total + n };
    const uint64_t w0 = in.base >> 5;
    const uint32_t nwords = (uint32_t)(((in.end - 1) >> 5) - w0 + 1);
    hipLaunchKernelGGL(fx_match_kernel, dim3((nwords + FX_MATCH_WORDS - 1) / FX_MATCH_WORDS, C), dim3(256), 0, s, in, f->d_m,
                       f->mstride, f->d_sum, f->sstride, w0, nwords);
    FX_TRY(hipGetLastError());
    const FxWalk W{ in, f->d_m, f->d_sum, f->mstride, f->sstride, nwords, w0, f->d_st, f->d_ev, f->d_fw, f->d_fd, f->d_counts, f->max_ev, f->max_fw, f->d_bch };
    hipLaunchKernelGGL(fx_walk_kernel, dim3(C), dim3(64), 0, s, W);
    FX_TRY(hipGetLastError());
    /* the frames' words: reads the same samples, so it runs before the history ring is refreshed */
    /* a call of n samples cannot end more than n / 28 672 + 2 frames per channel */
    const uint32_t fw_now = n / 28672u + 2u < f->max_fw ? n / 28672u + 2u : f->max_fw;
    hipLaunchKernelGGL(fx_gather_kernel, dim3(fw_now, C), dim3(FX_GATHER_THREADS), 0, s, W);
    FX_TRY(hipGetLastError());
    const uint32_t cnt = n < FX_HIST ? n : FX_HIST;
    hipLaunchKernelGGL(fx_hist_kernel, dim3((cnt + 255) / 256, C), dim3(256), 0, s, f->d_hist, d_pcm, in_stride, in.base, n);
    FX_TRY(hipGetLastError());
    f->total += n;
    return MFM_OK;
}

int mfm_flex_process_host(struct mfm_flex *f, const int16_t *pcm, size_t in_stride, size_t nr_in)
{
    if (!f || (!pcm && nr_in)) {
        return MFM_E_INVAL;
    }
    FX_TRY(hipSetDevice(f->cfg.device));
    const uint32_t C = f->cfg.nr_channels;
    int16_t *d_in = nullptr;
    FX_TRY(hipMalloc(&d_in, (size_t)C * (nr_in ? nr_in : 1) * 2));
    if (nr_in) {
        FX_TRY(hipMemcpy2D(d_in, nr_in * 2, pcm, in_stride * 2, nr_in * 2, C, hipMemcpyHostToDevice));
    }
    const int rc = mfm_flex_process_device(f, d_in, nr_in, nr_in, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipFree(d_in);
    return rc;
}

int mfm_flex_fetch_events(struct mfm_flex *f, struct mfm_flex_event *events, size_t max_events, size_t *nr_events,
                          struct mfm_flex_frame_words *frames, size_t max_frames, size_t *nr_frames)
{
    if (!f || !nr_events || !nr_frames || (!events && max_events) || (!frames && max_frames)) {
        return MFM_E_INVAL;
    }
    *nr_events = 0;
    *nr_frames = 0;
    if (!f->have_call) {
        return MFM_OK;
    }
    FX_TRY(hipSetDevice(f->cfg.device));
    FX_TRY(hipStreamSynchronize(f->last_stream));
    const uint32_t C = f->cfg.nr_channels;
    std::vector<uint32_t> cnt((size_t)C * 2);
    FX_TRY(hipMemcpy(cnt.data(), f->d_counts, (size_t)C * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    size_t tot_ev = 0, tot_fw = 0;
    bool overflow = false;
    for (uint32_t c = 0; c < C; c++) {
        overflow |= cnt[2 * c] > f->max_ev || cnt[2 * c + 1] > f->max_fw;
        tot_ev += cnt[2 * c] > f->max_ev ? f->max_ev : cnt[2 * c];
        tot_fw += cnt[2 * c + 1] > f->max_fw ? f->max_fw : cnt[2 * c + 1];
    }
    *nr_events = tot_ev;
    *nr_frames = tot_fw;
    if (tot_ev > max_events || tot_fw > max_frames) {
        return MFM_E_NOMEM;
    }
    size_t pe = 0, pw = 0;
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t ke = cnt[2 * c] > f->max_ev ? f->max_ev : cnt[2 * c];
        const uint32_t kw = cnt[2 * c + 1] > f->max_fw ? f->max_fw : cnt[2 * c + 1];
        if (ke) {
            FX_TRY(hipMemcpy(events + pe, f->d_ev + (size_t)c * f->max_ev, (size_t)ke * sizeof(mfm_flex_event),
                             hipMemcpyDeviceToHost));
        }
        if (kw) {
            FX_TRY(hipMemcpy(frames + pw, f->d_fw + (size_t)c * f->max_fw, (size_t)kw * sizeof(mfm_flex_frame_words),
                             hipMemcpyDeviceToHost));
        }
        for (uint32_t i = 0; i < ke; i++) {
            mfm_flex_event *e = &events[pe + i];
            if (e->type == MFM_FLEX_EV_FRAME) {
                if (e->frame_index < kw) {
                    e->frame_index += (uint32_t)pw;
                } else {
                    overflow = true; /* the frame's words found no room on the device */
                    e->frame_index = 0xffffffffu;
                }
            }
        }
        pe += ke;
        pw += kw;
    }
    if (overflow) {
        snprintf(g_fx_error, sizeof(g_fx_error), "a channel produced more than max_events=%u events in one call", f->max_ev);
        mfm_internal_set_error(g_fx_error);
        return MFM_E_STATE;
    }
    return MFM_OK;
}

} /* extern "C" */
