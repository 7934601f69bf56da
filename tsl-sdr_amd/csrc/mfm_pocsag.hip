/*
 * mfm_pocsag.hip - the pager stage behind the PCM resampler, batched over all channels on the GPU:
 * POCSAG slicer, three-rate sync search, batch collection and BCH(31,21) correction
 * (pager/pager_pocsag.c:81-117,434-543 and pager/bch_code.c:307-398).  See include/multifm_hip.h for the
 * boundary and the event format.
 *
 * The reference walks one sample at a time through a state machine.  What is data parallel in it, and how it
 * is laid out here (per channel, all in HBM, 1 bit per sample):
 *
 *   bits    sign of every PCM sample (the slicer, pager_pocsag.c:91).
 *   m[d]    "the eye detector of rate d would see a sync word at this sample": popcount(W ^ SYNC) <= 4 with
 *           W bit j = bits[n - j * samples_per_bit].  eye_detect[cur_word] of the reference is exactly that
 *           register, whichever slot it lives in.  Computed 32 samples at a time, bit-sliced: 32 shifted
 *           views of the bit stream (one funnel shift each) go through a carry-save adder tree.
 *   summ    one bit per 32 samples: some run of two or more matches touches this word (a detector needs more
 *           than eight in a row to fire).  Lets an idle channel be skipped 65536 samples per step.
 *
 * What stays sequential is the walk from event to event (sync found -> 512 strided bits -> 32 sync bits ->
 * ...): one workgroup of four waves per channel does it.  All waves carry the same state; each finds the first
 * "eye closes after more than spb/2 matches" with a segmented wave scan over the m words; in a transmission
 * every wave gathers its own share of the coming slots (544 strided bits each, nine ballots), corrects the
 * 16 words with a 10-bit syndrome -> flip-mask table in LDS, and the sync verdicts are exchanged through LDS.
 *
 * After a reset the reference's registers are zero-filled, so for 31 bit periods m differs from the
 * free-running bitmap; the walker recomputes those words itself with the pre-reset bits masked off (same
 * bit-sliced routine, "EXACT" mode).  A register can never match on two consecutive visits (the sync word is
 * at Hamming distance >= 14 from every shift of itself, a match allows 4), so runs are at most one bit period
 * long - the code does not rely on it.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/multifm_hip.h"
#include "mfm_bch.h"

extern "C" __attribute__((visibility("hidden"))) void mfm_internal_set_error(const char *msg);

namespace {

constexpr uint32_t PG_HIST = 65536;  /* samples of history kept in front of the newest block (>= 544 * 75 + 31 * 75) */
constexpr uint32_t PG_GROUP = 2048;  /* alignment unit: 64 words = one wave of the match kernel = 2 summary words */
constexpr uint32_t PG_SYNC = 0x7cd215d8u; /* pager_pocsag_priv.h:40 */
constexpr uint32_t PG_WALK_WAVES = 4; /* waves per channel in the walker */
constexpr uint32_t PG_SPEC = 4;       /* batch + sync slots each of them gathers per step */
constexpr uint32_t PG_SLOW_SPAN = 31 * 75; /* samples after a reset during which some register is zero-filled */

enum : uint32_t { PG_SEARCH = 0, PG_BATCH = 2, PG_SYNCWORD = 3 };

/* ---- BCH(31,21): tables ------------------------------------------------------------------------------- */

using BchTables = MfmBchTables;

BchTables build_bch_tables()
{
    BchTables t;
    memset(&t, 0, sizeof(t));
    /* GF(2^5) on x^5 + x^2 + 1 (pager_pocsag.c:150) */
    int ex[31], lg[32];
    int v = 1;
    for (int i = 0; i < 31; i++) {
        ex[i] = v;
        lg[v] = i;
        v <<= 1;
        if (v & 32) {
            v ^= 0x25;
        }
    }
    lg[0] = -1;
    /* word bit (30 - j) is the coefficient of x^j (bch_code.c:325-326): S1 += a^j, S3 += a^(3j) */
    for (int byte = 0; byte < 4; byte++) {
        for (int val = 0; val < 256; val++) {
            uint32_t s = 0;
            for (int b = 0; b < 8; b++) {
                const int bit = 8 * byte + b;
                if (((val >> b) & 1) && bit <= 30) {
                    const int j = 30 - bit;
                    s ^= (uint32_t)ex[j] | ((uint32_t)ex[(3 * j) % 31] << 5);
                }
            }
            t.syn[byte][val] = (uint16_t)s;
        }
    }
    /* what bch_code.c:341-394 does with each syndrome pair */
    for (uint32_t idx = 0; idx < 1024; idx++) {
        const int S1 = idx & 31, S3 = idx >> 5;
        uint32_t f = 0;
        if (S1 != 0) {
            const int l1 = lg[S1];
            const int cube = ex[(3 * l1) % 31];
            if (S3 == cube) {
                f = 1u << (30 - l1);
            } else {
                const int la = lg[cube ^ S3];
                int e1 = ((2 * l1) % 31 - la + 31) % 31; /* log(S2) = 2 log(S1): S2 = S1^2 for a binary word */
                int e2 = (l1 - la + 31) % 31;
                int loc[2], count = 0;
                for (int i = 1; i <= 31; i++) {
                    e1 = (e1 + 1) % 31;
                    e2 = (e2 + 2) % 31;
                    if ((1 ^ ex[e1] ^ ex[e2]) == 0) {
                        if (count < 2) {
                            loc[count] = i % 31;
                        }
                        count++;
                    }
                }
                if (count == 2) {
                    f = (1u << (30 - loc[0])) ^ (1u << (30 - loc[1]));
                } else {
                    f = 0x80000000u;
                }
            }
        }
        /* S1 == 0: nothing is flipped and 0 is returned even when S3 != 0 (bch_code.c:342,391) */
        t.flips[idx] = f;
    }
    return t;
}

const BchTables &bch_tables()
{
    static const BchTables t = build_bch_tables();
    return t;
}

template <class T>
__host__ __device__ __forceinline__ uint32_t pg_bch_fix(const T *t, uint32_t w, uint32_t *rc)
{
    return mfm_bch_fix(t, w, rc);
}

__global__ __launch_bounds__(256) void pg_bch_kernel(uint32_t *words, uint8_t *rc, size_t n, const BchTables *tab)
{
    __shared__ BchTables T;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tab);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&T);
        for (uint32_t i = threadIdx.x; i < sizeof(BchTables) / 4; i += blockDim.x) {
            dst[i] = src[i];
        }
    }
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t r;
        words[i] = pg_bch_fix(&T, words[i], &r);
        rc[i] = (uint8_t)r;
    }
}

/* ---- bit-sliced sync-word correlator -------------------------------------------------------------------- */

#define PG_FA(a, b, c, s, cy)                                                                                \
    do {                                                                                                     \
        const uint32_t x_ = (a) ^ (b);                                                                       \
        const uint32_t s_ = x_ ^ (c);                                                                        \
        const uint32_t c_ = (x_ & (c)) | ((a) & (b));                                                        \
        (s) = s_;                                                                                            \
        (cy) = c_;                                                                                           \
    } while (0)
#define PG_HA(a, b, s, cy)                                                                                   \
    do {                                                                                                     \
        const uint32_t s_ = (a) ^ (b);                                                                       \
        const uint32_t c_ = (a) & (b);                                                                       \
        (s) = s_;                                                                                            \
        (cy) = c_;                                                                                           \
    } while (0)

/* y[0..31]: 32 one-bit-per-sample mismatch vectors; returns, per sample, "at most 4 of them are set" */
__device__ __forceinline__ uint32_t pg_count_le4(const uint32_t *y)
{
    uint32_t s[12], c[16];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        PG_FA(y[3 * i], y[3 * i + 1], y[3 * i + 2], s[i], c[i]);
    }
    s[10] = y[30];
    s[11] = y[31];
    uint32_t t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        PG_FA(s[3 * i], s[3 * i + 1], s[3 * i + 2], t[i], c[10 + i]);
    }
    uint32_t u0, bit0;
    PG_FA(t[0], t[1], t[2], u0, c[14]);
    PG_HA(u0, t[3], bit0, c[15]);
    /* weight 2: 16 inputs */
    uint32_t v[6], d[8];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        PG_FA(c[3 * i], c[3 * i + 1], c[3 * i + 2], v[i], d[i]);
    }
    v[5] = c[15];
    uint32_t w0, w1, bit1;
    PG_FA(v[0], v[1], v[2], w0, d[5]);
    PG_FA(v[3], v[4], v[5], w1, d[6]);
    PG_HA(w0, w1, bit1, d[7]);
    /* weight 4: 8 inputs */
    uint32_t x0, x1, e0, e1, e2, e3, z0, bit2;
    PG_FA(d[0], d[1], d[2], x0, e0);
    PG_FA(d[3], d[4], d[5], x1, e1);
    PG_FA(x0, x1, d[6], z0, e2);
    PG_HA(z0, d[7], bit2, e3);
    const uint32_t ge8 = e0 | e1 | e2 | e3;
    return ~ge8 & ~(bit2 & (bit1 | bit0));
}

/*
 * m word for samples [32 * wi, 32 * wi + 32) of rate SPB.  ld(q) returns bit-stream word q (0 for q < 0).
 * EXACT: bits of samples before r_rel (window-relative sample index of the reset) read as zero, which is what
 * the reference's zero-filled registers hold (pager_pocsag.c:119-126).
 */
template <int SPB, bool EXACT, class LD>
__device__ __forceinline__ uint32_t pg_match32(LD ld, int32_t wi, int32_t r_rel)
{
    uint32_t y[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int32_t P = 32 * wi - j * SPB;
        const int32_t q = P >> 5;
        uint32_t x;
        if ((j * SPB) % 32 == 0) {
            x = ld(q);
        } else {
            x = __builtin_amdgcn_alignbit(ld(q + 1), ld(q), (uint32_t)P & 31u);
        }
        if (EXACT) {
            const int32_t th = r_rel + j * SPB - 32 * wi; /* first sample of the word that sees a real bit */
            const uint32_t valid = th <= 0 ? 0xffffffffu : (th >= 32 ? 0u : (0xffffffffu << th));
            x &= valid;
        }
        y[j] = ((PG_SYNC >> j) & 1u) ? ~x : x;
    }
    return pg_count_le4(y);
}

/* ---- device layout -------------------------------------------------------------------------------------- */

struct PgBuf {
    uint32_t *base; /* planes 0..3 ([plane][channel][BW]) then the summary ([channel][SW]) */
    uint32_t C, BW, SW;
    __host__ __device__ uint32_t *plane(uint32_t p, uint32_t c) const { return base + ((size_t)p * C + c) * BW; }
    __host__ __device__ uint32_t *summ(uint32_t c) const { return base + (size_t)4 * C * BW + (size_t)c * SW; }
};

struct PgChanState {
    uint64_t pos; /* SEARCH: next sample to look at */
    uint64_t r;   /* SEARCH: first sample after the last reset */
    uint64_t b0;  /* BATCH / SYNCWORD: sample that carries the first bit still to be collected */
    uint32_t mode, S, baud;
    uint32_t nr[3]; /* SEARCH: nr_eye_matches of the three detectors at pos */
};

/* carry the tail of the window over to the other buffer (the walker may lag one batch behind) */
__global__ __launch_bounds__(256) void pg_slide_kernel(PgBuf dst, PgBuf src, uint32_t shift_w, uint32_t keep_w)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (k < keep_w) {
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            dst.plane(p, c)[k] = src.plane(p, c)[k + shift_w];
        }
    }
    if (k < (keep_w + 31) / 32) {
        dst.summ(c)[k] = src.summ(c)[k + shift_w / 32];
    }
}

/*
 * The slicer: bit = sample < 0 (pager_pocsag.c:91,476,507).  HBM-bound: 2 bytes in per sample, 1/8 byte out.
 * One lane takes 8 consecutive samples with one 16-byte load (the address is only 2-byte aligned in general:
 * the window is aligned, the caller's rows are not), squeezes them to a byte, and four neighbouring lanes merge
 * their bytes into a word; a wave covers 512 samples per step and PG_SLICE_U steps are in flight together.
 */
constexpr uint32_t PG_SLICE_U = 4;

struct __attribute__((packed, aligned(2))) PgPcm8 {
    uint32_t d[4];
};

__global__ __launch_bounds__(256) void pg_slice_kernel(PgBuf buf, const int16_t *x, size_t stride, uint32_t n, uint32_t off0,
                                                      uint32_t nsteps)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t s0 = (blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * PG_SLICE_U;
    const uint32_t c = blockIdx.y;
    if (s0 >= nsteps) {
        return;
    }
    const int16_t *xc = x + (size_t)c * stride;
    const uint32_t base0 = (off0 & ~511u) + 512u * s0 + 8u * lane; /* window-relative index of my first sample */
    uint32_t d[PG_SLICE_U][4];
    const int64_t wave_first = (int64_t)((off0 & ~511u) + 512u * s0) - (int64_t)off0; /* input index of the wave's first sample */
    if (wave_first >= 0 && wave_first + 512 * (int64_t)PG_SLICE_U <= (int64_t)n) {
        /* interior (wave-uniform): all loads issued back to back */
#pragma unroll
        for (uint32_t k = 0; k < PG_SLICE_U; k++) {
            const PgPcm8 v = *reinterpret_cast<const PgPcm8 *>(xc + ((int64_t)base0 + 512 * k - (int64_t)off0));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                d[k][q] = v.d[q];
            }
        }
    } else { /* first / last samples of the call: element by element */
#pragma unroll
        for (uint32_t k = 0; k < PG_SLICE_U; k++) {
            const int64_t i = (int64_t)base0 + 512 * k - (int64_t)off0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t i0 = i + 2 * q, i1 = i0 + 1;
                const uint32_t lo = (i0 >= 0 && i0 < (int64_t)n) ? (uint16_t)xc[i0] : 0u;
                const uint32_t hi = (i1 >= 0 && i1 < (int64_t)n) ? (uint16_t)xc[i1] : 0u;
                d[k][q] = lo | (hi << 16);
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < PG_SLICE_U; k++) {
        uint32_t b = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            b |= (((d[k][q] >> 15) & 1u) | ((d[k][q] >> 30) & 2u)) << (2 * q);
        }
        b |= (uint32_t)__shfl_down((int)b, 1) << 8;  /* valid in even lanes */
        b |= (uint32_t)__shfl_down((int)b, 2) << 16; /* valid in lanes 0, 4, 8, ... */
        if (s0 + k < nsteps && (lane & 3u) == 0) {
            const uint32_t first = base0 + 512u * k; /* first sample of this word */
            uint32_t *dst = buf.plane(0, c) + (first >> 5);
            uint32_t word = b;
            if (first < off0) { /* the word straddles the old end: keep the bits that are already there */
                const uint32_t keep = (off0 - first >= 32) ? 0xffffffffu : ((1u << (off0 - first)) - 1u);
                word = (*dst & keep) | (word & ~keep);
            }
            *dst = word;
        }
    }
}

/* m[0..2] and the summary for words [w_first, w_first + 256 * gridDim.x) */
__global__ __launch_bounds__(256) void pg_match_kernel(PgBuf buf, uint32_t w_first)
{
    constexpr int BACK = 76; /* 31 * 75 bits = 72.7 words of history, plus the funnel-shift neighbour */
    __shared__ uint32_t tile[BACK + 256 + 4];
    const uint32_t c = blockIdx.y;
    const int32_t w0 = (int32_t)(w_first + blockIdx.x * 256u);
    const uint32_t *bits = buf.plane(0, c);
    for (int32_t k = (int32_t)threadIdx.x; k < BACK + 256 + 2; k += 256) {
        const int32_t q = w0 - BACK + k;
        tile[k] = q >= 0 ? bits[q] : 0u;
    }
    __syncthreads();
    const int32_t wi = w0 + (int32_t)threadIdx.x;
    auto ld = [&](int32_t q) { return tile[q - (w0 - BACK)]; };
    const uint32_t m0 = pg_match32<75, false>(ld, wi, 0);
    const uint32_t m1 = pg_match32<32, false>(ld, wi, 0);
    const uint32_t m2 = pg_match32<16, false>(ld, wi, 0);
    buf.plane(1, c)[wi] = m0;
    buf.plane(2, c)[wi] = m1;
    buf.plane(3, c)[wi] = m2;
    /* summary bit: "a run of two or more matches touches this word" - two adjacent matches inside it, or its last
     * sample and the next word's first both match (across a wave boundary: assume they do).  A detector can only
     * fire after more than 8 consecutive matches, so words without the bit can be skipped by the walker. */
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t flag = 0;
    {
        const uint32_t ms[3] = { m0, m1, m2 };
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const uint32_t nxt = (uint32_t)__shfl_down((int)ms[d], 1);
            const uint32_t next_first = lane == 63 ? 1u : (nxt & 1u);
            flag |= (ms[d] & (ms[d] >> 1)) | ((ms[d] >> 31) & next_first);
        }
    }
    const unsigned long long any = __ballot(flag != 0u);
    if (lane == 0 || lane == 32) {
        buf.summ(c)[(uint32_t)wi >> 5] = (uint32_t)(any >> lane);
    }
}

struct PgWalk {
    PgBuf buf;
    uint64_t ws;  /* absolute sample index of window word 0 */
    uint64_t end; /* absolute index one past the newest sample */
    PgChanState *st;
    mfm_pocsag_event *ev;
    uint32_t *ev_count;
    uint32_t max_ev;
    const BchTables *bch;
};

__device__ __forceinline__ uint32_t pg_wave_min(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off);
        v = o < v ? o : v;
    }
    return v;
}

/* one wave per channel: pager_pocsag_on_pcm (pager_pocsag.c:434-543) from event to event */
__global__ __launch_bounds__(64 * PG_WALK_WAVES) void pg_walk_kernel(const PgWalk L)
{
    __shared__ BchTables T;
    __shared__ uint32_t sh_sync[PG_WALK_WAVES * PG_SPEC];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(L.bch);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&T);
        for (uint32_t i = threadIdx.x; i < sizeof(BchTables) / 4; i += blockDim.x) {
            dst[i] = src[i];
        }
    }
    __syncthreads();
    const uint32_t c = blockIdx.x;
    PgChanState st = L.st[c];
    const uint32_t *bits = L.buf.plane(0, c);
    const uint32_t *summ = L.buf.summ(c);
    mfm_pocsag_event *ev = L.ev + (size_t)c * L.max_ev;
    uint32_t nev = 0;

    auto getbit = [&](uint64_t n) {
        const uint32_t o = (uint32_t)(n - L.ws);
        return (bits[o >> 5] >> (o & 31u)) & 1u;
    };
    auto getbit_rel = [&](uint32_t o) { return (bits[o >> 5] >> (o & 31u)) & 1u; }; /* window-relative index */
    /* every wave keeps the same state and takes the same decisions; wave 0 reports them, except in the slot-parallel
     * batch step where each wave reports its own slots */
    auto emit_at = [&](uint32_t idx, uint32_t type, uint32_t aux, uint64_t sample, uint32_t nr_ok, uint32_t fail_mask,
                       uint32_t raw, uint32_t fixed) {
        if (idx < L.max_ev) {
            mfm_pocsag_event *e = &ev[idx];
            if (lane == 0) {
                e->type = type;
                e->baud = st.baud;
                e->channel = c;
                e->aux = aux;
                e->sample = sample;
                e->nr_ok = nr_ok;
                e->fail_mask = fail_mask;
            }
            if (lane < 16) {
                e->raw[lane] = raw;
                e->corrected[lane] = fixed;
            }
        }
    };
    auto emit = [&](uint32_t type, uint32_t aux, uint64_t sample, uint32_t nr_ok, uint32_t fail_mask, uint32_t raw,
                    uint32_t fixed) {
        if (nev < L.max_ev && wv == 0) {
            mfm_pocsag_event *e = &ev[nev];
            if (lane == 0) {
                e->type = type;
                e->baud = st.baud;
                e->channel = c;
                e->aux = aux;
                e->sample = sample;
                e->nr_ok = nr_ok;
                e->fail_mask = fail_mask;
            }
            if (lane < 16) {
                e->raw[lane] = raw;
                e->corrected[lane] = fixed;
            }
        }
        nev++;
    };

    for (;;) {
        if (st.mode == PG_SEARCH) {
            if (st.pos >= L.end) {
                break;
            }
            /* ---- one aligned chunk of 64 words x 32 samples, all three detectors ---- */
            const uint64_t cb = st.pos & ~(uint64_t)(PG_GROUP - 1);
            const uint32_t wrel = (uint32_t)((cb - L.ws) >> 5) + lane;
            const int64_t lane_base = (int64_t)cb + 32 * (int64_t)lane;
            int64_t lo64 = (int64_t)st.pos - lane_base, hi64 = (int64_t)L.end - lane_base;
            const int lo = lo64 < 0 ? 0 : (lo64 > 32 ? 32 : (int)lo64);
            const int hi = hi64 < 0 ? 0 : (hi64 > 32 ? 32 : (int)hi64);
            const bool active = hi > lo;
            const uint32_t rm = active ? (((hi == 32) ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
            const bool exact = cb < st.r + PG_SLOW_SPAN;
            const int32_t r_rel = (int32_t)((int64_t)st.r - (int64_t)L.ws);
            auto ldg = [&](int32_t q) { return q >= 0 ? bits[q] : 0u; };
            uint32_t bestkey = 0xffffffffu, bestrun = 0;
            uint32_t endrun[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const uint32_t spb = d == 0 ? 75u : (d == 1 ? 32u : 16u);
                uint32_t m;
                if (exact) {
                    m = d == 0 ? pg_match32<75, true>(ldg, (int32_t)wrel, r_rel)
                               : (d == 1 ? pg_match32<32, true>(ldg, (int32_t)wrel, r_rel)
                                         : pg_match32<16, true>(ldg, (int32_t)wrel, r_rel));
                } else {
                    m = L.buf.plane(1 + d, c)[wrel];
                }
                const uint32_t z = ~m & rm; /* non-matching samples of my word that are in range */
                /* run of matches reaching the end of my range, and whether my whole range matches */
                uint32_t full = (z == 0u) ? 1u : 0u;
                uint32_t val = active ? ((z == 0u) ? (uint32_t)(hi - lo) : (uint32_t)(hi - 1 - (31 - __clz((int)z)))) : 0u;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t pv = (uint32_t)__shfl_up((int)val, off);
                    const uint32_t pf = (uint32_t)__shfl_up((int)full, off);
                    if ((int)lane >= off) {
                        val = full ? val + pv : val;
                        full = full & pf;
                    }
                }
                uint32_t c_in = (uint32_t)__shfl_up((int)val, 1);
                uint32_t f_in = (uint32_t)__shfl_up((int)full, 1);
                if (lane == 0) {
                    c_in = 0;
                    f_in = 1;
                }
                c_in += f_in ? st.nr[d] : 0u; /* nr_eye_matches when the detector reaches my range */
                const uint32_t v63 = (uint32_t)__shfl((int)val, 63), f63 = (uint32_t)__shfl((int)full, 63);
                endrun[d] = v63 + (f63 ? st.nr[d] : 0u);
                if (active) {
                    /* a detector fires on a non-matching sample that ends a run of more than spb/2 matches
                     * (pager_pocsag.c:96-108); only samples right behind a match can qualify */
                    uint32_t cand = z & (((m & rm) << 1) | ((c_in > 0u) ? (1u << lo) : 0u));
                    while (cand) {
                        const int k = __ffs((int)cand) - 1;
                        cand &= cand - 1u;
                        const uint32_t zb = z & ((1u << k) - 1u);
                        const uint32_t run = zb ? (uint32_t)(k - 1 - (31 - __clz((int)zb))) : (uint32_t)(k - lo) + c_in;
                        if (run > spb / 2u) {
                            const uint32_t key = ((lane * 32u + (uint32_t)k) << 2) | (uint32_t)(2 - d);
                            if (key < bestkey) {
                                bestkey = key;
                                bestrun = run;
                            }
                            break;
                        }
                    }
                }
            }
            const uint32_t minkey = pg_wave_min(bestkey);
            if (minkey != 0xffffffffu) {
                /* earliest sample wins; on the same sample the detector run last (2400 after 1200 after 512)
                 * leaves its settings behind (pager_pocsag.c:452-457) */
                const unsigned long long who = __ballot(bestkey == minkey);
                const uint32_t run = (uint32_t)__shfl((int)bestrun, __ffsll((long long)who) - 1);
                const int d = 2 - (int)(minkey & 3u);
                const uint64_t f = cb + (minkey >> 2);
                st.S = d == 0 ? 75u : (d == 1 ? 32u : 16u);
                st.baud = d == 0 ? 512u : (d == 1 ? 1200u : 2400u);
                emit(MFM_POCSAG_EV_SYNC_FOUND, run, f, 0, 0, 0, 0);
                /* batch.cur_sample_skip = matches / 2 (uint16), a bit is taken when ++skip == sample_skip */
                const uint32_t c0 = (run >> 1) & 0xffffu;
                st.b0 = f + (c0 < st.S ? st.S - c0 : 65536u + st.S - c0);
                st.mode = PG_BATCH;
            } else {
                st.nr[0] = endrun[0];
                st.nr[1] = endrun[1];
                st.nr[2] = endrun[2];
                const uint64_t nxt = cb + PG_GROUP;
                st.pos = nxt < L.end ? nxt : L.end;
                /* nothing pending and past the zero-filled span: jump to the next word with any match in it */
                while ((st.nr[0] | st.nr[1] | st.nr[2]) == 0u && st.pos < L.end && st.pos >= st.r + PG_SLOW_SPAN) {
                    const uint32_t sw0 = (uint32_t)((st.pos - L.ws) >> 10);
                    const uint32_t sidx = sw0 + lane;
                    const uint32_t sv = sidx < L.buf.SW ? summ[sidx] : 0xffffffffu;
                    const unsigned long long nz = __ballot(sv != 0u);
                    if (nz == 0ull) {
                        const uint64_t far = st.pos + 65536ull;
                        st.pos = far < L.end ? far : L.end;
                        continue;
                    }
                    const int l1 = __ffsll((long long)nz) - 1;
                    const uint32_t svw = (uint32_t)__shfl((int)sv, l1);
                    const uint64_t hit = L.ws + (((uint64_t)(sw0 + (uint32_t)l1) * 32u + (uint32_t)(__ffs((int)svw) - 1)) << 5);
                    st.pos = hit < L.end ? hit : L.end;
                    break;
                }
            }
        } else if (st.mode == PG_BATCH) {
            /* 512 bits, one every S samples, LSB first into 16 words (pager_pocsag.c:472-481), then 32 bits in the
             * sync slot, first one ending up in bit 31 (:506-513) */
            if (st.b0 + 511ull * st.S >= L.end) {
                break;
            }
            const uint64_t avail_bits = (L.end - 1 - st.b0) / st.S + 1;
            const uint32_t nslots = avail_bits / 544u < PG_WALK_WAVES * PG_SPEC ? (uint32_t)(avail_bits / 544u) : PG_WALK_WAVES * PG_SPEC;
            if (nslots == 0) {
                /* the batch is complete but its sync slot is not: report the batch now, as the reference does */
                uint32_t myraw = 0;
#pragma unroll
                for (uint32_t r = 0; r < 8; r++) {
                    const uint32_t bit = getbit(st.b0 + (uint64_t)(64u * r + lane) * st.S);
                    const unsigned long long b = __ballot((int)bit);
                    if (lane == 2 * r) {
                        myraw = (uint32_t)b;
                    }
                    if (lane == 2 * r + 1) {
                        myraw = (uint32_t)(b >> 32);
                    }
                }
                uint32_t rc = 0;
                const uint32_t fixed = pg_bch_fix(&T, myraw & 0x7fffffffu, &rc); /* pager_pocsag.c:332-334 */
                const uint32_t fail = (uint32_t)__ballot(lane < 16 && rc) & 0xffffu;
                const uint32_t nr_ok = fail ? (uint32_t)(__ffs((int)fail) - 1) : 16u;
                emit(MFM_POCSAG_EV_BATCH, 0, st.b0 + 511ull * st.S, nr_ok, fail, myraw, fixed);
                st.b0 += 512ull * st.S;
                st.mode = PG_SYNCWORD;
            } else {
                /* Whole slots (batch + sync word) are here.  While sync holds, the next slot starts exactly 544 bit
                 * periods later, so up to PG_SPEC slots per wave are gathered speculatively - every load in flight
                 * before the first ballot - by all waves at once (wave w takes slots w * PG_SPEC ...), the sync
                 * verdicts are exchanged through LDS, and everything behind the first lost sync is dropped. */
                const uint32_t rel0 = (uint32_t)(st.b0 - L.ws);
                uint32_t got[PG_SPEC][9];
#pragma unroll
                for (uint32_t j = 0; j < PG_SPEC; j++) {
                    const uint32_t g = wv * PG_SPEC + j;
                    const uint32_t sb = rel0 + (g < nslots ? 544u * g * st.S : 0u);
#pragma unroll
                    for (uint32_t r = 0; r < 8; r++) {
                        got[j][r] = getbit_rel(sb + (64u * r + lane) * st.S);
                    }
                    got[j][8] = getbit_rel(sb + (512u + (lane & 31u)) * st.S);
                }
                uint32_t raws[PG_SPEC], syncs[PG_SPEC];
#pragma unroll
                for (uint32_t j = 0; j < PG_SPEC; j++) {
                    raws[j] = 0;
#pragma unroll
                    for (uint32_t r = 0; r < 8; r++) {
                        const unsigned long long b = __ballot((int)got[j][r]);
                        /* word 2r -> lane 2r, word 2r+1 -> lane 2r+1.  There is no clang builtin for v_writelane here,
                         * and the compiler does not pad hazards around inline asm: the ballot is a v_cmp writing VCC,
                         * and v_writelane reading it straight away sees the previous value (measured), hence s_nop. */
                        asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"
                                     : "+v"(raws[j])
                                     : "s"((uint32_t)b), "n"(2 * r), "s"((uint32_t)(b >> 32)), "n"(2 * r + 1));
                    }
                    syncs[j] = __brev((uint32_t)__ballot((int)got[j][8]));
                    if (lane == 0) {
                        sh_sync[wv * PG_SPEC + j] = syncs[j];
                    }
                }
                __syncthreads();
                /* first slot whose sync word is bad (nslots = none) */
                uint32_t bad = nslots;
                {
                    const uint32_t sw = lane < PG_WALK_WAVES * PG_SPEC ? sh_sync[lane] : PG_SYNC;
                    const unsigned long long lost = __ballot(lane < nslots && __popc(sw ^ PG_SYNC) > 4);
                    if (lost) {
                        bad = (uint32_t)(__ffsll((long long)lost) - 1);
                    }
                }
                __syncthreads(); /* sh_sync is reused by the next step */
#pragma unroll
                for (uint32_t j = 0; j < PG_SPEC; j++) {
                    const uint32_t g = wv * PG_SPEC + j;
                    if (g < nslots && g <= bad) {
                        uint32_t rc = 0;
                        const uint32_t fixed = pg_bch_fix(&T, raws[j] & 0x7fffffffu, &rc); /* pager_pocsag.c:332-334 */
                        const uint32_t fail = (uint32_t)__ballot(lane < 16 && rc) & 0xffffu;
                        const uint32_t nr_ok = fail ? (uint32_t)(__ffs((int)fail) - 1) : 16u;
                        const uint64_t sb = st.b0 + (uint64_t)(544u * g) * st.S;
                        emit_at(nev + 2 * g, MFM_POCSAG_EV_BATCH, 0, sb + 511ull * st.S, nr_ok, fail, raws[j], fixed);
                        emit_at(nev + 2 * g + 1, g == bad ? MFM_POCSAG_EV_SYNC_LOST : MFM_POCSAG_EV_SYNC_KEPT, syncs[j],
                                sb + 543ull * st.S, 0, 0, 0, 0);
                    }
                }
                if (bad == nslots) {
                    nev += 2 * nslots;
                    st.b0 += (uint64_t)(544u * nslots) * st.S;
                } else {
                    nev += 2 * (bad + 1);
                    st.mode = PG_SEARCH; /* pager_pocsag.c:517-523: all three detectors start from zero */
                    st.pos = st.r = st.b0 + (uint64_t)(544u * bad + 543u) * st.S + 1;
                    st.nr[0] = st.nr[1] = st.nr[2] = 0;
                }
            }
        } else {
            /* a batch was reported before its sync slot had arrived; now the 32 bits are here */
            if (st.b0 + 31ull * st.S >= L.end) {
                break;
            }
            const uint32_t bit = lane < 32 ? getbit(st.b0 + (uint64_t)lane * st.S) : 0u;
            const uint32_t sw = __brev((uint32_t)__ballot((int)bit));
            const uint64_t at = st.b0 + 31ull * st.S;
            if (__popc(sw ^ PG_SYNC) <= 4) {
                emit(MFM_POCSAG_EV_SYNC_KEPT, sw, at, 0, 0, 0, 0);
                st.b0 += 32ull * st.S;
                st.mode = PG_BATCH;
            } else {
                emit(MFM_POCSAG_EV_SYNC_LOST, sw, at, 0, 0, 0, 0);
                st.mode = PG_SEARCH; /* pager_pocsag.c:517-523: all three detectors start from zero */
                st.pos = st.r = at + 1;
                st.nr[0] = st.nr[1] = st.nr[2] = 0;
            }
        }
    }
    if (threadIdx.x == 0) {
        L.st[c] = st;
        L.ev_count[c] = nev;
    }
}

thread_local char g_pg_error[256] = "";

std::mutex g_bch_mu;
BchTables *g_bch_dev[64] = {};

} /* namespace */

#define PG_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t err_ = (expr);                                                                            \
        if (err_ != hipSuccess) {                                                                            \
            snprintf(g_pg_error, sizeof(g_pg_error), "%s failed: %s", #expr, hipGetErrorString(err_));       \
            mfm_internal_set_error(g_pg_error);                                                              \
            return MFM_E_DEVICE;                                                                             \
        }                                                                                                    \
    } while (0)

static int pg_device_tables(int device, BchTables **out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || device >= 64) {
        return MFM_E_DEVICE; /* no CPU path */
    }
    std::lock_guard<std::mutex> lk(g_bch_mu);
    PG_TRY(hipSetDevice(device));
    if (!g_bch_dev[device]) {
        BchTables *d = nullptr;
        PG_TRY(hipMalloc(&d, sizeof(BchTables)));
        PG_TRY(hipMemcpy(d, &bch_tables(), sizeof(BchTables), hipMemcpyHostToDevice));
        g_bch_dev[device] = d;
    }
    *out = g_bch_dev[device];
    return MFM_OK;
}

struct mfm_pocsag {
    mfm_pocsag_config cfg{};
    uint32_t cap_samples = 0, max_ev = 0;
    PgBuf buf[2]{};
    int cur = 0;
    uint64_t ws = 0, total = 0;
    PgChanState *d_st = nullptr;
    mfm_pocsag_event *d_ev = nullptr;
    uint32_t *d_evcount = nullptr;
    BchTables *d_bch = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_call = false;
};

extern "C" {

int mfm_pocsag_create(struct mfm_pocsag **pp, const struct mfm_pocsag_config *cfg)
{
    if (!pp || !cfg) {
        return MFM_E_INVAL;
    }
    *pp = nullptr;
    if (cfg->abi_version != MFM_ABI_VERSION || 0 == cfg->nr_channels || 0 == cfg->max_in_samples ||
        cfg->max_in_samples > (1u << 28)) {
        return MFM_E_INVAL;
    }
    BchTables *d_bch = nullptr;
    int rc = pg_device_tables(cfg->device, &d_bch);
    if (rc != MFM_OK) {
        return rc;
    }
    mfm_pocsag *p = new (std::nothrow) mfm_pocsag();
    if (!p) {
        return MFM_E_NOMEM;
    }
    p->cfg = *cfg;
    p->d_bch = d_bch;
    const uint32_t in_round = (cfg->max_in_samples + PG_GROUP - 1) / PG_GROUP * PG_GROUP;
    p->cap_samples = PG_HIST + PG_GROUP + 2 * in_round;
    p->max_ev = cfg->max_events ? cfg->max_events : cfg->max_in_samples / 2048 + 16;
    const uint32_t BW = p->cap_samples / 32 + 256 + 8; /* the match kernel rounds its range up to 256 words */
    const uint32_t SW = BW / 32 + 4;
    const uint32_t C = cfg->nr_channels;
    *pp = p;
    for (int i = 0; i < 2; i++) {
        const size_t bytes = ((size_t)4 * C * BW + (size_t)C * SW) * 4;
        p->buf[i] = PgBuf{ nullptr, C, BW, SW };
        PG_TRY(hipMalloc(&p->buf[i].base, bytes));
        PG_TRY(hipMemset(p->buf[i].base, 0, bytes));
    }
    PG_TRY(hipMalloc(&p->d_st, (size_t)C * sizeof(PgChanState)));
    PG_TRY(hipMemset(p->d_st, 0, (size_t)C * sizeof(PgChanState))); /* SEARCH at sample 0, reset at 0 */
    PG_TRY(hipMalloc(&p->d_ev, (size_t)C * p->max_ev * sizeof(mfm_pocsag_event)));
    PG_TRY(hipMalloc(&p->d_evcount, (size_t)C * 4));
    PG_TRY(hipMemset(p->d_evcount, 0, (size_t)C * 4));
    PG_TRY(hipDeviceSynchronize());
    return MFM_OK;
}

void mfm_pocsag_destroy(struct mfm_pocsag **pp)
{
    if (!pp || !*pp) {
        return;
    }
    mfm_pocsag *p = *pp;
    (void)hipSetDevice(p->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(p->buf[0].base);
    (void)hipFree(p->buf[1].base);
    (void)hipFree(p->d_st);
    (void)hipFree(p->d_ev);
    (void)hipFree(p->d_evcount);
    delete p;
    *pp = nullptr;
}

int mfm_pocsag_process_device(struct mfm_pocsag *p, const int16_t *d_pcm, size_t in_stride, size_t nr_in, void *stream)
{
    if (!p || (!d_pcm && nr_in) || nr_in > p->cfg.max_in_samples) {
        return MFM_E_INVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t C = p->cfg.nr_channels, n = (uint32_t)nr_in;
    PG_TRY(hipSetDevice(p->cfg.device));
    if (p->have_call && p->last_stream != s) {
        PG_TRY(hipStreamSynchronize(p->last_stream)); /* state lives on the device; keep calls ordered */
    }
    uint32_t off0 = (uint32_t)(p->total - p->ws);
    if ((uint64_t)off0 + n > p->cap_samples) {
        const uint64_t new_ws = (p->total & ~(uint64_t)(PG_GROUP - 1)) - PG_HIST;
        const uint32_t shift_w = (uint32_t)((new_ws - p->ws) >> 5);
        const uint32_t used_w = ((off0 + 31) / 32 + 63) / 64 * 64;
        const uint32_t keep_w = used_w - shift_w;
        hipLaunchKernelGGL(pg_slide_kernel, dim3((keep_w + 255) / 256, C), dim3(256), 0, s, p->buf[p->cur ^ 1], p->buf[p->cur],
                           shift_w, keep_w);
        PG_TRY(hipGetLastError());
        p->cur ^= 1;
        p->ws = new_ws;
        off0 = (uint32_t)(p->total - p->ws);
    }
    const PgBuf buf = p->buf[p->cur];
    if (n) {
        const uint32_t nsteps = (off0 + n - (off0 & ~511u) + 511) / 512;
        hipLaunchKernelGGL(pg_slice_kernel, dim3((nsteps + 4 * PG_SLICE_U - 1) / (4 * PG_SLICE_U), C), dim3(256), 0, s, buf, d_pcm,
                           in_stride, n, off0, nsteps);
        PG_TRY(hipGetLastError());
        const uint32_t w_first = (off0 & ~(PG_GROUP - 1)) / 32;
        const uint32_t w_end = (off0 + n + 31) / 32;
        hipLaunchKernelGGL(pg_match_kernel, dim3((w_end - w_first + 255) / 256, C), dim3(256), 0, s, buf, w_first);
        PG_TRY(hipGetLastError());
    }
    PgWalk W{ buf, p->ws, p->total + n, p->d_st, p->d_ev, p->d_evcount, p->max_ev, p->d_bch };
    hipLaunchKernelGGL(pg_walk_kernel, dim3(C), dim3(64 * PG_WALK_WAVES), 0, s, W);
    PG_TRY(hipGetLastError());
    p->total += n;
    p->last_stream = s;
    p->have_call = true;
    return MFM_OK;
}

int mfm_pocsag_process_host(struct mfm_pocsag *p, const int16_t *pcm, size_t in_stride, size_t nr_in)
{
    if (!p || (!pcm && nr_in)) {
        return MFM_E_INVAL;
    }
    PG_TRY(hipSetDevice(p->cfg.device));
    const uint32_t C = p->cfg.nr_channels;
    int16_t *d_in = nullptr;
    PG_TRY(hipMalloc(&d_in, (size_t)C * (nr_in ? nr_in : 1) * 2));
    if (nr_in) {
        PG_TRY(hipMemcpy2D(d_in, nr_in * 2, pcm, in_stride * 2, nr_in * 2, C, hipMemcpyHostToDevice));
    }
    const int rc = mfm_pocsag_process_device(p, d_in, nr_in, nr_in, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipFree(d_in);
    return rc;
}

int mfm_pocsag_fetch_events(struct mfm_pocsag *p, struct mfm_pocsag_event *out, size_t max_events, size_t *nr_events)
{
    if (!p || !nr_events || (!out && max_events)) {
        return MFM_E_INVAL;
    }
    *nr_events = 0;
    if (!p->have_call) {
        return MFM_OK;
    }
    PG_TRY(hipSetDevice(p->cfg.device));
    PG_TRY(hipStreamSynchronize(p->last_stream));
    const uint32_t C = p->cfg.nr_channels;
    std::vector<uint32_t> cnt(C);
    PG_TRY(hipMemcpy(cnt.data(), p->d_evcount, (size_t)C * 4, hipMemcpyDeviceToHost));
    size_t total = 0;
    bool overflow = false;
    for (uint32_t c = 0; c < C; c++) {
        overflow |= cnt[c] > p->max_ev;
        total += cnt[c] > p->max_ev ? p->max_ev : cnt[c];
    }
    *nr_events = total;
    if (total > max_events) {
        return MFM_E_NOMEM;
    }
    size_t pos = 0;
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t k = cnt[c] > p->max_ev ? p->max_ev : cnt[c];
        if (k) {
            PG_TRY(hipMemcpy(out + pos, p->d_ev + (size_t)c * p->max_ev, (size_t)k * sizeof(mfm_pocsag_event),
                             hipMemcpyDeviceToHost));
            pos += k;
        }
    }
    if (overflow) {
        snprintf(g_pg_error, sizeof(g_pg_error), "a channel produced more than max_events=%u events in one call", p->max_ev);
        mfm_internal_set_error(g_pg_error);
        return MFM_E_STATE;
    }
    return MFM_OK;
}

int mfm_bch3121_decode_device(uint32_t *d_words, uint8_t *d_rc, size_t n, int device, void *stream)
{
    if ((!d_words || !d_rc) && n) {
        return MFM_E_INVAL;
    }
    BchTables *tab = nullptr;
    const int rc = pg_device_tables(device, &tab);
    if (rc != MFM_OK) {
        return rc;
    }
    if (0 == n) {
        return MFM_OK;
    }
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) {
        blocks = 8192;
    }
    hipLaunchKernelGGL(pg_bch_kernel, dim3((uint32_t)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_words, d_rc, n, tab);
    PG_TRY(hipGetLastError());
    return MFM_OK;
}

int mfm_bch3121_decode_host(uint32_t *words, uint8_t *rc, size_t n, int device)
{
    if ((!words || !rc) && n) {
        return MFM_E_INVAL;
    }
    BchTables *tab = nullptr;
    int r = pg_device_tables(device, &tab);
    if (r != MFM_OK || 0 == n) {
        return r;
    }
    uint32_t *d_w = nullptr;
    uint8_t *d_rc = nullptr;
    PG_TRY(hipMalloc(&d_w, n * 4));
    PG_TRY(hipMalloc(&d_rc, n));
    PG_TRY(hipMemcpy(d_w, words, n * 4, hipMemcpyHostToDevice));
    r = mfm_bch3121_decode_device(d_w, d_rc, n, device, nullptr);
    if (r == MFM_OK) {
        if (hipMemcpy(words, d_w, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(rc, d_rc, n, hipMemcpyDeviceToHost) != hipSuccess) {
            r = MFM_E_DEVICE;
        }
    }
    (void)hipFree(d_w);
    (void)hipFree(d_rc);
    return r;
}

int mfm_internal_bch_device_tables(int device, MfmBchTables **out)
{
    return pg_device_tables(device, out);
}

const MfmBchTables *mfm_internal_bch_host_tables(void)
{
    return &bch_tables();
}

int mfm_hosttwin_bch3121_decode(uint32_t *word)
{
    uint32_t rc = 0;
    *word = pg_bch_fix(&bch_tables(), *word, &rc);
    return (int)rc;
}

} /* extern "C" */
