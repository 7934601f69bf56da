#ifndef X
#define X 0
#endif
/*
 * mfm_f32.hip - the channel path on floating-point IQ (BASELINE.json configs[4]: "fp32 vs int16 IQ path").
 *
 * The reference has no floating-point path; this one is the integer path with the Q14 quantisation steps taken out
 * (oracle/f32_oracle.h is the fp64 restatement it is checked against, 1e-5 relative):
 *
 *   taps      c[i] = (gain * cexp(j*f_offs*i)) * h[i]        multifm/demod.c:210,232-243 before the int16 casts
 *   FIR       a[n] = sum_i c[i] * x[n*D + i]                 filter/direct_fir.c:363-384, fp32 FMA accumulation
 *   derotate  o[n] = a[n] * w^n, w = cexp(-j*2*pi*off*D/fs)  filter/direct_fir.c:72-79,151-172, closed form
 *   discrim.  s = o[n]*conj(o[n-1]); phi = fast_atan2f(s)    multifm/fm_demod.c:55-72, fast_atan2f.c:101-174
 *             pcm = phi/pi*16384 (float), and truncated to int16 for the stages behind (resampler, pager)
 *
 * Kernel shape.  A workgroup = 8 waves, one tile = 64 output columns (column 0 is the output before the tile's first
 * new one, recomputed, so that the discriminator's history is always the column to the left) by 64 channels (wave w
 * owns channels 8w..8w+7 of the group).  The input windows of the 64 columns are staged into LDS as 64 rows of
 * KT = 64 taps' worth of samples, tap chunk by tap chunk.  Two multiply variants (template parameter):
 *   MFMA  - v_mfma_f32_16x16x4_f32 on the wave's 16 rows (re, im of 8 channels) x 16 columns, exact fp32; a lane's
 *           four B values of four consecutive MFMAs are one ds_read_b128 of its column's row, the A fragments are
 *           laid out on the host to match and live in registers for a chunk.  Default.
 *   VALU  - lane = column, taps [tap][channel] read with scalar loads and fed to v_pk_fma_f32 as SGPR operands, two
 *           packed FMAs per complex tap (MFM_F32_PACKED_FMA; kept for A/B):
 *               acc(re,im) += (cr, cr) * (xr, xi)            op_sel_hi:[0,1,1]
 *               acc(re,im) += (-ci, ci) * (xi, xr)           op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]
 * Measured: DESIGN.md 3.5 (71-90 TFLOP/s of the 157 TFLOP/s fp32 peak).
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/multifm_hip.h"
#include "mfm_taps.h"

extern "C" void mfm_internal_set_error(const char *msg);

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr uint32_t F_NT = 512;  /* threads per workgroup */
constexpr uint32_t F_CB = 8;    /* channels per wave */
constexpr uint32_t F_CG = 64;   /* channels per workgroup */
constexpr uint32_t F_COLS = 64; /* columns per tile, 63 of them new */
constexpr uint32_t F_KT = 64;   /* taps per LDS chunk: 33 KB tiles, three workgroups per CU (128: 0.162 ms, 64: 0.153, 32: 0.156) */
constexpr uint32_t F_NQ = F_KT / 8; /* quads of 16 floats per chunk */
constexpr uint32_t F_PITCH = F_KT + 2; /* row pitch in samples (float2): an odd multiple of 16 B */
/* the tile, or the accumulators of 64 channels x 64 columns on their way to the epilogue, whichever is larger */
constexpr uint32_t F_LDS = F_COLS * F_PITCH * 8u > 64u * 64u * 8u ? F_COLS * F_PITCH * 8u : 64u * 64u * 8u;

struct F32Launch {
    const float2 *tail;   /* [tail_len] unconsumed samples of earlier calls */
    const float2 *blk;    /* [nr_in] this call's samples */
    const float2 *taps_t; /* [T + 8][cpad], rows >= T zero */
    const float4 *afrag;  /* matrix-core variant: [cpad / 8][chunks][16 quads][64 lanes] A fragments */
    const float2 *wlane;  /* [cpad][64]: w^lane per channel */
    const float2 *lut;    /* [256] {T[i], T[i+1]-T[i]} */
    const uint32_t *step_mod; /* [cpad] (off*D) mod fs */
    const float2 *prev_in; /* [cpad] last derotated sample of the previous call */
    float2 *prev_out;     /* [cpad] the same for the next call (another buffer: tile 0 reads while the last tile writes) */
    float2 *tail_out;
    float *pcm_f;         /* [C][out_cap] */
    int16_t *pcm_i;       /* [C][out_cap] */
    float2 *iq;           /* [C][out_cap] or null */
    uint32_t tail_len, nr_in, total; /* total = tail_len + nr_in */
    uint32_t nt, decim, fs, nchan, cpad, out_cap;
    uint32_t n_new;       /* outputs of this call */
    uint32_t n0_mod;      /* (absolute index of this call's first output) mod fs */
    uint32_t pos_end, new_tail;
    uint32_t ntiles, nchunks_p; /* persistent kernel: tiles of the call, tile chunks (= gridDim.x) */
};

static __device__ __forceinline__ float2 f_sample(const F32Launch &L, uint32_t v)
{
    /* clamped: samples past the end feed only columns that are not stored */
    v = v < L.total ? v : L.total - 1u;
    /* one load through a selected pointer (no branch: the loads of a staging pass can be issued back to back) */
    const float2 *p = v < L.tail_len ? L.tail + v : L.blk + (v - L.tail_len);
    return *p;
}

/* multifm/fast_atan2f.c:101-174 on floats; lut[i] = {T[i], T[i+1]-T[i]} */
static __device__ __forceinline__ float f_fast_atan2f(float y, float x, const float2 *lut)
{
    const float xa = fabsf(x), ya = fabsf(y);
    const float mx = fmaxf(xa, ya), mn = fminf(xa, ya);
    if (!(mx > 0.0f)) {
        return 0.0f; /* :111-112 */
    }
    const float z = mn / mx; /* :114-117 */
    float base = z;
    if (!(z < 0.003921569f)) { /* :121 */
        float alpha = z * 255.0f;
        const int idx = ((int)alpha) & 0xff;
        alpha -= (float)idx;
        const float2 e = lut[idx];
        base = e.x + e.y * alpha; /* :125-131 */
    }
    float ang;
    if (xa > ya) { /* :134-163 */
        ang = (x >= 0.0f) ? base : 3.14159265358979f - base;
    } else {
        ang = (x >= 0.0f) ? 1.57079632679490f - base : 1.57079632679490f + base;
    }
    return (y < 0.0f) ? -ang : ang;
}

template <bool MFMA>
__global__ __launch_bounds__(F_NT, 6) void mfm_f32_channel_kernel(const F32Launch L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t f_smem[];
    float2 *xs = reinterpret_cast<float2 *>(f_smem); /* [64][F_PITCH] */
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tile = blockIdx.x;
    const uint32_t ch0 = (blockIdx.y * (F_NT / 64u) + wave) * F_CB; /* < cpad */
    /* column 0 of the tile is output rel0 of this call (-1 for the first tile) */
    const int rel0 = (int)(tile * (F_COLS - 1u)) - 1;

    v2f acc[F_CB];
#pragma unroll
    for (uint32_t k = 0; k < F_CB; k++) {
        acc[k] = v2f{ 0.0f, 0.0f };
    }
    v4f macc[4]; /* matrix-core variant: four 16-column groups of the wave's 16 rows */
#pragma unroll
    for (uint32_t g = 0; g < 4; g++) {
        macc[g] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
    }

    for (uint32_t i0 = 0; i0 < L.nt; i0 += F_KT) {
        const uint32_t kt = L.nt - i0 < F_KT ? L.nt - i0 : F_KT;
        __syncthreads(); /* the previous chunk has been consumed */
        /* taps go in trips of eight; a chunk length that is not a multiple of 8 is rounded up - the taps array has
         * zero rows behind the last tap, the extra columns hold real (finite) samples */
        const uint32_t kte = MFMA ? F_KT : (kt + 7u) & ~7u;
        if (kte == F_KT) {
            /* full chunk: all 16 loads of a thread are in flight together (as a plain loop every iteration waited
             * for its own round trip: 16 of them per tile) */
            float2 r[F_COLS * F_KT / F_NT];
#pragma unroll
            for (uint32_t j = 0; j < F_COLS * F_KT / F_NT; j++) {
                const uint32_t idx = tid + j * F_NT, row = idx / F_KT, col = idx % F_KT;
                const int rel = rel0 + (int)row;
                r[j] = f_sample(L, rel < 0 ? 0u : (uint32_t)rel * L.decim + i0 + col);
            }
#pragma unroll
            for (uint32_t j = 0; j < F_COLS * F_KT / F_NT; j++) {
                const uint32_t idx = tid + j * F_NT, row = idx / F_KT, col = idx % F_KT;
                xs[row * F_PITCH + col] = r[j];
            }
        } else {
            for (uint32_t idx = tid; idx < F_COLS * kte; idx += F_NT) {
                const uint32_t row = idx / kte, col = idx - row * kte;
                const int rel = rel0 + (int)row;
                const uint32_t v = rel < 0 ? 0u : (uint32_t)rel * L.decim + i0 + col;
                xs[row * F_PITCH + col] = f_sample(L, v);
            }
        }
        __syncthreads();
        if (MFMA) {
            /* ---- matrix cores: acc[row][col] += sum_k W[row][k] * e[col][k], rows = (re, im) of the wave's 8 channels,
             * k = the 2 * kte floats of a column's window chunk.  v_mfma_f32_16x16x4_f32 takes one float of A and one
             * of B per lane (k = lane / 16); the k order inside a group of 16 is permuted so that a lane's four B
             * values of four consecutive MFMAs are one ds_read_b128 of its column's row (the A fragments are laid
             * out to match on the host).  Exact fp32 (an FMA chain per output). */
            const uint32_t kg = lane >> 4, n = lane & 15u;
            const float4 *ap = L.afrag + ((size_t)(ch0 / 8u) * ((L.nt + F_KT - 1u) / F_KT) + i0 / F_KT) * F_NQ * 64u + lane;
            float4 a[F_NQ];
#pragma unroll
            for (uint32_t q = 0; q < F_NQ; q++) {
                a[q] = ap[q * 64u]; /* quads past nq hold zeros and are not used */
            }
            const float4 *bbase = reinterpret_cast<const float4 *>(xs + n * F_PITCH) + kg;
            /* one straight run over the 4 x F_NQ (group, quad) steps with the next step's B fragment already requested
             * (as nested loops with a guard per quad, every ds_read_b128 was waited for right where it was issued).
             * A chunk is always staged whole in this variant: taps past the filter are zero in the A fragments, the
             * columns they meet hold real samples */
            {
                float4 b = bbase[0];
#pragma unroll
                for (uint32_t st = 0; st < 4u * F_NQ; st++) {
                    const uint32_t g = st / F_NQ, q = st % F_NQ;
                    float4 bn = b;
                    if (st + 1u < 4u * F_NQ) {
                        const uint32_t g1 = (st + 1u) / F_NQ, q1 = (st + 1u) % F_NQ;
                        bn = bbase[g1 * 16u * (F_PITCH / 2u) + q1 * 4u];
                    }
                    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b.x, macc[g], 0, 0, 0);
                    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b.y, macc[g], 0, 0, 0);
                    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b.z, macc[g], 0, 0, 0);
                    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b.w, macc[g], 0, 0, 0);
                    b = bn;
                }
            }
            continue;
        }
        const float2 *xrow = xs + lane * F_PITCH;
        const float2 *tp = L.taps_t + (size_t)i0 * L.cpad + ch0;
        /* Software pipeline, by hand: the taps of one pair are two s_load_dwordx16 (8 channels x (re, im) per tap),
         * issued from inline asm one pair ahead of the FMAs that use them, because the compiler would only wait for
         * them right where they are issued (scalar loads return out of order: its only wait is lgkmcnt(0) at the first
         * use).  What is in flight is invisible to the compiler, so every stage is settled explicitly: "settle" is an
         * s_waitcnt lgkmcnt(0) that the stage's registers (taps and the two LDS samples) are threaded through, which
         * orders it before their uses; the compiler's own LDS waits in between can under-wait (they do not count the
         * scalar loads) but are always followed by this one.  Four pairs per trip, two register stages; nothing scalar
         * is carried around the loop. */
        v16i c[2][2];
        v2f x[2][2];
        auto fetch = [&](uint32_t i, int st) {
            const float2 *t0 = tp + (size_t)i * L.cpad;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0"
                         : "=&s"(c[st][0]), "=&s"(c[st][1])
                         : "s"(t0), "s"(t0 + L.cpad));
            const float2 a = xrow[i], b = xrow[i + 1];
            x[st][0] = v2f{ a.x, a.y };
            x[st][1] = v2f{ b.x, b.y };
        };
        auto settle = [&](int st) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c[st][0]), "+s"(c[st][1]), "+v"(x[st][0]), "+v"(x[st][1]));
        };
        auto mac = [&](int st) {
#pragma unroll
            for (uint32_t u = 0; u < 2; u++) {
#pragma unroll
                for (uint32_t k = 0; k < F_CB; k += 2) {
                    const v2i ca = { c[st][u][2 * k], c[st][u][2 * k + 1] };
                    const v2i cb = { c[st][u][2 * k + 2], c[st][u][2 * k + 3] };
                    /* a VOP3P result needs one wait state before it is read again: the two channels of a block are
                     * interleaved, so dependent instructions are never adjacent - also across blocks */
                    asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[0,1,1]\n\t"
                        "v_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[0,1,1]\n\t"
                        "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
                        "v_pk_fma_f32 %1, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                        : "+v"(acc[k]), "+v"(acc[k + 1])
                        : "s"(ca), "s"(cb), "v"(x[st][u]));
                }
            }
        };
        for (uint32_t i = 0; i < kte; i += 8) {
            fetch(i, 0);
#pragma unroll
            for (uint32_t sp = 0; sp < 4; sp++) {
                const int st = (int)(sp & 1u);
                settle(st);
                if (sp < 3) {
                    fetch(i + 2u * (sp + 1u), st ^ 1);
                }
                mac(st);
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < F_CB; k++) {
        asm volatile("s_nop 0" : "+v"(acc[k])); /* the wait state for whoever reads the accumulators next */
    }
    if (MFMA) {
        /* C/D layout (lane (kg, n): rows 4kg..4kg+3 of column n) -> the epilogue's (lane = column, 8 channels in
         * registers), through the LDS tile, which nobody reads any more after the barrier */
        __syncthreads();
        float2 *os = xs + (size_t)wave * F_CB * 64u; /* [8 channels][64 columns] of this wave */
        const uint32_t kg = lane >> 4, n = lane & 15u;
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            os[(2u * kg) * 64u + 16u * g + n] = make_float2(macc[g][0], macc[g][1]);
            os[(2u * kg + 1u) * 64u + 16u * g + n] = make_float2(macc[g][2], macc[g][3]);
        }
        /* written and read by the same wave: LDS operations of a wave complete in order */
#pragma unroll
        for (uint32_t k = 0; k < F_CB; k++) {
            const float2 v = os[k * 64u + lane];
            acc[k] = v2f{ v.x, v.y };
        }
    }

    /* ---- epilogue: derotation, discriminator, stores ---- */
    const int rel = rel0 + (int)lane;
    /* phase of column 0: (step_mod * ((n0 + rel0) mod fs)) mod fs, exact; lanes 0..15 do one channel each */
    float2 wbase_l = make_float2(1.0f, 0.0f);
    if (lane < F_CB) {
        const uint32_t sm = L.step_mod[ch0 + lane];
        /* n0_mod + rel0 may be -1 for the first tile: add fs first */
        const uint64_t nmod = ((uint64_t)L.n0_mod + (uint64_t)L.fs + (uint64_t)(int64_t)rel0) % L.fs;
        const uint32_t ph = (uint32_t)(((uint64_t)sm * nmod) % L.fs);
        const double t = (double)ph / (double)L.fs; /* turns, [0, 1) */
        float s, c;
        sincospif((float)(-2.0 * t), &s, &c);
        wbase_l = make_float2(c, s);
    }
    const bool stored = lane != 0u && rel < (int)L.n_new;
#pragma unroll
    for (uint32_t k = 0; k < F_CB; k++) {
        const uint32_t ch = ch0 + k;
        const float wbr = __shfl(wbase_l.x, (int)k), wbi = __shfl(wbase_l.y, (int)k);
        const float2 wl = L.wlane[(size_t)ch * 64u + lane];
        /* w^(n_abs) = w^(n0 + rel0) * w^lane */
        const float wr = wbr * wl.x - wbi * wl.y, wi = wbr * wl.y + wbi * wl.x;
        float o_re = acc[k].x * wr - acc[k].y * wi;
        float o_im = acc[k].x * wi + acc[k].y * wr;
        if (tile == 0u && lane == 0u) {
            const float2 p = L.prev_in[ch]; /* the output before this call's first one */
            o_re = p.x;
            o_im = p.y;
        }
        const float p_re = __shfl_up(o_re, 1), p_im = __shfl_up(o_im, 1);
        /* multifm/fm_demod.c:63-64 */
        const float s_re = o_re * p_re + o_im * p_im;
        const float s_im = o_im * p_re - o_re * p_im;
        const float phi = f_fast_atan2f(s_im, s_re, L.lut);
        const float pcm = phi * (16384.0f / 3.14159265358979f);
        if (stored && ch < L.nchan) {
            const size_t at = (size_t)ch * L.out_cap + (uint32_t)rel;
            L.pcm_f[at] = pcm;
            L.pcm_i[at] = (int16_t)pcm; /* truncation, as multifm/fm_demod.c:72 */
            if (L.iq) {
                L.iq[at] = make_float2(o_re, o_im);
            }
            if (rel == (int)L.n_new - 1) {
                L.prev_out[ch] = make_float2(o_re, o_im);
            }
        }
    }
}

/*
 * The persistent form of the matrix-core variant (round 2; the default).  A workgroup keeps a run of consecutive
 * tiles: while the matrix instructions of one 32-tap chunk run out of one LDS buffer, the next chunk's samples are
 * already on their way into registers (issued before the multiply phase, stored to the other buffer behind it),
 * one barrier per chunk.  What the non-persistent kernel pays per tile and this one does not:
 *   - staging addresses: a thread's eight samples of a chunk are 8 D apart, one 64-bit add each; the clamps and the
 *     tail / block selection are decided once per chunk for the whole workgroup (only the first and last tiles of a
 *     call need them);
 *   - A fragments: the loop runs quad-major, so quad q of the NEXT chunk is requested as soon as this chunk is done
 *     with quad q - 32 registers hold both, the loads have a whole multiply phase to arrive;
 *   - the accumulators' trip through LDS: the epilogue works in the matrix result layout (lane (kg, n) holds channels
 *     2 kg, 2 kg + 1 of columns 16 g + n); the column to the left is one DPP row shift, plus one lane shuffle per group
 *     for n = 0; the per-column derotation factors of the lane's eight (channel, column) slots live in registers;
 *   - the arctangent table is in LDS, the quotient is v_rcp_f32 * min (1 ulp; the path's tolerance is 1e-5).
 */
constexpr uint32_t P_KT = 32;             /* taps per staged chunk: with 64 the A fragments (32 registers) and the samples in
                                           * flight (16) leave the register allocator no room - it spills into the loop */
constexpr uint32_t P_NQ = P_KT / 8;       /* quads of 16 floats per chunk; the A-fragment array is [..][quad][lane], so a
                                           * chunk of this kernel is simply four consecutive quads of it */
constexpr uint32_t P_PITCH = P_KT + 2;    /* row pitch in samples (float2): 272 B, an odd multiple of 16 B */
constexpr uint32_t P_ROWS_PER_PASS = F_NT / P_KT; /* 16 rows of a chunk per pass over the workgroup's threads */
constexpr uint32_t P_NLD = F_COLS / P_ROWS_PER_PASS; /* 4 samples per thread and chunk */
constexpr uint32_t P_BUF = F_COLS * P_PITCH * 8u;      /* one staged chunk: 64 rows x 34 float2 */
constexpr uint32_t P_LDS = 2u * P_BUF + 256u * 8u;     /* two buffers + the table */

static __device__ __forceinline__ float p_fast_atan2f(float y, float x, const float2 *lut_s)
{
    const float xa = fabsf(x), ya = fabsf(y);
    const float mx = fmaxf(xa, ya), mn = fminf(xa, ya);
    const float z = mn * __builtin_amdgcn_rcpf(mx); /* NaN for (0, 0): selected away at the end */
    float alpha = z * 255.0f;
    const int idx = ((int)alpha) & 0xff;
    alpha -= (float)idx;
    const float2 e = lut_s[idx];
    float base = fmaf(e.y, alpha, e.x);
    base = (z < 0.003921569f) ? z : base; /* fast_atan2f.c:121 */
    const float hpi = 1.57079632679490f, pi = 3.14159265358979f;
    /* fast_atan2f.c:134-163 */
    const float a1 = (x >= 0.0f) ? base : pi - base;
    const float a2 = (x >= 0.0f) ? hpi - base : hpi + base;
    float ang = (xa > ya) ? a1 : a2;
    ang = (y < 0.0f) ? -ang : ang;
    return (mx > 0.0f) ? ang : 0.0f;
}

/*
 * One step of the persistent kernel's multiply phase: column group g = ST % 4 of quad q = ST / 4.  The B fragment of
 * step ST + 1 is requested before the four matrix instructions of step ST are issued.  Left to the compiler the read
 * ends up right in front of its use (it rates the register pressure of the staged samples higher), so the reads and
 * their waits are written out: LDS returns in order and at most two reads are outstanding, so lgkmcnt(1) means the
 * older one has arrived.  When the chunk is done with quad q, the next chunk's quad q is requested into the same
 * registers (after the last chunk: a load nobody uses).
 */
template <uint32_t NG, uint32_t ST>
static __device__ __forceinline__ void p_step(float4 (&a)[P_NQ], v4f (&macc)[4], v4f &b_even, v4f &b_odd, uint32_t baddr,
                                              const float4 *anext)
{
    constexpr uint32_t q = ST / NG, g = ST % NG;
    v4f &cur = (ST & 1u) ? b_odd : b_even;
    v4f &nxt = (ST & 1u) ? b_even : b_odd;
    if constexpr (ST + 1u < NG * P_NQ) {
        constexpr uint32_t q1 = (ST + 1u) / NG, g1 = (ST + 1u) % NG;
        constexpr uint32_t OFF = g1 * 16u * P_PITCH * 8u + q1 * 64u; /* 16 rows per column group, 4 float4 per quad */
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nxt) : "v"(baddr), "n"(OFF) : "memory");
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur));
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur));
    }
    const v4f b = cur;
    if (X & 8) { macc[g][0] += a[q].x + b.x; macc[g][1] += a[q].y + b.y; } else {
    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b.x, macc[g], 0, 0, 0);
    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b.y, macc[g], 0, 0, 0);
    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b.z, macc[g], 0, 0, 0);
    macc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b.w, macc[g], 0, 0, 0); }
    if constexpr (g == NG - 1u && !(X & 16)) {
        a[q] = anext[q * 64u];
    }
}

/* NG = column groups of 16 the tile uses: 4, or fewer for the short tile that ends a workgroup's run */
template <uint32_t NG, uint32_t... ST>
static __device__ __forceinline__ void p_phase(std::integer_sequence<uint32_t, ST...>, float4 (&a)[P_NQ], v4f (&macc)[4], v4f &b_even,
                                               v4f &b_odd, uint32_t baddr, const float4 *anext)
{
    (p_step<NG, ST>(a, macc, b_even, b_odd, baddr, anext), ...);
}

__global__ __launch_bounds__(F_NT, 4) void mfm_f32_channel_kernel_p(const F32Launch L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t f_smem[];
    float2 *const xs0 = reinterpret_cast<float2 *>(f_smem);
    float2 *const lut_s = reinterpret_cast<float2 *>(f_smem + 2u * P_BUF);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kg = lane >> 4, n = lane & 15u;
    const uint32_t ch0 = (blockIdx.y * (F_NT / 64u) + wave) * F_CB; /* < cpad */
    const uint32_t nck = (L.nt + P_KT - 1u) / P_KT;
    /* this workgroup's run of outputs [o_first, o_end): whole tiles of 63 new columns, then one short tile that uses
     * only as many 16-column groups as the rest needs - every workgroup gets the same work to within a group */
    const uint32_t o_first = (uint32_t)(((uint64_t)blockIdx.x * L.n_new) / L.nchunks_p);
    const uint32_t o_end = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * L.n_new) / L.nchunks_p);
    if (o_first >= o_end) {
        return;
    }
    const uint32_t t_end = (o_end - o_first + F_COLS - 2u) / (F_COLS - 1u); /* tiles of the run, numbered from 0 */
    for (uint32_t i = tid; i < 256u; i += F_NT) {
        lut_s[i] = L.lut[i];
    }
    /* two waves of the workgroup share a SIMD and the arbiter favours the older one: without this the second wave of
     * every pair trails and the other seven wait for it at each chunk's barrier */
    if (wave >= 4u) {
        __builtin_amdgcn_s_setprio(1);
    }

    /* staging: thread (row0 = tid / 32, col = tid % 32) owns rows row0 + 16 j of every chunk */
    const uint32_t row0 = tid / P_KT, col = tid % P_KT;
    float2 r[P_NLD];
    auto stage_load = [&](uint32_t tile, uint32_t ck) {
        const int rel0 = (int)(o_first + tile * (F_COLS - 1u)) - 1;
        const uint32_t i0 = ck * P_KT;
        /* whole-workgroup decisions: does any row of the chunk start before the stream (tile 0), reach into the tail
         * of the previous call, or run past the last sample? */
        const int64_t v_lo = (int64_t)rel0 * (int64_t)L.decim + i0;
        const uint64_t v_hi = (uint64_t)(rel0 + 63) * L.decim + i0 + P_KT; /* one past the largest index (rel0 + 63 >= 62) */
        if (v_lo >= (int64_t)L.tail_len && v_hi <= L.total) {
            const float2 *p = L.blk + ((size_t)(v_lo - (int64_t)L.tail_len) + (size_t)row0 * L.decim + col);
            const size_t step = (size_t)P_ROWS_PER_PASS * L.decim;
#pragma unroll
            for (uint32_t j = 0; j < P_NLD; j++) {
                if (X & 1) { r[j] = make_float2((float)(tid + j), 1.0f); } else
                r[j] = p[j * step];
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < P_NLD; j++) {
                const int rel = rel0 + (int)(row0 + P_ROWS_PER_PASS * j);
                r[j] = f_sample(L, rel < 0 ? 0u : (uint32_t)rel * L.decim + i0 + col);
            }
        }
    };
    auto stage_store = [&](uint32_t buf) {
        float2 *xs = xs0 + buf * (P_BUF / 8u);
#pragma unroll
        for (uint32_t j = 0; j < P_NLD; j++) {
            xs[(row0 + P_ROWS_PER_PASS * j) * P_PITCH + col] = r[j];
        }
    };

    /* A fragments of (channel group of the wave, chunk ck): [P_NQ][64 lanes] float4 */
    const float4 *const afrag_w = L.afrag + (size_t)(ch0 / 8u) * ((L.nt + F_KT - 1u) / F_KT) * F_NQ * 64u + lane;
    float4 a[P_NQ];
#pragma unroll
    for (uint32_t q = 0; q < P_NQ; q++) {
        a[q] = afrag_w[q * 64u];
    }

    /* derotation: column 16 g + n of tile t turns by w^(n0 + rel0(t)) * w^n * (w^16)^g.  The lane keeps w^n, w^16 and
     * w^63 of its two channels (rows of the per-channel power table) and the running w^(n0 + rel0): exact (integer
     * phase, sincospif) at the first tile of the run and every 16 tiles, one complex multiply by w^63 in between. */
    float2 w_n[2], w_16[2], w_63[2], w_b[2];
    uint32_t step_mod[2];
#pragma unroll
    for (uint32_t j = 0; j < 2; j++) {
        const float2 *row = L.wlane + (size_t)(ch0 + 2u * kg + j) * 64u;
        w_n[j] = row[n];
        w_16[j] = row[16];
        w_63[j] = row[63];
        w_b[j] = make_float2(1.0f, 0.0f);
        step_mod[j] = L.step_mod[ch0 + 2u * kg + j];
    }

    v4f macc[4];
#pragma unroll
    for (uint32_t g = 0; g < 4; g++) {
        macc[g] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
    }

    stage_load(0, 0);
    stage_store(0);
    __syncthreads();

    uint32_t buf = 0;
    /* one tile: NG column groups of 16 (4 for a whole tile; the short tile that ends the run gets its own copy of the
     * code, outside the loop over the whole ones) */
    auto do_tile = [&](auto ng_const, const uint32_t tile) {
        constexpr uint32_t NG = decltype(ng_const)::value;
        /* new columns of this tile (column 0 is the recomputed one) */
        const uint32_t fresh = o_end - (o_first + tile * (F_COLS - 1u)) < F_COLS - 1u ? o_end - (o_first + tile * (F_COLS - 1u)) : F_COLS - 1u;
#pragma clang loop unroll(disable)
        for (uint32_t ck = 0; ck < nck; ck++) {
            /* the chunk after this one */
            uint32_t ntile = tile, nckk = ck + 1u;
            if (nckk == nck) {
                nckk = 0;
                ntile = tile + 1u;
            }
            const bool more = ntile < t_end;
            /* always the same number of loads, so that every path into the multiply phase has the same number of loads in flight
             * behind the A fragments (after the run's last chunk they fetch this chunk again and are dropped) */
            stage_load(more ? ntile : tile, more ? nckk : ck);
            /* ---- multiply: acc[row][col] += sum_k W[row][k] * e[col][k] over the chunk, quad-major ---- */
            {
                const uint32_t baddr = (uint32_t)(uintptr_t)(xs0 + buf * (P_BUF / 8u) + n * P_PITCH) + 16u * kg;
                v4f b_even, b_odd;
                asm volatile("ds_read_b128 %0, %1" : "=v"(b_even) : "v"(baddr) : "memory");
                const float4 *anext = afrag_w + (size_t)nckk * P_NQ * 64u;
                p_phase<NG>(std::make_integer_sequence<uint32_t, NG * P_NQ>{}, a, macc, b_even, b_odd, baddr, anext);
            }
            if (more) {
                stage_store(buf ^ 1u);
            }
            __syncthreads();
            buf ^= 1u;
            if (ck + 1u < nck) {
                continue;
            }

            /* ---- epilogue of the tile, in the matrix result layout ---- */
            const int rel0 = (int)(o_first + tile * (F_COLS - 1u)) - 1;
            if ((tile & 15u) == 0u) {
                /* phase of column 0: (step_mod * ((n0 + rel0) mod fs)) mod fs, exact */
                const uint64_t nmod = ((uint64_t)L.n0_mod + (uint64_t)L.fs + (uint64_t)(int64_t)rel0) % L.fs;
#pragma unroll
                for (uint32_t j = 0; j < 2; j++) {
                    const uint32_t ph = (uint32_t)(((uint64_t)step_mod[j] * nmod) % L.fs);
                    const double t = (double)ph / (double)L.fs; /* turns, [0, 1) */
                    float sn, cs;
                    sincospif((float)(-2.0 * t), &sn, &cs);
                    w_b[j] = make_float2(cs, sn);
                }
            }
            float o_re[4][2], o_im[4][2];
#pragma unroll
            for (uint32_t j = 0; j < 2; j++) {
                /* w^(n_abs) = w^(n0 + rel0) * w^n * (w^16)^g */
                float wr = w_b[j].x * w_n[j].x - w_b[j].y * w_n[j].y, wi = w_b[j].x * w_n[j].y + w_b[j].y * w_n[j].x;
#pragma unroll
                for (uint32_t g = 0; g < NG; g++) {
                    const float ar = macc[g][2 * j], ai = macc[g][2 * j + 1];
                    o_re[g][j] = ar * wr - ai * wi;
                    o_im[g][j] = ar * wi + ai * wr;
                    const float nr = wr * w_16[j].x - wi * w_16[j].y;
                    wi = wr * w_16[j].y + wi * w_16[j].x;
                    wr = nr;
                }
                /* the next tile starts 63 columns on */
                const float br = w_b[j].x * w_63[j].x - w_b[j].y * w_63[j].y;
                w_b[j].y = w_b[j].x * w_63[j].y + w_b[j].y * w_63[j].x;
                w_b[j].x = br;
            }
            if (rel0 < 0 && n == 0u) {
#pragma unroll
                for (uint32_t j = 0; j < 2; j++) {
                    const float2 p = L.prev_in[ch0 + 2u * kg + j]; /* the output before this call's first one */
                    o_re[0][j] = p.x;
                    o_im[0][j] = p.y;
                }
            }
#pragma unroll
            for (uint32_t g = 0; g < 4; g++) {
                macc[g] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
            }
            /* a whole tile, none of its columns the call's last output, every channel exists: plain stores */
            const bool plain = fresh == F_COLS - 1u && rel0 + 63 < (int)L.n_new - 1 && ch0 + F_CB <= L.nchan && !L.iq;
#pragma unroll
            for (uint32_t j = 0; j < 2; j++) {
                const uint32_t ch = ch0 + 2u * kg + j;
                float *pf = L.pcm_f + (size_t)ch * L.out_cap + rel0 + (int)n;
                int16_t *pi = L.pcm_i + (size_t)ch * L.out_cap + rel0 + (int)n;
#pragma unroll
                for (uint32_t g = 0; g < NG; g++) {
                    /* the column to the left: lane n - 1 of the row; for n = 0, lane 15 of the row in the group before */
                    float l_re = 0.0f, l_im = 0.0f;
                    if (g > 0) {
                        l_re = __shfl(o_re[g - 1][j], (int)(lane | 15u));
                        l_im = __shfl(o_im[g - 1][j], (int)(lane | 15u));
                    }
                    const float p_re = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, l_re),
                                           __builtin_bit_cast(int, o_re[g][j]), 0x111, 0xf, 0xf, false));
                    const float p_im = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, l_im),
                                           __builtin_bit_cast(int, o_im[g][j]), 0x111, 0xf, 0xf, false));
                    /* multifm/fm_demod.c:63-64 */
                    const float s_re = o_re[g][j] * p_re + o_im[g][j] * p_im;
                    const float s_im = o_im[g][j] * p_re - o_re[g][j] * p_im;
                    const float pcm = (X & 4) ? s_re + s_im : p_fast_atan2f(s_im, s_re, lut_s) * (16384.0f / 3.14159265358979f);
                    if (plain) {
                        if (X & 2) { if (pcm == 12345.678f) pf[0] = pcm; } else
                        if (g > 0 || n != 0u) {
                            pf[16 * (int)g] = pcm;
                            pi[16 * (int)g] = (int16_t)pcm; /* truncation, as multifm/fm_demod.c:72 */
                        }
                        continue;
                    }
                    const int rel = rel0 + (int)(16u * g + n);
                    if ((g | n) != 0u && rel < (int)o_end && ch < L.nchan) { /* columns past the run belong to the next workgroup */
                        pf[16 * (int)g] = pcm;
                        pi[16 * (int)g] = (int16_t)pcm;
                        if (L.iq) {
                            L.iq[(size_t)ch * L.out_cap + (uint32_t)rel] = make_float2(o_re[g][j], o_im[g][j]);
                        }
                        if (rel == (int)L.n_new - 1) {
                            L.prev_out[ch] = make_float2(o_re[g][j], o_im[g][j]);
                        }
                    }
                }
            }
        }
    };

    /* whole tiles, then the short one: its column count decides how many groups of 16 are multiplied at all */
    const uint32_t last_fresh = o_end - o_first - (t_end - 1u) * (F_COLS - 1u); /* 1..63 new columns in the last tile */
    const uint32_t t_whole = last_fresh == F_COLS - 1u ? t_end : t_end - 1u;
#pragma clang loop unroll(disable)
    for (uint32_t tile = 0; tile < t_whole; tile++) {
        do_tile(std::integral_constant<uint32_t, 4>{}, tile);
    }
    if (t_whole < t_end) {
        switch ((last_fresh + 16u) / 16u) {
        case 1:
            do_tile(std::integral_constant<uint32_t, 1>{}, t_whole);
            break;
        case 2:
            do_tile(std::integral_constant<uint32_t, 2>{}, t_whole);
            break;
        case 3:
            do_tile(std::integral_constant<uint32_t, 3>{}, t_whole);
            break;
        default:
            do_tile(std::integral_constant<uint32_t, 4>{}, t_whole);
            break;
        }
    }
}

/* what this call leaves unconsumed: samples pos_end .. pos_end + new_tail of the virtual stream */
__global__ void mfm_f32_tail_kernel(const F32Launch L)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L.new_tail; i += gridDim.x * blockDim.x) {
        L.tail_out[i] = f_sample(L, L.pos_end + i);
    }
}

thread_local char g_f_error[256] = "";

struct F32Chan {
    int32_t offset_hz;
    double gain;
    std::vector<double> lpf;
};

} /* namespace */

struct mfm_f32_engine {
    mfm_f32_config cfg{};
    std::vector<F32Chan> chans;
    bool committed = false;
    uint32_t nt = 0, cpad = 0, out_cap = 0, tail_cap = 0;
    float2 *d_taps = nullptr, *d_wlane = nullptr, *d_lut = nullptr;
    float4 *d_afrag = nullptr;
    bool use_mfma = true, persistent = true;
    uint32_t slots = 512;
    float2 *d_prev[2] = { nullptr, nullptr };
    int prev_cur = 0;
    uint32_t *d_step = nullptr;
    float2 *d_tail[2] = { nullptr, nullptr };
    float *d_pcm_f = nullptr;
    int16_t *d_pcm_i = nullptr;
    float2 *d_iq = nullptr;
    float2 *d_stage = nullptr; /* process_host */
    int cur = 0;
    uint32_t tail = 0;
    uint64_t n_out_total = 0;
};

#define F_TRY(expr)                                                                                          \
    do {                                                                                                     \
        hipError_t err_ = (expr);                                                                            \
        if (err_ != hipSuccess) {                                                                            \
            snprintf(g_f_error, sizeof(g_f_error), "%s failed: %s", #expr, hipGetErrorString(err_));         \
            mfm_internal_set_error(g_f_error);                                                               \
            return MFM_E_DEVICE;                                                                             \
        }                                                                                                    \
    } while (0)

extern "C" {

int mfm_f32_create(struct mfm_f32_engine **pe, const struct mfm_f32_config *cfg)
{
    if (!pe || !cfg) {
        return MFM_E_INVAL;
    }
    *pe = nullptr;
    if (cfg->abi_version != MFM_ABI_VERSION || 0 == cfg->sample_rate_hz || 0 == cfg->decimation ||
        0 == cfg->max_block_samples || cfg->sample_rate_hz >= (1u << 31)) {
        return MFM_E_INVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) {
        return MFM_E_DEVICE; /* no CPU path */
    }
    mfm_f32_engine *e = new (std::nothrow) mfm_f32_engine();
    if (!e) {
        return MFM_E_NOMEM;
    }
    e->cfg = *cfg;
    *pe = e;
    return MFM_OK;
}

int mfm_f32_add_channel(struct mfm_f32_engine *e, int32_t offset_hz, const double *lpf_taps, size_t nr_taps, double gain)
{
    if (!e || !lpf_taps || 0 == nr_taps) {
        return MFM_E_INVAL;
    }
    if (e->committed) {
        return MFM_E_STATE;
    }
    /* as the integer engine: one filter length per engine, at least one decimation step long */
    if ((e->nt && nr_taps != e->nt) || nr_taps < e->cfg.decimation) {
        return MFM_E_INVAL;
    }
    e->nt = (uint32_t)nr_taps;
    F32Chan c;
    c.offset_hz = offset_hz;
    c.gain = gain;
    c.lpf.assign(lpf_taps, lpf_taps + nr_taps);
    e->chans.push_back(std::move(c));
    return (int)e->chans.size() - 1;
}

int mfm_f32_commit(struct mfm_f32_engine *e)
{
    if (!e) {
        return MFM_E_INVAL;
    }
    if (e->committed || e->chans.empty()) {
        return MFM_E_STATE;
    }
    const uint32_t C = (uint32_t)e->chans.size(), T = e->nt, D = e->cfg.decimation, fs = e->cfg.sample_rate_hz;
    e->cpad = (C + F_CG - 1) / F_CG * F_CG;
    e->tail_cap = T + 8;
    e->out_cap = (e->cfg.max_block_samples + e->tail_cap) / D + 8;
    if ((uint64_t)e->out_cap * D >= (1ull << 31) || (uint64_t)e->cfg.max_block_samples + e->tail_cap >= (1ull << 31)) {
        return MFM_E_INVAL;
    }
    std::vector<float2> taps((size_t)(T + 8u) * e->cpad, make_float2(0.0f, 0.0f)); /* + zero rows: trips of 8 taps */
    std::vector<float2> wlane((size_t)e->cpad * 64, make_float2(1.0f, 0.0f));
    std::vector<uint32_t> step(e->cpad, 0);
    std::vector<double> re(T), im(T);
    for (uint32_t c = 0; c < C; c++) {
        const F32Chan &ch = e->chans[c];
        mfm_taps_rotate_f64(ch.lpf.data(), T, ch.offset_hz, fs, ch.gain, re.data(), im.data());
        for (uint32_t i = 0; i < T; i++) {
            taps[(size_t)i * e->cpad + c] = make_float2((float)re[i], (float)im[i]);
        }
        int64_t m = ((int64_t)ch.offset_hz * (int64_t)D) % (int64_t)fs;
        if (m < 0) {
            m += fs;
        }
        step[c] = (uint32_t)m;
        for (uint32_t l = 0; l < 64; l++) {
            const uint64_t ph = ((uint64_t)m * l) % fs;
            const double ang = -2.0 * M_PI * ((double)ph / (double)fs);
            wlane[(size_t)c * 64 + l] = make_float2((float)cos(ang), (float)sin(ang));
        }
    }
    /* A fragments of the matrix-core variant.  Rows of a wave: 2c = real part, 2c + 1 = imaginary part of its channel
     * c; k = 2 * tap + part of the sample: W[2c] = (cr, -ci, ...), W[2c+1] = (ci, cr, ...) (filter/complex.h:40-46).
     * Lane (kg, r) of quad q, MFMA m holds W[r][16 q + 4 kg + m] of the chunk. */
    const uint32_t nchunks = (T + F_KT - 1) / F_KT;
    std::vector<float4> afrag((size_t)(e->cpad / 8) * nchunks * F_NQ * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    for (uint32_t grp = 0; grp < e->cpad / 8; grp++) {
        for (uint32_t ck = 0; ck < nchunks; ck++) {
            for (uint32_t q = 0; q < F_NQ; q++) {
                for (uint32_t ln = 0; ln < 64; ln++) {
                    const uint32_t r = ln & 15u, kg = ln >> 4, c = grp * 8 + r / 2, part = r & 1u;
                    float v[4];
                    for (uint32_t m = 0; m < 4; m++) {
                        const uint32_t kk = 16 * q + 4 * kg + m, tap = ck * F_KT + kk / 2, comp = kk & 1u;
                        float val = 0.0f;
                        if (tap < T && c < C) {
                            const float2 t = taps[(size_t)tap * e->cpad + c];
                            val = part == 0 ? (comp == 0 ? t.x : -t.y) : (comp == 0 ? t.y : t.x);
                        }
                        v[m] = val;
                    }
                    afrag[(((size_t)grp * nchunks + ck) * F_NQ + q) * 64 + ln] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    e->use_mfma = !(e->cfg.flags & MFM_F32_PACKED_FMA); /* A/B: the packed-FMA variant */
    float tbl[257];
    mfm_hosttwin_atan_table(tbl);
    if (!mfm_hosttwin_atan_table_ok()) {
        mfm_internal_set_error("atan table self-check failed (host libm rounds atan() differently)");
        return MFM_E_INVAL;
    }
    std::vector<float2> lut(256);
    for (int i = 0; i < 256; i++) {
        lut[i] = make_float2(tbl[i], tbl[i + 1] - tbl[i]);
    }
    F_TRY(hipSetDevice(e->cfg.device));
    F_TRY(hipMalloc(&e->d_taps, taps.size() * sizeof(float2)));
    F_TRY(hipMemcpy(e->d_taps, taps.data(), taps.size() * sizeof(float2), hipMemcpyHostToDevice));
    F_TRY(hipMalloc(&e->d_afrag, afrag.size() * sizeof(float4)));
    F_TRY(hipMemcpy(e->d_afrag, afrag.data(), afrag.size() * sizeof(float4), hipMemcpyHostToDevice));
    F_TRY(hipMalloc(&e->d_wlane, wlane.size() * sizeof(float2)));
    F_TRY(hipMemcpy(e->d_wlane, wlane.data(), wlane.size() * sizeof(float2), hipMemcpyHostToDevice));
    F_TRY(hipMalloc(&e->d_lut, lut.size() * sizeof(float2)));
    F_TRY(hipMemcpy(e->d_lut, lut.data(), lut.size() * sizeof(float2), hipMemcpyHostToDevice));
    F_TRY(hipMalloc(&e->d_step, step.size() * sizeof(uint32_t)));
    F_TRY(hipMemcpy(e->d_step, step.data(), step.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (int i = 0; i < 2; i++) {
        F_TRY(hipMalloc(&e->d_prev[i], e->cpad * sizeof(float2)));
        F_TRY(hipMemset(e->d_prev[i], 0, e->cpad * sizeof(float2))); /* fm_demod.c:41-42: the history starts at 0 */
        F_TRY(hipMalloc(&e->d_tail[i], e->tail_cap * sizeof(float2)));
        F_TRY(hipMemset(e->d_tail[i], 0, e->tail_cap * sizeof(float2)));
    }
    F_TRY(hipMalloc(&e->d_pcm_f, (size_t)C * e->out_cap * sizeof(float)));
    F_TRY(hipMalloc(&e->d_pcm_i, (size_t)C * e->out_cap * sizeof(int16_t)));
    if (e->cfg.flags & MFM_F32_WANT_IQ) {
        F_TRY(hipMalloc(&e->d_iq, (size_t)C * e->out_cap * sizeof(float2)));
    }
    F_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mfm_f32_channel_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F_LDS)));
    F_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mfm_f32_channel_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F_LDS)));
    F_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mfm_f32_channel_kernel_p),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(P_LDS)));
    {
        hipDeviceProp_t prop;
        F_TRY(hipGetDeviceProperties(&prop, e->cfg.device));
        e->slots = 2u * (uint32_t)prop.multiProcessorCount; /* two workgroups of the persistent kernel per CU */
    }
    e->persistent = e->use_mfma && !(e->cfg.flags & MFM_F32_TILE_KERNEL);
    F_TRY(hipDeviceSynchronize());
    e->committed = true;
    return MFM_OK;
}

void mfm_f32_destroy(struct mfm_f32_engine **pe)
{
    if (!pe || !*pe) {
        return;
    }
    mfm_f32_engine *e = *pe;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(e->d_taps);
    (void)hipFree(e->d_wlane);
    (void)hipFree(e->d_afrag);
    (void)hipFree(e->d_lut);
    (void)hipFree(e->d_step);
    (void)hipFree(e->d_prev[0]);
    (void)hipFree(e->d_prev[1]);
    (void)hipFree(e->d_tail[0]);
    (void)hipFree(e->d_tail[1]);
    (void)hipFree(e->d_pcm_f);
    (void)hipFree(e->d_pcm_i);
    (void)hipFree(e->d_iq);
    (void)hipFree(e->d_stage);
    delete e;
    *pe = nullptr;
}

size_t mfm_f32_max_out(const struct mfm_f32_engine *e)
{
    return e ? e->out_cap : 0;
}

int mfm_f32_process_device(struct mfm_f32_engine *e, const float *d_iq, size_t nr_samples, void *stream,
                           struct mfm_f32_block *out)
{
    if (!e || !out || (!d_iq && nr_samples)) {
        return MFM_E_INVAL;
    }
    if (!e->committed) {
        return MFM_E_STATE;
    }
    if (nr_samples > e->cfg.max_block_samples) {
        return MFM_E_INVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    F_TRY(hipSetDevice(e->cfg.device));
    const uint32_t C = (uint32_t)e->chans.size(), T = e->nt, D = e->cfg.decimation, fs = e->cfg.sample_rate_hz;
    const uint32_t total = e->tail + (uint32_t)nr_samples;
    /* filter/direct_fir.c:455-472: an output while at least T samples are left */
    const uint32_t n_new = total >= T ? (total - T) / D + 1u : 0u;
    const uint32_t pos_end = n_new * D; /* <= total: T >= D */
    const uint32_t new_tail = total - pos_end;
    F32Launch L{};
    L.tail = e->d_tail[e->cur];
    L.blk = reinterpret_cast<const float2 *>(d_iq);
    L.taps_t = e->d_taps;
    L.afrag = e->d_afrag;
    L.wlane = e->d_wlane;
    L.lut = e->d_lut;
    L.step_mod = e->d_step;
    L.prev_in = e->d_prev[e->prev_cur];
    L.prev_out = e->d_prev[e->prev_cur ^ 1];
    L.tail_out = e->d_tail[e->cur ^ 1];
    L.pcm_f = e->d_pcm_f;
    L.pcm_i = e->d_pcm_i;
    L.iq = e->d_iq;
    L.tail_len = e->tail;
    L.nr_in = (uint32_t)nr_samples;
    L.total = total;
    L.nt = T;
    L.decim = D;
    L.fs = fs;
    L.nchan = C;
    L.cpad = e->cpad;
    L.out_cap = e->out_cap;
    L.n_new = n_new;
    L.n0_mod = (uint32_t)(e->n_out_total % fs);
    L.pos_end = pos_end;
    L.new_tail = new_tail;
    if (n_new) {
        const dim3 grid((n_new + F_COLS - 2u) / (F_COLS - 1u), e->cpad / F_CG);
        if (e->persistent) {
            const uint32_t per_group = e->slots / grid.y ? e->slots / grid.y : 1u;
            /* runs of outputs, not of tiles; no run shorter than half a tile */
            const uint32_t most = (n_new + 31u) / 32u;
            L.ntiles = grid.x;
            L.nchunks_p = most < per_group ? most : per_group;
            hipLaunchKernelGGL(mfm_f32_channel_kernel_p, dim3(L.nchunks_p, grid.y), dim3(F_NT), P_LDS, s, L);
        } else if (e->use_mfma) {
            hipLaunchKernelGGL(mfm_f32_channel_kernel<true>, grid, dim3(F_NT), F_LDS, s, L);
        } else {
            hipLaunchKernelGGL(mfm_f32_channel_kernel<false>, grid, dim3(F_NT), F_LDS, s, L);
        }
        F_TRY(hipGetLastError());
        e->prev_cur ^= 1;
    }
    if (total) {
        if (new_tail) {
            hipLaunchKernelGGL(mfm_f32_tail_kernel, dim3((new_tail + 255u) / 256u), dim3(256), 0, s, L);
            F_TRY(hipGetLastError());
        }
        e->cur ^= 1;
    }
    e->tail = new_tail;
    e->n_out_total += n_new;
    out->d_pcm_f32 = e->d_pcm_f;
    out->d_pcm_i16 = e->d_pcm_i;
    out->d_iq_f32 = reinterpret_cast<float *>(e->d_iq);
    out->stride = e->out_cap;
    out->nr_out = n_new;
    out->nr_channels = C;
    return MFM_OK;
}

int mfm_f32_process_host(struct mfm_f32_engine *e, const float *iq, size_t nr_samples, float *pcm_f32, int16_t *pcm_i16,
                         float *iq_f32, size_t out_stride, size_t *nr_out)
{
    if (!e || !nr_out || (!iq && nr_samples)) {
        return MFM_E_INVAL;
    }
    if (!e->committed) {
        return MFM_E_STATE;
    }
    if (nr_samples > e->cfg.max_block_samples) {
        return MFM_E_INVAL;
    }
    F_TRY(hipSetDevice(e->cfg.device));
    if (!e->d_stage) {
        F_TRY(hipMalloc(&e->d_stage, (size_t)e->cfg.max_block_samples * sizeof(float2)));
    }
    if (nr_samples) {
        F_TRY(hipMemcpy(e->d_stage, iq, nr_samples * sizeof(float2), hipMemcpyHostToDevice));
    }
    struct mfm_f32_block b;
    int rc = mfm_f32_process_device(e, reinterpret_cast<const float *>(e->d_stage), nr_samples, nullptr, &b);
    if (rc != MFM_OK) {
        return rc;
    }
    F_TRY(hipDeviceSynchronize());
    *nr_out = b.nr_out;
    if (b.nr_out > out_stride) {
        return MFM_E_INVAL;
    }
    if (b.nr_out) {
        const size_t C = b.nr_channels;
        if (pcm_f32) {
            F_TRY(hipMemcpy2D(pcm_f32, out_stride * 4, b.d_pcm_f32, b.stride * 4, b.nr_out * 4, C, hipMemcpyDeviceToHost));
        }
        if (pcm_i16) {
            F_TRY(hipMemcpy2D(pcm_i16, out_stride * 2, b.d_pcm_i16, b.stride * 2, b.nr_out * 2, C, hipMemcpyDeviceToHost));
        }
        if (iq_f32) {
            if (!b.d_iq_f32) {
                return MFM_E_STATE;
            }
            F_TRY(hipMemcpy2D(iq_f32, out_stride * 8, b.d_iq_f32, b.stride * 8, b.nr_out * 8, C, hipMemcpyDeviceToHost));
        }
    }
    return MFM_OK;
}

} /* extern "C" */
