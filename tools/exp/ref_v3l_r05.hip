/*
 * mfm_kernel_v3l.hip - the second-generation channel kernel for LONG filters (129..512 taps: the 256-tap low-passes of the
 * reference's etc/pocsag_1200khz_fs.json / etc/pocsag_narrow.json, the 512-tap one of etc/flex_25khz_lpf_3mhz.json) at any
 * decimation the first generation takes.
 *
 * Arithmetic: that of mfm_kernel_v3.hip and mfm_kernel_mfma.hip - filter/direct_fir.c:328-417 as byte-plane products on
 * v_mfma_i32_16x16x64_i8 with wrapping int32 accumulators, Q14 round, derotation by the tabulated rotator, Q14 round
 * (filter/direct_fir.c:151-172,406-413), s = q * conj(prev), fast_atan2f, PCM (multifm/fm_demod.c:53-79,
 * multifm/fast_atan2f.c:101-174).  The helpers are the same code (mfm_v3_device.h): bit for bit the oracle's results.
 *
 * Structure: the second generation's - 64-output tiles of a 64-channel slice, a lane ends up with FOUR CONSECUTIVE outputs
 * of its two channels (one 8-byte non-temporal PCM store per channel and tile, 16-byte rotator loads of 4-byte entries, the
 * four-at-a-time discriminator whose history is the previous output in the same lane), chunks of consecutive tiles per
 * workgroup with history and table position in registers, nothing carried between launches (mfm_launch_v3::hist, ::k_base).
 * What a long filter changes:
 *   - up to sixteen k-steps of taps per wave stay in registers (128 of them): two waves per SIMD, one workgroup per CU, so
 *     the B fragments are requested PF k-steps ahead by hand (inline ds_read_b128 + counted s_waitcnt), across the column
 *     groups of an image;
 *   - the LDS image is the first generation's: plain rows (one row = the D samples between two outputs, padded to 16 bytes
 *     with zero taps over the padding, odd multiple of 32 bytes stride), which takes any decimation and whose size does not
 *     depend on how outputs are dealt to lanes.  Column n of column group g of an image is its output 16 g + n; an image
 *     holds a whole tile (four groups) or, for large decimations (400 of configs[4]: a 64-row image of both planes is
 *     112 KB), half a tile;
 *   - the four consecutive outputs per lane come out of a wave-private transposition area in LDS: every column group's
 *     packed filtered samples are written as [channel][output], read back as 16 bytes per channel (10 LDS instructions per
 *     lane and tile, 1.25 per (channel, output)), no barrier - a wave reads what it wrote itself;
 *   - the output in front of a chunk (the discriminator's history) is recomputed from a one-group image staged into the
 *     idle buffer at the chunk's start (column 0 = that output).
 */
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mfm_kernel.h"
#include "mfm_numerics.h"

#include "mfm_v3_device.h"

#ifndef MFM3L_ONLY_KQ
#define MFM3L_ONLY_KQ 0 /* > 0: this translation unit holds the instances of that k-step count only (the Makefile compiles the
                           file once per count, side by side) */
#endif
#ifndef MFM3L_PF
#define MFM3L_PF 4 /* k-steps of B fragments in flight ahead of the matrix instructions (2 where all 128 tap registers are in use) */
#endif

/* "at most `younger` LDS requests issued after the ones that fill h (and l) are still outstanding": the wait in front of the
 * products of a B fragment that was requested by inline asm.  LDS returns in order and everything else that counts on
 * lgkmcnt only adds to it, so the wait can be too long, never too short.  The operands tie the wait to the registers. */
template <bool ONE_PLANE>
static __device__ __forceinline__ void mfm3l_wait_fragments(int younger, mfm_v4i &h, mfm_v4i &l)
{
#define MFM3L_WAIT_CASE(N_)                                                    \
    case N_:                                                                   \
        if (ONE_PLANE) {                                                       \
            asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(h)::"memory");    \
        } else {                                                               \
            asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(h), "+v"(l)::"memory"); \
        }                                                                      \
        break;
    switch (younger) {
        MFM3L_WAIT_CASE(1) MFM3L_WAIT_CASE(2) MFM3L_WAIT_CASE(3) MFM3L_WAIT_CASE(4) MFM3L_WAIT_CASE(5) MFM3L_WAIT_CASE(6)
        MFM3L_WAIT_CASE(7) MFM3L_WAIT_CASE(8)
    default:
        if (ONE_PLANE) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h)::"memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h), "+v"(l)::"memory");
        }
        break;
    }
#undef MFM3L_WAIT_CASE
}

/* KQ: k-steps of 64 elements held in registers (6 .. 16: mfm_v3l_built_kq); NG: column groups per staged image (4: a tile, 2: half a tile);
 * NCH: 4-sample staging chunks a thread owns per image (a surplus chunk is loaded and not stored); IN8: the input is 8-bit
 * IQ off the wire (one sample plane, two products per k-step, the first rounding's shift in L.in8: mfm_kernel_v3.hip has
 * the arithmetic).
 * NH: how many k-steps have a high-byte tap plane that is not all zero.  The engine multiplies the k-steps of a window in the
 * order L.kperm - those NH first - so "which planes" is a count, not a mask: a windowed low-pass of 129..512 taps at multifm's
 * gains has taps beyond one byte in a few middle k-steps only (configs[4]'s 512-tap filter in none), the planes that are
 * all zero are neither held (four registers each) nor multiplied (two matrix instructions each), and the matrix phase is
 * straight-line code: run-time tests of a mask between the matrix instructions cost the compiler's lane-mask arithmetic
 * and a full LDS wait per k-step (first measurement of this file, profiles/r05_long_filters.txt). */
/* SHIFT: decimations 1, 2, 4 (etc/multifm_file.json channelises without decimating) - a row of 2 D plane bytes is shorter than
 * the 16 bytes a B fragment reads, and the window of output o starts at plane byte 2 D o, aligned to 16 bytes only for every
 * (8 / D)-th output.  The image is kept 8 / D times, copy c shifted by 2 D c bytes (copy_c[j] = plane[j + 2 D c]), so that
 * column n = (8 / D) a + c reads copy c at the aligned offset 16 a: every fragment read is again one aligned ds_read_b128,
 * "lane register + group offset".  The image of a tile is a few hundred bytes per copy; the taps are the unpadded window. */
/* waves per SIMD an instance is built for: four (two workgroups per CU) where the taps are few - the four-k-step shifted-copies
 * form on 8-bit input (128 taps at decimation 1: etc/multifm_file.json is a cs8 capture) - else two */
constexpr int mfm3l_waves_per_simd(int KQ, int NH, int RB, bool SHIFT, bool IN8)
{
    return (SHIFT && IN8 && KQ == 4 && RB == 1 && NH <= 2) ? 4 : 2; /* (the int16 forms would spill at 128 registers) */
}

template <int KQ, int NH, int NG, int NCH, bool IN8, int RB, bool SHIFT = false>
__global__ __launch_bounds__(MFM3_NT, mfm3l_waves_per_simd(KQ, NH, RB, SHIFT, IN8)) void mfm_channel_kernel_v3l(const mfm_launch_v3 L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    static_assert(NG == 1 || NG == 2 || NG == 4, "an image is a tile, half or a quarter of one");
    static_assert(RB == 1 || RB == 2, "row blocks (of 16 rows = 8 channels) per wave");
    constexpr uint32_t NSUB = 4u / (uint32_t)NG; /* images per tile */
    constexpr uint32_t OPI = 16u * (uint32_t)NG; /* outputs per image */
    static_assert(NH >= 0 && NH <= KQ, "planes held");
    /* two k-steps ahead where the taps take 112 registers or more and the fragments are pairs */
    /* (and with four waves per SIMD, which have each other to hide an LDS round trip and 128 registers each) */
    constexpr int PF = ((4 * RB * (KQ + NH) >= 112 && !IN8) || mfm3l_waves_per_simd(KQ, NH, RB, SHIFT, IN8) == 4) ? (MFM3L_PF < 2 ? MFM3L_PF : 2) : MFM3L_PF;
    struct one_sample { uint32_t x; };
    using chunk_t = typename std::conditional<SHIFT, one_sample, typename std::conditional<IN8, uint2, uint4>::type>::type;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kg = lane >> 4, n = lane & 15u;
    const uint32_t D = L.decim, row_bytes = L.row_bytes, rs = L.rs;
    const uint32_t plane_pitch = L.plane_pitch, buf_pitch = L.buf_pitch;
    const bool split_rows = L.split_rows != 0u;
    const uint32_t in8_sh = (uint32_t)__builtin_amdgcn_readfirstlane(L.in8);
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(smem + L.lut_off);

    /* atan LUT, once per workgroup: {T[i], T[i+1]-T[i]} pairs as the engine holds them */
    static_assert(MFM3_NT == 512, "one table dword per thread");
    reinterpret_cast<uint32_t *>(smem + L.lut_off)[MFM3_LUT_SLOT(tid)] = reinterpret_cast<const uint32_t *>(L.lut)[tid];

    /* staging: this thread owns the 4-sample chunks q = tid + j * 512 of every image; where they go never changes:
     * samples 4q .. 4q + 3 of the image sit in row (4q) / D, two plane bytes per sample; bits 16..18: how many of the four
     * still belong to that row (fewer than 4 only when D is not a multiple of 4) */
    /* (16-bit offsets - LDS is what limits the image size at large decimations -, the in-row counts behind them as bytes: only
     * decimations that are not multiples of 4 read those) */
    uint16_t *sta16_s = reinterpret_cast<uint16_t *>(smem + L.sta_off);
    uint8_t *sta_in_row_s = smem + L.sta_off + NCH * MFM3_NT * 2u;
#pragma unroll
    for (int j = 0; j < NCH && !SHIFT; j++) {
        const uint32_t s0 = (tid + (uint32_t)j * MFM3_NT) * 4u;
        const uint32_t r0 = s0 / D, c0 = s0 % D;
        sta16_s[j * MFM3_NT + tid] = (uint16_t)(r0 * rs + 2u * c0);
        if (split_rows) {
            sta_in_row_s[j * MFM3_NT + tid] = (uint8_t)min(4u, D - c0);
        }
    }

    /* B fragments: column n of the image's first column group, k-step kq, lane group kg reads 16 bytes at element
     * 64 kq + 16 kg of the window that starts at row n: row n + e / row_bytes, byte e % row_bytes */
    uint32_t boff[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; kq++) {
        const uint32_t step = (L.kperm[kq >> 2] >> (8 * (kq & 3))) & 0xffu; /* the kq-th k-step multiplied is this one of the window */
        const uint32_t e = 64u * step + 16u * kg;
        if constexpr (SHIFT) {
            /* column n = nc * a + c (nc = 8 / D copies): copy c, aligned offset 16 a; L.sp_pitch = bytes between two copies */
            const uint32_t nc = 8u / D;
            boff[kq] = (n % nc) * L.sp_pitch + 16u * (n / nc) + e;
        } else {
            boff[kq] = (n + e / row_bytes) * rs + e % row_bytes;
        }
    }

    /* sample index (from L.x) of the first sample of the image that starts at output `out` of this launch */
    auto image_start = [&](int out) -> int { return (int)L.hist + out * (int)D; };
    /* SHIFT: a thread-chunk is ONE sample (the image of a tile is a few hundred of them): its two plane bytes go to every copy */
    auto stage_load1 = [&](int s_first, int j) -> uint32_t {
        int gs = s_first + (int)(tid + (uint32_t)j * MFM3_NT);
        gs = gs < 0 ? 0 : gs;
        gs = gs > (int)L.x_last4 ? (int)L.x_last4 : gs;
        if constexpr (IN8) {
            return (uint32_t)reinterpret_cast<const uint16_t *>(L.x)[gs];
        } else {
            return L.x[gs];
        }
    };
    auto stage_store1 = [&](uint32_t buf, int j, uint32_t v, uint32_t nsamp) {
        const uint32_t p = tid + (uint32_t)j * MFM3_NT; /* sample index within the image */
        if (p >= nsamp) {
            return;
        }
        uint32_t hi, lo = 0;
        if constexpr (IN8) {
            hi = (v ^ L.in8_xor) & 0xffffu;
        } else {
            hi = __builtin_amdgcn_perm(v, v, 0x0c0c0301u);           /* (I_hi, Q_hi) */
            lo = __builtin_amdgcn_perm(v, v, 0x0c0c0200u) ^ 0x8080u; /* (I_lo, Q_lo) - 128 */
        }
        uint8_t *img = smem + buf * buf_pitch;
        const uint32_t nc = 8u / D;
        for (uint32_t c = 0; c < nc; c++) {
            const int off = 2 * (int)p - (int)(2u * D * c); /* copy_c[j] = plane[j + 2 D c] */
            if (off >= 0) {
                *reinterpret_cast<uint16_t *>(img + c * L.sp_pitch + (uint32_t)off) = (uint16_t)hi;
                if (!IN8) {
                    *reinterpret_cast<uint16_t *>(img + plane_pitch + c * L.sp_pitch + (uint32_t)off) = (uint16_t)lo;
                }
            }
        }
    };
    auto stage_load = [&](int s_first, int j) -> chunk_t {
        if constexpr (SHIFT) {
            chunk_t r{};
            r.x = stage_load1(s_first, j);
            return r;
        } else {
        /* 4 samples of an image.  Only a readable address is needed: samples past n_avail feed only outputs >= n_new (never
         * stored) or zero-padded taps; an image never starts in front of the buffer (the output in front of a launch has
         * its first row there: L.hist). */
        int gs = s_first + 4 * (int)(tid + (uint32_t)j * MFM3_NT);
        gs = gs < 0 ? 0 : gs;
        gs = gs > (int)L.x_last4 ? (int)L.x_last4 : gs;
        if constexpr (IN8) {
            return *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(L.x) + ((uint32_t)gs << 1));
        } else {
            return *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(L.x) + ((uint32_t)gs << 2));
        }
        }
    };
    auto stage_store = [&](uint32_t buf, int j, const chunk_t &v, uint32_t nchunk) {
        if constexpr (SHIFT) {
            stage_store1(buf, j, v.x, nchunk);
            return;
        } else {
        if (tid + (uint32_t)j * MFM3_NT >= nchunk) {
            return;
        }
        uint8_t *base = smem + buf * buf_pitch + (uint32_t)sta16_s[j * MFM3_NT + tid]; /* own slot: no barrier needed */
        const uint32_t in_row = split_rows ? (uint32_t)sta_in_row_s[j * MFM3_NT + tid] : 4u, hop = rs - 2u * D;
        if constexpr (IN8) {
            const uint32_t m = L.in8_xor; /* 0x80808080: unsigned bytes -> int8 */
            const uint2 hi = make_uint2(v.x ^ m, v.y ^ m);
            if (!split_rows) {
                *reinterpret_cast<uint2 *>(base) = hi;
            } else {
                const uint32_t h[4] = { hi.x & 0xffffu, hi.x >> 16, hi.y & 0xffffu, hi.y >> 16 };
#pragma unroll
                for (uint32_t m4 = 0; m4 < 4; m4++) {
                    *reinterpret_cast<uint16_t *>(base + 2u * m4 + (m4 >= in_row ? hop : 0u)) = (uint16_t)h[m4];
                }
            }
        } else {
            /* dword = [lo0 hi0 lo1 hi1]: gather high / low bytes of four int16 into one dword */
            uint2 hi, lo;
            hi.x = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
            hi.y = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
            lo.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u) ^ 0x80808080u;
            lo.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u) ^ 0x80808080u;
            if (!split_rows) {
                *reinterpret_cast<uint2 *>(base) = hi;
                *reinterpret_cast<uint2 *>(base + plane_pitch) = lo;
            } else {
                /* the four samples may straddle two rows: sample by sample, two plane bytes each */
                const uint32_t h[4] = { hi.x & 0xffffu, hi.x >> 16, hi.y & 0xffffu, hi.y >> 16 };
                const uint32_t l[4] = { lo.x & 0xffffu, lo.x >> 16, lo.y & 0xffffu, lo.y >> 16 };
#pragma unroll
                for (uint32_t m = 0; m < 4; m++) {
                    uint8_t *p = base + 2u * m + (m >= in_row ? hop : 0u);
                    *reinterpret_cast<uint16_t *>(p) = (uint16_t)h[m];
                    *reinterpret_cast<uint16_t *>(p + plane_pitch) = (uint16_t)l[m];
                }
            }
        }
        }
    };

    /* the unconsumed samples at the end of this block are the head of the next one */
    if (blockIdx.x == 0) {
        for (uint32_t i = tid; i < L.tail_n; i += MFM3_NT) {
            if (IN8) {
                reinterpret_cast<uint16_t *>(L.tail_dst)[i] = reinterpret_cast<const uint16_t *>(L.x)[L.tail_src + i];
            } else {
                L.tail_dst[i] = L.x[L.tail_src + i];
            }
        }
    }

    uint32_t item = blockIdx.x, chunk, slice;
    if (!mfm3_decode_item(L, item, &chunk, &slice)) {
        return;
    }
    const uint32_t stamp_t0 = mfm3_stamp_lo(L.cyc ? __builtin_amdgcn_s_memtime() : 0ull), stamp_r0 = mfm3_stamp_lo(L.cyc ? __builtin_amdgcn_s_memrealtime() : 0ull);
    /* chunk j = tiles [j * ntiles / nchunks, (j + 1) * ntiles / nchunks): lengths differ by at most one tile */
    uint32_t tile = (uint32_t)(((uint64_t)chunk * L.ntiles) / L.nchunks);
    uint32_t tend = (uint32_t)(((uint64_t)(chunk + 1u) * L.ntiles) / L.nchunks);

    /* LDS behind the images: atan table, staging offsets, then: the waves' transposition areas ([8 * RB channels][TP]
     * dwords each), fold constants, exact-rotator tables */
    uint8_t *aux = smem + L.tp_off;
    uint32_t *tp_s = reinterpret_cast<uint32_t *>(aux) + wave * (8u * RB * MFM_V3L_TP);
    /* what this lane writes after a column group (row block r, channel 2 kg + c at + (8 r + c) * TP dwords, output 16 g + n
     * at + 16 g) and what it reads back for the epilogue (outputs 4n .. 4n + 3) */
    uint32_t *tp_w = tp_s + (2u * kg) * MFM_V3L_TP + n;
    const uint4 *tp_r = reinterpret_cast<const uint4 *>(tp_s + (2u * kg) * MFM_V3L_TP + 4u * n);
    uint8_t *per_wave = aux + 8u * 8u * RB * MFM_V3L_TP * 4u;
    uint2 *fold_s = reinterpret_cast<uint2 *>(per_wave) + (wave * 4u + kg) * (2u * RB);
    uint32_t *xq_s = reinterpret_cast<uint32_t *>(per_wave + 512u * RB) + (wave * 4u + kg) * (16u * RB);

    mfm_v4i a_h[RB][NH > 0 ? NH : 1], a_l[RB][KQ];
    mfm_v4i krow[RB]; /* 128 * sum(W) + 8192 (or the 8-bit form's constant) of the lane's rows */
    uint32_t slice_loaded = 0xffffffffu;

    /* ---- the workgroup's first image, staged synchronously into buffer 0 ---- */
    {
        chunk_t v[NCH];
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            v[j] = stage_load(image_start((int)(tile * MFM_V3_OT)), j);
        }
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            stage_store(0, j, v[j], L.nstage4);
        }
    }
    __syncthreads();

    const int prio_matrix = wave >= 4 ? 1 : 0; /* mfm_kernel_v3.hip: the younger half of a workgroup loses every arbitration */
    uint32_t cur = 0;
    bool first_of_chunk = true;

    /* per-lane state of the chunk: two channels per row block */
    uint32_t kb8[RB][2];   /* byte offset into the rotator table of the entry of (this tile's first output + 4n) */
    uint32_t voff[RB][2];  /* byte offset into pcm of (channel, this tile's first output + 4n) */
    uint32_t hist[RB][2];  /* lanes n = 0: filtered sample of the output in front of this tile */
    bool ch_ok[RB][2];
    bool w_exact[RB];      /* wave uniform: all eight channels of the row block have exact rotators */
#pragma unroll
    for (int r = 0; r < RB; r++) {
        kb8[r][0] = kb8[r][1] = voff[r][0] = voff[r][1] = hist[r][0] = hist[r][1] = 0;
        ch_ok[r][0] = ch_ok[r][1] = w_exact[r] = false;
    }

    /* One image's matrix phase: NGC column groups of (RB x 16) rows x 16 columns x 64 * KQ elements, four byte-plane products
     * per k-step and row block (two with one sample plane), B fragments requested PF k-steps ahead across the groups and
     * used by every row block of the wave.  sink(g, r, f) takes the packed filtered samples (first Q14 rounding done) of
     * column group g, row block r: lane (kg, n) holds channels 2 kg, 2 kg + 1 of column n. */
    auto matrix_phase = [&](auto ngc_tag, uint32_t lds_h, auto &&sink) {
        constexpr int NGC = decltype(ngc_tag)::value;
        constexpr int RPK = IN8 ? 1 : 2, SLOTS = PF + 1, NS = NGC * KQ;
        static_assert(PF * RPK <= 8, "the wait helper counts up to eight younger requests");
        mfm_v4i hh[RB], md[RB], ll[RB];
        mfm_v4i bh[SLOTS], bl[SLOTS];
        /* (the address arithmetic is inline asm as well: left to the compiler, "fragment offset + group offset (+ plane
         * pitch)" is loop invariant and gets hoisted - a register per (column group, k-step, plane), which the instances
         * with 128 tap registers do not have) */
        const uint32_t lds_u = (uint32_t)__builtin_amdgcn_readfirstlane(lds_h);
        auto request = [&](int st) { /* step st = column group st / KQ, k-step st % KQ */
            const uint32_t gbase = lds_u + (uint32_t)(st / KQ) * 16u * rs;
            uint32_t at;
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(at) : "s"(gbase), "v"(boff[st % KQ]));
            asm volatile("ds_read_b128 %0, %1" : "=v"(bh[st % SLOTS]) : "v"(at) : "memory");
            if constexpr (!IN8) {
                if constexpr (SHIFT) {
                    const uint32_t gbase_l = gbase + plane_pitch;
                    uint32_t at_l;
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(at_l) : "s"(gbase_l), "v"(boff[st % KQ]));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bl[st % SLOTS]) : "v"(at_l) : "memory");
                } else {
                    /* the low plane lies a constant behind the high one (mfm_v3l_plane_pitch): the same address register */
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[st % SLOTS]) : "v"(at), "n"(RB == 2 ? 24576 : 31744) : "memory");
                }
            }
        };
#pragma unroll
        for (int st = 0; st < PF && st < NS; st++) {
            request(st);
        }
#pragma unroll
        for (int st = 0; st < NS; st++) {
            const int gq = st / KQ, kq = st % KQ, cb = st % SLOTS;
            if (kq == 0) {
#pragma unroll
                for (int r = 0; r < RB; r++) {
                    hh[r] = mfm_v4i{ 0, 0, 0, 0 };
                    md[r] = mfm_v4i{ 0, 0, 0, 0 };
                    ll[r] = krow[r];
                }
            }
            if (st + PF < NS) {
                request(st + PF);
            }
            mfm3l_wait_fragments<IN8>((NS - 1 - st < PF ? NS - 1 - st : PF) * RPK, bh[cb], bl[cb]);
#pragma unroll
            for (int r = 0; r < RB; r++) {
                if constexpr (IN8) {
                    if (kq < NH) {
                        hh[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[r][kq < NH ? kq : 0], bh[cb], hh[r], 0, 0, 0);
                    }
                    ll[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[r][kq], bh[cb], ll[r], 0, 0, 0);
                } else {
                    if (kq < NH) { /* the k-steps whose high-byte tap plane is not all zero come first (L.kperm) */
                        hh[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[r][kq < NH ? kq : 0], bh[cb], hh[r], 0, 0, 0);
                        md[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[r][kq < NH ? kq : 0], bl[cb], md[r], 0, 0, 0);
                    }
                    ll[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[r][kq], bl[cb], ll[r], 0, 0, 0);
                    md[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[r][kq], bh[cb], md[r], 0, 0, 0);
                }
            }
            if (kq == KQ - 1) {
                /* MFMA -> VALU read hazard: 16 wait states cover a 16x16x64 MFMA (hipcc has been seen to leave it unpadded) */
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RB; r++) {
                    uint32_t a_re[2], a_im[2], f[2];
                    if constexpr (IN8) {
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a_re[c]) : "v"(hh[r][2 * c]), "v"(ll[r][2 * c]));
                            asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a_im[c]) : "v"(hh[r][2 * c + 1]), "v"(ll[r][2 * c + 1]));
                        }
                        mfm3_round_pack2_s(a_re, a_im, in8_sh, f);
                    } else {
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            a_re[c] = mfm3_combine(hh[r][2 * c], md[r][2 * c], ll[r][2 * c]);
                            a_im[c] = mfm3_combine(hh[r][2 * c + 1], md[r][2 * c + 1], ll[r][2 * c + 1]);
                        }
                        mfm3_round_pack2(a_re, a_im, f);
                    }
                    sink(gq, r, f);
                }
            }
        }
    };

    while (true) {
        /* this wave's row blocks: RB consecutive ones of the slice's 8 * RB; a row block past the last one recomputes the
         * last (its channels are past the end: nothing of it is stored) */
        const uint32_t rb0 = (slice * 8u + wave) * RB;
        const bool rb_valid = rb0 < L.nrb; /* wave uniform */
        const uint32_t first_out = tile * MFM_V3_OT;

        if (rb_valid && slice != slice_loaded) {
            /* A operand: 16 rows x (64 * KQ) elements, both byte planes, in fragment order, the k-steps in the order L.kperm */
#pragma unroll
            for (int r = 0; r < RB; r++) {
                const uint32_t rb = rb0 + r < L.nrb ? rb0 + r : L.nrb - 1u;
                const mfm_v4i *ap = reinterpret_cast<const mfm_v4i *>(L.afrag) + (size_t)rb * KQ * 2 * 64 + mfm3_opaque(lane);
#pragma unroll
                for (int kq = 0; kq < KQ; kq++) {
                    if (kq < NH) {
                        a_h[r][kq < NH ? kq : 0] = ap[(kq * 2 + 0) * 64];
                    }
                    a_l[r][kq] = ap[(kq * 2 + 1) * 64];
                }
                krow[r] = *reinterpret_cast<const mfm_v4i *>(L.krow + (size_t)rb * 16 + 4 * mfm3_opaque(kg));
            }
            /* settled here: they stay live across the whole chunk */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < RB; r++) {
#pragma unroll
                for (int kq = 0; kq < KQ; kq++) {
                    if (kq < NH) {
                        asm volatile("" : "+v"(a_h[r][kq < NH ? kq : 0]));
                    }
                    asm volatile("" : "+v"(a_l[r][kq]));
                }
                asm volatile("" : "+v"(krow[r]));
            }
            slice_loaded = slice;
        }

        if (first_of_chunk) {
            const bool has_front = first_out != 0 || L.hist != 0; /* uniform over the workgroup */
            uint32_t wrx[RB][2], wry[RB][2];
            if (rb_valid) {
                /* ---- chunk set-up: where the lane's channels stand in their rotator tables and in the output ---- */
#pragma unroll
                for (int r = 0; r < RB; r++) {
                    const uint32_t ch0 = (rb0 + r) * 8u + 2u * kg;
                    uint32_t kbg[2], cls[2], selq[2], sgq[2];
                    uint2 fog[2];
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const uint32_t chn = ch0 + c;
                        ch_ok[r][c] = chn < L.nchan;
                        const uint32_t chs = ch_ok[r][c] ? chn : 0u;
                        const uint32_t *ip = reinterpret_cast<const uint32_t *>(L.info) + (size_t)chs * 8;
                        const uint4 inf = *reinterpret_cast<const uint4 *>(ip);
                        const uint32_t lam_magic = ip[4];
                        const uint32_t mu = inf.z, lam = inf.w;
                        /* where output k_base + first_out stands in the channel's rotator table (wave uniform which way) */
                        const uint64_t kabs = L.k_base + first_out;
                        const uint32_t k0 = (kabs >> 32) == 0 ? mfm3_fold((uint32_t)kabs, 0u, mu, lam, lam_magic)
                                                              : mfm3_fold64(kabs, mu, lam, lam_magic);
                        kbg[c] = (inf.x + k0 + 4u * n) * MFM3_ES;
                        fog[c] = make_uint2((inf.x + mu + lam) * MFM3_ES, lam * MFM3_ES);
                        const uint32_t rcw = ch_ok[r][c] ? ip[7] : MFM_RC_IDENT;
                        cls[c] = rcw & 15u;
                        /* an exact rotator (mfm_kernel_v3.hip): output k0 + g is rotated by m = turns * (k0 + g) quarter turns -
                         * selector and sign word of the lane's output g, computed by lane n = g */
                        const uint32_t mq = ((rcw >> 4) * (k0 + n)) & 3u;
                        selq[c] = (mq & 1u) ? 0x01000302u : 0x03020100u;
                        sgq[c] = mq == 0u ? 0x00010001u : mq == 1u ? 0x0001ffffu : mq == 2u ? 0xffffffffu : 0xffff0001u;
                        voff[r][c] = (ip[6] * L.out_stride + first_out + 4u * n) * 2u; /* ip[6]: the row this channel's output goes to */
                        wrx[r][c] = wry[r][c] = 0;
                        if (has_front) {
                            /* rotator entry of the output in front (the entry in front of a period is not the period's last
                             * one: position mu is reached from mu - 1 the first time and from mu + lam - 1 ever after) */
                            const uint32_t kw = (k0 != mu || kabs == (uint64_t)mu) ? k0 - 1u : mu + lam - 1u;
#if MFM3_ROT4
                            const uint32_t rr = reinterpret_cast<const uint32_t *>(L.rot)[inf.x + kw];
                            wrx[r][c] = mfm3_rot_x(rr);
                            wry[r][c] = mfm3_rot_y(rr);
#else
                            const uint2 e = reinterpret_cast<const uint2 *>(L.rot)[inf.x + kw];
                            wrx[r][c] = e.x;
                            wry[r][c] = e.y;
#endif
                        }
                        hist[r][c] = 0; /* nothing in front: multifm/fm_demod.c:16-17,29, the last sample starts at zero */
                    }
                    /* rows are ordered by rotator class (the engine): a row block whose eight channels are all exact derotates
                     * with the permute-and-sign form and reads the table's first line only */
                    w_exact[r] = MFM3_WAVE_EXACT &&
                                 __builtin_amdgcn_ballot_w64(cls[0] == MFM_RC_GENERAL || cls[1] == MFM_RC_GENERAL) == 0;
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        kb8[r][c] = w_exact[r] ? 0u : kbg[c];
                        if (n == 0) {
                            fold_s[r * 2 + c] = fog[c]; /* only this wave reads it */
                        }
                    }
                    if (w_exact[r] && n < 4u) {
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            xq_s[r * 16 + c * 8 + n] = selq[c]; /* only this wave reads them */
                            xq_s[r * 16 + c * 8 + 4 + n] = sgq[c];
                        }
                    }
                }
            }
            if (has_front) {
                /* ---- the output in front of the chunk - for the first chunk of a launch the last output of the launch before,
                 *      from the L.hist samples kept in front of the first unconsumed one: a one-group image whose column 0 is
                 *      that output, staged into the idle buffer (the current one holds the chunk's first image) ---- */
                {
                    chunk_t v[NCH];
#pragma unroll
                    for (int j = 0; j < NCH; j++) {
                        v[j] = stage_load(image_start((int)first_out - 1), j);
                    }
#pragma unroll
                    for (int j = 0; j < NCH; j++) {
                        stage_store(cur ^ 1u, j, v[j], L.nstage_p);
                    }
                }
                __syncthreads();
                if (rb_valid) {
                    uint32_t fw[RB][2];
                    matrix_phase(std::integral_constant<int, 1>{}, (uint32_t)(uintptr_t)(smem + (cur ^ 1u) * buf_pitch),
                                 [&](int, int r, const uint32_t (&f)[2]) {
                                     fw[r][0] = f[0];
                                     fw[r][1] = f[1];
                                 });
#pragma unroll
                    for (int r = 0; r < RB; r++) {
                        uint32_t qw[2];
                        derotate2(fw[r], wrx[r], wry[r], qw);
                        hist[r][0] = qw[0];
                        hist[r][1] = qw[1];
                    }
                }
                __syncthreads(); /* the idle buffer is free again: the first image's successor goes there */
            }
        }

        /* ---- the tile: NSUB images, the next one staged into the other buffer while this one multiplies ---- */
        uint32_t n_item = item, n_chunk = chunk, n_slice = slice, n_tile = tile + 1u, n_tend = tend;
        bool n_first = false, n_valid = true;
        if (n_tile >= n_tend) {
            n_item = item + gridDim.x;
            n_valid = mfm3_decode_item(L, n_item, &n_chunk, &n_slice);
            n_tile = (uint32_t)(((uint64_t)n_chunk * L.ntiles) / L.nchunks);
            n_tend = (uint32_t)(((uint64_t)(n_chunk + 1u) * L.ntiles) / L.nchunks);
            n_first = true;
        }
        uint4 rva[RB][2];
        /* (not unrolled for part-tile images: several copies of the matrix phase cost the compiler 20-60 registers) */
#pragma unroll 1
        for (uint32_t h = 0; h < NSUB; h++) {
            /* the image behind this one: the tile's next, or the first of the workgroup's next tile (a workgroup's last image
             * re-reads its own samples into the idle buffer: loads and stores of the loop are unconditional) */
            const int next_out = h + 1u < NSUB ? (int)(first_out + (h + 1u) * OPI)
                                               : (int)((n_valid ? n_tile : tile) * MFM_V3_OT);
            chunk_t pre[NCH];
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                pre[j] = stage_load(image_start(next_out), j);
            }
            __builtin_amdgcn_sched_barrier(MFM3_SCHED_ALL_BUT_VMEM);

            if (prio_matrix) {
                __builtin_amdgcn_s_setprio(1);
            } else {
                __builtin_amdgcn_s_setprio(0);
            }
            if (rb_valid) {
                matrix_phase(std::integral_constant<int, NG>{}, (uint32_t)(uintptr_t)(smem + cur * buf_pitch),
                             [&](int gq, int r, const uint32_t (&f)[2]) {
                                 uint32_t *w = tp_w + (uint32_t)r * (8u * MFM_V3L_TP) + 16u * (h * (uint32_t)NG + (uint32_t)gq);
                                 w[0] = f[0];
                                 w[MFM_V3L_TP] = f[1];
                             });
                if (RB == 1 && h + 1u == NSUB) {
                    /* rotator entries of this tile, four consecutive ones per channel: requested behind the tile's last matrix
                     * phase, needed behind the staging stores and the barrier.  (In front of it - a matrix phase more to
                     * arrive - measured 1 % slower: profiles/r05_long_filters.txt.)  With two row blocks per wave the sixteen
                     * registers are not there across the staging stores - the compiler parked them in scratch, which waits for
                     * the loads on the spot - so those instances ask at the top of the epilogue. */
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        rva[0][c] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(L.rot) + mfm3_opaque(kb8[0][c]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            /* the next image goes to the other buffer; after the barrier nobody reads the current one any more */
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                stage_store(cur ^ 1u, j, pre[j], L.nstage4);
            }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0); /* the next image's loads stay behind this one's stores (their registers) */
            cur ^= 1u;
        }

        __builtin_amdgcn_s_setprio(2); /* epilogue */
        if (rb_valid) {
            static_assert(MFM3_ROT4, "the long-filter kernel is written for 4-byte rotator entries");
            const uint32_t n_left = L.n_new - first_out; /* >= 1 */
            if (RB > 1) {
#pragma unroll
                for (int r = 0; r < RB; r++) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        rva[r][c] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(L.rot) + mfm3_opaque(kb8[r][c]));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RB; r++) {
                uint32_t q[4][2];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    /* this lane's four consecutive outputs of channel 2 kg + c of row block r, as the column groups left them */
                    const uint4 fv = tp_r[(r * 8 + c) * (MFM_V3L_TP / 4u)];
                    const uint32_t f[4] = { fv.x, fv.y, fv.z, fv.w };
                    if (w_exact[r]) {
                        /* exact rotators: r14(f * rot) = f * j^m - swap the halves for odd m, then two signs */
                        const uint4 sel4 = *reinterpret_cast<const uint4 *>(xq_s + r * 16 + c * 8);
                        const uint4 sg4 = *reinterpret_cast<const uint4 *>(xq_s + r * 16 + c * 8 + 4);
                        const uint32_t sel[4] = { sel4.x, sel4.y, sel4.z, sel4.w }, sg[4] = { sg4.x, sg4.y, sg4.z, sg4.w };
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const uint32_t t = __builtin_amdgcn_perm(f[g], f[g], sel[g]);
                            asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(q[g][c]) : "v"(t), "v"(sg[g]));
                        }
                    } else {
                        /* derotation + second rounding, two outputs per call */
#pragma unroll
                        for (int h2 = 0; h2 < 2; h2++) {
                            const uint32_t r0 = h2 ? rva[r][c].z : rva[r][c].x, r1 = h2 ? rva[r][c].w : rva[r][c].y;
                            const uint32_t fin[2] = { f[2 * h2], f[2 * h2 + 1] };
                            const uint32_t rx[2] = { mfm3_rot_x(r0), mfm3_rot_x(r1) }, ry[2] = { mfm3_rot_y(r0), mfm3_rot_y(r1) };
                            uint32_t qo[2];
                            derotate2(fin, rx, ry, qo);
                            q[2 * h2][c] = qo[0];
                            q[2 * h2 + 1][c] = qo[1];
                        }
                    }
                    /* discriminator: previous output = the one before in the same lane; for the first the neighbouring lane's
                     * last, and for lane n = 0 the last output of the previous tile */
                    const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp((int)hist[r][c], (int)q[3][c], 0x111 /* row_shr:1 */,
                                                                              0xf, 0xf, false);
                    const uint32_t pp[4] = { p0, q[0][c], q[1][c], q[2][c] };
                    int s_re[4], s_im[4], pcm[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        mfm3_conj_mul(q[g][c], pp[g], &s_re[g], &s_im[g]);
                    }
                    mfm3_discriminate4(s_re, s_im, lut_addr, pcm);
                    /* lane 0 of each row of 16 lanes gets lane 15's last sample: the next tile's history */
                    hist[r][c] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q[3][c], 0x121 /* row_ror:1 */, 0xf, 0xf, true);
                    if (RB > 1 && c == 0) {
                        __builtin_amdgcn_sched_barrier(0); /* ... and one channel after the other */
                    }
                    if (n_left >= MFM_V3_OT) {
                        if (ch_ok[r][c]) {
                            uint2 w;
                            w.x = __builtin_amdgcn_perm((uint32_t)pcm[1], (uint32_t)pcm[0], 0x05040100u);
                            w.y = __builtin_amdgcn_perm((uint32_t)pcm[3], (uint32_t)pcm[2], 0x05040100u);
                            mfm3_store_pcm4(L.pcm, voff[r][c], w.x, w.y);
                        }
                    } else {
                        /* the last tile of the pass, partly filled */
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            if (ch_ok[r][c] && 4u * n + (uint32_t)g < n_left) {
                                *reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(L.pcm) + voff[r][c] + 2u * g) = (int16_t)pcm[g];
                            }
                        }
                    }
                }
                if (RB > 1) {
                    /* two row blocks' taps leave the epilogue some 100 registers: one row block after the other */
                    __builtin_amdgcn_sched_barrier(0);
                }
                /* next tile of the chunk: 64 outputs on */
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    if (!w_exact[r]) {
                        const uint2 fo = fold_s[r * 2 + c]; /* at or past table position mu + lam the position folds back by lam */
                        kb8[r][c] += MFM_V3_OT * MFM3_ES;
                        kb8[r][c] = kb8[r][c] >= fo.x + 4u * MFM3_ES * n ? kb8[r][c] - fo.y : kb8[r][c];
                    }
                    voff[r][c] += MFM_V3_OT * 2u;
                }
            }
        }

        if (!n_valid) {
            break;
        }
        item = n_item;
        chunk = n_chunk;
        slice = n_slice;
        tile = n_tile;
        tend = n_tend;
        first_of_chunk = n_first;
    }
    mfm3_stamp_end(L, stamp_t0, stamp_r0);
}

/* Not built: the instances that would need more than 256 registers - many k-steps with most high-byte planes held, int16
 * input and eight staging chunks in flight - and two row blocks per wave where their taps alone take more than 128.  No
 * instance may spill: a fragment register saved to scratch between its request and its wait would save what was in it
 * before the data arrived.  The engine asks (mfm_select_channel_kernel_v3) and runs what is not built on one row block per
 * wave, or on the first generation (tests/test_abi.py checks the spill counts of everything that IS built). */
template <int KQ, int NH, int NG, int NCH, bool IN8, int RB>
constexpr bool mfm3l_fits()
{
    if (RB == 2) {
        /* 128-channel slices: whole- and half-tile images of two row blocks' transposition areas do not fit LDS at the
         * decimations that want them; built for quarter-tile images */
        return NG == 1 && 8 * (KQ + NH) <= 128; /* (half-tile images with eight staging chunks per thread spill in the tile loop:
                                                   0.26 -> 0.36 ms at configs[4]'s share, profiles/r05_long_filters.txt) */
    }
    return NG != 1 && (IN8 || 4 * (KQ + NH) + (NCH == 8 ? 32 : 16) + (NG == 2 ? 8 : 0) <= 160);
}

template <int KQ, int NH, int NG, int NCH, bool IN8, int RB>
static const void *mfm3l_instance_ptr()
{
    if constexpr (mfm3l_fits<KQ, NH, NG, NCH, IN8, RB>()) {
        return reinterpret_cast<const void *>(&mfm_channel_kernel_v3l<KQ, NH, NG, NCH, IN8, RB>);
    } else {
        return nullptr;
    }
}

template <int KQ, int NH, int NG, int RB>
static const void *mfm3l_instance_fmt(const mfm_launch_v3 *L, uint32_t nch)
{
    const bool big = mfm_v3l_built_nch(nch) == 8u;
    if (L->in8) {
        return big ? mfm3l_instance_ptr<KQ, NH, NG, 8, true, RB>() : mfm3l_instance_ptr<KQ, NH, NG, 4, true, RB>();
    }
    return big ? mfm3l_instance_ptr<KQ, NH, NG, 8, false, RB>() : mfm3l_instance_ptr<KQ, NH, NG, 4, false, RB>();
}

template <int KQ, int NH>
static const void *mfm3l_instance_geo(const mfm_launch_v3 *L, uint32_t nch)
{
    /* decimations 1, 2, 4 on shifted copies of the image: whole-tile images, one row block per wave, k-step counts 4, 8, 16 */
    if constexpr (KQ == 4 || KQ == 8 || KQ == 16) {
        if (L->shift) {
            if (L->ng != 4u || L->rb != 1u || mfm_v3l_built_nch(nch) != 4u) {
                return nullptr;
            }
            return L->in8 ? reinterpret_cast<const void *>(&mfm_channel_kernel_v3l<KQ, NH, 4, 4, true, 1, true>)
                          : reinterpret_cast<const void *>(&mfm_channel_kernel_v3l<KQ, NH, 4, 4, false, 1, true>);
        }
    }
    if constexpr (KQ == 4) {
        return nullptr; /* (the other layouts of this file start at five k-steps) */
    } else {
    if (L->shift) {
        return nullptr;
    }
    if (L->rb == 2u) {
        return L->ng == 1u ? mfm3l_instance_fmt<KQ, NH, 1, 2>(L, nch) : nullptr;
    }
    return L->ng == 4u ? mfm3l_instance_fmt<KQ, NH, 4, 1>(L, nch) : L->ng == 2u ? mfm3l_instance_fmt<KQ, NH, 2, 1>(L, nch) : nullptr;
    }
}

/* the instance for a launch description (geometry fields only, all fixed at commit): L->kq k-steps (a built count), the
 * built count of held planes at or above L->nh */
template <int KQ>
static const void *mfm3l_instance(const mfm_launch_v3 *L, uint32_t nch)
{
    switch (mfm_v3l_built_nh((uint32_t)KQ, L->nh)) {
    case 0: return mfm3l_instance_geo<KQ, 0>(L, nch);
    case 2: return mfm3l_instance_geo<KQ, 2>(L, nch);
    case 4: return mfm3l_instance_geo<KQ, 4>(L, nch);
    case 8: return mfm3l_instance_geo<KQ, (KQ < 8 ? KQ : 8)>(L, nch);
    default: return mfm3l_instance_geo<KQ, KQ>(L, nch);
    }
}

#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 4
/* workgroups per CU the instance a launch description selects is built for */
extern "C" uint32_t mfm_v3l_wg_per_cu(const mfm_launch_v3 *L)
{
    return (L->shift && L->in8 && L->kq == 4u && L->rb == 1u && mfm_v3l_built_nh(4u, L->nh) <= 2u) ? 2u : 1u;
}
#endif

#define MFM3L_EXPORT(KQ_)                                                                  \
    extern "C" const void *mfm_v3l_instance_kq##KQ_(const mfm_launch_v3 *L, uint32_t nch)  \
    {                                                                                      \
        return mfm3l_instance<KQ_>(L, nch);                                                \
    }
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 4
MFM3L_EXPORT(4)
#endif
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 6
MFM3L_EXPORT(6)
#endif
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 8
MFM3L_EXPORT(8)
#endif
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 9
MFM3L_EXPORT(9)
#endif
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 10
MFM3L_EXPORT(10)
#endif
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 11
MFM3L_EXPORT(11)
#endif
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 12
MFM3L_EXPORT(12)
#endif
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 14
MFM3L_EXPORT(14)
#endif
#if MFM3L_ONLY_KQ == 0 || MFM3L_ONLY_KQ == 16
MFM3L_EXPORT(16)
#endif
