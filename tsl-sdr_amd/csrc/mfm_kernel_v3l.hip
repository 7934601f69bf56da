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

#include <tuple>
#include <type_traits>

#include "mfm_kernel.h"
#include "mfm_numerics.h"

#include "mfm_v3_device.h"
#include "mfm_v3l_plan.h"

typedef unsigned int mfm_v2u __attribute__((ext_vector_type(2)));

#ifndef MFM3L_ONLY_KQ
#define MFM3L_ONLY_KQ 0 /* > 0: this translation unit holds the instances of that k-step count only (the Makefile compiles the
                           file once per count, side by side) */
#endif
#ifndef MFM3L_KNOCK
#define MFM3L_KNOCK 0 /* TIMING-ONLY builds (wrong results), tools/exp/variant_l.sh: bit 2 = no barrier behind an image, bit 3 = no
                         epilogue, bit 4 = no matrix phase (and with it no staging), bit 5 = no rotator-table loads, bit 6 = no
                         discriminator arithmetic, bit 7 = no PCM stores */
#endif
#ifndef MFM3L_LUT_ASM
#define MFM3L_LUT_ASM 1     /* the discriminator's table reads as asm statements (mfm_v3_device.h: mfm3_discriminate4<ASM_READS>), instances of one
                               row block per wave: 1 = the four reads in one batch behind the four divisions */
#endif
#ifndef MFM3L_LUT_ASM_RB2
#define MFM3L_LUT_ASM_RB2 2 /* ... of two row blocks per wave: 2 = each read behind its own division.  Both by measurement:
                               profiles/r06_ab_asm_reads.txt */
#endif
#ifndef MFM3L_PF
#define MFM3L_PF 4 /* k-steps of B fragments in flight ahead of the matrix instructions (2 where all 128 tap registers are in use) */
#endif

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

/* ---- compile-time loops: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) ---- */
template <typename F, int... Is>
static __device__ __forceinline__ void mfm3l_for_seq(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
static __device__ __forceinline__ void mfm3l_for(F &&f)
{
    mfm3l_for_seq(f, std::make_integer_sequence<int, N>{});
}

/* word K of a staging chunk (a template, so that the one-sample chunks of the shifted-copies form are never asked for a .y) */
template <int K, typename T>
static __device__ __forceinline__ uint32_t mfm3l_word(const T &c)
{
    if constexpr (K == 0) {
        return c.x;
    } else if constexpr (K == 1) {
        return c.y;
    } else if constexpr (K == 2) {
        return c.z;
    } else {
        return c.w;
    }
}

/* dst = v where dst IS sixteen bytes of staging registers (int16 input); nothing elsewhere (never reached there) */
template <typename T>
static __device__ __forceinline__ void mfm3l_put16(T &dst, const uint4 &v)
{
    if constexpr (std::is_same<T, uint4>::value) {
        dst = v;
    }
}
template <typename T>
static __device__ __forceinline__ uint4 mfm3l_get16(const T &src)
{
    if constexpr (std::is_same<T, uint4>::value) {
        return src;
    } else {
        return make_uint4(0, 0, 0, 0);
    }
}

/* the schedule of one phase as a compile-time object (only ever used in constant expressions: nothing of it exists on the device) */
template <int KQ, int NH, int NGC, int RB, bool IN8, int PF, bool SHIFTRD, bool DB, bool PI, bool CO, int NST, int STGM>
struct mfm3l_plan_of {
    static constexpr auto value = mfm3l_make_plan<KQ, NH, NGC, RB, IN8, PF, SHIFTRD, DB, PI, CO, NST, STGM>();
};

/* two accumulator sets (a column group's recombination in the gaps of the next group's matrix instructions) where the taps leave
 * the registers for them: 12 * RB more (8 * RB without a high tap plane) */
constexpr bool mfm3l_two_acc_sets(int KQ, int NH, int RB, bool SHIFT, bool IN8)
{
    return mfm3l_waves_per_simd(KQ, NH, RB, SHIFT, IN8) == 2 && 4 * RB * (KQ + NH) <= 80;
}

/* SPLIT: the decimation is not a multiple of 4 - a 4-sample staging chunk can straddle two rows of the image and is stored sample
 * by sample (such decimations are small - the image is a few hundred chunks -: instances of one and of four chunks per thread) */
template <int KQ, int NH, int NG, int NCH, bool IN8, int RB, bool SHIFT = false, bool SPLIT = false>
__global__ __launch_bounds__(MFM3_NT, mfm3l_waves_per_simd(KQ, NH, RB, SHIFT, IN8)) void mfm_channel_kernel_v3l(const mfm_launch_v3 L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    static_assert(NG == 1 || NG == 2 || NG == 4, "an image is a tile, half or a quarter of one");
    static_assert(RB == 1 || RB == 2, "row blocks (of 16 rows = 8 channels) per wave");
    constexpr int NSUB = 4 / NG;                     /* images per tile */
    constexpr uint32_t OPI = 16u * (uint32_t)NG;     /* outputs per image */
    static_assert(NH >= 0 && NH <= KQ, "planes held");
    /* two k-steps ahead where the taps take 112 registers or more and the fragments are pairs */
    /* (and with four waves per SIMD, which have each other to hide an LDS round trip and 128 registers each) */
    /* (two row blocks: a k-step is four matrix instructions and more per fragment pair - two k-steps are lead enough - and from 96
     * tap registers on the five fragment buffers of the deeper pipeline are not there) */
    constexpr int PF = ((4 * RB * (KQ + NH) >= (RB == 2 ? 96 : 112) && !IN8) || mfm3l_waves_per_simd(KQ, NH, RB, SHIFT, IN8) == 4) ? (MFM3L_PF < 2 ? MFM3L_PF : 2) : MFM3L_PF;
    constexpr int SLOTS = PF + 1;
    /* SHADOW: the next image's staging stores sit in the gaps of this image's matrix instructions, and the image after next is
     * requested from memory as soon as a chunk's registers are free (a whole phase to arrive) */
    constexpr bool SHADOW = !SHIFT;
    constexpr int STGM = SPLIT ? (IN8 ? MFM3L_STG_I8_S : MFM3L_STG_I16_S) : (IN8 ? MFM3L_STG_I8 : MFM3L_STG_I16);
    constexpr bool DB = mfm3l_two_acc_sets(KQ, NH, RB, SHIFT, IN8);
    constexpr int NACC = DB ? 2 : 1;
    /* the rotator entries of a tile travel in the staging registers of its last image (see the matrix phase) */
    constexpr bool ROT_IN_PRE = SHADOW && !IN8 && NCH >= 2 * RB;
    /* the epilogue's channels strictly one after the other (scheduling barriers) where two row blocks' taps leave it some 100
     * registers; where the taps are few the compiler may interleave the dependent chains of several channels */
#ifndef MFM3L_EPI_FREE_BELOW
#define MFM3L_EPI_FREE_BELOW 64 /* tap registers below which the barriers are dropped (-1.2 % at 1024 channels on 128-channel slices) */
#endif
    constexpr bool EPI_SERIAL = RB > 1 && 4 * RB * (KQ + NH) >= MFM3L_EPI_FREE_BELOW;
    struct one_sample { uint32_t x; };
    using chunk_t = typename std::conditional<SHIFT, one_sample, typename std::conditional<IN8, uint2, uint4>::type>::type;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kg = lane >> 4, n = lane & 15u;
    const uint32_t D = L.decim, row_bytes = L.row_bytes, rs = L.rs;
    const uint32_t plane_pitch = L.plane_pitch, buf_pitch = L.buf_pitch;
    const bool split_rows = SPLIT;
    const uint32_t in8_sh = (uint32_t)__builtin_amdgcn_readfirstlane(L.in8);
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(smem + L.lut_off);
    /* some channel of the set wants its filtered IQ (signalDebugFile, multifm/demod.c:75-81): a run-time switch here - a
     * wave-uniform branch around a 16-byte store per channel and tile - where mfm_kernel_v3.hip has instances */
    const bool want_iq = L.iq_dbg != nullptr;
    const bool pcm_sys = L.pcm_scope != 0u; /* launches of many channels write their PCM through (mfm3_store_pcm4) */

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
    /* SHADOW: the offsets stay in registers, and a chunk past the image's last one is the last one again (the same bytes to the
     * same place: no lane is ever masked off in the matrix phase) */
    /* SPLIT: every sample of the chunk has its own offset (2 m on, and rs - 2 D further from the sample that begins the next
     * row), two 16-bit offsets per register */
    uint32_t sta_r[NCH][SPLIT ? 2 : 1];
#pragma unroll
    for (int j = 0; j < NCH && !SHIFT; j++) {
        uint32_t q = tid + (uint32_t)j * MFM3_NT;
        if (SHADOW) {
            q = q < L.nstage4 ? q : L.nstage4 - 1u;
        }
        const uint32_t s0 = q * 4u;
        const uint32_t r0 = s0 / D, c0 = s0 % D;
        const uint32_t o0 = r0 * rs + 2u * c0, in_row = min(4u, D - c0), hop = rs - 2u * D;
        if constexpr (SPLIT) {
            const uint32_t o1 = o0 + 2u + (1u >= in_row ? hop : 0u), o2 = o0 + 4u + (2u >= in_row ? hop : 0u), o3 = o0 + 6u + (3u >= in_row ? hop : 0u);
            sta_r[j][0] = o0 | (o1 << 16);
            sta_r[j][SPLIT ? 1 : 0] = o2 | (o3 << 16);
        } else {
            sta_r[j][0] = o0;
        }
        /* (the tables of the synchronous staging - a workgroup's first image, the image in front of a chunk - by the plain chunk) */
        const uint32_t sq = (tid + (uint32_t)j * MFM3_NT) * 4u;
        sta16_s[j * MFM3_NT + tid] = (uint16_t)((sq / D) * rs + 2u * (sq % D));
        if (split_rows) {
            sta_in_row_s[j * MFM3_NT + tid] = (uint8_t)min(4u, D - sq % D);
        }
    }

    /* B fragments: column n of the image's first column group, k-step kq, lane group kg reads 16 bytes at element
     * 64 kq + 16 kg of the window that starts at row n: row n + e / row_bytes, byte e % row_bytes */
    /* (where the taps take 96 registers or more: two 16-bit offsets per register, taken apart by the address add's operand
     * selection - an image plane is less than 32 KB) */
    constexpr bool BPACK = 4 * RB * (KQ + NH) >= 96;
    uint32_t boff[BPACK ? (KQ + 1) / 2 : KQ];
#pragma unroll
    for (int kq = 0; kq < (BPACK ? (KQ + 1) / 2 : KQ); kq++) {
        boff[kq] = 0;
    }
#pragma unroll
    for (int kq = 0; kq < KQ; kq++) {
        const uint32_t step = (L.kperm[kq >> 2] >> (8 * (kq & 3))) & 0xffu; /* the kq-th k-step multiplied is this one of the window */
        const uint32_t e = 64u * step + 16u * kg;
        if constexpr (SHIFT) {
            /* column n = nc * a + c (nc = 8 / D copies): copy c, aligned offset 16 a; L.sp_pitch = bytes between two copies */
            const uint32_t nc = 8u / D;
            const uint32_t o = (n % nc) * L.sp_pitch + 16u * (n / nc) + e;
            boff[BPACK ? kq / 2 : kq] |= BPACK ? o << (16 * (kq & 1)) : o;
        } else {
            const uint32_t o = (n + e / row_bytes) * rs + e % row_bytes;
            boff[BPACK ? kq / 2 : kq] |= BPACK ? o << (16 * (kq & 1)) : o;
        }
    }
    /* at = base + the lane's offset of k-step KQI */
    auto frag_addr = [&](auto kq_tag, uint32_t &dst, uint32_t base) {
        constexpr int KQI = decltype(kq_tag)::value;
        (void)&boff; /* (named outside a dependent expression: the generic lambda captures it when it is defined) */
        if constexpr (!BPACK) {
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(dst) : "s"(base), "v"(boff[KQI]));
        } else if constexpr ((KQI & 1) == 0) {
            asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(dst) : "s"(base), "v"(boff[KQI / 2]));
        } else {
            asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(dst) : "s"(base), "v"(boff[KQI / 2]));
        }
    };

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
    /* 4 samples of an image, chunk q of it.  Only a readable address is needed: samples past n_avail feed only outputs >= n_new
     * (never stored) or zero-padded taps; an image never starts in front of the buffer (the output in front of a launch has
     * its first row there: L.hist). */
    auto stage_load_q = [&](int s_first, uint32_t q) -> chunk_t {
        int gs = s_first + 4 * (int)q;
        gs = gs < 0 ? 0 : gs;
        gs = gs > (int)L.x_last4 ? (int)L.x_last4 : gs;
        if constexpr (SHIFT) {
            return chunk_t{};
        } else if constexpr (IN8) {
            return *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(L.x) + ((uint32_t)gs << 1));
        } else {
            return *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(L.x) + ((uint32_t)gs << 2));
        }
    };
    /* chunk j of this thread in the matrix phase's numbering (a chunk past the image's last one is the last one again) */
    auto chunk_of = [&](int j) -> uint32_t {
        const uint32_t q = tid + (uint32_t)j * MFM3_NT;
        return q < L.nstage4 ? q : L.nstage4 - 1u;
    };
    auto stage_load = [&](int s_first, int j) -> chunk_t {
        if constexpr (SHIFT) {
            chunk_t r{};
            r.x = stage_load1(s_first, j);
            return r;
        } else {
            return stage_load_q(s_first, tid + (uint32_t)j * MFM3_NT);
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
    const uint32_t tp_w_addr = (uint32_t)(uintptr_t)(tp_s + (2u * kg) * MFM_V3L_TP + n);
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

    /* ---- registers of the matrix phases (mfm_v3l_plan.h has the schedule) ---- */
    mfm_v4i acc[NACC][RB][3];           /* hh, md, ll of the column group in flight (and, DB, of the one before) */
    mfm_v4i bh[SLOTS], bl[IN8 ? 1 : SLOTS]; /* B fragments, rotating */
    uint32_t tq[RB][4], fq[RB][2];      /* recombined sums / packed filtered samples of the group being finished */
    chunk_t pre[NCH];                   /* SHADOW: the image behind the one in the idle buffer, on its way from memory */
#pragma unroll
    for (int r = 0; r < RB; r++) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                acc[a][r][k] = mfm_v4i{ 0, 0, 0, 0 };
            }
        }
        tq[r][0] = tq[r][1] = tq[r][2] = tq[r][3] = fq[r][0] = fq[r][1] = 0;
    }

    /*
     * One image's matrix phase: NGC column groups of (RB x 16) rows x 16 columns x 64 * KQ elements, four byte-plane products
     * per k-step and row block (two with one sample plane), the B fragments used by every row block of the wave.  Everything in
     * it is a single-instruction asm statement in the order of the plan: the compiler allocates the registers and nothing else.
     *   MAIN: image H of the tile.  Its column groups' packed filtered samples (first Q14 rounding done; lane (kg, n) holds
     *   channels 2 kg, 2 kg + 1 of column n) go to the wave's transposition area; with two accumulator sets the last group is
     *   left to the next image of the tile (its recombination sits in that phase's gaps).  SHADOW: the chunks in `pre` go to
     *   the idle buffer in the gaps, and as soon as a chunk is stored its successor (the image that starts at output n2_out) is
     *   requested into the same registers.
     *   FRONT: the one-group image in front of a chunk; the samples stay in fq.
     */
    auto phase = [&](auto ngc_tag, auto h_tag, auto main_tag, uint32_t lds_h, uint32_t lds_other, int n2_out, auto &&behind_matrix) {
        constexpr int NGC = decltype(ngc_tag)::value, H = decltype(h_tag)::value;
        constexpr bool MAIN = decltype(main_tag)::value;
        constexpr bool PDB = DB && MAIN;
        constexpr bool PEND_IN = PDB && H > 0, CARRY_OUT = PDB && H + 1 < NSUB;
        constexpr int P0 = PDB ? (H * NG) & 1 : 0;
        constexpr int NST = (MAIN && SHADOW) ? NCH : 0;
        using plan_t = mfm3l_plan_of<KQ, NH, NGC, RB, IN8, PF, SHIFT, PDB, PEND_IN, CARRY_OUT, NST, STGM>;
        constexpr int NMF = mfm3l_nmf(KQ, NH, NGC, RB, IN8), NS = NGC * KQ;
        constexpr int PL = RB == 2 ? 24576 : 31744; /* mfm_v3l_plane_pitch: the low plane lies a constant behind the high one */

        const uint32_t lds_u = (uint32_t)__builtin_amdgcn_readfirstlane(lds_h);
        const uint32_t sto_u = (uint32_t)__builtin_amdgcn_readfirstlane(lds_other);
        uint32_t at = 0, at_l = 0, sa = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0, sam[4] = { 0, 0, 0, 0 };
        /* (named here so that every level of these nested generic lambdas captures them when it is defined) */
#define MFM3L_CAPTURES (void)&bl, (void)&bh, (void)&acc, (void)&tq, (void)&fq, (void)&pre, (void)&a_h, (void)&a_l, (void)&krow, (void)&boff, (void)&sta_r; (void)&at, (void)&at_l, (void)&sa, (void)&s0, (void)&s1, (void)&s2, (void)&s3, (void)&sam
        MFM3L_CAPTURES;

        auto filler = [&](auto idx_tag) {
            MFM3L_CAPTURES;
            constexpr int I = decltype(idx_tag)::value;
            constexpr mfm3l_fl F = plan_t::value.fl[I];
            if constexpr (F.kind == MFM3L_F_ADD || F.kind == MFM3L_F_ADDL) {
                constexpr int ST = F.a;
                const uint32_t gbase = lds_u + (uint32_t)(ST / KQ) * 16u * rs + (F.kind == MFM3L_F_ADDL ? plane_pitch : 0u);
                if constexpr (F.kind == MFM3L_F_ADD) {
                    frag_addr(std::integral_constant<int, ST % KQ>{}, at, gbase);
                } else {
                    frag_addr(std::integral_constant<int, ST % KQ>{}, at_l, gbase);
                }
            } else if constexpr (F.kind == MFM3L_F_RDH) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(bh[F.a % SLOTS]) : "v"(at) : "memory");
            } else if constexpr (F.kind == MFM3L_F_RDL) {
                if constexpr (SHIFT) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bl[IN8 ? 0 : F.a % SLOTS]) : "v"(at_l) : "memory");
                } else {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[IN8 ? 0 : F.a % SLOTS]) : "v"(at), "n"(PL) : "memory");
                }
            } else if constexpr (F.kind == MFM3L_F_LA) {
                /* t = (x << 8) + y over the group's accumulators: (hh, md) then (t, ll); without a high tap plane (md, ll); one
                 * sample plane: (hh, ll) */
                constexpr int A = !PDB ? 0 : F.c == MFM3L_G_PEND ? (P0 + 1) & 1 : (P0 + F.c) & 1; /* which accumulator set */
                constexpr int R = F.a, E = F.b, LV = F.d;
                constexpr int NLA = mfm3l_la_levels(IN8, NH);
                if constexpr (NLA == 2 && LV == 0) {
                    asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(tq[R][E]) : "v"(acc[A][R][0][E]), "v"(acc[A][R][1][E]));
                } else if constexpr (NLA == 2) {
                    asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(tq[R][E]) : "v"(acc[A][R][2][E]));
                } else if constexpr (IN8) {
                    asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(tq[R][E]) : "v"(acc[A][R][0][E]), "v"(acc[A][R][2][E]));
                } else {
                    asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(tq[R][E]) : "v"(acc[A][R][1][E]), "v"(acc[A][R][2][E]));
                }
            } else if constexpr (F.kind == MFM3L_F_SH0 || F.kind == MFM3L_F_SH1) {
                /* bits 29:14 (8-bit input: the format's shift) of the biased sums as (re | im << 16) */
                constexpr int A = !PDB ? 0 : F.c == MFM3L_G_PEND ? (P0 + 1) & 1 : (P0 + F.c) & 1; /* which accumulator set */
                constexpr int R = F.a, C = F.b, E = 2 * F.b + (F.kind == MFM3L_F_SH1 ? 1 : 0);
                constexpr bool DIRECT = mfm3l_la_levels(IN8, NH) == 0; /* one sample plane, no high tap plane: the sum is ll itself */
                if constexpr (F.kind == MFM3L_F_SH0) {
                    if constexpr (IN8 && DIRECT) {
                        asm volatile("v_lshrrev_b32_sdwa %0, %2, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(fq[R][C]) : "v"(acc[A][R][2][E]), "s"(in8_sh));
                    } else if constexpr (IN8) {
                        asm volatile("v_lshrrev_b32_sdwa %0, %2, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(fq[R][C]) : "v"(tq[R][E]), "s"(in8_sh));
                    } else {
                        asm volatile("v_lshrrev_b32_sdwa %0, 14, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(fq[R][C]) : "v"(tq[R][E]));
                    }
                } else {
                    if constexpr (IN8 && DIRECT) {
                        asm volatile("v_lshrrev_b32_sdwa %0, %2, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(fq[R][C]) : "v"(acc[A][R][2][E]), "s"(in8_sh));
                    } else if constexpr (IN8) {
                        asm volatile("v_lshrrev_b32_sdwa %0, %2, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(fq[R][C]) : "v"(tq[R][E]), "s"(in8_sh));
                    } else {
                        asm volatile("v_lshrrev_b32_sdwa %0, 14, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(fq[R][C]) : "v"(tq[R][E]));
                    }
                }
            } else if constexpr (F.kind == MFM3L_F_TPW) {
                if constexpr (MAIN) {
                    /* column group (H * NG + g) of the tile, row block R, channel 2 kg + C of it: [channel][output] */
                    constexpr int POS = F.c == MFM3L_G_PEND ? H * NG - 1 : H * NG + F.c;
                    static_assert(POS >= 0 && POS < 4, "a group left to the next image stays inside its tile");
                    constexpr int OFF = (POS * 16 + (F.a * 8 + F.b) * (int)MFM_V3L_TP) * 4;
                    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(tp_w_addr), "v"(fq[F.a][F.b]), "n"(OFF) : "memory");
                }
            } else if constexpr (F.kind == MFM3L_F_STG) {
                constexpr int J = F.a, OP = F.b;
                if constexpr (SPLIT) {
                    /* sample by sample: s0, s1 = the high (or only) plane's two dwords - sample m is half m & 1 of dword m >> 1 -,
                     * s2, s3 the low plane's; sam[m] = where sample m goes */
                    constexpr int NPREP = IN8 ? 2 : 6; /* operations in front of the address adds */
                    if constexpr (IN8 && OP == 0) {
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(s0) : "s"(L.in8_xor), "v"(mfm3l_word<0>(pre[J])));
                    } else if constexpr (IN8 && OP == 1) {
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(s1) : "s"(L.in8_xor), "v"(mfm3l_word<1>(pre[J])));
                    } else if constexpr (!IN8 && OP == 0) {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s0) : "v"(mfm3l_word<1>(pre[J])), "v"(mfm3l_word<0>(pre[J])), "s"(0x07050301u));
                    } else if constexpr (!IN8 && OP == 1) {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s1) : "v"(mfm3l_word<3>(pre[J])), "v"(mfm3l_word<2>(pre[J])), "s"(0x07050301u));
                    } else if constexpr (!IN8 && OP == 2) {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s2) : "v"(mfm3l_word<1>(pre[J])), "v"(mfm3l_word<0>(pre[J])), "s"(0x06040200u));
                    } else if constexpr (!IN8 && OP == 3) {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s3) : "v"(mfm3l_word<3>(pre[J])), "v"(mfm3l_word<2>(pre[J])), "s"(0x06040200u));
                    } else if constexpr (!IN8 && OP == 4) {
                        asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(s2));
                    } else if constexpr (!IN8 && OP == 5) {
                        asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(s3));
                    } else if constexpr (OP < NPREP + 4) {
                        constexpr int M = OP - NPREP;
                        if constexpr ((M & 1) == 0) {
                            asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(sam[M]) : "s"(sto_u), "v"(sta_r[J][M >> 1]));
                        } else {
                            asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(sam[M]) : "s"(sto_u), "v"(sta_r[J][M >> 1]));
                        }
                    } else {
                        constexpr int W = OP - NPREP - 4, M = W & 3; /* W < 4: the high (or only) plane, else the low one */
                        const uint32_t &src = W < 4 ? (M < 2 ? s0 : s1) : (M < 2 ? s2 : s3);
                        if constexpr (W < 4 && (M & 1) == 0) {
                            asm volatile("ds_write_b16 %0, %1" ::"v"(sam[M]), "v"(src) : "memory");
                        } else if constexpr (W < 4) {
                            asm volatile("ds_write_b16_d16_hi %0, %1" ::"v"(sam[M]), "v"(src) : "memory");
                        } else if constexpr ((M & 1) == 0) {
                            asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(sam[M]), "v"(src), "n"(PL) : "memory");
                        } else {
                            asm volatile("ds_write_b16_d16_hi %0, %1 offset:%2" ::"v"(sam[M]), "v"(src), "n"(PL) : "memory");
                        }
                    }
                } else if constexpr (OP == 0) {
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(sa) : "s"(sto_u), "v"(sta_r[J][0]));
                } else if constexpr (IN8) {
                    if constexpr (OP == 1) {
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(s0) : "s"(L.in8_xor), "v"(mfm3l_word<0>(pre[J])));
                    } else if constexpr (OP == 2) {
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(s1) : "s"(L.in8_xor), "v"(mfm3l_word<1>(pre[J])));
                    } else {
                        const mfm_v2u w = { s0, s1 };
                        asm volatile("ds_write_b64 %0, %1" ::"v"(sa), "v"(w) : "memory");
                    }
                } else {
                    /* dword = [lo0 hi0 lo1 hi1]: the high bytes of four int16 into one dword, the low bytes (- 128) into another */
                    if constexpr (OP == 1) {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s0) : "v"(mfm3l_word<1>(pre[J])), "v"(mfm3l_word<0>(pre[J])), "s"(0x07050301u));
                    } else if constexpr (OP == 2) {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s1) : "v"(mfm3l_word<3>(pre[J])), "v"(mfm3l_word<2>(pre[J])), "s"(0x07050301u));
                    } else if constexpr (OP == 3) {
                        const mfm_v2u w = { s0, s1 };
                        asm volatile("ds_write_b64 %0, %1" ::"v"(sa), "v"(w) : "memory");
                    } else if constexpr (OP == 4) {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s0) : "v"(mfm3l_word<1>(pre[J])), "v"(mfm3l_word<0>(pre[J])), "s"(0x06040200u));
                    } else if constexpr (OP == 5) {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s1) : "v"(mfm3l_word<3>(pre[J])), "v"(mfm3l_word<2>(pre[J])), "s"(0x06040200u));
                    } else if constexpr (OP == 6) {
                        asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(s0));
                    } else if constexpr (OP == 7) {
                        asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(s1));
                    } else {
                        const mfm_v2u w = { s0, s1 };
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(sa), "v"(w), "n"(PL) : "memory");
                    }
                }
            } else if constexpr (F.kind == MFM3L_F_NOP16) {
                /* matrix result -> vector instruction: 16 wait states cover a 16x16x64 matrix instruction */
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            }
        };
        /* the successor of a stored chunk: the same registers, a whole phase to arrive */
        auto reload = [&](auto m_tag) {
            MFM3L_CAPTURES;
            if constexpr (NST > 0) {
                constexpr int M = decltype(m_tag)::value;
                mfm3l_for<NST>([&](auto j_tag) {
                    MFM3L_CAPTURES;
                    constexpr int J = decltype(j_tag)::value;
                    if constexpr (plan_t::value.stg_done[J] == M) {
                        if constexpr (ROT_IN_PRE && H + 1 == NSUB && J < 2 * RB) {
                            /* the tile's last image: the registers of the first chunks carry the tile's rotator entries to the
                             * epilogue (four consecutive 4-byte entries per channel - a chunk's 16 bytes), requested here, half a
                             * phase and more ahead of their use; the chunk's successor is requested behind the last matrix
                             * instruction instead (the epilogue to arrive) */
#if MFM3L_KNOCK & 32
                            mfm3l_put16(pre[J], make_uint4(kb8[J / 2][J % 2], 16384u, 16384u, 16384u));
#else
                            mfm3l_put16(pre[J], *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(L.rot) + mfm3_opaque(kb8[J / 2][J % 2])));
#endif
                        } else {
                            pre[J] = stage_load_q(image_start(n2_out), chunk_of(J));
                        }
                    }
                });
            }
        };

        /* the first PF k-steps' fragments */
        mfm3l_for<(PF < NS ? PF : NS)>([&](auto st_tag) {
            MFM3L_CAPTURES;
            constexpr int ST = decltype(st_tag)::value;
            const uint32_t gbase = lds_u + (uint32_t)(ST / KQ) * 16u * rs;
            frag_addr(std::integral_constant<int, ST % KQ>{}, at, gbase);
            asm volatile("ds_read_b128 %0, %1" : "=v"(bh[ST % SLOTS]) : "v"(at) : "memory");
            if constexpr (!IN8) {
                if constexpr (SHIFT) {
                    const uint32_t gbase_l = gbase + plane_pitch;
                    frag_addr(std::integral_constant<int, ST % KQ>{}, at_l, gbase_l);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bl[IN8 ? 0 : ST % SLOTS]) : "v"(at_l) : "memory");
                } else {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[IN8 ? 0 : ST % SLOTS]) : "v"(at), "n"(PL) : "memory");
                }
            }
        });
        mfm3l_for<NMF>([&](auto m_tag) {
            MFM3L_CAPTURES;
            constexpr int M = decltype(m_tag)::value;
            constexpr mfm3l_mf X = plan_t::value.mf[M];
            constexpr int SL = X.step % SLOTS, A = PDB ? (P0 + X.g) & 1 : 0, R = X.r;
            if constexpr (X.wait != 0xff) {
                /* "at most `wait` LGKM operations issued behind this k-step's reads are still outstanding"; the operands tie
                 * the wait to the registers */
                if constexpr (IN8) {
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bh[SL]) : "n"(X.wait) : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[SL]), "+v"(bl[IN8 ? 0 : SL]) : "n"(X.wait) : "memory");
                }
            }
            /* the product: which tap plane, which sample plane, which accumulator */
            constexpr int AC = X.prod == MFM3L_P_HH ? 0 : X.prod == MFM3L_P_LL ? 2 : 1;
            constexpr bool TAP_H = X.prod == MFM3L_P_HH || X.prod == MFM3L_P_MDH;
            constexpr bool SMP_H = IN8 || X.prod == MFM3L_P_HH || X.prod == MFM3L_P_MD;
            const mfm_v4i &ta = TAP_H ? a_h[R][X.kq < NH ? X.kq : 0] : a_l[R][X.kq];
            const mfm_v4i &sb = SMP_H ? bh[SL] : bl[IN8 ? 0 : SL];
            if constexpr (X.init && AC == 2) {
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(acc[A][R][AC]) : "v"(ta), "v"(sb), "v"(krow[R]));
            } else if constexpr (X.init) {
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(acc[A][R][AC]) : "v"(ta), "v"(sb));
            } else {
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[A][R][AC]) : "v"(ta), "v"(sb));
            }
            mfm3l_for<plan_t::value.gap_hi[M] - plan_t::value.gap_lo[M]>([&](auto i_tag) {
                filler(std::integral_constant<int, plan_t::value.gap_lo[M] + decltype(i_tag)::value>{});
            });
            reload(m_tag);
        });
        /* behind the last matrix instruction, in front of what is left to finish (a group's recombination behind its 16 wait
         * states): the fragment registers are free from here on */
        /* (first what is left of the staging - instances with more staging operations than gaps -, and the successors of the
         * chunks that only now are through) */
        mfm3l_for<plan_t::value.tail_mid - plan_t::value.tail_lo>([&](auto i_tag) {
            filler(std::integral_constant<int, plan_t::value.tail_lo + decltype(i_tag)::value>{});
        });
        reload(std::integral_constant<int, NMF>{});
        behind_matrix();
        mfm3l_for<plan_t::value.tail_hi - plan_t::value.tail_mid>([&](auto i_tag) {
            filler(std::integral_constant<int, plan_t::value.tail_mid + decltype(i_tag)::value>{});
        });
#undef MFM3L_CAPTURES
    };

    /* what follows a tile in the workgroup's stream: the next tile of the chunk, or the first tile of the workgroup's next item
     * (false: nothing - the workgroup's last tile) */
    auto advance = [&](uint32_t &it, uint32_t &ch, uint32_t &sl, uint32_t &ti, uint32_t &te, bool &first) -> bool {
        ti += 1u;
        first = false;
        if (ti >= te) {
            it += gridDim.x;
            const bool v = mfm3_decode_item(L, it, &ch, &sl);
            ti = (uint32_t)(((uint64_t)ch * L.ntiles) / L.nchunks);
            te = (uint32_t)(((uint64_t)(ch + 1u) * L.ntiles) / L.nchunks);
            first = true;
            return v;
        }
        return true;
    };
    if constexpr (SHADOW) {
        /* the image behind the first one: on its way while the first tile is set up */
        int n1_out = (int)(tile * MFM_V3_OT + OPI);
        if (NSUB == 1) {
            uint32_t a_item = item, a_chunk = chunk, a_slice = slice, a_tile = tile, a_tend = tend;
            bool a_first = false;
            const bool a_valid = advance(a_item, a_chunk, a_slice, a_tile, a_tend, a_first);
            n1_out = (int)((a_valid ? a_tile : tile) * MFM_V3_OT);
        }
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            pre[j] = stage_load_q(image_start(n1_out), chunk_of(j));
        }
    }

    while (true) {
        /* this wave's row blocks: RB consecutive ones of the slice's 8 * RB; a row block past the last one recomputes the
         * last (its channels are past the end: nothing of it is stored) */
        const uint32_t rb0 = (slice * 8u + wave) * RB;
        const uint32_t first_out = tile * MFM_V3_OT;

        if (slice != slice_loaded) {
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
            {
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
                phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::false_type{},
                      (uint32_t)(uintptr_t)(smem + (cur ^ 1u) * buf_pitch), 0u, 0, [] {});
#pragma unroll
                for (int r = 0; r < RB; r++) {
                    uint32_t qw[2];
                    derotate2(fq[r], wrx[r], wry[r], qw);
                    hist[r][0] = qw[0];
                    hist[r][1] = qw[1];
                }
                __syncthreads(); /* the idle buffer is free again: the first image's successor goes there */
            }
        }

        /* ---- the tile: NSUB images, the next one staged into the other buffer while this one multiplies ---- */
        uint32_t n_item = item, n_chunk = chunk, n_slice = slice, n_tile = tile, n_tend = tend;
        bool n_first = false;
        const bool n_valid = advance(n_item, n_chunk, n_slice, n_tile, n_tend, n_first);
        const uint32_t t1 = n_valid ? n_tile : tile; /* the tile behind this one (a workgroup's last tile: itself - loads and
                                                        stores of the loop are unconditional) */
        uint32_t t2 = t1;                            /* ... and, whole-tile images only, the one behind that */
        if (SHADOW && NSUB == 1 && n_valid) {
            uint32_t a_item = n_item, a_chunk = n_chunk, a_slice = n_slice, a_tile = n_tile, a_tend = n_tend;
            bool a_first = false;
            const bool a_valid = advance(a_item, a_chunk, a_slice, a_tile, a_tend, a_first);
            t2 = a_valid ? a_tile : t1;
        }
        uint4 rva[RB][2];
        if (prio_matrix) {
            __builtin_amdgcn_s_setprio(1);
        } else {
            __builtin_amdgcn_s_setprio(0);
        }
        /* rotator entries of this tile, four consecutive ones per channel: requested behind the tile's last matrix instruction -
         * the B fragments' registers are free then - and needed behind the group's recombination and the barrier (at the top of
         * the epilogue, where two-row-block instances asked until round 6, their latency was 8 % of configs[4]'s launch:
         * profiles/r06_knockout_v3l.txt) */
        auto request_rotators = [&]() {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RB; r++) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
#if MFM3L_KNOCK & 32
                    rva[r][c] = make_uint4(kb8[r][c], 16384u, 16384u, 16384u);
#else
                    rva[r][c] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(L.rot) + mfm3_opaque(kb8[r][c]));
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        mfm3l_for<NSUB>([&](auto h_tag) {
            constexpr int H = decltype(h_tag)::value;
            int n2_here = 0; /* (set below) */
            auto behind = [&]() {
                if constexpr (H + 1 == NSUB) {
                    if constexpr (ROT_IN_PRE) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 2 * RB; j++) {
                            rva[j / 2][j % 2] = mfm3l_get16(pre[j]);
                            pre[j] = stage_load_q(image_start(n2_here), chunk_of(j));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        request_rotators();
                    }
                }
            };
            if constexpr (SHADOW) {
                /* the image behind the one in `pre` */
                const int n2_out = H + 2 < NSUB ? (int)(first_out + (uint32_t)(H + 2) * OPI)
                                                : NSUB == 1 ? (int)(t2 * MFM_V3_OT) : (int)(t1 * MFM_V3_OT + (uint32_t)(H + 2 - NSUB) * OPI);
                n2_here = n2_out;
#if !(MFM3L_KNOCK & 16)
                phase(std::integral_constant<int, NG>{}, h_tag, std::true_type{}, (uint32_t)(uintptr_t)(smem + cur * buf_pitch),
                      (uint32_t)(uintptr_t)(smem + (cur ^ 1u) * buf_pitch), n2_out, behind);
#endif
            } else {
                /* the image behind this one: the tile's next, or the first of the workgroup's next tile */
                const int next_out = H + 1 < NSUB ? (int)(first_out + (uint32_t)(H + 1) * OPI) : (int)(t1 * MFM_V3_OT);
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    pre[j] = stage_load(image_start(next_out), j);
                }
                __builtin_amdgcn_sched_barrier(MFM3_SCHED_ALL_BUT_VMEM);
                phase(std::integral_constant<int, NG>{}, h_tag, std::true_type{}, (uint32_t)(uintptr_t)(smem + cur * buf_pitch), 0u, 0, behind);
                /* the next image goes to the other buffer; after the barrier nobody reads the current one any more */
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    stage_store(cur ^ 1u, j, pre[j], L.nstage4);
                }
            }
            /* the stores of the phase are inline asm: the compiler does not know of them, the barrier needs them done */
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(MFM3L_KNOCK & 4)
            __syncthreads();
#endif
            __builtin_amdgcn_sched_barrier(0);
            cur ^= 1u;
        });

        __builtin_amdgcn_s_setprio(2); /* epilogue */
#if !(MFM3L_KNOCK & 8)
        {
            static_assert(MFM3_ROT4, "the long-filter kernel is written for 4-byte rotator entries");
            const uint32_t n_left = L.n_new - first_out; /* >= 1 */
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
#if MFM3L_KNOCK & 64
                    pcm[0] = s_re[0] + s_im[1], pcm[1] = s_re[1] + s_im[2], pcm[2] = s_re[2] + s_im[3], pcm[3] = s_re[3] + s_im[0];
#else
                    mfm3_discriminate4<(RB == 1) ? MFM3L_LUT_ASM : MFM3L_LUT_ASM_RB2>(s_re, s_im, lut_addr, pcm);
#endif
                    /* lane 0 of each row of 16 lanes gets lane 15's last sample: the next tile's history */
                    hist[r][c] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q[3][c], 0x121 /* row_ror:1 */, 0xf, 0xf, true);
                    if (EPI_SERIAL && c == 0) {
                        __builtin_amdgcn_sched_barrier(0); /* ... and one channel after the other */
                    }
                    if (n_left >= MFM_V3_OT) {
                        if (ch_ok[r][c]) {
                            uint2 w;
                            w.x = __builtin_amdgcn_perm((uint32_t)pcm[1], (uint32_t)pcm[0], 0x05040100u);
                            w.y = __builtin_amdgcn_perm((uint32_t)pcm[3], (uint32_t)pcm[2], 0x05040100u);
#if MFM3L_KNOCK & 128
                            asm volatile("" ::"v"(w.x), "v"(w.y));
#else
                            mfm3_store_pcm4(L.pcm, voff[r][c], w.x, w.y, pcm_sys);
#endif
                            if (want_iq) {
                                /* the filtered samples beside the PCM: multifm/demod.c:75-81 writes them to signalDebugFile */
                                *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(L.iq_dbg) + 2u * (size_t)voff[r][c]) =
                                    make_uint4(q[0][c], q[1][c], q[2][c], q[3][c]);
                            }
                        }
                    } else {
                        /* the last tile of the pass, partly filled */
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            if (ch_ok[r][c] && 4u * n + (uint32_t)g < n_left) {
                                *reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(L.pcm) + voff[r][c] + 2u * g) = (int16_t)pcm[g];
                                if (want_iq) {
                                    *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(L.iq_dbg) + 2u * (size_t)voff[r][c] + 4u * g) = q[g][c];
                                }
                            }
                        }
                    }
                }
                if (EPI_SERIAL) {
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
#endif

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
        /* 128-channel slices: quarter-tile images (large decimations: configs[4]'s 400), or - few k-steps, small decimations:
         * the 128-tap filters of 128 and more channels, multifm/receiver.c:195-244 builds as many as the configuration lists -
         * whole-tile images, whose two row blocks' transposition areas then fit LDS beside them */
        /* (a high tap plane brings a third accumulator per row block and a second recombination level: registers the fullest instances do not have) */
        return 8 * (KQ + NH) + (NH > 0 ? 16 : 0) <= 128 && NCH == 4 && (NG == 1 || (NG == 4 && KQ <= 4));
    }
    /* (eight staging chunks per thread: half-tile images of large decimations only - a whole-tile image that would need them
     * runs as two half-tile ones) */
    return NG != 1 && !(NCH == 8 && NG == 4) && (IN8 || 4 * (KQ + NH) + (NCH == 8 ? 32 : 16) + (NG == 2 ? 8 : 0) <= 160);
}

/* experiment builds (tools/exp/variant_l.sh): -DMFM3L_ONLY_INSTANCE=16,0,1,4,2,false (KQ, NH, NG, NCH, RB, SPLIT) compiles that instance only, for both input formats */
template <int KQ, int NH, int NG, int NCH, bool IN8, int RB, bool SPLIT>
constexpr bool mfm3l_wanted()
{
#ifdef MFM3L_ONLY_INSTANCE
    (void)IN8; /* (both input formats of the instance: the engine asks for every format at commit) */
    return std::make_tuple(KQ, NH, NG, NCH, RB, SPLIT) == std::make_tuple(MFM3L_ONLY_INSTANCE);
#else
    return true;
#endif
}

template <int KQ, int NH, int NG, int NCH, bool IN8, int RB, bool SPLIT>
static const void *mfm3l_instance_ptr()
{
    if constexpr (mfm3l_wanted<KQ, NH, NG, NCH, IN8, RB, SPLIT>() && mfm3l_fits<KQ, NH, NG, (NCH == 1 ? 4 : NCH), IN8, RB>() &&
                  !(SPLIT && (NCH == 8 || RB == 2 || NG != 4)) && (NCH != 1 || SPLIT)) {
        return reinterpret_cast<const void *>(&mfm_channel_kernel_v3l<KQ, NH, NG, NCH, IN8, RB, false, SPLIT>);
    } else {
        return nullptr;
    }
}

template <int KQ, int NH, int NG, int RB>
static const void *mfm3l_instance_fmt(const mfm_launch_v3 *L, uint32_t nch)
{
    const bool big = mfm_v3l_built_nch(nch) == 8u;
    if (L->split_rows) { /* decimations that are not multiples of 4 (small ones: one or four staging chunks, one row block per wave) */
        if (big) {
            return nullptr;
        }
        if (nch <= 1u) {
            return L->in8 ? mfm3l_instance_ptr<KQ, NH, NG, 1, true, RB, true>() : mfm3l_instance_ptr<KQ, NH, NG, 1, false, RB, true>();
        }
        return L->in8 ? mfm3l_instance_ptr<KQ, NH, NG, 4, true, RB, true>() : mfm3l_instance_ptr<KQ, NH, NG, 4, false, RB, true>();
    }
    if (L->in8) {
        return big ? mfm3l_instance_ptr<KQ, NH, NG, 8, true, RB, false>() : mfm3l_instance_ptr<KQ, NH, NG, 4, true, RB, false>();
    }
    return big ? mfm3l_instance_ptr<KQ, NH, NG, 8, false, RB, false>() : mfm3l_instance_ptr<KQ, NH, NG, 4, false, RB, false>();
}

template <int KQ, int NH>
static const void *mfm3l_instance_geo(const mfm_launch_v3 *L, uint32_t nch)
{
    /* decimations 1, 2, 4 on shifted copies of the image: whole-tile images, one row block per wave, k-step counts 4, 8, 16 */
#ifndef MFM3L_ONLY_INSTANCE
    if constexpr (KQ == 4 || KQ == 8 || KQ == 16) {
        if (L->shift) {
            if (L->ng != 4u || L->rb != 1u || mfm_v3l_built_nch(nch) != 4u) {
                return nullptr;
            }
            return L->in8 ? reinterpret_cast<const void *>(&mfm_channel_kernel_v3l<KQ, NH, 4, 4, true, 1, true>)
                          : reinterpret_cast<const void *>(&mfm_channel_kernel_v3l<KQ, NH, 4, 4, false, 1, true>);
        }
    }
#endif
    if (L->shift) {
        return nullptr;
    }
    if constexpr (KQ == 4) {
        /* four k-steps: 128-tap filters on 128-channel slices (the layouts of mfm_kernel_v3.hip take them on slices of 64) */
        return (L->rb == 2u && L->ng == 4u) ? mfm3l_instance_fmt<KQ, NH, 4, 2>(L, nch) : nullptr;
    } else {
        if (L->rb == 2u) {
            return L->ng == 1u ? mfm3l_instance_fmt<KQ, NH, 1, 2>(L, nch) : L->ng == 4u ? mfm3l_instance_fmt<KQ, NH, 4, 2>(L, nch) : nullptr;
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
    case 4: return mfm3l_instance_geo<KQ, (KQ < 4 ? KQ : 4)>(L, nch);
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
