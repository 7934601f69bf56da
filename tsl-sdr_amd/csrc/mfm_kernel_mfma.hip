/*
 * mfm_kernel_mfma.hip - the multifm channel kernel with the FIR evaluated on the matrix cores,
 * bit-exactly.
 *
 * The bank of C complex-tap decimating FIRs (filter/direct_fir.c:328-417) is a dense contraction
 *
 *     acc[row][n] = sum_k W[row][k] * e[2*n*D + k]          (int16 x int16 -> wrapping int32)
 *
 * over the raw interleaved int16 stream e[] (re, im, re, im ...), with two rows per channel:
 * W[2c] = (cr0, -ci0, cr1, -ci1, ...) and W[2c+1] = (ci0, cr0, ci1, cr1, ...)  (filter/complex.h:40-46).
 * gfx950 has no int16 MFMA, and v_dot2_i32_i16 is a half-rate VALU op (measured 36 T lane-ops/s, see
 * tools/ubench_dot2.hip), so the products are split into bytes:
 *
 *     W = 256*Wh + Wl          Wh, Wl in [-128, 127]   (needs |taps| <= 32639, checked by the engine)
 *     e = 256*Eh + El + 128    Eh = e >> 8, El = (e & 255) - 128
 *
 *     sum W*e = 65536*sum Wh*Eh + 256*(sum Wh*El + sum Wl*Eh) + sum Wl*El + 128*sum W     (mod 2^32)
 *
 * i.e. four v_mfma_i32_16x16x64_i8 per 16 rows x 16 outputs x 64 elements, with int32 accumulators
 * that wrap exactly like the reference's int32 sums (tools/ubench_mfma_i8.hip checks wrap-around, the
 * k-slot pairing and the C/D layout on hardware).  The last term is a per-row constant.
 *
 * Geometry.  A workgroup = 8 waves; wave w owns GEMM rows 16w..16w+15 = channels 8w..8w+7 of the
 * workgroup's 64-channel slice, its taps (A operand) stay in registers for the whole launch.  All
 * waves share one LDS image of the input tile, stored as two byte planes (Eh, El) in rows of 2*D
 * bytes with an odd 16-byte row stride, so every lane's B operand is one ds_read_b128.  One loop
 * iteration of a wave covers 32 output columns as two 16-column MFMA groups (interleaved so no
 * accumulator is reused back to back); column 0 is the output before the first new one, recomputed,
 * which makes the discriminator's one-sample history always "the lane to the left" (DPP row_shr).
 * In the C/D layout a lane holds re/im of 2 channels for one column, so the epilogue - Q14 round,
 * derotation by the tabulated rotator, fast_atan2f discriminator, PCM store (the same exact scalar
 * arithmetic as the v_dot2 kernel, mfm_numerics.h) - handles four (channel, output) pairs per lane
 * per iteration.  The small register footprint (<= 128 VGPRs) is what lets four waves share a SIMD
 * so that one wave's matrix work overlaps another's VALU epilogue.
 */
#include <hip/hip_runtime.h>

#ifndef MFM_M_NONTEMPORAL
#define MFM_M_NONTEMPORAL 0 /* 1: A/B builds - the 2-byte PCM stores carry the non-temporal hint.  What helps the second generation's
                               8-byte stores (mfm_kernel_v3.hip) is wrong here: 0.543 -> 0.861 ms at decimation 25, 0.282 -> 0.515 ms
                               at the configs[4] share (profiles/r04_traffic_1024ch.txt) - a hinted 2-byte store is a fabric write
                               of its own */
#endif
#include "mfm_kernel.h"
#include "mfm_numerics.h"

typedef int mfm_v4i __attribute__((ext_vector_type(4)));

#define MFM_M_NT (MFM_MFMA_NW * 64)

#define MFM_M_NEW 31 /* new outputs per 32-column iteration */
#ifndef MFM_RES_PF
#define MFM_RES_PF 4 /* resident long-filter instances: k-steps of B fragments in flight ahead of the matrix instructions */
#endif
#ifndef MFM_M_RES_SPLIT_ALL
#define MFM_M_RES_SPLIT_ALL 0 /* 1: A/B builds - every resident instance carries the per-sample staging path of decimations that
                               * are not multiples of 4, not only the twelve-k-step ones: 6-7 % slower at configs[4]'s geometry
                               * (decimation 400, eight staging chunks per thread), 1.5 % at 512 taps / decimation 96 */
#endif
#ifndef MFM_RES_EARLY
#define MFM_RES_EARLY 1 /* resident instances, single-iteration tiles: next tile's samples requested in front of the matrix phase */
#endif
/* sched_barrier mask: ALU | VALU | SALU | MFMA | DS | DS-read | DS-write may cross, vector memory may not */
#define MFM_SCHED_ALL_BUT_VMEM 0x38F

/* (hh << 16) + (md << 8) + ll: the recombined sum, K + 8192 riding in through ll.  Two v_lshl_add_u32; left to
 * itself the compiler canonicalises the expression into two shifts and a three-operand add. */
static __device__ __forceinline__ uint32_t mfm_combine(int hh, int md, int ll)
{
    uint32_t t, a;
    asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(hh), "v"(md));
    asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a) : "v"(t), "v"(ll));
    return a;
}

/*
 * Twice "bits 29:14 of re_b and of im_b as (re | im << 16)" (round_q30_q15 + int16 truncation of biased sums): two
 * sub-dword shifts per pair - v_lshrrev_b32 with dst_sel WORD_0 / WORD_1 writes the 16 result bits straight into its
 * half of the destination - instead of shift, shift, merge.  A VALU write with dst_sel needs one wait state before the
 * register is read again (the second shift preserves, i.e. reads, the other half); the two pairs are interleaved so
 * that an independent instruction sits in between, and the block ends with the wait state for whoever reads p[1]
 * next.  (The compiler does not pad hazards around inline asm.)
 */
static __device__ __forceinline__ void mfm_round_pack2(const uint32_t re_b[2], const uint32_t im_b[2], uint32_t p[2])
{
    uint32_t p0, p1;
    asm("v_lshrrev_b32_sdwa %0, 14, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %1, 14, %3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %0, 14, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %1, 14, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(p0), "=&v"(p1)
        : "v"(re_b[0]), "v"(re_b[1]), "v"(im_b[0]), "v"(im_b[1]));
    p[0] = p0;
    p[1] = p1;
}

/* the same with the shift in an SGPR (8-bit input: 7 for the RTL-SDR scaling, 14 for cs8 / cu8) */
static __device__ __forceinline__ void mfm_round_pack2_s(const uint32_t re_b[2], const uint32_t im_b[2], uint32_t sh, uint32_t p[2])
{
    uint32_t p0, p1;
    asm("v_lshrrev_b32_sdwa %0, %6, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %1, %6, %3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %0, %6, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %1, %6, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(p0), "=&v"(p1)
        : "v"(re_b[0]), "v"(re_b[1]), "v"(im_b[0]), "v"(im_b[1]), "s"(sh));
    p[0] = p0;
    p[1] = p1;
}

/*
 * o = f * r + 8192 for a packed complex f and the two packed rotator operands: two VOP3P v_dot2_i32_i16 with the
 * bias as an SGPR operand.  The builtin compiles to v_dot2c_i32_i16, which needs its accumulator preloaded by a
 * v_mov.  The compiler does not pad hazards around inline asm, so the 3 wait states a DOT result needs before
 * another VALU instruction reads it are part of the block.
 */
static __device__ __forceinline__ void mfm_rotate_biased(uint32_t f, uint2 r, uint32_t *o_re, uint32_t *o_im)
{
    asm("v_dot2_i32_i16 %0, %2, %3, %5\n\tv_dot2_i32_i16 %1, %2, %4, %5\n\ts_nop 2"
        : "=&v"(*o_re), "=&v"(*o_im)
        : "v"(f), "v"(r.x), "v"(r.y), "s"(8192));
}

/* s = q * conj(p) (multifm/fm_demod.c:55-64), wrapping int32: s_re by dot2, the two cross products by
 * v_mad_i32_i16 with op_sel; same hazard padding as above (two instructions + s_nop 0 after the DOT) */
static __device__ __forceinline__ void mfm_conj_mul(uint32_t q, uint32_t p, int *s_re, int *s_im)
{
    int u, t;
    asm("v_dot2_i32_i16 %0, %3, %4, 0\n\t"
        "v_mad_i32_i16 %1, %3, %4, 0 op_sel:[1,0,0,0]\n\t"
        "v_mad_i32_i16 %2, %3, %4, 0 op_sel:[0,1,0,0]\n\t"
        "s_nop 0"
        : "=&v"(*s_re), "=&v"(u), "=&v"(t)
        : "v"(q), "v"(p));
    *s_im = (int)((uint32_t)u - (uint32_t)t); /* q_im*p_re - q_re*p_im */
}

/* "at most `younger` LDS requests issued after the ones that fill h (and l) are still outstanding": the wait in front of the
 * products of a B fragment that was requested by inline asm.  The operands tie the wait to the registers, so the
 * products cannot be scheduled in front of it. */
template <bool ONE_PLANE>
static __device__ __forceinline__ void mfm_wait_fragments(int younger, mfm_v4i &h, mfm_v4i &l)
{
#define MFM_WAIT_CASE(N_)                                                      \
    case N_:                                                                   \
        if (ONE_PLANE) {                                                       \
            asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(h)::"memory");    \
        } else {                                                               \
            asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(h), "+v"(l)::"memory"); \
        }                                                                      \
        break;
    switch (younger) {
        MFM_WAIT_CASE(1) MFM_WAIT_CASE(2) MFM_WAIT_CASE(3) MFM_WAIT_CASE(4) MFM_WAIT_CASE(5) MFM_WAIT_CASE(6) MFM_WAIT_CASE(7)
        MFM_WAIT_CASE(8) MFM_WAIT_CASE(9) MFM_WAIT_CASE(10) MFM_WAIT_CASE(11) MFM_WAIT_CASE(12) MFM_WAIT_CASE(13)
        MFM_WAIT_CASE(14) MFM_WAIT_CASE(15)
    default:
        if (ONE_PLANE) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h)::"memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h), "+v"(l)::"memory");
        }
        break;
    }
#undef MFM_WAIT_CASE
}

/* Launder a value the compiler would otherwise use to hoist address arithmetic of rarely executed code (slice
 * change, first / last tile of a pass) out of the tile loop: those 64-bit addresses then sit in VGPRs for the whole
 * loop and push hot values into scratch, and every scratch reload is a VMEM access that costs a vmcnt(0). */
static __device__ __forceinline__ uint32_t mfm_opaque(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}

/* decode a persistent-loop item into (tile, slice): XCD-aware, the slices of one tile run back to back
 * on one XCD (see mfm_kernel.hip) */
static __device__ __forceinline__ bool mfm_decode_item(const mfm_launch_mfma &L, uint32_t item, uint32_t *tile,
                                                       uint32_t *slice)
{
    const uint32_t xcd = item & 7u, seq = item >> 3;
    *tile = (seq / L.nslices) * 8u + xcd;
    *slice = seq % L.nslices;
    return item < L.nitems && *tile < L.ntiles;
}

/* KQ = k-steps of 64 elements (32 complex taps); FIXP: planes at a fixed pitch; NCH = 16-byte staging chunks a thread
 * owns per tile = ceil(samples per tile / 4 / 512).  A compile-time count: a chunk nobody needs would still be loaded
 * (loads in this loop are unconditional), and with 62 outputs x 96 samples that was a fourth chunk per thread, 8 KB of
 * somebody else's tile per tile - a third on top of the input traffic.
 * KC = 1: the taps (A operand, KQ k-steps) stay in registers for the whole launch.  KC > 1 (filters of 129..512 taps):
 * KQ = 4 and the A operand is re-read from L2 in KC chunks of four k-steps in every iteration.
 * NIT = iterations (31 new outputs each) per tile: 2, or 1 when a 62-output tile does not fit LDS (large decimations).
 * AHM >= 0: L.ah_mask as a compile-time constant (no branches between the MFMAs of a k-step); -1: read at run time.
 * IN8: the input is 8-bit IQ off the wire (two bytes per sample): one sample plane, two products per k-step, the first
 * rounding's shift in L.in8 (mfm_kernel_v3.hip has the arithmetic).  A staging chunk stays 4 samples - an 8-byte load.
 *
 * Waves per SIMD.  Instances that keep up to four (int16) or eight (8-bit) k-steps of taps in registers are built for 128
 * vector registers, i.e. two workgroups per CU.  The RESIDENT long-filter instances (KC = 1 with KQ = 8, 12 or 16: filters of
 * 129..512 taps, all of whose taps stay in registers - up to 64 + 64 of them) are built for 256 registers and one workgroup
 * per CU: what a long filter loses in occupancy it more than gets back by not re-reading 16..32 KB of taps per wave and
 * iteration from L2. */
constexpr int mfm_m_waves_per_simd(int KQ, int KC, bool IN8)
{
    return (KC == 1 && (KQ >= 12 || (KQ >= 8 && !IN8))) ? 2 : 4;
}

template <int KQ, bool DBG_IQ, bool FIXP, int NCH, int KC, int AHM, int NIT, bool IN8>
__global__ __launch_bounds__(MFM_M_NT, mfm_m_waves_per_simd(KQ, KC, IN8)) void mfm_channel_kernel_mfma(const mfm_launch_mfma L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kg = lane >> 4, n = lane & 15u;
    /* row_bytes: plane bytes of one LDS row = 2 * D rounded up to 16 (decimations that are not multiples of 8 pad
     * their rows; the engine lays zero taps over the padding, so what the pad bytes hold does not matter) */
    const uint32_t D = L.decim, row_bytes = L.row_bytes, rs = L.rs;
    const uint32_t ah_mask = AHM >= 0 ? (uint32_t)AHM : (uint32_t)__builtin_amdgcn_readfirstlane(L.ah_mask);
    const uint32_t nchunk = L.nstage >> 2; /* 16-byte chunks (4 samples) per tile */
    const uint32_t in8_sh = (uint32_t)__builtin_amdgcn_readfirstlane(L.in8);
    /* the resident long-filter instances (two waves per SIMD, 256 registers): the twelve-k-step ones take decimations that
     * are not multiples of 4 too (etc/pocsag_rtlsdr.json's 25 with the 256-tap low-pass of etc/pocsag_1200khz_fs.json); the
     * eight- and sixteen-step ones are built without the per-sample store path and the shuffles of the straddling load */
    constexpr bool RESIDENT = mfm_m_waves_per_simd(KQ, KC, IN8) == 2;
    const bool split_rows = (RESIDENT && KQ != 12 && !MFM_M_RES_SPLIT_ALL) ? false : L.split_rows != 0u;
    /* one staging buffer = H plane + L plane; with FIXP the distances are compile-time constants and end up in the
     * offset field of the LDS instructions instead of costing a v_add each (ds_read has no SGPR offset) */
    const uint32_t plane_dist = FIXP ? MFM_M_PLANE_DIST : L.plane_bytes;
    const uint32_t buf_bytes = 2u * plane_dist;
    const float *lut_t = reinterpret_cast<const float *>(smem + L.lut_off), *lut_d = lut_t + 256;

    /* atan LUT, once per workgroup (its LDS region is never restaged): global {T[i], T[i+1]-T[i]} pairs are
     * split into T[256] followed by dT[256] so that a look-up is two ds_read_b32 into the halves of register pairs */
    {
        uint32_t *lut_s = reinterpret_cast<uint32_t *>(smem + L.lut_off);
        const uint32_t *lut_g = reinterpret_cast<const uint32_t *>(L.lut);
        for (uint32_t i = tid; i < 512; i += MFM_M_NT) {
            lut_s[(i >> 1) + ((i & 1u) << 8)] = lut_g[i];
        }
    }

    /* rotator constants of every channel {rot_base, -, mu, lam, lam_magic, kb, -, -}: each tile needs them to place
     * its first column in the rotator tables.  Read from global memory they were the only vector-memory loads at
     * the head of a tile, and waiting for them (in-order vmcnt) also waited for the previous tile's PCM stores. */
    const uint32_t *tbl = L.tbl_off ? reinterpret_cast<const uint32_t *>(smem + L.tbl_off) : nullptr;
    if (L.tbl_off) {
        uint32_t *tbl_s = reinterpret_cast<uint32_t *>(smem + L.tbl_off);
        const uint32_t *info_g = reinterpret_cast<const uint32_t *>(L.info);
        for (uint32_t i = tid; i < L.nchan * 8u; i += MFM_M_NT) {
            tbl_s[i] = ((i & 7u) == 5u) ? L.st_in[i >> 3].kb : info_g[i];
        }
    }

    /* per-lane LDS byte offset of the B fragment of k-step kq for column n of the first group */
    if (KC > 1) {
        /* B-fragment offsets of all 4 * KC k-steps: too many for registers, a division each to recompute -> LDS */
        uint16_t *bof_w = reinterpret_cast<uint16_t *>(smem + L.bof_off);
        for (uint32_t i = tid; i < 64u * 16u; i += MFM_M_NT) {
            const uint32_t ln = i >> 4, kqi = i & 15u;
            const uint32_t e = 64u * kqi + 16u * (ln >> 4);
            bof_w[i] = (uint16_t)(((ln & 15u) + e / row_bytes) * rs + e % row_bytes);
        }
    }
    const uint16_t *bof_s = reinterpret_cast<const uint16_t *>(smem + L.bof_off) + lane * 16u;
    uint32_t boff[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; kq++) {
        const uint32_t e = 64u * kq + 16u * kg;
        boff[kq] = (n + e / row_bytes) * rs + e % row_bytes;
    }

    /* staging: this thread owns chunks q = tid + j * MFM_M_NT, j = 0 .. NCH-1 of every tile; their
     * place in the LDS image never changes */
    /* where this thread's chunks go in the LDS image never changes; the offsets cost a division by the row length,
     * so they are computed once - and parked in LDS rather than in VGPRs, which are all taken while the matrix
     * phase runs (a spilled offset would come back through scratch, i.e. through vmcnt) */
    uint32_t *sta_s = reinterpret_cast<uint32_t *>(smem + L.sta_off);
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        /* samples 4q .. 4q+3 of the tile: row (4q) / D, 2 plane bytes per sample; bits 16..18: how many of the four
         * samples still belong to that row (fewer than 4 only when D is not a multiple of 4) */
        const uint32_t s0 = (tid + (uint32_t)j * MFM_M_NT) * 4u;
        const uint32_t r0 = s0 / D, c0 = s0 % D;
        sta_s[j * MFM_M_NT + tid] = (r0 * rs + 2u * c0) | (min(4u, D - c0) << 16);
    }

    auto stage_load = [&](uint32_t tile, int j) -> uint4 {
        /* 4 samples of tile `tile`.  No bounds masking is needed, only a readable address: samples before
         * the stream start feed nothing but column 0 of tile 0 (replaced by the carried sample), samples
         * past n_avail feed only columns >= n_new (never stored) or zero-padded taps.  Sample indices fit
         * 31 bits (the engine caps a block at 2^30 samples). */
        const int g0 = (int)(tile * L.ot * D) + (4 * (int)(tid + (uint32_t)j * MFM_M_NT) - (int)D);
        int gs = g0 < 0 ? 0 : g0;
        gs = gs > (int)L.x_last4 ? (int)L.x_last4 : gs;
        /* uniform base + 32-bit byte offset: one VGPR of address instead of a 64-bit pair */
        if (IN8) {
            const uint2 w2 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(L.x) + ((uint32_t)gs << 1));
            uint64_t w = (uint64_t)w2.x | ((uint64_t)w2.y << 32);
            const int sh8 = -g0;
            if (split_rows && sh8 > 0 && sh8 < 4) {
                w <<= 16 * sh8; /* the chunk that straddles the stream start keeps its real samples in place */
            }
            return make_uint4((uint32_t)w, (uint32_t)(w >> 32), 0u, 0u);
        }
        uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(L.x) + ((uint32_t)gs << 2));
        if (split_rows) {
            /* chunks start at any sample here: the one that straddles the stream start (g0 = -1 .. -3) holds real
             * samples behind the missing ones and must keep them in place (the clamp above moved them) */
            const int sh = -g0;
            if (sh > 0 && sh < 4) {
                v = sh == 1 ? make_uint4(0, v.x, v.y, v.z) : sh == 2 ? make_uint4(0, 0, v.x, v.y) : make_uint4(0, 0, 0, v.x);
            }
        }
        return v;
    };
    auto stage_store = [&](uint32_t buf, int j, const uint4 &v) {
        if (IN8) {
            if (tid + (uint32_t)j * MFM_M_NT < nchunk) {
                const uint32_t m = L.in8_xor; /* 0x80808080: unsigned bytes -> int8 */
                const uint2 hi = make_uint2(v.x ^ m, v.y ^ m);
                const uint32_t st = sta_s[j * MFM_M_NT + tid];
                uint8_t *base = smem + buf * buf_bytes + (st & 0xffffu);
                if (!split_rows) {
                    *reinterpret_cast<uint2 *>(base) = hi;
                } else {
                    const uint32_t in_row = st >> 16, hop = rs - 2u * D;
                    const uint32_t h[4] = { hi.x & 0xffffu, hi.x >> 16, hi.y & 0xffffu, hi.y >> 16 };
#pragma unroll
                    for (uint32_t m4 = 0; m4 < 4; m4++) {
                        *reinterpret_cast<uint16_t *>(base + 2u * m4 + (m4 >= in_row ? hop : 0u)) = (uint16_t)h[m4];
                    }
                }
            }
        } else if (tid + (uint32_t)j * MFM_M_NT < nchunk) {
            /* dword = [lo0 hi0 lo1 hi1]: gather high / low bytes of four int16 into one dword */
            uint2 hi, lo;
            hi.x = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
            hi.y = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
            lo.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u) ^ 0x80808080u;
            lo.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u) ^ 0x80808080u;
            const uint32_t st = sta_s[j * MFM_M_NT + tid];
            uint8_t *base = smem + buf * buf_bytes + (st & 0xffffu); /* own slot: no barrier needed */
            if (!split_rows) {
                *reinterpret_cast<uint2 *>(base) = hi;
                *reinterpret_cast<uint2 *>(base + plane_dist) = lo;
            } else {
                /* the four samples may straddle two rows: sample by sample, two plane bytes each */
                const uint32_t in_row = st >> 16, hop = rs - 2u * D;
                const uint32_t h[4] = { hi.x & 0xffffu, hi.x >> 16, hi.y & 0xffffu, hi.y >> 16 };
                const uint32_t l[4] = { lo.x & 0xffffu, lo.x >> 16, lo.y & 0xffffu, lo.y >> 16 };
#pragma unroll
                for (uint32_t m = 0; m < 4; m++) {
                    uint8_t *p = base + 2u * m + (m >= in_row ? hop : 0u);
                    *reinterpret_cast<uint16_t *>(p) = (uint16_t)h[m];
                    *reinterpret_cast<uint16_t *>(p + plane_dist) = (uint16_t)l[m];
                }
            }
        }
    };
    /* where column 0 of tile `tile` sits in this lane's two channels' rotator tables */
    auto rot_offsets = [&](uint32_t tile, uint32_t ch0, bool valid, uint32_t koff[2]) {
        const int rel_first = (int)(tile * L.ot) - 1;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t chn = ch0 + c;
            const uint32_t chs = (valid && chn < L.nchan) ? chn : 0u;
            uint4 inf;
            uint32_t lam_magic, kb;
            if (tbl) {
                const uint4 *tp = reinterpret_cast<const uint4 *>(tbl + chs * 8u);
                inf = tp[0];
                const uint2 hi = *reinterpret_cast<const uint2 *>(tp + 1);
                lam_magic = hi.x;
                kb = hi.y;
            } else {
                const uint32_t *ip = reinterpret_cast<const uint32_t *>(L.info) + (size_t)mfm_opaque(chs) * 8;
                inf = *reinterpret_cast<const uint4 *>(ip);
                lam_magic = ip[4];
                /* volatile: otherwise the compiler merges this load with the LDS one above into a flat_load through a
                 * selected pointer, and a flat load makes every tile wait for vmcnt(0) before it can place its rotator
                 * entries (no measurable difference in the end, but the tile prologue is free of vector memory now) */
                kb = *reinterpret_cast<const volatile uint32_t *>(&L.st_in[mfm_opaque(chs)].kb);
            }
            const uint32_t mu = inf.z, lam = inf.w;
            int k = (int)kb + rel_first; /* >= -1; entry -1 of every table is a readable dummy */
            if (k >= (int)mu) {
                const uint32_t x = (uint32_t)k - mu;
                uint32_t m = x - __umulhi(x, lam_magic) * lam;
                m = (m >= lam) ? m - lam : m;
                k = (int)(mu + m);
            }
            /* every table runs 128 entries past mu + lam, so a tile never wraps once its first column is
             * folded; the byte offset fits 32 bits (engine checks) */
            koff[c] = (inf.x + (uint32_t)k + n) * 8u;
        }
    };

    mfm_v4i a_h[KQ], a_l[KQ];
    mfm_v4i krow = { 0, 0, 0, 0 };
    uint32_t slice_loaded = 0xffffffffu;

    /* the unconsumed samples at the end of this block are the head of the next one: carried over here instead of
     * by a separate copy behind the kernel (one stream operation less per block) */
    if (blockIdx.x == 0) {
        for (uint32_t i = tid; i < L.tail_n; i += MFM_M_NT) {
            if (IN8) {
                reinterpret_cast<uint16_t *>(L.tail_dst)[i] = reinterpret_cast<const uint16_t *>(L.x)[L.tail_src + i];
            } else {
                L.tail_dst[i] = L.x[L.tail_src + i];
            }
        }
    }

    /* ---- first tile of this workgroup: staged synchronously into buffer 0 ---- */
    uint32_t item = blockIdx.x, tile, slice;
    bool have = mfm_decode_item(L, item, &tile, &slice);
    if (have) {
        uint4 v[NCH];
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            v[j] = stage_load(tile, j);
        }
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            stage_store(0, j, v[j]);
        }
    }
    __syncthreads();
    uint32_t k_off[2] = { 0, 0 };
    if (have) {
        rot_offsets(tile, (slice * MFM_MFMA_NW + wave) * 8u + 2u * kg, slice * MFM_MFMA_NW + wave < L.nrb, k_off);
    }

    /* Vector memory in this loop is straight-line code: every load and store below is issued on every path
     * (row blocks past the end are clamped, outputs that must not be written go to a dump slot behind the output
     * buffer, staging runs even when there is no next tile).  s_waitcnt vmcnt is one in-order counter; with
     * loads or stores under branches the compiler cannot count what is in flight and falls back to vmcnt(0) in
     * front of every reuse of a register - which put the full store round trip on the critical path several
     * times per iteration.  The rotator entries of iteration i+1 are requested right after the epilogue of
     * iteration i, ahead of its PCM stores, so waiting for them never waits for those stores. */
    auto rot_load = [&](const uint32_t koff[2], uint32_t it, uint2 out[2][2]) {
        const uint8_t *rot_it = reinterpret_cast<const uint8_t *>(L.rot) + (size_t)it * MFM_M_NEW * 8u;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            out[0][c] = *reinterpret_cast<const uint2 *>(rot_it + koff[c]);
            out[1][c] = *reinterpret_cast<const uint2 *>(rot_it + koff[c] + 16u * 8u);
        }
    };
    uint2 rv[2][2];
    rot_load(k_off, 0, rv);

    uint32_t cur = 0;
    while (have) {
        /* the tile after this one (persistent loop, stride = grid) is staged while this one computes */
        uint32_t tile_n, slice_n;
        const uint32_t item_n = item + gridDim.x;
        const bool have_n = mfm_decode_item(L, item_n, &tile_n, &slice_n);

        const uint32_t rb = slice * MFM_MFMA_NW + wave; /* this wave's block of 16 rows = 8 channels */
        const bool rb_valid = rb < L.nrb;
        const uint32_t rbc = rb_valid ? rb : L.nrb - 1u; /* a wave past the last row block recomputes it, stores nothing */
        const uint32_t ch0 = rbc * 8u + 2u * kg;       /* this lane's channels: ch0 (regs 0,1), ch0+1 (regs 2,3) */
        const int rel_first = (int)(tile * L.ot) - 1;  /* output index (this pass) of column 0 */
        const bool last_tile = (tile + 1u) * L.ot >= L.n_new; /* holds output n_new - 1 */

        if (slice != slice_loaded) {
            /* A operand: 16 rows x (64*KQ) elements, both byte planes, already in fragment order */
            if (KC == 1) {
                /* L.kq k-steps are laid out per row block; this instance multiplies the first KQ of them (the rest, if
                 * any, is the zero padding a filter is rounded up by) */
                const mfm_v4i *ap = reinterpret_cast<const mfm_v4i *>(L.afrag) + (size_t)rbc * L.kq * 2 * 64 + mfm_opaque(lane);
#pragma unroll
                for (int kq = 0; kq < KQ; kq++) {
                    a_h[kq] = ap[(kq * 2 + 0) * 64];
                    a_l[kq] = ap[(kq * 2 + 1) * 64];
                }
            }
            /* 128 * sum_k W[row][k] + 8192 for rows 4kg..4kg+3 */
            krow = *reinterpret_cast<const mfm_v4i *>(L.krow + (size_t)rbc * 16 + 4 * mfm_opaque(kg));
            /* settle these loads now: they stay live across the whole tile loop, and without this the
             * compiler waits vmcnt(0) at their first use in every iteration */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (KC == 1) {
#pragma unroll
                for (int kq = 0; kq < KQ; kq++) {
                    asm volatile("" : "+v"(a_h[kq]), "+v"(a_l[kq]));
                }
            }
            asm volatile("" : "+v"(krow));
            slice_loaded = slice;
        }

        uint32_t k_off_n[2] = { k_off[0], k_off[1] };
        if (have_n) {
            const uint32_t rb_n = slice_n * MFM_MFMA_NW + wave;
            rot_offsets(tile_n, (rb_n < L.nrb ? rb_n : L.nrb - 1u) * 8u + 2u * kg, true, k_off_n);
        }

        const uint8_t *plane_h = smem + cur * buf_bytes, *plane_l = plane_h + plane_dist;

        /* first tile of the pass: its column 0 is the previous pass's last filtered sample */
        const bool use_carry = (tile == 0) && (n == 0);
        uint32_t carry[2] = { 0, 0 };
        if (tile == 0) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t chn = mfm_opaque(ch0) + c;
                carry[c] = L.st_in[chn < L.nchan ? chn : 0u].carry_q;
            }
        }

        /* The whole next tile is requested now and written to LDS at the end of this tile: the loads stay in flight
         * for the full tile instead of one iteration (a workgroup's last tile re-reads its own samples and stages them
         * into the idle buffer). */
        uint4 pre[NCH];
        if (KC == 1 && (NIT == 2 || (RESIDENT && MFM_RES_EARLY))) { /* (the resident instances have the registers with one iteration too) */
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                pre[j] = stage_load(have_n ? tile_n : tile, j);
            }
            __builtin_amdgcn_sched_barrier(MFM_SCHED_ALL_BUT_VMEM); /* keep them ahead of the matrix work */
        }

#pragma unroll
        for (uint32_t it = 0; it < NIT; it++) {
            /* a wave in its epilogue goes ahead of waves in their matrix phase (mfm_kernel_v3.hip has the reasoning) */
            __builtin_amdgcn_s_setprio(0);
            uint32_t q[2][2];
            /* The two 16-column groups of an iteration one after the other: three accumulators (hh, md, ll) and one
             * pair of B fragments are live at a time instead of six and four - the registers that buys go into the
             * whole-tile prefetch above and into fetching the next k-step's fragments while this one multiplies.
             * (md is hit twice per k-step; tools/ubench_mfma_dep.hip: hh, md, ll, md runs at the full MFMA rate.) */
            /* recombine, r14, derotate, r14 for one column group: lane (kg, n) holds channels ch0, ch0+1 of column
             * 16 gq + n */
            auto finish_group = [&](int gq, const mfm_v4i &hh, const mfm_v4i &md, const mfm_v4i &ll) {
                uint32_t a_re[2], a_im[2], f[2], o_re[2], o_im[2];
                if (IN8) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a_re[c]) : "v"(hh[2 * c]), "v"(ll[2 * c]));
                        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a_im[c]) : "v"(hh[2 * c + 1]), "v"(ll[2 * c + 1]));
                    }
                    mfm_round_pack2_s(a_re, a_im, in8_sh, f);
                } else {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    /* a + 8192 (mod 2^32); r14(a) truncated to int16 is bits 29:14 (filter/complex.h:30-34) */
                    a_re[c] = mfm_combine(hh[2 * c], md[2 * c], ll[2 * c]);
                    a_im[c] = mfm_combine(hh[2 * c + 1], md[2 * c + 1], ll[2 * c + 1]);
                }
                mfm_round_pack2(a_re, a_im, f);
                }
                /* filter/direct_fir.c:406-413: o = f * rot, then r14 again (bias folded into the dot2) */
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    mfm_rotate_biased(f[c], rv[gq][c], &o_re[c], &o_im[c]);
                }
                mfm_round_pack2(o_re, o_im, q[gq]);
            };
            if (KC > 1) {
                /* ---- long filters: A operand streamed.  Both column groups accumulate while a chunk of taps is in
                 *      registers, so every chunk is read once per iteration (8 KB per wave, L2 hits) ---- */
                mfm_v4i hh2[2], md2[2], ll2[2];
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    hh2[gq] = mfm_v4i{ 0, 0, 0, 0 };
                    md2[gq] = mfm_v4i{ 0, 0, 0, 0 };
                    ll2[gq] = krow;
                }
                const uint32_t ibase = it * MFM_M_NEW * rs;
#pragma unroll 1
                for (int ck = 0; ck < KC; ck++) {
                    const mfm_v4i *ap = reinterpret_cast<const mfm_v4i *>(L.afrag) +
                                        ((size_t)rbc * (KQ * KC) + (size_t)ck * KQ) * 2 * 64 + lane;
#pragma unroll
                    for (int kq = 0; kq < KQ; kq++) {
                        a_h[kq] = ap[(kq * 2 + 0) * 64];
                        a_l[kq] = ap[(kq * 2 + 1) * 64];
                    }
                    const uint2 bo = *reinterpret_cast<const uint2 *>(bof_s + ck * KQ); /* four uint16 offsets */
                    const uint32_t bo4[4] = { bo.x & 0xffffu, bo.x >> 16, bo.y & 0xffffu, bo.y >> 16 };
#pragma unroll
                    for (int kq = 0; kq < KQ; kq++) {
#pragma unroll
                        for (int gq = 0; gq < 2; gq++) {
                            const uint32_t at = ibase + bo4[kq] + (uint32_t)gq * 16u * rs;
                            const mfm_v4i b_h = *reinterpret_cast<const mfm_v4i *>(plane_h + at);
                            if (IN8) {
                                if ((ah_mask >> (ck * KQ + kq)) & 1u) {
                                    hh2[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], b_h, hh2[gq], 0, 0, 0);
                                }
                                ll2[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], b_h, ll2[gq], 0, 0, 0);
                                continue;
                            }
                            const mfm_v4i b_l = *reinterpret_cast<const mfm_v4i *>(plane_l + at);
                            if ((ah_mask >> (ck * KQ + kq)) & 1u) {
                                hh2[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], b_h, hh2[gq], 0, 0, 0);
                                md2[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], b_l, md2[gq], 0, 0, 0);
                            }
                            ll2[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], b_l, ll2[gq], 0, 0, 0);
                            md2[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], b_h, md2[gq], 0, 0, 0);
                        }
                    }
                }
                if (it == 0) {
                    /* the next tile's samples: requested behind this iteration's tap loads, so that waiting for taps
                     * (vmcnt is in order) does not wait for HBM */
#pragma unroll
                    for (int j = 0; j < NCH; j++) {
                        pre[j] = stage_load(have_n ? tile_n : tile, j);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                finish_group(0, hh2[0], md2[0], ll2[0]);
                finish_group(1, hh2[1], md2[1], ll2[1]);
            } else if constexpr (RESIDENT) {
                /* ---- resident long filters: two waves per SIMD do not cover an LDS round trip between one k-step's reads
                 *      and its products the way four do, so the B fragments are requested MFM_RES_PF k-steps ahead - across
                 *      the boundary of the two column groups - with the requests and their waits spelled out (left to
                 *      itself the compiler sinks every read to just in front of its use).  lgkmcnt counts in order for
                 *      LDS: a wait for "at most N younger requests outstanding" can only wait too long when the compiler
                 *      has LDS traffic of its own in flight, never too briefly. ---- */
                /* (the int16 instances that keep every high-byte tap plane have 128 tap registers: two k-steps ahead there;
                 * measured, two and four and six are the same to within the noise - profiles/r03_resident_taps.txt.  No
                 * instance may spill: a fragment register saved to scratch between its request and its wait would save
                 * what was in it before the data arrived) */
                constexpr bool kAllPlanes = !IN8 && AHM == (1 << KQ) - 1 && KQ == 16;
                constexpr int PF = kAllPlanes ? (NCH >= 8 ? 1 : 2) : MFM_RES_PF, RPK = IN8 ? 1 : 2, SLOTS = PF + 1, NS = 2 * KQ;
                static_assert(PF * RPK <= 15, "lgkmcnt is a 4-bit counter");
                mfm_v4i hh[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } }, md[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } }, ll[2] = { krow, krow };
                mfm_v4i bh[SLOTS], bl[SLOTS];
                const uint32_t lds_h = (uint32_t)(uintptr_t)plane_h + it * MFM_M_NEW * rs;
                auto request = [&](int st) { /* step st = column group st / KQ, k-step st % KQ */
                    const uint32_t at = lds_h + (uint32_t)(st / KQ) * 16u * rs + boff[st % KQ];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bh[st % SLOTS]) : "v"(at) : "memory");
                    if (!IN8) {
                        const uint32_t at_l = at + plane_dist;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(bl[st % SLOTS]) : "v"(at_l) : "memory");
                    }
                };
#pragma unroll
                for (int st = 0; st < PF && st < NS; st++) {
                    request(st);
                }
#pragma unroll
                for (int st = 0; st < NS; st++) {
                    const int gq = st / KQ, kq = st % KQ, cb = st % SLOTS;
                    if (st + PF < NS) {
                        request(st + PF);
                    }
                    mfm_wait_fragments<IN8>((NS - 1 - st < PF ? NS - 1 - st : PF) * RPK, bh[cb], bl[cb]);
                    if (IN8) {
                        if ((ah_mask >> kq) & 1u) {
                            hh[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], bh[cb], hh[gq], 0, 0, 0);
                        }
                        ll[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], bh[cb], ll[gq], 0, 0, 0);
                    } else {
                        if ((ah_mask >> kq) & 1u) {
                            hh[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], bh[cb], hh[gq], 0, 0, 0);
                            md[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], bl[cb], md[gq], 0, 0, 0);
                        }
                        ll[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], bl[cb], ll[gq], 0, 0, 0);
                        md[gq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], bh[cb], md[gq], 0, 0, 0);
                    }
                    if (kq == KQ - 1) {
                        /* MFMA -> VALU read hazard, see below; group 0 is finished while group 1's fragments arrive */
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        finish_group(gq, hh[gq], md[gq], ll[gq]);
                    }
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    /* ---- GEMM: 16 rows x 16 columns x (64*KQ) elements, four byte-plane products ---- */
                    mfm_v4i hh = { 0, 0, 0, 0 }, md = { 0, 0, 0, 0 }, ll = krow;
                    const uint32_t gbase = (it * MFM_M_NEW + 16u * (uint32_t)gq) * rs;
                    mfm_v4i bh[2], bl[2];
                    bh[0] = *reinterpret_cast<const mfm_v4i *>(plane_h + gbase + boff[0]);
                    if (!IN8) {
                        bl[0] = *reinterpret_cast<const mfm_v4i *>(plane_l + gbase + boff[0]);
                    }
    #pragma unroll
                    for (int kq = 0; kq < KQ; kq++) {
                        const int cb = kq & 1, nb = cb ^ 1;
                        if (kq + 1 < KQ) {
                            bh[nb] = *reinterpret_cast<const mfm_v4i *>(plane_h + gbase + boff[kq + 1]);
                            if (!IN8) {
                                bl[nb] = *reinterpret_cast<const mfm_v4i *>(plane_l + gbase + boff[kq + 1]);
                            }
                        }
                        if (IN8) { /* one sample plane: hh = sum Wh * s, ll = sum Wl * s + row constant */
                            if ((ah_mask >> kq) & 1u) {
                                hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], bh[cb], hh, 0, 0, 0);
                            }
                            ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], bh[cb], ll, 0, 0, 0);
                            continue;
                        }
                        if ((ah_mask >> kq) & 1u) { /* uniform: skipped where the high-byte tap plane is all zero */
                            hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], bh[cb], hh, 0, 0, 0);
                            md = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], bl[cb], md, 0, 0, 0);
                        }
                        ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], bl[cb], ll, 0, 0, 0);
                        md = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], bh[cb], md, 0, 0, 0);
                    }
                    /* The accumulators are read by VALU code right below.  hipcc (ROCm 7.2) has been seen to leave the
                     * MFMA -> VALU read hazard unpadded here (caught by the parity tests: tile 0 passed, later tiles did
                     * not).  16 wait states cover a 16x16x64 MFMA. */
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);

                    finish_group(gq, hh, md, ll);
                }
            }
            if (KC == 1 && NIT == 1 && !(RESIDENT && MFM_RES_EARLY)) {
                /* single-iteration tiles are the big ones (up to 8 chunks per thread): requested behind the matrix
                 * work, when the accumulators are about to die */
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    pre[j] = stage_load(have_n ? tile_n : tile, j);
                }
            }

            __builtin_amdgcn_s_setprio(2);
            int pcm[2][2];
            if (it == 0) {
                /* column 0 of the pass is the last filtered sample of the previous pass */
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    q[0][c] = use_carry ? carry[c] : q[0][c];
                }
            }
#pragma unroll
            for (int c = 0; c < 2; c++) {
                /* previous output of the same channel: the lane to the left; column 16's is column 15.
                 * bound_ctrl: lanes without a source read 0 and no "old" value has to be set up (column 0 of
                 * group 0 is never stored, so what its lane gets does not matter) */
                const uint32_t p0 =
                    (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q[0][c], 0x111 /* row_shr:1 */, 0xf, 0xf, true);
                const int wrap = __builtin_amdgcn_update_dpp(0, (int)q[0][c], 0x121 /* row_ror:1 */, 0xf, 0xf, true);
                const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(wrap, (int)q[1][c], 0x111 /* row_shr:1 */,
                                                                          0xf, 0xf, false);
                const uint32_t pp[2] = { p0, p1 };
                int s_re[2], s_im[2], out[2];
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    mfm_conj_mul(q[gq][c], pp[gq], &s_re[gq], &s_im[gq]);
                }
                mfm_discriminate2(s_re, s_im, lut_t, lut_d, out);
                pcm[0][c] = out[0];
                pcm[1][c] = out[1];
            }
            /* rotator entries of the next iteration (of the next tile after the last one) */
            uint2 rvn[2][2];
            if (it + 1 < NIT) {
                rot_load(k_off, it + 1, rvn);
            } else {
                rot_load(k_off_n, 0, rvn);
            }
            __builtin_amdgcn_sched_barrier(MFM_SCHED_ALL_BUT_VMEM);

            if (it + 1 == NIT) {
                /* the prefetched samples go to the other staging buffer */
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    stage_store(cur ^ 1u, j, pre[j]);
                }
                __builtin_amdgcn_sched_barrier(MFM_SCHED_ALL_BUT_VMEM);
            }
            /* PCM: every lane stores its four values; the ones that are not outputs (column 0 of group 0, columns
             * past n_new, channels past the end) go to the dump slot */
            const uint32_t dump = L.nchan * L.out_stride + lane;
#pragma unroll
            for (int gq = 0; gq < 2; gq++) {
                const int rel = rel_first + (int)(it * MFM_M_NEW + 16u * gq + n);
                const bool col_ok = rb_valid && (gq != 0 || n != 0) && (rel < (int)L.n_new);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const uint32_t chn = ch0 + c;
                    const bool ok = col_ok && chn < L.nchan;
                    const uint32_t at = ok ? chn * L.out_stride + (uint32_t)rel : dump; /* fits 32 bits (engine checks) */
#if MFM_M_NONTEMPORAL
                    __builtin_nontemporal_store((int16_t)pcm[gq][c], reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(L.pcm) + (at << 1)));
#else
                    *reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(L.pcm) + (at << 1)) = (int16_t)pcm[gq][c];
#endif
                    if (DBG_IQ) {
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(L.iq_dbg) + (at << 2)) = q[gq][c];
                    }
                }
            }
            if (last_tile) {
                /* the one sample the next pass starts from */
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    const int rel = rel_first + (int)(it * MFM_M_NEW + 16u * gq + n);
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const uint32_t chn = mfm_opaque(ch0) + c;
                        if (rb_valid && (gq != 0 || n != 0) && rel == (int)L.n_new - 1 && chn < L.nchan) {
                            L.st_out[chn].carry_q = q[gq][c];
                        }
                    }
                }
            }
#pragma unroll
            for (int gq = 0; gq < 2; gq++) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    rv[gq][c] = rvn[gq][c];
                }
            }
        }

        if (rb_valid && tile == 0 && n == 0) {
            /* rotator index of the next pass's first output, one lane per channel pair */
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t chn = mfm_opaque(ch0) + c;
                if (chn < L.nchan) {
                    const uint32_t *ip = reinterpret_cast<const uint32_t *>(L.info) + (size_t)chn * 8;
                    const uint32_t mu = ip[2], lam = ip[3], lam_magic = ip[4];
                    uint32_t kn = L.st_in[chn].kb + L.n_new;
                    if (kn >= mu) {
                        const uint32_t x = kn - mu;
                        uint32_t m = x - __umulhi(x, lam_magic) * lam;
                        m = (m >= lam) ? m - lam : m;
                        kn = mu + m;
                    }
                    L.st_out[chn].kb = kn;
                }
            }
        }
        __syncthreads(); /* next tile's image is complete and nobody reads the current one any more */
        cur ^= 1u;
        item = item_n;
        tile = tile_n;
        slice = slice_n;
        have = have_n;
        k_off[0] = k_off_n[0];
        k_off[1] = k_off_n[1];
    }
}

/* Which instance runs a launch description - decided by geometry fields that are fixed at commit (and the input format);
 * the engine asks once per format at commit, raises the instance's LDS limit there and launches through the pointer
 * (mfm_kernel_v3.hip has the same pair). */
/*
 * The resident long-filter instance for a launch description, or nullptr when none is built for it.  The tap-plane mask
 * of an instance only has to COVER the filter's (a k-step whose high-byte plane is zero multiplies zeros, exactly): none,
 * the four and the eight middle k-steps of sixteen (two and four of eight) are what windowed low-pass filters of 129..512
 * taps produce at the gains multifm runs them at; anything else keeps every plane.  Staging chunks per thread round up
 * to a built count (a surplus chunk is loaded and not stored).
 */
template <int KQ, int AHM, int NIT, bool IN8>
static const void *mfm_resident_instance_nch(uint32_t nch)
{
    if constexpr (NIT == 2) {
        return nch <= 2 ? reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ, false, false, 2, 1, AHM, NIT, IN8>)
                        : reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ, false, false, 4, 1, AHM, NIT, IN8>);
    } else {
        return nch <= 4   ? reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ, false, false, 4, 1, AHM, NIT, IN8>)
               : nch <= 6 ? reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ, false, false, 6, 1, AHM, NIT, IN8>)
               : nch == 7 ? reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ, false, false, 7, 1, AHM, NIT, IN8>)
                          : reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ, false, false, 8, 1, AHM, NIT, IN8>);
    }
}

template <int KQ, int AHM, bool IN8>
static const void *mfm_resident_instance_nit(const mfm_launch_mfma *L, uint32_t nch)
{
    return L->ot == MFM_M_NEW ? mfm_resident_instance_nch<KQ, AHM, 1, IN8>(nch) : mfm_resident_instance_nch<KQ, AHM, 2, IN8>(nch);
}

template <int KQ, bool IN8>
static const void *mfm_resident_instance_mask(const mfm_launch_mfma *L, uint32_t nch)
{
    constexpr int kMid4 = KQ == 16 ? 0x03c0 : 0x18, kMid8 = KQ == 16 ? 0x0ff0 : 0x3c, kAll = KQ == 16 ? 0xffff : 0xff;
    if (L->ah_mask == 0u) { /* every tap fits one byte (the 512-tap 12.5 kHz low-pass at 10 MS/s: largest tap 82): two products per k-step */
        return mfm_resident_instance_nit<KQ, 0, IN8>(L, nch);
    }
    if ((L->ah_mask & ~(uint32_t)kMid4) == 0u) {
        return mfm_resident_instance_nit<KQ, kMid4, IN8>(L, nch);
    }
    if ((L->ah_mask & ~(uint32_t)kMid8) == 0u) {
        return mfm_resident_instance_nit<KQ, kMid8, IN8>(L, nch);
    }
    return mfm_resident_instance_nit<KQ, kAll, IN8>(L, nch);
}

static const void *mfm_resident_instance(const mfm_launch_mfma *L, int dbg_iq, uint32_t nch)
{
    if (dbg_iq || L->stream_taps || L->fixed_planes || (L->kq != 8u && L->kq != 16u)) {
        return nullptr;
    }
    if (L->ot != MFM_M_NEW && nch > 4u) {
        return nullptr;
    }
    if (L->kq == 16u && L->kq_used >= 9u && L->kq_used <= 12u) {
        /* nine to twelve k-steps of taps in a sixteen-step layout (256 taps at decimation 25 = 11, at decimation 100 = 9:
         * the reference's POCSAG configurations with their own low-pass files): an instance of twelve multiplies the
         * first twelve and leaves the all-zero rest alone.  Where the taps beyond one byte sit depends on the filter's
         * centre and the channel gains: none of them (instance without the high-byte products) or the mask at run time. */
        if (L->ah_mask == 0u) {
            return L->in8 ? mfm_resident_instance_nit<12, 0, true>(L, nch) : mfm_resident_instance_nit<12, 0, false>(L, nch);
        }
        return L->in8 ? mfm_resident_instance_nit<12, -1, true>(L, nch) : mfm_resident_instance_nit<12, -1, false>(L, nch);
    }
    if (L->split_rows && !MFM_M_RES_SPLIT_ALL) {
        return nullptr; /* streamed instances (MFM_M_RES_SPLIT_ALL has the price of doing otherwise) */
    }
    if (L->kq == 16u) {
        return L->in8 ? mfm_resident_instance_mask<16, true>(L, nch) : mfm_resident_instance_mask<16, false>(L, nch);
    }
    /* eight k-steps: the 8-bit form has 128-register instances that hold them (below); the int16 form gets them here */
    return L->in8 ? nullptr : mfm_resident_instance_mask<8, false>(L, nch);
}

/* waves_per_simd_out: what the chosen instance is built for (4: two workgroups of 8 waves per CU where LDS allows, 2: one) */
/* The packed discriminator of this file on caller-supplied products (tests, the engine's division self-test): thread t
 * takes s[2t], s[2t + 1]. */
__global__ __launch_bounds__(512) void mfm_disc_test_kernel_mfma(const int *s_re, const int *s_im, int *pcm, uint32_t n2, const float2 *lut)
{
    __shared__ __attribute__((aligned(16))) float tbl[512]; /* T[256] followed by dT[256] */
    reinterpret_cast<uint32_t *>(tbl)[(threadIdx.x >> 1) + ((threadIdx.x & 1u) << 8)] = reinterpret_cast<const uint32_t *>(lut)[threadIdx.x];
    __syncthreads();
    const uint32_t t = blockIdx.x * 512u + threadIdx.x;
    if (t >= n2) {
        return;
    }
    const int32_t re[2] = { s_re[2 * t], s_re[2 * t + 1] }, im[2] = { s_im[2 * t], s_im[2 * t + 1] };
    int32_t out[2];
    mfm_discriminate2(re, im, tbl, tbl + 256, out);
    pcm[2 * t] = out[0];
    pcm[2 * t + 1] = out[1];
}

extern "C" hipError_t mfm_disc_test_mfma(const int *s_re, const int *s_im, int *pcm, uint32_t n, const float2 *lut, hipStream_t stream)
{
    const uint32_t n2 = n / 2u;
    hipLaunchKernelGGL(mfm_disc_test_kernel_mfma, dim3((n2 + 511u) / 512u), dim3(512), 0, stream, s_re, s_im, pcm, n2, lut);
    return hipGetLastError();
}

extern "C" hipError_t mfm_select_channel_kernel_mfma(const mfm_launch_mfma *L, int dbg_iq, const void **kfn_out,
                                                     uint32_t *waves_per_simd_out)
{
    *kfn_out = nullptr;
    *waves_per_simd_out = 4;
    if (L->in8 && dbg_iq) {
        return hipErrorInvalidValue;
    }
    {
        const uint32_t nch_r = ((L->nstage >> 2) + MFM_M_NT - 1) / MFM_M_NT;
        /* the 128-register int16 instances of five and six k-steps (below) multiply less than a resident eight would */
        const void *fn = nch_r >= 1 && nch_r <= MFM_M_CH_MAX ? mfm_resident_instance(L, dbg_iq, nch_r) : nullptr;
        if (fn && !(L->kq == 8u && (L->kq_used == 5u || L->kq_used == 6u) && nch_r == 1 && L->ot != MFM_M_NEW)) {
            *kfn_out = fn;
            *waves_per_simd_out = 2;
            return hipSuccess;
        }
    }
#define MFM_LAUNCH_N(KQ_, DBG_, FIXP_, NCH_)                                                                 \
    do {                                                                                                     \
        /* the usual case - 128-tap low-pass, only the two middle k-steps carry taps beyond one byte - and the  \
         * all-planes case get branch-free kernels; anything else reads the mask at run time */               \
        if (KQ_ == 4 && !DBG_ && L->ah_mask == 0x6u) {                                                       \
            MFM_LAUNCH_A(KQ_, DBG_, FIXP_, NCH_, 1, (KQ_ == 4 && !DBG_) ? 0x6 : -1);                         \
        } else if (KQ_ == 4 && !DBG_ && L->ah_mask == 0xfu) {                                                \
            MFM_LAUNCH_A(KQ_, DBG_, FIXP_, NCH_, 1, (KQ_ == 4 && !DBG_) ? 0xf : -1);                         \
        } else {                                                                                             \
            MFM_LAUNCH_A(KQ_, DBG_, FIXP_, NCH_, 1, -1);                                                     \
        }                                                                                                    \
    } while (0)
#define MFM_LAUNCH_C(KQ_, DBG_, FIXP_, NCH_, KC_) MFM_LAUNCH_A(KQ_, DBG_, FIXP_, NCH_, KC_, -1)
#define MFM_LAUNCH_A(KQ_, DBG_, FIXP_, NCH_, KC_, AHM_) MFM_LAUNCH_T(KQ_, DBG_, FIXP_, NCH_, KC_, AHM_, 2)
#define MFM_LAUNCH_T(KQ_, DBG_, FIXP_, NCH_, KC_, AHM_, NIT_)                                                \
    do {                                                                                                     \
        /* 8-bit input (L->in8: the engine only asks when no channel wants its filtered IQ) */                  \
        *kfn_out = (L->in8 && !DBG_)                                                                         \
                       ? reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ_, DBG_, FIXP_, NCH_, KC_, AHM_, NIT_, !DBG_>) \
                       : reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ_, DBG_, FIXP_, NCH_, KC_, AHM_, NIT_, false>); \
    } while (0)
#define MFM_LAUNCH_M(KQ_, DBG_, FIXP_)                                                                       \
    do {                                                                                                     \
        switch (nch) {                                                                                       \
        case 1: MFM_LAUNCH_N(KQ_, DBG_, FIXP_, 1); break;                                                    \
        case 2: MFM_LAUNCH_N(KQ_, DBG_, FIXP_, 2); break;                                                    \
        case 3: MFM_LAUNCH_N(KQ_, DBG_, FIXP_, 3); break;                                                    \
        default: MFM_LAUNCH_N(KQ_, DBG_, FIXP_, 4); break;                                                   \
        }                                                                                                    \
    } while (0)
#define MFM_LAUNCH_KQ(KQ_)                                                                                   \
    do {                                                                                                     \
        if (dbg_iq) {                                                                                        \
            if (L->fixed_planes) {                                                                           \
                MFM_LAUNCH_M(KQ_, true, true);                                                               \
            } else {                                                                                         \
                MFM_LAUNCH_M(KQ_, true, false);                                                              \
            }                                                                                                \
        } else if (L->fixed_planes) {                                                                        \
            MFM_LAUNCH_M(KQ_, false, true);                                                                  \
        } else {                                                                                             \
            MFM_LAUNCH_M(KQ_, false, false);                                                                 \
        }                                                                                                    \
    } while (0)

    const uint32_t nch = ((L->nstage >> 2) + MFM_M_NT - 1) / MFM_M_NT; /* 16-byte chunks per thread and tile */
    if (nch < 1 || nch > MFM_M_CH_MAX) {
        return hipErrorInvalidValue;
    }
#define MFM_LAUNCH_S(KC_)                                                                                    \
    do {                                                                                                     \
        if (dbg_iq) {                                                                                        \
            switch (nch) {                                                                                   \
            case 1: MFM_LAUNCH_C(4, true, false, 1, KC_); break;                                             \
            case 2: MFM_LAUNCH_C(4, true, false, 2, KC_); break;                                             \
            case 3: MFM_LAUNCH_C(4, true, false, 3, KC_); break;                                             \
            default: MFM_LAUNCH_C(4, true, false, 4, KC_); break;                                            \
            }                                                                                                \
        } else {                                                                                             \
            switch (nch) {                                                                                   \
            case 1: MFM_LAUNCH_C(4, false, false, 1, KC_); break;                                            \
            case 2: MFM_LAUNCH_C(4, false, false, 2, KC_); break;                                            \
            case 3: MFM_LAUNCH_C(4, false, false, 3, KC_); break;                                            \
            default: MFM_LAUNCH_C(4, false, false, 4, KC_); break;                                           \
            }                                                                                                \
        }                                                                                                    \
    } while (0)

    if (L->ot == MFM_M_NEW) {
        /* single-iteration tiles (large decimations): 128-tap-class and streamed filters, packed planes */
        if (L->fixed_planes || (L->kq != 4 && L->kq != 8 && L->kq != 16)) {
            return hipErrorInvalidValue;
        }
#define MFM_LAUNCH_1N(DBG_, KC_)                                                                             \
    do {                                                                                                     \
        switch (nch) {                                                                                       \
        case 1: MFM_LAUNCH_T(4, DBG_, false, 1, KC_, -1, 1); break;                                          \
        case 2: MFM_LAUNCH_T(4, DBG_, false, 2, KC_, -1, 1); break;                                          \
        case 3: MFM_LAUNCH_T(4, DBG_, false, 3, KC_, -1, 1); break;                                          \
        case 4: MFM_LAUNCH_T(4, DBG_, false, 4, KC_, -1, 1); break;                                          \
        case 5: MFM_LAUNCH_T(4, DBG_, false, 5, KC_, -1, 1); break;                                          \
        case 6: MFM_LAUNCH_T(4, DBG_, false, 6, KC_, -1, 1); break;                                          \
        case 7: MFM_LAUNCH_T(4, DBG_, false, 7, KC_, -1, 1); break;                                          \
        default: MFM_LAUNCH_T(4, DBG_, false, 8, KC_, -1, 1); break;                                         \
        }                                                                                                    \
    } while (0)
#define MFM_LAUNCH_1(KC_)                                                                                    \
    do {                                                                                                     \
        if (dbg_iq) {                                                                                        \
            MFM_LAUNCH_1N(true, KC_);                                                                        \
        } else {                                                                                             \
            MFM_LAUNCH_1N(false, KC_);                                                                       \
        }                                                                                                    \
    } while (0)
        switch (L->kq) {
        case 4: MFM_LAUNCH_1(1); break;
        case 8: MFM_LAUNCH_1(2); break;
        default: MFM_LAUNCH_1(4); break;
        }
#undef MFM_LAUNCH_1
#undef MFM_LAUNCH_1N
        return hipSuccess;
    }
    if (nch > 4) {
        return hipErrorInvalidValue; /* two-iteration tiles are built for up to 4 chunks per thread */
    }
    switch (L->kq) {
    case 1: MFM_LAUNCH_KQ(1); break;
    case 2: MFM_LAUNCH_KQ(2); break;
    case 4: MFM_LAUNCH_KQ(4); break;
    case 8:
        if (L->fixed_planes) {
            return hipErrorInvalidValue; /* the streaming variants are built for packed planes only */
        }
#define MFM_LAUNCH_X(KQ_, NCH_, IN8_)                                                                        \
    do {                                                                                                     \
        *kfn_out = reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<KQ_, false, false, NCH_, 1, -1, 2, IN8_>); \
    } while (0)
        /* Five to seven k-steps of taps (the D = 25 plan of etc/pocsag_rtlsdr.json: 326 elements = 6) fit the registers
         * beside one or two staging chunks: nothing is streamed, nothing is multiplied with the padding up to the eighth.
         * The int16 form has the registers for six of them and one chunk. */
        if (!dbg_iq && L->in8 && nch <= 2 && L->kq_used >= 5u && L->kq_used <= 7u) {
            if (nch == 1) {
                switch (L->kq_used) {
                case 5: MFM_LAUNCH_X(5, 1, true); break;
                case 6: MFM_LAUNCH_X(6, 1, true); break;
                default: MFM_LAUNCH_X(7, 1, true); break;
                }
            } else {
                switch (L->kq_used) {
                case 5: MFM_LAUNCH_X(5, 2, true); break;
                case 6: MFM_LAUNCH_X(6, 2, true); break;
                default: MFM_LAUNCH_X(7, 2, true); break;
                }
            }
            break;
        }
        if (!dbg_iq && !L->in8 && nch == 1 && (L->kq_used == 5u || L->kq_used == 6u)) {
            if (L->kq_used == 5u) {
                MFM_LAUNCH_X(5, 1, false);
            } else {
                MFM_LAUNCH_X(6, 1, false);
            }
            break;
        }
#undef MFM_LAUNCH_X
        if (L->in8 && !dbg_iq) {
            /* 8-bit input: no low sample plane, no middle accumulator - the registers that frees hold all eight k-steps
             * of taps, so nothing is streamed (the D = 25 plan of etc/pocsag_rtlsdr.json: 6 k-steps run as 8) */
#define MFM_LAUNCH_R8(NCH_)                                                                                  \
    do {                                                                                                     \
        *kfn_out = reinterpret_cast<const void *>(&mfm_channel_kernel_mfma<8, false, false, NCH_, 1, -1, 2, true>); \
    } while (0)
            if (nch <= 2) { /* more staging registers than that and the taps no longer fit beside them */
                if (nch == 1) {
                    MFM_LAUNCH_R8(1);
                } else {
                    MFM_LAUNCH_R8(2);
                }
                break;
            }
#undef MFM_LAUNCH_R8
        }
        MFM_LAUNCH_S(2);
        break;
    case 16:
        if (L->fixed_planes) {
            return hipErrorInvalidValue;
        }
        MFM_LAUNCH_S(4);
        break;
    default: return hipErrorInvalidValue;
    }
#undef MFM_LAUNCH_S
#undef MFM_LAUNCH_KQ
#undef MFM_LAUNCH_M
#undef MFM_LAUNCH_N
#undef MFM_LAUNCH_C
#undef MFM_LAUNCH_A
#undef MFM_LAUNCH_T
    return hipSuccess;
}

extern "C" hipError_t mfm_launch_channel_kernel_mfma(const void *kfn, const mfm_launch_mfma *L, uint32_t lds_bytes,
                                                     uint32_t grid, hipStream_t stream)
{
    if (L->ntiles == 0) {
        return hipSuccess;
    }
    void *args[] = { const_cast<mfm_launch_mfma *>(L) };
    return hipLaunchKernel(kfn, dim3(grid), dim3(MFM_M_NT), args, lds_bytes, stream);
}
