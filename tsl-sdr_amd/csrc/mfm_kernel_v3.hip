/*
 * mfm_kernel_v3.hip - the multifm channel kernel on the matrix cores, second generation.
 *
 * Arithmetic: exactly that of mfm_kernel_mfma.hip - the bank of complex-tap decimating FIRs
 * (filter/direct_fir.c:328-417, filter/complex.h:40-46) as four int8 byte-plane products per k-step on
 * v_mfma_i32_16x16x64_i8 with wrapping int32 accumulators, then Q14 round, derotation by the tabulated
 * rotator, Q14 round (filter/direct_fir.c:151-172,406-413), s = q * conj(prev), fast_atan2f, PCM
 * (multifm/fm_demod.c:53-79, multifm/fast_atan2f.c:101-174).  Bit for bit the oracle's results.
 *
 * What is different is the assignment of outputs to lanes and of tiles to workgroups (mfm_kernel.h has the
 * summary).  Measured on MI355X (profiles/r02_*): the first generation was held at ~105 us per 2^26-sample
 * block by its 2-byte PCM stores alone (one 64-lane store instruction moves 128 bytes and takes as long as one
 * that moves 1 KB), matrix instructions and other VALU instructions of one SIMD do not overlap (only two
 * instructions between two MFMAs of a wave issue in the MFMA's shadow), and most VALU instructions other than
 * 32-bit add / logic / shift-right / fp32 mul-add-fma take two issue slots.  Hence:
 *   - 8-byte PCM stores (4 consecutive outputs per lane), 16-byte rotator loads;
 *   - chunks of consecutive tiles per workgroup: history and table position stay in registers, no per-tile
 *     index arithmetic, no recomputed column;
 *   - workgroups stage 64 + 4 + (window overhang) rows per tile; the 4 rows in front are only read by the
 *     first tile of a chunk (its column group 3 shifted one sub-plane row down yields output -1).
 *
 * Geometry: workgroup = 8 waves, wave w = GEMM rows 16w..16w+15 = channels 8w..8w+7 of the 64-channel slice,
 * taps in registers.  Lane (kg = lane >> 4, n = lane & 15) holds, after column group g, re/im of channels
 * 2kg, 2kg+1 (of the wave's eight) for output 64*tile + 4n + g.
 */
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mfm_kernel.h"
#include "mfm_numerics.h"

#include "mfm_v3_device.h"

/*
 * Hand-scheduled column group for the decimation-96 / 128-tap geometry (DFIX = 96, KQ = 4, AHM = 0x6: the high-byte tap
 * plane is zero in k-steps 0 and 3 -> 12 MFMAs).  Why by hand: on gfx950 an MFMA and other VALU work of one SIMD do not
 * overlap, EXCEPT that up to two instructions behind an MFMA of the same wave issue in its shadow
 * (profiles/r02_ubench_shadow.txt: 4 x (MFMA, 2 VALU) takes as long as 4 MFMAs; a third instruction per gap and
 * everything is paid in full).  The compiler does not move inline asm between MFMAs, and in plain C the recombination
 * costs a third instruction per sum.  So the block interleaves, never more than two per gap:
 *   - the 12 MFMAs of THIS column group,
 *   - the B-fragment reads of its k-steps 2, 3 and of k-steps 0, 1 of the NEXT group (those of its own k-steps 0, 1
 *     were requested by the previous block), with counted s_waitcnt lgkmcnt (LDS returns in order),
 *   - the recombination + first Q14 rounding of the PREVIOUS group (8 v_lshl_add_u32 + 4 SDWA shifts), which also
 *     keeps every accumulator three MFMAs away from its first reader (no hazard padding).
 * Two asm statements per group (an asm statement takes at most 30 operands).  B fragments: X0, X1 = k-steps 0, 1
 * (in: this group's, out: the next group's, still in flight at exit), Y0, Y1 = k-steps 2, 3.
 * acc = {hh, md, ll}; accP = the previous group's; tP = its recombined sums, in the end its packed samples in tP[0], tP[2].
 */
#define MFM3_MF(d, a, b, c) "v_mfma_i32_16x16x64_i8 %[" #d "], %[" #a "], %[" #b "], " c "\n\t"
#define MFM3_RD(d, o) "ds_read_b128 %[" #d "], %[lb] offset:%[" #o "]\n\t"
#define MFM3_LA(d, a, b) "v_lshl_add_u32 %[" #d "], %[" #a "], 8, %[" #b "]\n\t"
#define MFM3_SH0(d, a) "v_lshrrev_b32_sdwa %[" #d "], 14, %[" #a "] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define MFM3_SH1(d, a) "v_lshrrev_b32_sdwa %[" #d "], 14, %[" #a "] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"

template <bool HAS_PREV, bool HAS_NEXT, int LO, int O2, int O3, int N0, int N1, bool REUSE = false>
static __device__ __forceinline__ void mfm3_group_d96(uint32_t lb, uint32_t ka, const mfm_v4i &al0, const mfm_v4i &al1,
                                                      const mfm_v4i &al2, const mfm_v4i &al3, const mfm_v4i &ah1,
                                                      const mfm_v4i &ah2, mfm_v4i &b0h, mfm_v4i &b0l, mfm_v4i &b1h,
                                                      mfm_v4i &b1l, mfm_v4i &b2h, mfm_v4i &b2l, const mfm_v4i (&accP)[3],
                                                      mfm_v4i (&acc)[3], uint32_t (&tP)[4])
{
    /* Three fragment buffers (h + l each), rotating: in: b0 = k-step 0, b1 = k-step 1 of this group (requested by the
     * previous block); b2 takes k-step 2, b0 k-step 3 once k-step 0 is through; out: b1 = k-step 0, b2 = k-step 1 of the
     * next group, still in flight.  The caller rotates (b0, b1, b2) <- (b1, b2, b0) from group to group. */
    /* LO: high-byte plane -> low-byte plane */
    mfm_v4i hh, md, ll;
    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (HAS_PREV) {
        asm volatile(
            "ds_read_b128 %[ll], %[ka]\n\t"                                  /* 128 * sum(W) + 8192 of the lane's rows */
            MFM3_RD(b2h, o2) MFM3_RD(b2l, o2l)
            "s_waitcnt lgkmcnt(3)\n\t"                                       /* b0, b1 (requested by the previous block) */
            MFM3_MF(md, al0, b0h, "0")
            "s_waitcnt lgkmcnt(2)\n\t"
            MFM3_MF(ll, al0, b0l, "%[ll]")
            MFM3_RD(b0h, o3) MFM3_RD(b0l, o3l)                               /* k-step 3 replaces k-step 0 */
            MFM3_MF(hh, ah1, b1h, "0")
            MFM3_LA(t0, h0, m0) MFM3_LA(t1, h1, m1)
            MFM3_MF(md, ah1, b1l, "%[md]")
            MFM3_LA(t2, h2, m2) MFM3_LA(t3, h3, m3)
            MFM3_MF(ll, al1, b1l, "%[ll]")
            MFM3_MF(md, al1, b1h, "%[md]")
            : [hh] "=&v"(hh), [md] "=&v"(md), [ll] "=&v"(ll), [b2h] "=&v"(b2h), [b2l] "=&v"(b2l), [b0h] "+v"(b0h),
              [b0l] "+v"(b0l), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
            : [lb] "v"(lb), [ka] "v"(ka), [al0] "v"(al0), [al1] "v"(al1), [ah1] "v"(ah1), [b1h] "v"(b1h), [b1l] "v"(b1l),
              [h0] "v"(accP[0][0]), [h1] "v"(accP[0][1]), [h2] "v"(accP[0][2]), [h3] "v"(accP[0][3]), [m0] "v"(accP[1][0]),
              [m1] "v"(accP[1][1]), [m2] "v"(accP[1][2]), [m3] "v"(accP[1][3]), [o2] "n"(O2), [o2l] "n"(O2 + LO),
              [o3] "n"(O3), [o3l] "n"(O3 + LO)
            : "memory");
    } else {
        asm volatile(
            "ds_read_b128 %[ll], %[ka]\n\t"
            MFM3_RD(b2h, o2) MFM3_RD(b2l, o2l)
            "s_waitcnt lgkmcnt(3)\n\t"
            MFM3_MF(md, al0, b0h, "0")
            "s_waitcnt lgkmcnt(2)\n\t"
            MFM3_MF(ll, al0, b0l, "%[ll]")
            MFM3_RD(b0h, o3) MFM3_RD(b0l, o3l)
            MFM3_MF(hh, ah1, b1h, "0")
            MFM3_MF(md, ah1, b1l, "%[md]")
            MFM3_MF(ll, al1, b1l, "%[ll]")
            MFM3_MF(md, al1, b1h, "%[md]")
            : [hh] "=&v"(hh), [md] "=&v"(md), [ll] "=&v"(ll), [b2h] "=&v"(b2h), [b2l] "=&v"(b2l), [b0h] "+v"(b0h),
              [b0l] "+v"(b0l)
            : [lb] "v"(lb), [ka] "v"(ka), [al0] "v"(al0), [al1] "v"(al1), [ah1] "v"(ah1), [b1h] "v"(b1h), [b1l] "v"(b1l),
              [o2] "n"(O2), [o2l] "n"(O2 + LO), [o3] "n"(O3), [o3l] "n"(O3 + LO)
            : "memory");
    }
    /* second half: k-steps 2, 3; the next group's k-step 0 goes into b1 (free since MFMA 6), its k-step 1 into b2 (free
     * behind MFMA 10) */
    if (REUSE && HAS_PREV && HAS_NEXT) {
        /* The Toeplitz overlap (MFM3_TOEPLITZ): at D = 96, T = 128 the windows of consecutive outputs overlap by 64 plane bytes =
         * one k-step, so the next column group's k-step 0 IS this group's k-step 3 - the same LDS bytes, already in b0 when
         * this block ends: only the next group's k-step 1 is requested (into b2), and the caller hands (b0, b2, b1) on.  Two
         * requests fewer in flight: the counted waits are 2 where the form above has 4. */
        asm volatile(
            "s_waitcnt lgkmcnt(2)\n\t"                                       /* b2 = k-step 2 (behind it: b0 = k-step 3) */
            MFM3_MF(hh, ah2, b2h, "%[hh]")
            MFM3_LA(t0, t0, l0) MFM3_LA(t1, t1, l1)
            MFM3_MF(md, ah2, b2l, "%[md]")
            MFM3_LA(t2, t2, l2) MFM3_LA(t3, t3, l3)
            MFM3_MF(ll, al2, b2l, "%[ll]")
            MFM3_MF(md, al2, b2h, "%[md]")
            MFM3_RD(b2h, n1) MFM3_RD(b2l, n1l)
            "s_waitcnt lgkmcnt(2)\n\t"                                       /* b0 = k-step 3 */
            MFM3_MF(ll, al3, b0l, "%[ll]")
            MFM3_SH0(t0, t0) MFM3_SH0(t2, t2)
            MFM3_MF(md, al3, b0h, "%[md]")
            MFM3_SH1(t0, t1) MFM3_SH1(t2, t3)
            : [hh] "+v"(hh), [md] "+v"(md), [ll] "+v"(ll), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3),
              [b2h] "+v"(b2h), [b2l] "+v"(b2l)
            : [lb] "v"(lb), [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b0h] "v"(b0h), [b0l] "v"(b0l),
              [l0] "v"(accP[2][0]), [l1] "v"(accP[2][1]), [l2] "v"(accP[2][2]), [l3] "v"(accP[2][3]),
              [n1] "n"(N1), [n1l] "n"(N1 + LO)
            : "memory");
    } else if (REUSE && HAS_NEXT) {
        asm volatile(
            "s_waitcnt lgkmcnt(2)\n\t"
            MFM3_MF(hh, ah2, b2h, "%[hh]")
            MFM3_MF(md, ah2, b2l, "%[md]")
            MFM3_MF(ll, al2, b2l, "%[ll]")
            MFM3_MF(md, al2, b2h, "%[md]")
            MFM3_RD(b2h, n1) MFM3_RD(b2l, n1l)
            "s_waitcnt lgkmcnt(2)\n\t"
            MFM3_MF(ll, al3, b0l, "%[ll]")
            MFM3_MF(md, al3, b0h, "%[md]")
            : [hh] "+v"(hh), [md] "+v"(md), [ll] "+v"(ll), [b2h] "+v"(b2h), [b2l] "+v"(b2l)
            : [lb] "v"(lb), [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b0h] "v"(b0h), [b0l] "v"(b0l),
              [n1] "n"(N1), [n1l] "n"(N1 + LO)
            : "memory");
    } else if (HAS_PREV && HAS_NEXT) {
        asm volatile(
            MFM3_RD(b1h, n0) MFM3_RD(b1l, n0l)
            "s_waitcnt lgkmcnt(4)\n\t"                                       /* b2 = k-step 2 */
            MFM3_MF(hh, ah2, b2h, "%[hh]")
            MFM3_LA(t0, t0, l0) MFM3_LA(t1, t1, l1)
            MFM3_MF(md, ah2, b2l, "%[md]")
            MFM3_LA(t2, t2, l2) MFM3_LA(t3, t3, l3)
            MFM3_MF(ll, al2, b2l, "%[ll]")
            MFM3_MF(md, al2, b2h, "%[md]")
            MFM3_RD(b2h, n1) MFM3_RD(b2l, n1l)
            "s_waitcnt lgkmcnt(4)\n\t"                                       /* b0 = k-step 3 */
            MFM3_MF(ll, al3, b0l, "%[ll]")
            MFM3_SH0(t0, t0) MFM3_SH0(t2, t2)
            MFM3_MF(md, al3, b0h, "%[md]")
            MFM3_SH1(t0, t1) MFM3_SH1(t2, t3)
            : [hh] "+v"(hh), [md] "+v"(md), [ll] "+v"(ll), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3),
              [b1h] "=&v"(b1h), [b1l] "=&v"(b1l), [b2h] "+v"(b2h), [b2l] "+v"(b2l)
            : [lb] "v"(lb), [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b0h] "v"(b0h), [b0l] "v"(b0l),
              [l0] "v"(accP[2][0]), [l1] "v"(accP[2][1]), [l2] "v"(accP[2][2]), [l3] "v"(accP[2][3]),
              [n0] "n"(N0), [n0l] "n"(N0 + LO), [n1] "n"(N1), [n1l] "n"(N1 + LO)
            : "memory");
    } else if (HAS_NEXT) {
        asm volatile(
            MFM3_RD(b1h, n0) MFM3_RD(b1l, n0l)
            "s_waitcnt lgkmcnt(4)\n\t"
            MFM3_MF(hh, ah2, b2h, "%[hh]")
            MFM3_MF(md, ah2, b2l, "%[md]")
            MFM3_MF(ll, al2, b2l, "%[ll]")
            MFM3_MF(md, al2, b2h, "%[md]")
            MFM3_RD(b2h, n1) MFM3_RD(b2l, n1l)
            "s_waitcnt lgkmcnt(4)\n\t"
            MFM3_MF(ll, al3, b0l, "%[ll]")
            MFM3_MF(md, al3, b0h, "%[md]")
            : [hh] "+v"(hh), [md] "+v"(md), [ll] "+v"(ll), [b1h] "=&v"(b1h), [b1l] "=&v"(b1l), [b2h] "+v"(b2h), [b2l] "+v"(b2l)
            : [lb] "v"(lb), [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b0h] "v"(b0h), [b0l] "v"(b0l),
              [n0] "n"(N0), [n0l] "n"(N0 + LO), [n1] "n"(N1), [n1l] "n"(N1 + LO)
            : "memory");
    } else {
        asm volatile(
            "s_waitcnt lgkmcnt(2)\n\t"                                       /* b2; nothing is requested for a next group */
            MFM3_MF(hh, ah2, b2h, "%[hh]")
            MFM3_LA(t0, t0, l0) MFM3_LA(t1, t1, l1)
            MFM3_MF(md, ah2, b2l, "%[md]")
            MFM3_LA(t2, t2, l2) MFM3_LA(t3, t3, l3)
            MFM3_MF(ll, al2, b2l, "%[ll]")
            MFM3_MF(md, al2, b2h, "%[md]")
            "s_waitcnt lgkmcnt(0)\n\t"
            MFM3_MF(ll, al3, b0l, "%[ll]")
            MFM3_SH0(t0, t0) MFM3_SH0(t2, t2)
            MFM3_MF(md, al3, b0h, "%[md]")
            MFM3_SH1(t0, t1) MFM3_SH1(t2, t3)
            : [hh] "+v"(hh), [md] "+v"(md), [ll] "+v"(ll), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3)
            : [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b2h] "v"(b2h), [b2l] "v"(b2l), [b0h] "v"(b0h), [b0l] "v"(b0l),
              [l0] "v"(accP[2][0]), [l1] "v"(accP[2][1]), [l2] "v"(accP[2][2]), [l3] "v"(accP[2][3])
            : "memory");
    }
    acc[0] = hh;
    acc[1] = md;
    acc[2] = ll;
    tP[0] = t0;
    tP[1] = t1;
    tP[2] = t2;
    tP[3] = t3;
}

/*
 * The same for 8-bit input (IN8): one sample plane, so a column group is 6 MFMAs - ll over k-steps 0..3, hh over k-steps
 * 1, 2 - and the recombination of the previous group one shift-add per sum.  Twelve shadow slots, and what wants them:
 * four fragment reads, four shift-adds, four SDWA shifts - but the previous group's last MFMA wrote ll, and a VALU read
 * of an MFMA result wants three MFMAs in between, so the recombination starts behind the third MFMA and its last two
 * shifts are paid for at the end of the block.  Fragment buffers b0, b1, b2 rotate as above (in: k-steps 0, 1 of this
 * group in b0, b1; out: k-steps 0, 1 of the next in b1, b2, in flight).  acc = {hh, -, ll}; SH = the first rounding's
 * shift.
 */
#define MFM3_SHN0(d, a) "v_lshrrev_b32_sdwa %[" #d "], %[sh], %[" #a "] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define MFM3_SHN1(d, a) "v_lshrrev_b32_sdwa %[" #d "], %[sh], %[" #a "] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"

template <bool HAS_PREV, bool HAS_NEXT, int SH, int O2, int O3, int N0, int N1, bool REUSE = false>
static __device__ __forceinline__ void mfm3_group_d96_b8(uint32_t lb, uint32_t ka, const mfm_v4i &al0, const mfm_v4i &al1,
                                                         const mfm_v4i &al2, const mfm_v4i &al3, const mfm_v4i &ah1,
                                                         const mfm_v4i &ah2, mfm_v4i &b0, mfm_v4i &b1, mfm_v4i &b2,
                                                         const mfm_v4i (&accP)[3], mfm_v4i (&acc)[3], uint32_t (&tP)[4])
{
    mfm_v4i hh, ll;
    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    asm volatile(
        "ds_read_b128 %[ll], %[ka]\n\t"                                      /* sum(W)-constant + rounding bias of the lane's rows */
        MFM3_RD(b2, o2)                                                      /* k-step 2 */
        "s_waitcnt lgkmcnt(1)\n\t"                                           /* b0, b1 (requested by the previous block), the constant */
        MFM3_MF(ll, al0, b0, "%[ll]")
        MFM3_RD(b0, o3)                                                      /* k-step 3 replaces k-step 0 */
        MFM3_MF(hh, ah1, b1, "0")
        MFM3_MF(ll, al1, b1, "%[ll]")
        : [hh] "=&v"(hh), [ll] "=&v"(ll), [b2] "=&v"(b2), [b0] "+v"(b0)
        : [lb] "v"(lb), [ka] "v"(ka), [al0] "v"(al0), [al1] "v"(al1), [ah1] "v"(ah1), [b1] "v"(b1), [o2] "n"(O2), [o3] "n"(O3)
        : "memory");
    if (REUSE && HAS_PREV && HAS_NEXT) {
        /* the Toeplitz overlap (mfm3_group_d96): the next group's k-step 0 is this group's k-step 3, already in b0 - one request
         * fewer in flight, the counted waits are 1 where the form below has 2; the caller hands (b0, b2, b1) on */
        asm volatile(
            MFM3_LA(t0, h0, l0)
            "s_waitcnt lgkmcnt(1)\n\t"                                       /* b2 = k-step 2 */
            MFM3_MF(hh, ah2, b2, "%[hh]")
            MFM3_LA(t1, h1, l1) MFM3_LA(t2, h2, l2)
            MFM3_MF(ll, al2, b2, "%[ll]")
            MFM3_RD(b2, n1)                                                  /* the next group's k-step 1 */
            MFM3_LA(t3, h3, l3)
            "s_waitcnt lgkmcnt(1)\n\t"                                       /* b0 = k-step 3 */
            MFM3_MF(ll, al3, b0, "%[ll]")
            MFM3_SHN0(t0, t0) MFM3_SHN0(t2, t2)
            MFM3_SHN1(t0, t1) MFM3_SHN1(t2, t3)
            : [hh] "+v"(hh), [ll] "+v"(ll), [b2] "+v"(b2), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
            : [lb] "v"(lb), [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b0] "v"(b0), [h0] "v"(accP[0][0]),
              [h1] "v"(accP[0][1]), [h2] "v"(accP[0][2]), [h3] "v"(accP[0][3]), [l0] "v"(accP[2][0]), [l1] "v"(accP[2][1]),
              [l2] "v"(accP[2][2]), [l3] "v"(accP[2][3]), [n1] "n"(N1), [sh] "n"(SH)
            : "memory");
    } else if (REUSE && HAS_NEXT) {
        asm volatile(
            "s_waitcnt lgkmcnt(1)\n\t"
            MFM3_MF(hh, ah2, b2, "%[hh]")
            MFM3_MF(ll, al2, b2, "%[ll]")
            MFM3_RD(b2, n1)
            "s_waitcnt lgkmcnt(1)\n\t"
            MFM3_MF(ll, al3, b0, "%[ll]")
            : [hh] "+v"(hh), [ll] "+v"(ll), [b2] "+v"(b2)
            : [lb] "v"(lb), [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b0] "v"(b0), [n1] "n"(N1)
            : "memory");
    } else if (HAS_PREV && HAS_NEXT) {
        asm volatile(
            MFM3_RD(b1, n0)                                                  /* the next group's k-step 0 */
            MFM3_LA(t0, h0, l0)
            "s_waitcnt lgkmcnt(2)\n\t"                                       /* b2 = k-step 2 */
            MFM3_MF(hh, ah2, b2, "%[hh]")
            MFM3_LA(t1, h1, l1) MFM3_LA(t2, h2, l2)
            MFM3_MF(ll, al2, b2, "%[ll]")
            MFM3_RD(b2, n1)                                                  /* the next group's k-step 1 */
            MFM3_LA(t3, h3, l3)
            "s_waitcnt lgkmcnt(2)\n\t"                                       /* b0 = k-step 3 */
            MFM3_MF(ll, al3, b0, "%[ll]")
            MFM3_SHN0(t0, t0) MFM3_SHN0(t2, t2)
            MFM3_SHN1(t0, t1) MFM3_SHN1(t2, t3)
            : [hh] "+v"(hh), [ll] "+v"(ll), [b1] "=&v"(b1), [b2] "+v"(b2), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
              [t3] "=&v"(t3)
            : [lb] "v"(lb), [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b0] "v"(b0), [h0] "v"(accP[0][0]),
              [h1] "v"(accP[0][1]), [h2] "v"(accP[0][2]), [h3] "v"(accP[0][3]), [l0] "v"(accP[2][0]), [l1] "v"(accP[2][1]),
              [l2] "v"(accP[2][2]), [l3] "v"(accP[2][3]), [n0] "n"(N0), [n1] "n"(N1), [sh] "n"(SH)
            : "memory");
    } else if (HAS_NEXT) {
        asm volatile(
            MFM3_RD(b1, n0)
            "s_waitcnt lgkmcnt(2)\n\t"
            MFM3_MF(hh, ah2, b2, "%[hh]")
            MFM3_MF(ll, al2, b2, "%[ll]")
            MFM3_RD(b2, n1)
            "s_waitcnt lgkmcnt(2)\n\t"
            MFM3_MF(ll, al3, b0, "%[ll]")
            : [hh] "+v"(hh), [ll] "+v"(ll), [b1] "=&v"(b1), [b2] "+v"(b2)
            : [lb] "v"(lb), [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b0] "v"(b0), [n0] "n"(N0), [n1] "n"(N1)
            : "memory");
    } else {
        asm volatile(
            MFM3_LA(t0, h0, l0)
            "s_waitcnt lgkmcnt(1)\n\t"                                       /* b2; nothing is requested for a next group */
            MFM3_MF(hh, ah2, b2, "%[hh]")
            MFM3_LA(t1, h1, l1) MFM3_LA(t2, h2, l2)
            MFM3_MF(ll, al2, b2, "%[ll]")
            MFM3_LA(t3, h3, l3)
            "s_waitcnt lgkmcnt(0)\n\t"
            MFM3_MF(ll, al3, b0, "%[ll]")
            MFM3_SHN0(t0, t0) MFM3_SHN0(t2, t2)
            MFM3_SHN1(t0, t1) MFM3_SHN1(t2, t3)
            : [hh] "+v"(hh), [ll] "+v"(ll), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
            : [al2] "v"(al2), [al3] "v"(al3), [ah2] "v"(ah2), [b2] "v"(b2), [b0] "v"(b0), [h0] "v"(accP[0][0]),
              [h1] "v"(accP[0][1]), [h2] "v"(accP[0][2]), [h3] "v"(accP[0][3]), [l0] "v"(accP[2][0]), [l1] "v"(accP[2][1]),
              [l2] "v"(accP[2][2]), [l3] "v"(accP[2][3]), [sh] "n"(SH)
            : "memory");
    }
    acc[0] = hh;
    acc[2] = ll;
    tP[0] = t0;
    tP[1] = t1;
    tP[2] = t2;
    tP[3] = t3;
}

/* LDS row stride of a decimation: 2 * D plane bytes rounded up to an odd multiple of 32 (the engine's rule) */
static constexpr uint32_t mfm3_row_stride(uint32_t decim)
{
    const uint32_t r = (2u * decim + 31u) / 32u * 32u;
    return ((r / 32u) & 1u) ? r : r + 32u;
}

/*
 * The LDS geometry of a decimation as compile-time constants (the DFIX instances: every B-fragment address is lane
 * register + instruction immediate).  Decimations that are multiples of 32 use the sub-plane layout at the fixed 4096-byte
 * pitch, the other multiples of 8 the chunk-row layout (mfm_kernel.h); the engine computes the same numbers at run time
 * and the launcher only picks a DFIX instance when they agree.
 */
template <int DFIX>
struct mfm3_geo {
    static constexpr uint32_t D = DFIX > 0 ? (uint32_t)DFIX : 32u;
    /* decimation 25 (etc/pocsag_rtlsdr.json): rows of 50 plane bytes padded to 64, the taps carry zeros over the padding
     * (mfm_kernel.h, layout 2) - a k-step IS a row, a window spans six of them */
    static constexpr bool padded = DFIX == 25;
    static constexpr uint32_t row_bytes = padded ? 64u : 2u * D;
    static constexpr bool chunk_rows = !padded && (2u * D) % 64u != 0u;
    static constexpr uint32_t rows = padded ? MFM_V3_LEAD + MFM_V3_OT + 5u
                                            : MFM_V3_LEAD + MFM_V3_OT + (64u * 4u - 1u) / (2u * D); /* 128-tap class: four k-steps */
    /* chunk rows */
    static constexpr uint32_t cpo = D / 8u, per = 4u * cpo;
    static constexpr uint32_t nchunks16 = (rows * 2u * D + 15u) / 16u;
    static constexpr uint32_t pitch_a = (nchunks16 + per - 1u) / per + 1u, pitch_b = 16u + 1u + (cpo * 3u + 12u) / per + 1u;
    static constexpr uint32_t pitch = pitch_a > pitch_b ? pitch_a : pitch_b;
    static constexpr uint32_t plane_t = ((per + 3u) * pitch * 16u + 63u) & ~63u;
    /* sub-planes */
    static constexpr uint32_t rs = padded ? 96u : mfm3_row_stride(D), sp = MFM3_SP;
    static constexpr uint32_t plane_pitch = chunk_rows ? plane_t : 4u * sp, buf_pitch = 2u * plane_pitch;
    /* byte offset (from the lane's base) of the B fragment of column group g, k-step kq */
    static constexpr uint32_t ofs(uint32_t g, uint32_t kq)
    {
        if (chunk_rows) {
            const uint32_t x = cpo * g + 4u * kq;
            return ((x % per) * pitch + 1u + x / per) * 16u;
        }
        const uint32_t c = (64u * kq) / row_bytes, w = (64u * kq) % row_bytes;
        return ((g + c) & 3u) * sp + (1u + ((g + c) >> 2)) * rs + w;
    }
};

/* KQ: k-steps of 64 elements; NCH: 16-byte staging chunks per thread and tile; AHM >= 0: ah_mask as a compile-time constant;
 * DFIX > 0: the decimation as a compile-time constant with the sub-planes at the fixed 4096-byte pitch - every B-fragment
 * address is then "lane register + instruction immediate" (ds_read has no SGPR operand: with run-time geometry each of
 * the 32 reads of a tile costs a v_add). */
/* IN8 > 0: the input is 8-bit IQ as it came off the wire (RTL-SDR, cs8 / cu8 files: multifm/rtl_sdr_if.c:146-148,
 * multifm/file_if.c:66-157), two bytes per sample, and IN8 is the shift of the first rounding.  The reference widens
 * such a sample to x = alpha * s + beta (s the byte as int8, after ^ 0x80 for the RTL-SDR's unsigned bytes; alpha = 128,
 * beta = 128 there, alpha = 1, beta = 0 / -127 for cs8 / cu8), so
 *     sum W * x = alpha * (256 * sum Wh * s + sum Wl * s) + beta * sum W      (mod 2^32)
 * - ONE byte plane of samples, which is the wire format itself: the image is a 16-byte copy, a k-step is two MFMAs
 * instead of four (6 instead of 12 for the decimation-96 geometry), the recombination one shift-add instead of two.  The
 * per-row constant (L.krow) carries beta * sum W and the rounding bias, divided by alpha: for alpha = 128 bits 29:14 of
 * the reference's sum are bits 22:7 of (hh << 8) + ll, IN8 = 7. */
/* RC: the lowest rotator class (MFM_RC_*) of the launch's channels.  MFM_RC_IDENT: every rotator is (16384, 0) for ever and
 * r14(f * 16384) = f - no rotator-table loads, no derotation, no second rounding (filter/direct_fir.c:406-413 done by doing
 * nothing); MFM_RC_FLIP: rotators are +-(16384, 0) - one packed multiply by +-1 per sample; MFM_RC_QUARTER: rotators on the
 * four axis points - a byte permute and a packed multiply; MFM_RC_GENERAL: tabulated, but a WAVE whose eight channels are
 * all exact still takes the permute-and-multiply form (the engine orders the rows by class). */
template <int KQ, bool DBG_IQ, int NCH, int AHM, int DFIX, int IN8, uint32_t RC = MFM_RC_GENERAL>
__global__ __launch_bounds__(MFM3_NT, 4) void mfm_channel_kernel_v3(const mfm_launch_v3 L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kg = lane >> 4, n = lane & 15u;
    using G = mfm3_geo<DFIX>;
    constexpr bool PADDED = DFIX != 0 && G::padded; /* decimation 25: 64-byte rows of 50 sample bytes, sample-wise staging */
    constexpr uint32_t NCHS = PADDED ? 4u : (uint32_t)NCH; /* dwords of staging-offset table per thread (the engine sizes LDS) */
    const uint32_t D = DFIX ? (uint32_t)DFIX : L.decim, row_bytes = PADDED ? 64u : 2u * D;
    const uint32_t rs = DFIX ? G::rs : L.rs;
    const uint32_t sp_pitch = DFIX ? G::sp : L.sp_pitch;
    const uint32_t plane_pitch = DFIX ? G::plane_pitch : L.plane_pitch, buf_pitch = DFIX ? G::buf_pitch : L.buf_pitch;
    const bool chunk_rows = DFIX ? G::chunk_rows : L.layout == 1u; /* mfm_kernel.h: decimations that are not multiples of 32 */
    const uint32_t t_per = DFIX ? G::per : L.t_per, t_pitch = DFIX ? G::pitch : L.t_pitch;
    const uint32_t ah_mask = AHM >= 0 ? (uint32_t)AHM : (uint32_t)__builtin_amdgcn_readfirstlane(L.ah_mask);
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(smem + L.lut_off) | (MFM3_LUT_MODE == 2 ? 1u : 0u);
    const bool pcm_sys = L.pcm_scope != 0u; /* launches of many channels write their PCM through (mfm3_store_pcm4) */

    /* atan LUT, once per workgroup: {T[i], T[i+1]-T[i]} pairs as the engine holds them (MFM3_LUT_MODE).  Requested here,
     * stored behind the first image (MFM3_PROLOGUE: everything a workgroup needs before its first matrix phase - table,
     * first image, tap fragments, row constants - is requested before anything is waited for; the kernel's start is a
     * chain of memory round trips during which every SIMD of the chip idles, 6 % of a 2^26-sample launch in round 2) */
    static_assert(MFM3_NT == 512, "one table dword per thread");
    const uint32_t lut_v = reinterpret_cast<const uint32_t *>(L.lut)[tid];
#if !MFM3_PROLOGUE
    reinterpret_cast<uint32_t *>(smem + L.lut_off)[MFM3_LUT_SLOT(tid)] = lut_v;
#endif

    /* staging: this thread owns the 16-byte chunks q = tid + j * 512 of every tile (4 samples = 8 bytes per byte
     * plane); where they go in the image never changes: row = 8q / row_bytes, sub-plane row & 3, slot row >> 2 */
    uint32_t *sta_s = reinterpret_cast<uint32_t *>(smem + L.sta_off);
    /* chunk-row layout: where the second copy of a chunk goes (rows 0..2 of a slot again behind the last row of the slot
     * before), or ~0 - behind the first table, only there when the layout is in use */
    uint32_t *sta2_s = sta_s + NCHS * MFM3_NT + 768u; /* behind the row constants, fold constants and exact-rotator table */
    /* decimation 25: a 16-byte chunk (4 samples; 8 as bytes) does not sit in one row, and where the tile's image starts in
     * it depends on the launch (the first unconsumed sample is L.hist samples into the buffer): sample k of this thread's
     * chunk is image sample C * tid + k - delta, delta = (L.hist - LEAD * D) mod C; its two plane bytes go to row i / 25,
     * column 2 * (i % 25).  Eight 16-bit LDS offsets per thread (samples in front of or behind the image: a padding
     * column nobody reads). */
    uint16_t *sta16_s = reinterpret_cast<uint16_t *>(sta_s);
    if constexpr (PADDED) {
        constexpr int C = IN8 ? 8 : 4;
        const int delta = ((int)L.hist - (int)(MFM_V3_LEAD * 25u)) & (C - 1);
#pragma unroll
        for (int k = 0; k < C; k++) {
            const int i = C * (int)tid + k - delta;
            uint32_t off = 64u; /* behind the 64 bytes of row 0 that the fragments read */
            if (i >= 0 && i < (int)(G::rows * 25u)) {
                const uint32_t row = (uint32_t)i / 25u, col = ((uint32_t)i % 25u) * 2u;
                off = (row & 3u) * sp_pitch + (row >> 2) * rs + col;
            }
            sta16_s[tid * 8u + (uint32_t)k] = (uint16_t)off;
        }
    }
#pragma unroll
    for (int j = 0; j < NCH && !PADDED; j++) {
        const uint32_t p8 = (tid + (uint32_t)j * MFM3_NT) * (IN8 ? 16u : 8u); /* plane bytes in front of the chunk */
        if (chunk_rows) {
            const uint32_t sc = p8 >> 4, r = sc % t_per, c = sc / t_per;
            sta_s[j * MFM3_NT + tid] = (r * t_pitch + c) * 16u + (p8 & 15u);
            sta2_s[j * MFM3_NT + tid] = (r < 3u && c >= 1u) ? ((r + t_per) * t_pitch + c - 1u) * 16u + (p8 & 15u) : 0xffffffffu;
        } else {
            const uint32_t row = p8 / row_bytes, colb = p8 % row_bytes;
            sta_s[j * MFM3_NT + tid] = (row & 3u) * sp_pitch + (row >> 2) * rs + colb;
        }
    }

    /* B fragments: lane part of the address; the (column group, k-step) part is wave uniform:
     * row = LEAD + 4n + g + cross -> sub-plane (g + cross) & 3, slot 1 + n + ((g + cross) >> 2) */
    uint32_t ofs[4][KQ], ofs_w[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; kq++) {
        const uint32_t c = DFIX ? (64u * kq) / row_bytes : L.cross[kq], w = DFIX ? (64u * kq) % row_bytes : L.within[kq];
        static_assert(!PADDED || (KQ == 6 && NCH == 1), "decimation 25: six rows per window, one staging chunk per thread");
#pragma unroll
        for (int g = 0; g < 4; g++) {
            ofs[g][kq] = ((g + c) & 3u) * sp_pitch + (1u + ((g + c) >> 2)) * rs + w;
        }
        ofs_w[kq] = ((3u + c) & 3u) * sp_pitch + ((3u + c) >> 2) * rs + w; /* column group 3, four rows earlier */
        if (chunk_rows) {
            /* the window of output 4n + g starts t_per * (n + 1) + (t_per / 4) * g chunks into the image; k-step kq, lane
             * part kg: 4 kq + kg chunks further - row (x % t_per) + kg, slot n + 1 + x / t_per */
            const uint32_t cpo = t_per >> 2;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t x = cpo * (uint32_t)g + 4u * (uint32_t)kq;
                ofs[g][kq] = ((x % t_per) * t_pitch + 1u + x / t_per) * 16u;
            }
            const uint32_t xw = cpo * 3u + 4u * (uint32_t)kq; /* column group 3, four outputs earlier: slot n */
            ofs_w[kq] = ((xw % t_per) * t_pitch + xw / t_per) * 16u;
        }
    }
    const uint32_t lb0 = chunk_rows ? 16u * n + 16u * t_pitch * kg : n * rs + 16u * kg;

    auto stage_load = [&](uint32_t tile, int j) -> uint4 {
        /* 4 samples of the image of `tile`, which starts LEAD rows in front of the tile's first output (the first
         * unconsumed sample sits L.hist samples into the buffer).  Only a readable address is needed: samples before
         * the buffer feed nothing but outputs in front of the recomputed one (and, at the start of a stream, that one:
         * replaced by the zero history), samples past n_avail only outputs >= n_new (never stored) or zero-padded
         * taps.  Chunks past the image all read the tile's first line (one cache line per wave). */
        const uint32_t q = tid + (uint32_t)j * MFM3_NT;
        int gs = (int)(tile * MFM_V3_OT * D + L.hist) - (int)(MFM_V3_LEAD * D) + (IN8 ? 8 : 4) * (int)(q < L.nstage4 ? q : 0u);
        if constexpr (PADDED) {
            gs -= ((int)L.hist - (int)(MFM_V3_LEAD * 25u)) & ((IN8 ? 8 : 4) - 1); /* chunks are 16-byte aligned in the buffer */
        }
        gs = gs < 0 ? 0 : gs;
        gs = gs > (int)L.x_last4 ? (int)L.x_last4 : gs;
#if MFM3_NONTEMPORAL & 1
        const mfm_v4i ld = __builtin_nontemporal_load(
            reinterpret_cast<const mfm_v4i *>(reinterpret_cast<const uint8_t *>(L.x) + ((uint32_t)gs << (IN8 ? 1 : 2))));
        return make_uint4((uint32_t)ld[0], (uint32_t)ld[1], (uint32_t)ld[2], (uint32_t)ld[3]);
#else
        return *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(L.x) + ((uint32_t)gs << (IN8 ? 1 : 2)));
#endif
    };
    auto stage_store = [&](uint32_t buf, int j, const uint4 &v) {
        if constexpr (PADDED) {
            if (tid < L.nstage4) {
                const uint4 o = *reinterpret_cast<const uint4 *>(sta16_s + tid * 8u);
                uint8_t *img = smem + buf * buf_pitch;
                const uint32_t vv[4] = { v.x, v.y, v.z, v.w }, oo[4] = { o.x, o.y, o.z, o.w };
                if (IN8) {
                    const uint32_t m = L.in8_xor;
#pragma unroll
                    for (int k = 0; k < 4; k++) { /* two samples (two bytes each) per dword */
                        const uint32_t w = vv[k] ^ m;
                        *reinterpret_cast<uint16_t *>(img + (oo[k] & 0xffffu)) = (uint16_t)w;
                        *reinterpret_cast<uint16_t *>(img + (oo[k] >> 16)) = (uint16_t)(w >> 16);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) { /* one sample per dword: (I_lo, I_hi, Q_lo, Q_hi) */
                        const uint32_t off = (k & 1) ? oo[k >> 1] >> 16 : oo[k >> 1] & 0xffffu;
                        const uint32_t hi = __builtin_amdgcn_perm(vv[k], vv[k], 0x0c0c0301u);
                        const uint32_t lo = __builtin_amdgcn_perm(vv[k], vv[k], 0x0c0c0200u) ^ 0x8080u;
                        *reinterpret_cast<uint16_t *>(img + off) = (uint16_t)hi;
                        *reinterpret_cast<uint16_t *>(img + off + plane_pitch) = (uint16_t)lo;
                    }
                }
            }
            (void)j;
            return;
        }
        if (IN8) {
            if (tid + (uint32_t)j * MFM3_NT < L.nstage4) {
                const uint32_t m = L.in8_xor; /* 0x80808080: unsigned bytes -> int8 */
                const uint4 w = make_uint4(v.x ^ m, v.y ^ m, v.z ^ m, v.w ^ m);
                *reinterpret_cast<uint4 *>(smem + buf * buf_pitch + sta_s[j * MFM3_NT + tid]) = w;
                if (chunk_rows) {
                    const uint32_t a2 = sta2_s[j * MFM3_NT + tid];
                    if (a2 != 0xffffffffu) {
                        *reinterpret_cast<uint4 *>(smem + buf * buf_pitch + a2) = w;
                    }
                }
            }
        } else if (tid + (uint32_t)j * MFM3_NT < L.nstage4) {
            uint2 hi, lo;
            hi.x = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
            hi.y = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
            lo.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u) ^ 0x80808080u;
            lo.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u) ^ 0x80808080u;
            uint8_t *base = smem + buf * buf_pitch + sta_s[j * MFM3_NT + tid];
            *reinterpret_cast<uint2 *>(base) = hi;
            *reinterpret_cast<uint2 *>(base + plane_pitch) = lo;
            if (chunk_rows) {
                const uint32_t a2 = sta2_s[j * MFM3_NT + tid];
                if (a2 != 0xffffffffu) {
                    uint8_t *b2 = smem + buf * buf_pitch + a2;
                    *reinterpret_cast<uint2 *>(b2) = hi;
                    *reinterpret_cast<uint2 *>(b2 + plane_pitch) = lo;
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
    mfm_v4i a_h[KQ], a_l[KQ];
    /* 128 * sum(W) + 8192 of the wave's 16 rows: 64 bytes of LDS per wave (read back as the initial value of the low
     * accumulator of every column group; four registers that need not be live through the epilogue) */
    mfm_v4i *krow_s = reinterpret_cast<mfm_v4i *>(smem + L.sta_off + NCHS * MFM3_NT * 4u) + wave * 4u + kg;
    uint32_t slice_loaded = 0xffffffffu;
    {
        uint4 v[NCH];
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            v[j] = stage_load(tile, j);
        }
#if MFM3_PROLOGUE
        /* the tap fragments and row constants of the first item, behind the image in the queue */
        const uint32_t rb0 = slice * 8u + wave;
        mfm_v4i krow0 = { 0, 0, 0, 0 };
        if (rb0 < L.nrb) {
            const mfm_v4i *ap = reinterpret_cast<const mfm_v4i *>(L.afrag) + (size_t)rb0 * KQ * 2 * 64 + mfm3_opaque(lane);
#pragma unroll
            for (int kq = 0; kq < KQ; kq++) {
                a_h[kq] = ap[(kq * 2 + 0) * 64];
                a_l[kq] = ap[(kq * 2 + 1) * 64];
            }
            krow0 = *reinterpret_cast<const mfm_v4i *>(L.krow + (size_t)rb0 * 16 + 4 * mfm3_opaque(kg));
        }
        reinterpret_cast<uint32_t *>(smem + L.lut_off)[MFM3_LUT_SLOT(tid)] = lut_v;
#endif
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            stage_store(0, j, v[j]);
        }
#if MFM3_PROLOGUE
        if (rb0 < L.nrb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (n == 0) {
                *krow_s = krow0; /* only this wave reads it */
            }
#pragma unroll
            for (int kq = 0; kq < KQ; kq++) {
                asm volatile("" : "+v"(a_h[kq]), "+v"(a_l[kq]));
            }
            slice_loaded = slice;
        }
#endif
    }
    __syncthreads();

    /* Wave priorities.  (1) The second half of a workgroup loses the arbitration for the SIMD to the first (older) half on
     * every phase and reaches each barrier ~1300 cycles later (s_memtime stamps, profiles/r02_v3_phases.txt): priority 1
     * instead of 0 for it in the matrix phase.  (2) A wave in its epilogue goes ahead of waves in their matrix phase
     * (priority 2): the epilogue is a stream of VALU instructions that needs every issue slot it can get, the matrix phase
     * issues one MFMA every 32 cycles and hides what it has in between in the MFMA's shadow - it loses little by standing
     * back, the epilogue's wave (and with it the workgroup's next barrier) gains.  Worth 3-4 % at 64 channels, 5 % at 1024
     * (same-box A/B against the static form, profiles/r02_bench_table.txt's run). */
    const int prio_matrix = wave >= 4 ? 1 : 0;
    /* behind them, per (wave, kg, channel of the lane): {byte offset of table position mu + lam, 8 * lam} */
    uint2 *fold_s = reinterpret_cast<uint2 *>(smem + L.sta_off + NCHS * MFM3_NT * 4u + 512u) + (wave * 4u + kg) * 2u;
    /* behind them, per (wave, kg): exact rotators in their general form - per channel of the lane four byte selectors and
     * four sign words, one per column group (2 KB) */
    uint32_t *xq_s = reinterpret_cast<uint32_t *>(smem + L.sta_off + NCHS * MFM3_NT * 4u + 1024u) + (wave * 4u + kg) * 16u;
    uint32_t cur = 0;
    bool first_of_chunk = true;

    /* per-lane state of the chunk: two channels */
    uint32_t kb8[2] = { 0, 0 };   /* byte offset into the rotator table of the entry of (this tile's first output + 4n) */
    uint32_t voff[2] = { 0, 0 };  /* byte offset into pcm of (channel, this tile's first output + 4n) */
    uint32_t hist[2] = { 0, 0 };  /* lanes n = 0: filtered sample of the output in front of this tile */
    bool ch_ok[2] = { false, false };
    bool w_exact = false;         /* wave uniform: all eight channels of the wave have exact rotators (MFM_RC_FLIP / _IDENT) */

    /* what follows a tile in the workgroup's stream: the next tile of the chunk, or the first tile of the workgroup's
     * next item */
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
    /* EARLY (the 8-bit form has the registers for it): the image of the tile after next is requested as soon as the
     * staging registers are free - behind the barrier in the middle of a tile - and so has the epilogue and the next
     * matrix phase to arrive, instead of one matrix phase.  (Knock-out builds put the wait for the image at a fifth of
     * the int16 kernel's time, profiles/r02_knockout_v3.txt.) */
    constexpr bool EARLY = IN8 != 0 || MFM3_EARLY16;
    uint4 pre[NCH];
    if (EARLY) {
        uint32_t a_item = item, a_chunk = chunk, a_slice = slice, a_tile = tile, a_tend = tend;
        bool a_first = false;
        const bool a_valid = advance(a_item, a_chunk, a_slice, a_tile, a_tend, a_first);
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            pre[j] = stage_load(a_valid ? a_tile : tile, j);
        }
    }

    while (true) {
        /* its image is written to the other buffer between this tile's matrix phase and its epilogue */
        uint32_t n_item = item, n_chunk = chunk, n_slice = slice, n_tile = tile, n_tend = tend;
        bool n_first = false;
        const bool n_valid = advance(n_item, n_chunk, n_slice, n_tile, n_tend, n_first);
        if (!EARLY) {
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                pre[j] = stage_load(n_valid ? n_tile : tile, j);
            }
            __builtin_amdgcn_sched_barrier(MFM3_SCHED_ALL_BUT_VMEM);
        }

        if (prio_matrix) {
            __builtin_amdgcn_s_setprio(1);
        } else {
            __builtin_amdgcn_s_setprio(0);
        }
        const uint32_t rb = slice * 8u + wave;
        const bool rb_valid = rb < L.nrb; /* wave uniform */
        const uint32_t lb = lb0 + cur * buf_pitch;
        const uint8_t *img = smem;
        const uint32_t first_out = tile * MFM_V3_OT;

        uint32_t f[4][2];
        uint4 rva[2][MFM3_ROT4 ? 1 : 2];
        if (rb_valid) {
            const uint32_t ch0 = rb * 8u + 2u * kg;
            if (slice != slice_loaded) {
                const mfm_v4i *ap = reinterpret_cast<const mfm_v4i *>(L.afrag) + (size_t)rb * KQ * 2 * 64 + mfm3_opaque(lane);
#pragma unroll
                for (int kq = 0; kq < KQ; kq++) {
                    a_h[kq] = ap[(kq * 2 + 0) * 64];
                    a_l[kq] = ap[(kq * 2 + 1) * 64];
                }
                const mfm_v4i krow = *reinterpret_cast<const mfm_v4i *>(L.krow + (size_t)rb * 16 + 4 * mfm3_opaque(kg));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (n == 0) {
                    *krow_s = krow; /* only this wave reads it */
                }
#pragma unroll
                for (int kq = 0; kq < KQ; kq++) {
                    asm volatile("" : "+v"(a_h[kq]), "+v"(a_l[kq]));
                }
                slice_loaded = slice;
            }

            /* one column group: 16 rows x 16 columns x 64*KQ elements, four byte-plane products (acc = hh, md, ll) */
            auto mfma_chain = [&](const uint32_t (&o)[KQ], mfm_v4i (&acc)[3]) {
                mfm_v4i hh = { 0, 0, 0, 0 }, md = { 0, 0, 0, 0 }, ll = *krow_s;
                if constexpr (IN8 != 0) {
                    /* one sample plane: hh = sum Wh * s, ll = sum Wl * s + row constant */
                    mfm_v4i b[2];
                    b[0] = *reinterpret_cast<const mfm_v4i *>(img + lb + o[0]);
#pragma unroll
                    for (int kq = 0; kq < KQ; kq++) {
                        const int cb = kq & 1, nb = cb ^ 1;
                        if (kq + 1 < KQ) {
                            b[nb] = *reinterpret_cast<const mfm_v4i *>(img + lb + o[kq + 1]);
                        }
                        if ((ah_mask >> kq) & 1u) {
                            hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], b[cb], hh, 0, 0, 0);
                        }
                        ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], b[cb], ll, 0, 0, 0);
                    }
                    acc[0] = hh;
                    acc[1] = md;
                    acc[2] = ll;
                    return;
                }
                mfm_v4i bh[2], bl[2];
                bh[0] = *reinterpret_cast<const mfm_v4i *>(img + lb + o[0]);
                bl[0] = *reinterpret_cast<const mfm_v4i *>(img + lb + o[0] + plane_pitch);
#pragma unroll
                for (int kq = 0; kq < KQ; kq++) {
                    const int cb = kq & 1, nb = cb ^ 1;
                    if (kq + 1 < KQ) {
                        bh[nb] = *reinterpret_cast<const mfm_v4i *>(img + lb + o[kq + 1]);
                        bl[nb] = *reinterpret_cast<const mfm_v4i *>(img + lb + o[kq + 1] + plane_pitch);
                    }
                    if ((ah_mask >> kq) & 1u) {
                        hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], bh[cb], hh, 0, 0, 0);
                        md = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[kq], bl[cb], md, 0, 0, 0);
                    }
                    ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], bl[cb], ll, 0, 0, 0);
                    md = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[kq], bh[cb], md, 0, 0, 0);
                }
                acc[0] = hh;
                acc[1] = md;
                acc[2] = ll;
            };
            /* recombination and the first Q14 rounding -> packed filtered samples of the lane's two channels */
            auto finish = [&](const mfm_v4i (&acc)[3], uint32_t fout[2]) {
                uint32_t a_re[2], a_im[2];
                if constexpr (IN8 != 0) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a_re[c]) : "v"(acc[0][2 * c]), "v"(acc[2][2 * c]));
                        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a_im[c]) : "v"(acc[0][2 * c + 1]), "v"(acc[2][2 * c + 1]));
                    }
                    mfm3_round_pack2_sh<IN8 ? IN8 : 14>(a_re, a_im, fout);
                    return;
                }
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    a_re[c] = mfm3_combine(acc[0][2 * c], acc[1][2 * c], acc[2][2 * c]);
                    a_im[c] = mfm3_combine(acc[0][2 * c + 1], acc[1][2 * c + 1], acc[2][2 * c + 1]);
                }
                mfm3_round_pack2(a_re, a_im, fout);
            };
            /* MFMA -> VALU read hazard: 16 wait states cover a 16x16x64 MFMA (hipcc has been seen to leave it unpadded) */
            auto settle = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            auto column_group = [&](const uint32_t (&o)[KQ], uint32_t fout[2]) {
                mfm_v4i acc[3];
                mfma_chain(o, acc);
                settle();
                finish(acc, fout);
            };

            if (first_of_chunk) {
                /* ---- chunk set-up: where the lane's two channels stand in their rotator tables and in the output,
                 *      and the filtered sample in front of the chunk ---- */
                uint32_t wrx[2], wry[2], kbg[2], sgn[2], cls[2], selq[2], sgq[2];
                uint2 fog[2];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const uint32_t chn = ch0 + c;
                    ch_ok[c] = chn < L.nchan;
                    const uint32_t chs = ch_ok[c] ? chn : 0u;
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
                    /* an exact rotator: (16384, 0) for ever, or alternating with (-16384, 0) - then table position k holds
                     * (-1)^k * 16384 (pre-period 0, even period) and a tile's first output (64 * tile + 4n) has the parity
                     * of k0 throughout the chunk.  Low half: the factor of the lane's even outputs (column groups 0, 2),
                     * high half: of the odd ones. */
                    const uint32_t rcw = ch_ok[c] ? ip[7] : MFM_RC_IDENT;
                    cls[c] = rcw & 15u;
                    const uint32_t even = (cls[c] == MFM_RC_FLIP && (k0 & 1u)) ? 0xffffu : 1u;
                    const uint32_t odd = (cls[c] == MFM_RC_FLIP && !(k0 & 1u)) ? 0xffffu : 1u;
                    sgn[c] = even | (odd << 16);
                    /* the general form of an exact rotator: output k0 + g (g = 0..3, the lane's column groups) is rotated by
                     * m = turns * (k0 + g) quarter turns, f * j^m = halves swapped for odd m, then signs - selector and sign
                     * word of column group g, computed by lane n = g */
                    const uint32_t mq = ((rcw >> 4) * (k0 + n)) & 3u;
                    selq[c] = (mq & 1u) ? 0x01000302u : 0x03020100u;
                    sgq[c] = mq == 0u ? 0x00010001u : mq == 1u ? 0x0001ffffu : mq == 2u ? 0xffffffffu : 0xffff0001u;
                    voff[c] = (ip[6] * L.out_stride + first_out + 4u * n) * 2u; /* ip[6]: the row this channel's output goes to */
                    if (first_out == 0 && L.hist == 0) {
                        hist[c] = 0; /* nothing in front: multifm/fm_demod.c:16-17,29, the last sample starts at zero */
                        wrx[c] = wry[c] = 0;
                    } else {
                        /* rotator entry of the output in front (the entry in front of a period is not the period's last
                         * one: position mu is reached from mu - 1 the first time and from mu + lam - 1 ever after) */
                        const uint32_t kw = (k0 != mu || kabs == (uint64_t)mu) ? k0 - 1u : mu + lam - 1u;
#if MFM3_ROT4
                        const uint32_t r = reinterpret_cast<const uint32_t *>(L.rot)[inf.x + kw];
                        wrx[c] = mfm3_rot_x(r);
                        wry[c] = mfm3_rot_y(r);
#else
                        const uint2 e = reinterpret_cast<const uint2 *>(L.rot)[inf.x + kw];
                        wrx[c] = e.x;
                        wry[c] = e.y;
#endif
                    }
                }
                if constexpr (RC == MFM_RC_GENERAL) {
                    /* Rows are ordered by rotator class (the engine), so all but one or two waves of a mixed channel set
                     * hold exact rotators only: such a wave derotates with the packed sign multiply, its table loads all
                     * read the table's first line (the loads themselves stay - every vector memory operation of a tile
                     * stays countable for s_waitcnt vmcnt) and its sign words take the place of the
                     * fold constants. */
                    w_exact = MFM3_WAVE_EXACT &&
                              __builtin_amdgcn_ballot_w64(cls[0] == MFM_RC_GENERAL || cls[1] == MFM_RC_GENERAL) == 0;
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        kb8[c] = w_exact ? 0u : kbg[c];
                        if (n == 0) {
                            fold_s[c] = fog[c]; /* only this wave reads it */
                        }
                    }
                } else if constexpr (RC == MFM_RC_FLIP) {
                    kb8[0] = sgn[0];
                    kb8[1] = sgn[1];
                }
                if ((RC == MFM_RC_GENERAL && w_exact) || RC == MFM_RC_QUARTER) {
                    if (n < 4u) {
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            xq_s[c * 8 + n] = selq[c]; /* only this wave reads them */
                            xq_s[c * 8 + 4 + n] = sgq[c];
                        }
                    }
                }
                if (first_out != 0 || L.hist != 0) {
                    /* column group 3 one sub-plane row down: lane n computes output first_out - 4 + 4n + 3; n = 0 is
                     * the output in front of the chunk - for the first chunk of a launch the last output of the launch
                     * before, from the L.hist samples kept in front of the first unconsumed one */
                    uint32_t fw[2], qw[2];
                    uint32_t ow[KQ];
#pragma unroll
                    for (int kq = 0; kq < KQ; kq++) {
                        ow[kq] = ofs_w[kq];
                    }
                    column_group(ow, fw);
                    derotate2(fw, wrx, wry, qw);
                    hist[0] = qw[0];
                    hist[1] = qw[1];
                }
            }

            if constexpr (DFIX != 0 && KQ == 4 && AHM == 0x6 && IN8 == 0) {
                /* ---- the four column groups as hand-scheduled blocks (mfm3_group_d96): software pipelined by one group,
                 *      B fragments requested two k-steps ahead ---- */
                /* ofs[g][kq] as compile-time constants */
#define MFM3_OFS(g, kq) ((int)G::ofs((g), (kq)))
                constexpr int LO = (int)G::plane_pitch;
                const uint32_t ka = (uint32_t)(uintptr_t)krow_s;
                mfm_v4i ph, pl, qh, ql, rh = { 0, 0, 0, 0 }, rl = { 0, 0, 0, 0 }; /* three rotating fragment buffers */
                /* k-steps 0, 1 of group 0 (the blocks request those of the following group themselves) */
                asm volatile("ds_read_b128 %[x0h], %[lb] offset:%[a]\n\t"
                             "ds_read_b128 %[x0l], %[lb] offset:%[al]\n\t"
                             "ds_read_b128 %[x1h], %[lb] offset:%[b]\n\t"
                             "ds_read_b128 %[x1l], %[lb] offset:%[bl]\n\t"
                             : [x0h] "=&v"(ph), [x0l] "=&v"(pl), [x1h] "=&v"(qh), [x1l] "=&v"(ql)
                             : [lb] "v"(lb), [a] "n"(MFM3_OFS(0, 0)), [al] "n"(MFM3_OFS(0, 0) + LO), [b] "n"(MFM3_OFS(0, 1)),
                               [bl] "n"(MFM3_OFS(0, 1) + LO)
                             : "memory");
                mfm_v4i acc0[3], acc1[3];
                uint32_t t[4];
                const mfm_v4i none[3] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
                /* (decimation 96: the windows of consecutive outputs overlap by exactly one k-step; decimation 40 does not) */
                constexpr bool TOEP = MFM3_TOEPLITZ != 0 && MFM3_OFS(0, 3) == MFM3_OFS(1, 0) && MFM3_OFS(1, 3) == MFM3_OFS(2, 0) &&
                                      MFM3_OFS(2, 3) == MFM3_OFS(3, 0);
                if constexpr (TOEP) {
                    /* the next group's k-step 0 is this group's k-step 3 (same LDS bytes): p serves both, q and r alternate */
                    mfm3_group_d96<false, true, LO, MFM3_OFS(0, 2), MFM3_OFS(0, 3), MFM3_OFS(1, 0), MFM3_OFS(1, 1), true>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], ph, pl, qh, ql, rh, rl, none, acc0, t);
                    mfm3_group_d96<true, true, LO, MFM3_OFS(1, 2), MFM3_OFS(1, 3), MFM3_OFS(2, 0), MFM3_OFS(2, 1), true>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], ph, pl, rh, rl, qh, ql, acc0, acc1, t);
                    f[0][0] = t[0];
                    f[0][1] = t[2];
                    mfm3_group_d96<true, true, LO, MFM3_OFS(2, 2), MFM3_OFS(2, 3), MFM3_OFS(3, 0), MFM3_OFS(3, 1), true>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], ph, pl, qh, ql, rh, rl, acc1, acc0, t);
                    f[1][0] = t[0];
                    f[1][1] = t[2];
                    mfm3_group_d96<true, false, LO, MFM3_OFS(3, 2), MFM3_OFS(3, 3), 0, 0>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], ph, pl, rh, rl, qh, ql, acc0, acc1, t);
                    f[2][0] = t[0];
                    f[2][1] = t[2];
                } else {
                    mfm3_group_d96<false, true, LO, MFM3_OFS(0, 2), MFM3_OFS(0, 3), MFM3_OFS(1, 0), MFM3_OFS(1, 1)>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], ph, pl, qh, ql, rh, rl, none, acc0, t);
                    mfm3_group_d96<true, true, LO, MFM3_OFS(1, 2), MFM3_OFS(1, 3), MFM3_OFS(2, 0), MFM3_OFS(2, 1)>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], qh, ql, rh, rl, ph, pl, acc0, acc1, t);
                    f[0][0] = t[0];
                    f[0][1] = t[2];
                    mfm3_group_d96<true, true, LO, MFM3_OFS(2, 2), MFM3_OFS(2, 3), MFM3_OFS(3, 0), MFM3_OFS(3, 1)>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], rh, rl, ph, pl, qh, ql, acc1, acc0, t);
                    f[1][0] = t[0];
                    f[1][1] = t[2];
                    mfm3_group_d96<true, false, LO, MFM3_OFS(3, 2), MFM3_OFS(3, 3), 0, 0>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], ph, pl, qh, ql, rh, rl, acc0, acc1, t);
                    f[2][0] = t[0];
                    f[2][1] = t[2];
                }
                /* rotator entries of this tile, four consecutive ones per channel: requested at the end of the matrix phase
                 * (its registers are all taken until here), needed behind the staging stores and the barrier */
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (RC == MFM_RC_GENERAL) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const uint8_t *rp = reinterpret_cast<const uint8_t *>(L.rot) + kb8[c];
                        rva[c][0] = *reinterpret_cast<const uint4 *>(rp);
#if !MFM3_ROT4
                        rva[c][1] = *reinterpret_cast<const uint4 *>(rp + 16);
#endif
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                settle();
                finish(acc1, f[3]);
#undef MFM3_OFS
            } else if constexpr (DFIX != 0 && KQ == 4 && AHM == 0x6 && IN8 != 0) {
                /* ---- the same pipeline for one sample plane (mfm3_group_d96_b8) ---- */
#define MFM3_OFS(g, kq) ((int)G::ofs((g), (kq)))
                const uint32_t ka = (uint32_t)(uintptr_t)krow_s;
                mfm_v4i pb, qb, rb2 = { 0, 0, 0, 0 }; /* three rotating fragment buffers */
                asm volatile("ds_read_b128 %[x0], %[lb] offset:%[a]\n\t"
                             "ds_read_b128 %[x1], %[lb] offset:%[b]\n\t"
                             : [x0] "=&v"(pb), [x1] "=&v"(qb)
                             : [lb] "v"(lb), [a] "n"(MFM3_OFS(0, 0)), [b] "n"(MFM3_OFS(0, 1))
                             : "memory");
                mfm_v4i acc0[3], acc1[3];
                uint32_t t[4];
                const mfm_v4i none[3] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
                constexpr bool TOEP = MFM3_TOEPLITZ != 0 && MFM3_OFS(0, 3) == MFM3_OFS(1, 0) && MFM3_OFS(1, 3) == MFM3_OFS(2, 0) &&
                                      MFM3_OFS(2, 3) == MFM3_OFS(3, 0);
                if constexpr (TOEP) {
                    mfm3_group_d96_b8<false, true, IN8, MFM3_OFS(0, 2), MFM3_OFS(0, 3), MFM3_OFS(1, 0), MFM3_OFS(1, 1), true>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], pb, qb, rb2, none, acc0, t);
                    mfm3_group_d96_b8<true, true, IN8, MFM3_OFS(1, 2), MFM3_OFS(1, 3), MFM3_OFS(2, 0), MFM3_OFS(2, 1), true>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], pb, rb2, qb, acc0, acc1, t);
                    f[0][0] = t[0];
                    f[0][1] = t[2];
                    mfm3_group_d96_b8<true, true, IN8, MFM3_OFS(2, 2), MFM3_OFS(2, 3), MFM3_OFS(3, 0), MFM3_OFS(3, 1), true>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], pb, qb, rb2, acc1, acc0, t);
                    f[1][0] = t[0];
                    f[1][1] = t[2];
                } else {
                    mfm3_group_d96_b8<false, true, IN8, MFM3_OFS(0, 2), MFM3_OFS(0, 3), MFM3_OFS(1, 0), MFM3_OFS(1, 1)>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], pb, qb, rb2, none, acc0, t);
                    mfm3_group_d96_b8<true, true, IN8, MFM3_OFS(1, 2), MFM3_OFS(1, 3), MFM3_OFS(2, 0), MFM3_OFS(2, 1)>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], qb, rb2, pb, acc0, acc1, t);
                    f[0][0] = t[0];
                    f[0][1] = t[2];
                    mfm3_group_d96_b8<true, true, IN8, MFM3_OFS(2, 2), MFM3_OFS(2, 3), MFM3_OFS(3, 0), MFM3_OFS(3, 1)>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], rb2, pb, qb, acc1, acc0, t);
                    f[1][0] = t[0];
                    f[1][1] = t[2];
                }
                /* rotator entries of this tile: requested in front of the last column group instead of behind the matrix
                 * phase - a column group more time to arrive (the knock-out build without these loads was 8 us faster: what
                 * they cost is their own latency).  This form has the 16 registers for it; the int16 form spills with it
                 * and loses more than it gains (126-129 against 123 us, same box). */
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (RC == MFM_RC_GENERAL) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const uint8_t *rp = reinterpret_cast<const uint8_t *>(L.rot) + kb8[c];
                        rva[c][0] = *reinterpret_cast<const uint4 *>(rp);
#if !MFM3_ROT4
                        rva[c][1] = *reinterpret_cast<const uint4 *>(rp + 16);
#endif
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TOEP) {
                    mfm3_group_d96_b8<true, false, IN8, MFM3_OFS(3, 2), MFM3_OFS(3, 3), 0, 0>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], pb, rb2, qb, acc0, acc1, t);
                } else {
                    mfm3_group_d96_b8<true, false, IN8, MFM3_OFS(3, 2), MFM3_OFS(3, 3), 0, 0>(
                        lb, ka, a_l[0], a_l[1], a_l[2], a_l[3], a_h[1], a_h[2], pb, qb, rb2, acc0, acc1, t);
                }
                f[2][0] = t[0];
                f[2][1] = t[2];
                settle();
                finish(acc1, f[3]);
#undef MFM3_OFS
            } else {
            /* The four column groups, one after the other (run-time geometry; the hand-scheduled blocks above are the
             * software-pipelined form for the fixed one). */
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint32_t og[KQ];
#pragma unroll
                for (int kq = 0; kq < KQ; kq++) {
                    og[kq] = ofs[g][kq];
                }
                column_group(og, f[g]);
                if (g == 1) {
                    /* rotator entries of this tile, four consecutive ones per channel: requested half-way through
                     * the matrix phase (16 registers that need not be live before), needed behind it */
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RC == MFM_RC_GENERAL) {
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const uint8_t *rp = reinterpret_cast<const uint8_t *>(L.rot) + kb8[c];
                            rva[c][0] = *reinterpret_cast<const uint4 *>(rp);
#if !MFM3_ROT4
                            rva[c][1] = *reinterpret_cast<const uint4 *>(rp + 16);
#endif
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
        }

        /* The next image goes to the other buffer between the matrix phase and the epilogue; after the barrier nobody
         * reads the current buffer any more.  (Measured against storing it at the very end of the tile, which gives the
         * loads a whole tile to arrive: this order is 4 % faster.) */
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            stage_store(cur ^ 1u, j, pre[j]);
        }
#if !MFM3_BARRIER_LATE
        __syncthreads();
#endif
        __builtin_amdgcn_s_setprio(2); /* epilogue */
        if (EARLY) {
            uint32_t a_item = n_item, a_chunk = n_chunk, a_slice = n_slice, a_tile = n_tile, a_tend = n_tend;
            bool a_first = false;
            const bool a_valid = n_valid && advance(a_item, a_chunk, a_slice, a_tile, a_tend, a_first);
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                pre[j] = stage_load(a_valid ? a_tile : tile, j);
            }
            __builtin_amdgcn_sched_barrier(MFM3_SCHED_ALL_BUT_VMEM);
        }
        if (rb_valid) {
            /* ---- one channel after the other (register pressure): derotation, discriminator, stores ---- */
            const uint32_t n_left = L.n_new - first_out; /* >= 1 */
            uint32_t q[4][2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if ((RC == MFM_RC_GENERAL && w_exact) || RC == MFM_RC_QUARTER) {
                    /* exact rotators, any of them (a wave of such channels in a launch that has others, or a whole launch
                     * with quarter turns among them): r14(f * rot) = f * j^m - swap the halves for odd m, then two signs */
                    const uint4 sel4 = *reinterpret_cast<const uint4 *>(xq_s + c * 8);
                    const uint4 sg4 = *reinterpret_cast<const uint4 *>(xq_s + c * 8 + 4);
                    const uint32_t sel[4] = { sel4.x, sel4.y, sel4.z, sel4.w }, sg[4] = { sg4.x, sg4.y, sg4.z, sg4.w };
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const uint32_t t = __builtin_amdgcn_perm(f[g][c], f[g][c], sel[g]);
                        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(q[g][c]) : "v"(t), "v"(sg[g]));
                    }
                } else if constexpr (RC == MFM_RC_GENERAL) {
                    /* derotation + second rounding, two column groups per call */
#pragma unroll
                    for (int h = 0; h < 2; h++) {
#if MFM3_ROT4
                        const uint32_t r0 = h ? rva[c][0].z : rva[c][0].x, r1 = h ? rva[c][0].w : rva[c][0].y;
                        const uint32_t fin[2] = { f[2 * h][c], f[2 * h + 1][c] };
                        const uint32_t rx[2] = { mfm3_rot_x(r0), mfm3_rot_x(r1) }, ry[2] = { mfm3_rot_y(r0), mfm3_rot_y(r1) };
#else
                        const uint4 e = rva[c][h];
                        const uint32_t fin[2] = { f[2 * h][c], f[2 * h + 1][c] };
                        const uint32_t rx[2] = { e.x, e.z }, ry[2] = { e.y, e.w };
#endif
                        uint32_t qo[2];
                        derotate2(fin, rx, ry, qo);
                        q[2 * h][c] = qo[0];
                        q[2 * h + 1][c] = qo[1];
                    }
                } else if constexpr (RC == MFM_RC_FLIP) {
                    /* the rotator is +-(16384, 0): r14(f * rot) = +-f with the int16 cast's wrap, one packed multiply by the
                     * sign of this output's parity (kb8 holds both signs here) */
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        q[g][c] = mfm3_sign_flip(f[g][c], kb8[c], g & 1);
                    }
                } else {
                    /* the rotator is (16384, 0) for ever: r14(f * 16384) = f */
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        q[g][c] = f[g][c];
                    }
                }
                /* discriminator: previous output = previous column group; for group 0 the neighbouring lane's group 3,
                 * and for lane n = 0 the last output of the previous tile */
                const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp((int)hist[c], (int)q[3][c], 0x111 /* row_shr:1 */,
                                                                          0xf, 0xf, false);
                const uint32_t pp[4] = { p0, q[0][c], q[1][c], q[2][c] };
                int s_re[4], s_im[4], pcm[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    mfm3_conj_mul(q[g][c], pp[g], &s_re[g], &s_im[g]);
                }
                mfm3_discriminate4(s_re, s_im, lut_addr, pcm);
                /* lane 0 of each row of 16 lanes gets lane 15's last sample: the next tile's history */
                hist[c] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q[3][c], 0x121 /* row_ror:1 */, 0xf, 0xf, true);
                if (n_left >= MFM_V3_OT) {
                    if (ch_ok[c]) {
                        uint2 w;
                        w.x = __builtin_amdgcn_perm((uint32_t)pcm[1], (uint32_t)pcm[0], 0x05040100u);
                        w.y = __builtin_amdgcn_perm((uint32_t)pcm[3], (uint32_t)pcm[2], 0x05040100u);
#if MFM3_NONTEMPORAL & 2
                        mfm3_store_pcm4(L.pcm, voff[c], w.x, w.y, pcm_sys);
#else
                        *reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(L.pcm) + voff[c]) = w;
#endif
                        if (DBG_IQ) {
                            *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(L.iq_dbg) + 2u * (size_t)voff[c]) =
                                make_uint4(q[0][c], q[1][c], q[2][c], q[3][c]);
                        }
                    }
                } else {
                    /* the last tile of the pass, partly filled */
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        if (ch_ok[c] && 4u * n + (uint32_t)g < n_left) {
                            *reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(L.pcm) + voff[c] + 2u * g) = (int16_t)pcm[g];
                            if (DBG_IQ) {
                                *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(L.iq_dbg) + 2u * (size_t)voff[c] + 4u * g) = q[g][c];
                            }
                        }
                    }
                }
#if MFM3_EPI_SERIAL
                /* A/B builds: the second channel's epilogue strictly behind the first's (rounds 2-3: a scheduling barrier kept
                 * the register pressure down).  Without it the compiler interleaves the two dependent chains - the division
                 * of one under the table reads of the other - at the same 128 registers: 1.5 % faster
                 * (profiles/r04_ab_epilogue_interleave.txt) */
                if (c == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            }
            /* nothing is carried to the next launch: it recomputes the output in front of it and folds its own
             * rotator position (mfm_launch_v3::hist, ::k_base) */
            /* next tile of the chunk: 64 outputs on */
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (RC == MFM_RC_GENERAL && !w_exact) {
                    const uint2 fo = fold_s[c]; /* at or past table position mu + lam the position folds back by lam */
                    kb8[c] += MFM_V3_OT * MFM3_ES;
                    kb8[c] = kb8[c] >= fo.x + 4u * MFM3_ES * n ? kb8[c] - fo.y : kb8[c];
                }
                voff[c] += MFM_V3_OT * 2u;
            }
        }

#if MFM3_BARRIER_LATE
        __syncthreads();
#endif
        if (!n_valid) {
            break;
        }
        cur ^= 1u;
        item = n_item;
        chunk = n_chunk;
        slice = n_slice;
        tile = n_tile;
        tend = n_tend;
        first_of_chunk = n_first;
    }
    mfm3_stamp_end(L, stamp_t0, stamp_r0);
}

/* The discriminator of this file on caller-supplied products (tests, and the engine's division self-test at commit):
 * thread t takes s[4t .. 4t + 3], exactly as a lane of the channel kernel takes its four outputs. */
__global__ __launch_bounds__(MFM3_NT) void mfm3_disc_test_kernel(const int *s_re, const int *s_im, int *pcm, uint32_t n4, const float2 *lut)
{
    __shared__ __attribute__((aligned(16))) uint32_t tbl[512];
    tbl[MFM3_LUT_SLOT(threadIdx.x)] = reinterpret_cast<const uint32_t *>(lut)[threadIdx.x];
    __syncthreads();
    const uint32_t t = blockIdx.x * MFM3_NT + threadIdx.x;
    if (t >= n4) {
        return;
    }
    int re[4], im[4], out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        re[i] = s_re[4 * t + i];
        im[i] = s_im[4 * t + i];
    }
    mfm3_discriminate4(re, im, (uint32_t)(uintptr_t)tbl, out);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        pcm[4 * t + i] = out[i];
    }
}

extern "C" hipError_t mfm_disc_test_v3(const int *s_re, const int *s_im, int *pcm, uint32_t n, const float2 *lut, hipStream_t stream)
{
    const uint32_t n4 = n / 4u; /* n is a multiple of 4 (the caller pads) */
    hipLaunchKernelGGL(mfm3_disc_test_kernel, dim3((n4 + MFM3_NT - 1u) / MFM3_NT), dim3(MFM3_NT), 0, stream, s_re, s_im, pcm, n4, lut);
    return hipGetLastError();
}

/*
 * Which instance runs a launch description: decided by its geometry fields only (kq, nstage4, decim, rs, sp_pitch,
 * ah_mask, in8), all fixed at commit.  The engine asks once per input format at commit, raises the instance's LDS limit
 * there (hipFuncSetAttribute is a driver call; round 2 paid it on every launch) and launches through the pointer.
 */
/* layout 3 (long filters): the instances live in mfm_kernel_v3l.hip, one translation unit per k-step count */
#define MFM3L_IMPORT(KQ_) extern "C" const void *mfm_v3l_instance_kq##KQ_(const mfm_launch_v3 *L, uint32_t nch);
MFM3L_IMPORT(4) MFM3L_IMPORT(6) MFM3L_IMPORT(8) MFM3L_IMPORT(9) MFM3L_IMPORT(10) MFM3L_IMPORT(11) MFM3L_IMPORT(12) MFM3L_IMPORT(14) MFM3L_IMPORT(16)
#undef MFM3L_IMPORT

extern "C" hipError_t mfm_select_channel_kernel_v3(const mfm_launch_v3 *L, int dbg_iq, const void **kfn_out)
{
    *kfn_out = nullptr;
    if (L->layout == 3u) {
        /* L->kq is a built count of k-steps (mfm_v3l_built_kq), the tap fragments are laid out for exactly that many */
        const uint32_t nch4 = (L->nstage4 + MFM3_NT - 1) / MFM3_NT;
        if (nch4 < 1 || nch4 > MFM_V3_CH_MAX || (L->ng != 1u && L->ng != 2u && L->ng != 4u) || (L->rb != 1u && L->rb != 2u) ||
            L->kq_used > L->kq || L->nh > L->kq ||
            L->kq != mfm_v3l_built_kq(L->kq) || (L->in8 != 0u && L->in8 != 7u && L->in8 != 14u) ||
            /* (a description with the LDS layout filled in: the plane pitch the instances' immediates assume) */
            (!L->shift && L->plane_pitch != 0u && L->plane_pitch != mfm_v3l_plane_pitch(L->rb))) {
            return hipErrorInvalidValue;
        }
        switch (L->kq) {
        case 4: *kfn_out = mfm_v3l_instance_kq4(L, nch4); break;
        case 6: *kfn_out = mfm_v3l_instance_kq6(L, nch4); break;
        case 8: *kfn_out = mfm_v3l_instance_kq8(L, nch4); break;
        case 9: *kfn_out = mfm_v3l_instance_kq9(L, nch4); break;
        case 10: *kfn_out = mfm_v3l_instance_kq10(L, nch4); break;
        case 11: *kfn_out = mfm_v3l_instance_kq11(L, nch4); break;
        case 12: *kfn_out = mfm_v3l_instance_kq12(L, nch4); break;
        case 14: *kfn_out = mfm_v3l_instance_kq14(L, nch4); break;
        case 16: *kfn_out = mfm_v3l_instance_kq16(L, nch4); break;
        default: break;
        }
        return *kfn_out ? hipSuccess : hipErrorInvalidValue;
    }
    const uint32_t nch = (L->nstage4 + MFM3_NT - 1) / MFM3_NT;
    if (nch < 1 || nch > MFM_V3_CH_MAX) {
        return hipErrorInvalidValue;
    }
#define MFM3_LAUNCH(KQ_, DBG_, NCH_, AHM_) MFM3_LAUNCH_F(KQ_, DBG_, NCH_, AHM_, 0, 0)
#define MFM3_LAUNCH_F(KQ_, DBG_, NCH_, AHM_, DFIX_, IN8_)                                                    \
    do {                                                                                                     \
        *kfn_out = reinterpret_cast<const void *>(&mfm_channel_kernel_v3<KQ_, DBG_, NCH_, AHM_, DFIX_, IN8_>); \
    } while (0)
    /* the fixed geometry also has instances for channel sets whose rotators are all exact (L->rc) */
#define MFM3_LAUNCH_RC(KQ_, NCH_, AHM_, DFIX_, IN8_)                                                         \
    do {                                                                                                     \
        if (L->rc == MFM_RC_IDENT) {                                                                         \
            *kfn_out = reinterpret_cast<const void *>(                                                       \
                &mfm_channel_kernel_v3<KQ_, false, NCH_, AHM_, DFIX_, IN8_, MFM_RC_IDENT>);                  \
        } else if (L->rc == MFM_RC_QUARTER) {                                                                \
            *kfn_out = reinterpret_cast<const void *>(                                                       \
                &mfm_channel_kernel_v3<KQ_, false, NCH_, AHM_, DFIX_, IN8_, MFM_RC_QUARTER>);                \
        } else if (L->rc == MFM_RC_FLIP) {                                                                   \
            *kfn_out = reinterpret_cast<const void *>(                                                       \
                &mfm_channel_kernel_v3<KQ_, false, NCH_, AHM_, DFIX_, IN8_, MFM_RC_FLIP>);                   \
        } else {                                                                                             \
            MFM3_LAUNCH_F(KQ_, false, NCH_, AHM_, DFIX_, IN8_);                                              \
        }                                                                                                    \
    } while (0)
#define MFM3_LAUNCH_N(KQ_, DBG_, AHM_)                                                                       \
    do {                                                                                                     \
        switch (nch) {                                                                                       \
        case 1: MFM3_LAUNCH(KQ_, DBG_, 1, AHM_); break;                                                      \
        case 2: MFM3_LAUNCH(KQ_, DBG_, 2, AHM_); break;                                                      \
        case 3: MFM3_LAUNCH(KQ_, DBG_, 3, AHM_); break;                                                      \
        case 4: MFM3_LAUNCH(KQ_, DBG_, 4, AHM_); break;                                                      \
        case 5: MFM3_LAUNCH(KQ_, DBG_, 5, AHM_); break;                                                      \
        case 6: MFM3_LAUNCH(KQ_, DBG_, 6, AHM_); break;                                                      \
        case 7: MFM3_LAUNCH(KQ_, DBG_, 7, AHM_); break;                                                      \
        default: MFM3_LAUNCH(KQ_, DBG_, 8, AHM_); break;                                                     \
        }                                                                                                    \
    } while (0)
#define MFM3_LAUNCH_D(KQ_, AHM_)                                                                             \
    do {                                                                                                     \
        if (dbg_iq) {                                                                                        \
            MFM3_LAUNCH_N(KQ_, true, -1);                                                                    \
        } else {                                                                                             \
            MFM3_LAUNCH_N(KQ_, false, AHM_);                                                                 \
        }                                                                                                    \
    } while (0)
    /* 8-bit input (L->in8 = 7 or 14, the first rounding's shift; the engine only asks when dbg_iq is off): half the
     * staging chunks per thread (16 bytes are 8 samples), run-time tap-plane mask except for the fixed geometry */
#define MFM3_LAUNCH_8(KQ_, IN8_)                                                                             \
    do {                                                                                                     \
        switch (nch) {                                                                                       \
        case 1: MFM3_LAUNCH_F(KQ_, false, 1, -1, 0, IN8_); break;                                            \
        case 2: MFM3_LAUNCH_F(KQ_, false, 2, -1, 0, IN8_); break;                                            \
        case 3: MFM3_LAUNCH_F(KQ_, false, 3, -1, 0, IN8_); break;                                            \
        default: MFM3_LAUNCH_F(KQ_, false, 4, -1, 0, IN8_); break;                                           \
        }                                                                                                    \
    } while (0)
#define MFM3_LAUNCH_8K(IN8_)                                                                                 \
    do {                                                                                                     \
        if (L->decim == 96 && L->kq == 4 && nch == 2 && L->ah_mask == 0x6u && L->rs == mfm3_row_stride(96) &&      \
            L->sp_pitch == MFM3_SP) {                                                                        \
            MFM3_LAUNCH_RC(4, 2, 0x6, 96, IN8_);                                                             \
        } else if (L->decim == 40 && L->kq == 4 && nch == 1 && L->ah_mask == 0x6u && geo40) {                \
            MFM3_LAUNCH_F(4, false, 1, 0x6, 40, IN8_);                                                       \
        } else {                                                                                             \
            switch (L->kq) {                                                                                 \
            case 1: MFM3_LAUNCH_8(1, IN8_); break;                                                           \
            case 2: MFM3_LAUNCH_8(2, IN8_); break;                                                           \
            case 4: MFM3_LAUNCH_8(4, IN8_); break;                                                           \
            default: return hipErrorInvalidValue;                                                            \
            }                                                                                                \
        }                                                                                                    \
    } while (0)
    /* decimation 40 (etc/multifm.json, etc/multifm_1ch.json at 1 MS/s) with the engine's chunk-row numbers */
    const bool geo40 = L->layout == 1u && L->t_per == mfm3_geo<40>::per && L->t_pitch == mfm3_geo<40>::pitch &&
                       L->plane_pitch == mfm3_geo<40>::plane_pitch && L->buf_pitch == mfm3_geo<40>::buf_pitch;
    if (L->layout == 2u) {
        /* decimation 25 (etc/pocsag_rtlsdr.json) on padded rows: fixed geometry, six k-steps; instances by which k-steps carry a
         * high-byte tap plane (a mask that is a subset of an instance's runs on it: a zero plane multiplies zeros) */
        if (L->decim != 25u || L->kq != 6u || L->rs != mfm3_geo<25>::rs || L->sp_pitch != mfm3_geo<25>::sp || (dbg_iq && L->in8 != 0u)) {
            return hipErrorInvalidValue;
        }
        if (dbg_iq) {
            /* a channel wants its filtered IQ beside the PCM (signalDebugFile, multifm/demod.c:75-81): int16 input only */
            if ((L->ah_mask & ~0xeu) == 0u) {
                *kfn_out = reinterpret_cast<const void *>(&mfm_channel_kernel_v3<6, true, 1, 0xe, 25, 0>);
            } else {
                *kfn_out = reinterpret_cast<const void *>(&mfm_channel_kernel_v3<6, true, 1, -1, 25, 0>);
            }
            return hipSuccess;
        }
#define MFM3_LAUNCH_25(IN8_)                                                                                 \
    do {                                                                                                     \
        if ((L->ah_mask & ~0x4u) == 0u) {                                                                    \
            MFM3_LAUNCH_F(6, false, 1, 0x4, 25, IN8_);                                                       \
        } else if ((L->ah_mask & ~0xeu) == 0u) {                                                             \
            MFM3_LAUNCH_F(6, false, 1, 0xe, 25, IN8_);                                                       \
        } else {                                                                                             \
            MFM3_LAUNCH_F(6, false, 1, -1, 25, IN8_);                                                        \
        }                                                                                                    \
    } while (0)
        if (L->in8 == 7u) {
            MFM3_LAUNCH_25(7);
        } else if (L->in8 == 14u) {
            MFM3_LAUNCH_25(14);
        } else if (L->in8 == 0u) {
            MFM3_LAUNCH_25(0);
        } else {
            return hipErrorInvalidValue;
        }
#undef MFM3_LAUNCH_25
        return hipSuccess;
    }
    if (L->in8) {
        if (dbg_iq || nch > 4 || (L->in8 != 7u && L->in8 != 14u)) {
            return hipErrorInvalidValue;
        }
        if (L->in8 == 7u) {
            MFM3_LAUNCH_8K(7);
        } else {
            MFM3_LAUNCH_8K(14);
        }
        return hipSuccess;
    }
    /* the 2.4 MS/s -> 25 kS/s geometry of the reference's configurations (decimation 96, 128-tap low-pass whose
     * outer k-steps fit one byte) with every address a compile-time constant */
#ifndef MFM3_NO_FIX
    if (L->decim == 96 && L->kq == 4 && nch == 4 && !dbg_iq && L->ah_mask == 0x6u && L->rs == mfm3_row_stride(96) &&
        L->sp_pitch == MFM3_SP) {
        MFM3_LAUNCH_RC(4, 4, 0x6, 96, 0);
        return hipSuccess;
    }
    if (L->decim == 40 && L->kq == 4 && nch == 2 && !dbg_iq && L->ah_mask == 0x6u && geo40) {
        MFM3_LAUNCH_F(4, false, 2, 0x6, 40, 0);
        return hipSuccess;
    }
#endif
    switch (L->kq) {
    case 1: MFM3_LAUNCH_D(1, -1); break;
    case 2: MFM3_LAUNCH_D(2, -1); break;
    case 4:
        if (L->ah_mask == 0x6u) {
            MFM3_LAUNCH_D(4, 0x6);
        } else {
            MFM3_LAUNCH_D(4, -1);
        }
        break;
    default: return hipErrorInvalidValue;
    }
#undef MFM3_LAUNCH_D
#undef MFM3_LAUNCH_8K
#undef MFM3_LAUNCH_8
#undef MFM3_LAUNCH_N
#undef MFM3_LAUNCH_RC
#undef MFM3_LAUNCH_F
#undef MFM3_LAUNCH
    return hipSuccess;
}

/* the sub-plane pitch the fixed-geometry instances of this kernel file are built for */
extern "C" uint32_t mfm_sp_pitch_v3(void)
{
    return MFM3_SP;
}

/* what the engine's rotator table for this kernel file has to look like */
extern "C" uint32_t mfm_rot_entry_bytes_v3(void)
{
    return MFM3_ES;
}

extern "C" hipError_t mfm_launch_channel_kernel_v3(const void *kfn, const mfm_launch_v3 *L, uint32_t lds_bytes, uint32_t grid,
                                                   hipStream_t stream)
{
    if (L->ntiles == 0) {
        return hipSuccess;
    }
    void *args[] = { const_cast<mfm_launch_v3 *>(L) };
    return hipLaunchKernel(kfn, dim3(grid), dim3(MFM3_NT), args, lds_bytes, stream);
}
