/*
 * mfm_v3_device.h - device-side pieces shared by the second-generation channel kernels (mfm_kernel_v3.hip: filters of up to
 * 128 taps; mfm_kernel_v3l.hip: 129..512 taps): the build switches, the exact Q14 / derotation / discriminator helpers
 * (filter/complex.h:30-62, filter/direct_fir.c:151-172,406-413, multifm/fm_demod.c:53-79, multifm/fast_atan2f.c:101-174)
 * and the work-item decoding.  Both kernel files produce the same bits because they run the same code from here.
 * Device code only; include after <hip/hip_runtime.h>, mfm_kernel.h and mfm_numerics.h.
 */
#pragma once

typedef int mfm_v4i __attribute__((ext_vector_type(4)));

#define MFM3_NT 512u
#ifndef MFM3_EARLY16
#define MFM3_EARLY16 1 /* 0: A/B builds - the int16 form requests the next image at the top of a tile, as before */
#endif
#ifndef MFM3_WAVE_EXACT
#define MFM3_WAVE_EXACT 1 /* 0: A/B builds - waves of exact rotators inside a general launch derotate through the table too */
#endif
#ifndef MFM3_PROLOGUE
#define MFM3_PROLOGUE 1 /* 0: A/B builds - table, first image, tap fragments fetched one after the other as in round 2 */
#endif
#ifndef MFM3_BARRIER_LATE
#define MFM3_BARRIER_LATE 0 /* 1: A/B builds - the tile's barrier behind the epilogue instead of in front of it */
#endif
#ifndef MFM3_ROT4
#define MFM3_ROT4 1 /* rotator-table entries of 4 bytes (rr | ri << 16) instead of 8 ({(rr, -ri), (ri, rr)}; 0: A/B builds): half
                       the table bytes per output for two cheap and one expensive instruction per entry (the two dot-product
                       operands are rebuilt in registers).  At 1024 channels the tables are what misses L2 (each workgroup
                       re-reads its 64 channels' periodic table parts every other tile while input and output stream
                       through): 5.25 -> 2.21 GB per launch together with the stores' hint below, and 3 % faster
                       (profiles/r04_traffic_1024ch.txt).  The engine builds what mfm_rot_entry_bytes_v3() says. */
#endif
#define MFM3_ES (MFM3_ROT4 ? 4u : 8u) /* bytes per rotator-table entry */
#ifndef MFM3_NONTEMPORAL
#define MFM3_NONTEMPORAL 2 /* bit 1: the PCM stores carry the non-temporal hint - the 8-byte stores of a tile leave L2 as whole
                              lines instead of being written back piecemeal (WRITE_SIZE 1.23 x the PCM bytes without it,
                              0.98 x with it) (mfm3_store_pcm4); bit 0 (A/B builds): the image loads
                              too - wrong, the slices of a chunk share the image through L2 (fetch + 40 %, 4 % slower) */
#endif
#ifndef MFM3_SP
#define MFM3_SP 4096u /* bytes between the four sub-planes of a byte plane at the fixed geometries (A/B builds: 4160 - not a
                         multiple of the LDS bank cycle, so that the image stores of rows r and r + 1 do not meet in one bank) */
#endif
#ifndef MFM3_EPI_SERIAL
#define MFM3_EPI_SERIAL 0
#endif
#ifndef MFM3_DIV_STEPS
#define MFM3_DIV_STEPS 1 /* 2: A/B builds - the division with a second residual step, as in rounds 1-3 */
#endif
#ifndef MFM3_TOEPLITZ
#define MFM3_TOEPLITZ 1 /* 0: A/B builds; 1: the int16 and the 8-bit D = 96 / 128-tap instance does not read a column group's k-step 0 again when it is the
                           previous group's k-step 3 (6 of a tile's 32 B-fragment reads per wave and plane) */
#endif
#ifndef MFM3_LUT_MODE
#define MFM3_LUT_MODE 1 /* the arctangent table in LDS: 1 {T, dT} pairs, one 8-byte read per output (banked over 64 dwords);
                         * 0 (rounds 2-3, A/B builds) T[256] then dT[256], one ds_read2st64_b32 (banked over 32): the table's
                         * data-dependent addresses were 88 % of the kernel's SQ_LDS_BANK_CONFLICT cycles, the pair form halves
                         * them (25 % -> 17 % of LDS cycles) at the same instruction count; 2 counter builds only - every lane
                         * reads its own bank (wrong PCM): the kernel without those conflicts, 1.5 % faster, which a table
                         * replicated per bank would give and LDS (2 x 78.8 KB per CU) has no room for.
                         * profiles/r04_lds_conflicts.txt */
#endif
#define MFM3_LUT_SLOT(t) (MFM3_LUT_MODE == 1 ? (t) : ((t) >> 1) + (((t)&1u) << 8))
#define MFM3_SCHED_ALL_BUT_VMEM 0x38F

/* (hh << 16) + (md << 8) + ll, two v_lshl_add_u32 (left to itself the compiler makes it two shifts and a three-operand
 * add).  The accumulators come straight from MFMAs and the compiler does not pad the MFMA -> VALU read hazard for what it
 * cannot see inside inline asm: callers put 16 wait states between the last MFMA and this. */
static __device__ __forceinline__ uint32_t mfm3_combine(int hh, int md, int ll)
{
    uint32_t t, a;
    asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(hh), "v"(md));
    asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a) : "v"(t), "v"(ll));
    return a;
}

/* bits 29:14 of re_b and im_b (biased sums) as (re | im << 16), for two pairs: four SDWA shifts (see
 * mfm_kernel_mfma.hip for the hazard notes: one wait state between a dst_sel write and the PRESERVE read) */
static __device__ __forceinline__ void mfm3_round_pack2(const uint32_t re_b[2], const uint32_t im_b[2], uint32_t p[2])
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

/* the same with the shift as a template argument (8-bit input: the sums are 128 times smaller for the RTL-SDR scaling) */
template <int SH>
static __device__ __forceinline__ void mfm3_round_pack2_sh(const uint32_t re_b[2], const uint32_t im_b[2], uint32_t p[2])
{
    uint32_t p0, p1;
    asm("v_lshrrev_b32_sdwa %0, %6, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %1, %6, %3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %0, %6, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_lshrrev_b32_sdwa %1, %6, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(p0), "=&v"(p1)
        : "v"(re_b[0]), "v"(re_b[1]), "v"(im_b[0]), "v"(im_b[1]), "n"(SH));
    p[0] = p0;
    p[1] = p1;
}

/* the same with the shift in an SGPR (the long-filter kernel's 8-bit form: 7 for the RTL-SDR scaling, 14 for cs8 / cu8) */
static __device__ __forceinline__ void mfm3_round_pack2_s(const uint32_t re_b[2], const uint32_t im_b[2], uint32_t sh, uint32_t p[2])
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

/* o = f * r + 8192: two VOP3P v_dot2_i32_i16 (3 wait states before the results are read) */
static __device__ __forceinline__ void mfm3_rotate_biased(uint32_t f, uint32_t rx, uint32_t ry, uint32_t *o_re,
                                                          uint32_t *o_im)
{
    asm("v_dot2_i32_i16 %0, %2, %3, %5\n\tv_dot2_i32_i16 %1, %2, %4, %5\n\ts_nop 2"
        : "=&v"(*o_re), "=&v"(*o_im)
        : "v"(f), "v"(rx), "v"(ry), "s"(8192));
}

/* s = q * conj(p), wrapping int32 (multifm/fm_demod.c:55-64) */
static __device__ __forceinline__ void mfm3_conj_mul(uint32_t q, uint32_t p, int *s_re, int *s_im)
{
    int u, t;
    asm("v_dot2_i32_i16 %0, %3, %4, 0\n\t"
        "v_mad_i32_i16 %1, %3, %4, 0 op_sel:[1,0,0,0]\n\t"
        "v_mad_i32_i16 %2, %3, %4, 0 op_sel:[0,1,0,0]\n\t"
        "s_nop 0"
        : "=&v"(*s_re), "=&v"(u), "=&v"(t)
        : "v"(q), "v"(p));
    *s_im = (int)((uint32_t)u - (uint32_t)t);
}

/* derotation + second rounding of two packed samples (filter/direct_fir.c:406-413) */
static __device__ __forceinline__ void derotate2(const uint32_t fin[2], const uint32_t rx[2], const uint32_t ry[2], uint32_t qout[2])
{
    uint32_t o_re[2], o_im[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        mfm3_rotate_biased(fin[c], rx[c], ry[c], &o_re[c], &o_im[c]);
    }
    mfm3_round_pack2(o_re, o_im, qout);
}

/* f * (+-1) per int16 half, wrapping: r14(f * +-16384) of filter/direct_fir.c:406-413 (the int16 cast of filter/complex.h:33
 * wraps -(-32768) to -32768, and so does the low half of a 16-bit product).  sg = (factor of even outputs) | (factor of odd
 * outputs) << 16, each 1 or 0xffff. */
static __device__ __forceinline__ uint32_t mfm3_sign_flip(uint32_t f, uint32_t sg, int odd)
{
    uint32_t q;
    if (odd) {
        asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(q) : "v"(f), "v"(sg));
    } else {
        asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(q) : "v"(f), "v"(sg));
    }
    return q;
}

/* A lane's four PCM samples of one channel (8 bytes; the 16 lanes of a row write one 128-byte line between them), stored
 * with the non-temporal hint.  sys (mfm_launch_v3::pcm_scope, wave uniform): system scope as well - the lines are written through,
 * so that WRITE_SIZE is the PCM bytes to the byte and the PCM stream - 5.3 x the input's bytes at 1024 channels - pushes fewer
 * rotator-table lines out of L2 between two uses (L2-miss traffic 1.27 -> 1.155 x algorithmic there at unchanged time,
 * profiles/r05_store_policy.txt), while the 64-channel shapes, whose writes are a larger share of a shorter launch, run 1.5-5 %
 * slower with it: the engine asks for it from 512 channels per launch on. */
static __device__ __forceinline__ void mfm3_store_pcm4(void *base, uint32_t byte_off, uint32_t w0, uint32_t w1, bool sys)
{
    typedef unsigned int mfm_v2u_t __attribute__((ext_vector_type(2)));
    const mfm_v2u_t wv = { w0, w1 };
    if (sys) {
        asm volatile("global_store_dwordx2 %0, %1, %2 sc0 sc1 nt" ::"v"(byte_off), "v"(wv), "s"(base) : "memory");
    } else {
        __builtin_nontemporal_store(wv, reinterpret_cast<mfm_v2u_t *>(reinterpret_cast<uint8_t *>(base) + byte_off));
    }
}

static __device__ __forceinline__ uint32_t mfm3_opaque(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}

/* a 4-byte rotator entry r = (rr | ri << 16) as the two dot-product operands of the derotation: (rr, -ri) and (ri, rr).  The
 * engine refuses rotators that reach ri = -32768, so the negated half never wraps. */
static __device__ __forceinline__ uint32_t mfm3_rot_x(uint32_t r)
{
    return (r ^ 0xffff0000u) + 0x00010000u;
}

static __device__ __forceinline__ uint32_t mfm3_rot_y(uint32_t r)
{
    return __builtin_amdgcn_alignbit(r, r, 16);
}

/* rotator-table position of output index (kb + d) of a channel: pre-period as is, then folded into the period */
static __device__ __forceinline__ uint32_t mfm3_fold(uint32_t kb, uint32_t d, uint32_t mu, uint32_t lam, uint32_t lam_magic)
{
    uint32_t k = kb + d;
    if (k >= mu) {
        const uint32_t x = k - mu;
        uint32_t m = x - __umulhi(x, lam_magic) * lam;
        m = (m >= lam) ? m - lam : m;
        k = mu + m;
    }
    return k;
}

/* t mod lam for any 32-bit t, lam_magic = floor(2^32 / lam): the quotient estimate is at most one short */
static __device__ __forceinline__ uint32_t mfm3_mod_step(uint32_t t, uint32_t lam, uint32_t lam_magic)
{
    const uint32_t m = t - __umulhi(t, lam_magic) * lam;
    return (m >= lam) ? m - lam : m;
}

/* the same for a 64-bit output index: a stream at the bench's rate passes 2^32 outputs per channel within a second.
 * k - mu is reduced mod lam (< 2^27: the engine's table limit) five bits at a time, so that everything stays 32-bit. */
static __device__ __forceinline__ uint32_t mfm3_fold64(uint64_t k, uint32_t mu, uint32_t lam, uint32_t lam_magic)
{
    if (k < (uint64_t)mu) {
        return (uint32_t)k;
    }
    const uint64_t x = k - mu;
    const uint32_t xl = (uint32_t)x;
    uint32_t a = mfm3_mod_step((uint32_t)(x >> 32), lam, lam_magic);
#pragma unroll
    for (int s = 27; s >= 2; s -= 5) {
        a = mfm3_mod_step((a << 5) | ((xl >> s) & 31u), lam, lam_magic);
    }
    a = mfm3_mod_step((a << 2) | (xl & 3u), lam, lam_magic);
    return mu + a;
}

/*
 * Four discriminators (multifm/fm_demod.c:68-72 on fast_atan2f.c:101-174), same operations in the same order as
 * mfm_discriminate() in mfm_numerics.h (the form the host twin proves against the oracle), written for the issue
 * costs measured on gfx950 (profiles/r02_ubench_ops.txt): fp32 mul / add / fma and 32-bit add / and / xor / shift-right
 * take one issue slot, conversions, min / max, compares, selects, v_fract, SDWA / DPP and packed forms two, v_rcp_f32
 * four.  So the division is scalar FMAs (no packed math: nothing to gain, registers to lose), and the table index comes
 * out of the float adder: floor(alpha) + 2^21 has 4 * floor(alpha) in its low mantissa bits - the byte offset of
 * T[floor(alpha)] - which saves the conversion and the shift.
 * lut_addr: LDS byte address of the table: {T[i], dT[i]} pairs (MFM3_LUT_MODE 0: T[0], with dT[0] 1024 bytes behind it).
 */
/* ASM_READS (the long-filter kernel, two waves per SIMD): the four table reads of a call are single asm statements, waited for ONCE,
 * behind the quadrant arithmetic of all four outputs.  Left to the compiler (0), ONE of the four reads is sunk into an exec-masked
 * branch of its own - s_and_saveexec / ds_read / s_waitcnt lgkmcnt(0) / s_or - a whole LDS round trip with nothing else of the wave
 * in flight, per channel and four outputs.  1: the reads go out in one batch behind the four divisions; 2: each behind its own
 * division, the later divisions to arrive behind.  Measured (tools/exp/ab.py, profiles/r06_ab_asm_reads.txt): instances of one row
 * block per wave -0.6 ... -1.5 % with 1 (2: the same within the noise); two row blocks per wave: 1 is neutral (configs[4]'s share) or
 * +4 % (128-channel slices at 1024 channels), 2 is -0.6 ... -0.8 % on both; mfm_kernel_v3.hip (four waves per SIMD) +0.7 ... +1.3 %
 * with 1 - so: 1, 2 and 0 there. */
template <int ASM_READS = 0> /* 0: the compiler's reads; 1: asm reads in one batch behind the four divisions; 2: each behind its own division */
static __device__ __forceinline__ void mfm3_discriminate4(const int s_re[4], const int s_im[4], uint32_t lut_addr, int pcm[4])
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(off)
    float x[4], y[4], mx[4], mn[4], z[4], fr[4], t0[4], dt[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        x[i] = (float)s_re[i];
        y[i] = (float)s_im[i];
        asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(mx[i]) : "v"(x[i]), "v"(y[i]));
        asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(mn[i]) : "v"(x[i]), "v"(y[i]));
    }
    typedef float mfm3_f2 __attribute__((ext_vector_type(2)));
    mfm3_f2 pr[4];
    (void)pr;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if constexpr (ASM_READS == 2 && MFM3_LUT_MODE == 1) {
            if (i > 0) {
                /* (this division stays behind the previous output's read: the reads go out one by one, each with the later
                 * divisions to arrive behind) */
                asm volatile("" : "+v"(mx[i]), "+v"(mn[i]));
            }
        }
        /* mfm_div_unit: correctly rounded mn / mx (one residual step behind the quotient estimate: mfm_numerics.h has the
         * argument and tools/div_proof.c the enumeration behind it) */
        const float r0 = __builtin_amdgcn_rcpf(mx[i]);
        const float e0 = __builtin_fmaf(-mx[i], r0, 1.0f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        const float q0 = mn[i] * r1;
        const float e1 = __builtin_fmaf(-mx[i], q0, mn[i]);
        z[i] = __builtin_fmaf(e1, r1, q0);
#if MFM3_DIV_STEPS > 1
        z[i] = __builtin_fmaf(__builtin_fmaf(-mx[i], z[i], mn[i]), r1, z[i]);
#endif
        if constexpr (ASM_READS == 2 && MFM3_LUT_MODE == 1) {
            const float alpha = z[i] * 255.0f;             /* fast_atan2f.c:125 */
            fr[i] = __builtin_amdgcn_fractf(alpha);        /* :127 (alpha - floor(alpha), exact) */
            const float fl = alpha - fr[i];                /* floor(alpha), 0..255 (NaN for (0, 0)) */
            const float m = fl + 1048576.0f;               /* bits 0x49800000 + 8 * floor(alpha) */
            const uint32_t addr = __float_as_uint(m) + (lut_addr - 0x49800000u);
            asm volatile("ds_read_b64 %0, %1" : "=v"(pr[i]) : "v"(addr));
        }
    }
    if constexpr (ASM_READS != 0 && MFM3_LUT_MODE == 1) {
    if constexpr (ASM_READS == 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float alpha = z[i] * 255.0f;             /* fast_atan2f.c:125 */
        fr[i] = __builtin_amdgcn_fractf(alpha);        /* :127 (alpha - floor(alpha), exact) */
        const float fl = alpha - fr[i];                /* floor(alpha), 0..255 (NaN for (0, 0)) */
        const float m = fl + 1048576.0f;               /* bits 0x49800000 + 8 * floor(alpha) */
        /* (0, 0): NaN bits land far outside LDS; such a read returns 0 and the result is discarded below */
        const uint32_t addr = __float_as_uint(m) + (lut_addr - 0x49800000u);
        asm volatile("ds_read_b64 %0, %1" : "=v"(pr[i]) : "v"(addr));
    }
    }
    /* :134-163: sign(y) * (K + u), K in {0, pi, pi/2}, u = +-base - what of it does not need the table, while the reads are
     * under way (inputs of the wait below, so that it is computed in front of it) */
    float k[4];
    /* (x and y pass through an empty statement behind the reads: what is computed from them from here on stays behind the reads) */
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool x_nonneg = s_re[i] >= 0, wide = __builtin_fabsf(x[i]) > __builtin_fabsf(y[i]);
        k[i] = wide ? (x_nonneg ? 0.0f : MFM_PI_F) : MFM_HALF_PI_F;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]) : "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]));
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float prod = pr[i].y * fr[i];
        const float interp = pr[i].x + prod;           /* :130-131, unfused */
        const float base = (z[i] < MFM_TAN_MAP_RES_F) ? z[i] : interp;
        const bool x_nonneg = s_re[i] >= 0, wide = __builtin_fabsf(x[i]) > __builtin_fabsf(y[i]);
        const float u = (x_nonneg ^ wide) ? -base : base;
        const float mag = k[i] + u;
        const float lo = mag * MFM_Q14_OVER_PI_LO;
        const float sc = __builtin_fmaf(mag, MFM_Q14_OVER_PI_HI, lo);
        int signed_sc;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(signed_sc) : "s"(0x7fffffff), "v"(sc), "v"(s_im[i]));
        pcm[i] = (int)__int_as_float(signed_sc);       /* NaN (from (0, 0)) converts to 0 = fast_atan2f.c:111-112 */
    }
    (void)t0, (void)dt;
    } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float alpha = z[i] * 255.0f;             /* fast_atan2f.c:125 */
        fr[i] = __builtin_amdgcn_fractf(alpha);        /* :127 (alpha - floor(alpha), exact) */
        const float fl = alpha - fr[i];                /* floor(alpha), 0..255 (NaN for (0, 0)) */
#if MFM3_LUT_MODE == 1
        const float m = fl + 1048576.0f;               /* bits 0x49800000 + 8 * floor(alpha) */
        /* (0, 0): NaN bits land far outside LDS; such a read returns 0 and the result is discarded below */
        const uint32_t addr = __float_as_uint(m) + (lut_addr - 0x49800000u);
        typedef const __attribute__((address_space(3))) float2 *lds_f2p;
        const float2 pr = *(lds_f2p)(uintptr_t)addr;
        t0[i] = pr.x;
        dt[i] = pr.y;
#else
        const float m = fl + 2097152.0f;               /* bits 0x4A000000 + 4 * floor(alpha) */
        /* (0, 0): NaN bits land far outside LDS; such a read returns 0 and the result is discarded below */
#if MFM3_LUT_MODE == 2
        /* bit 0 of lut_addr set by the channel kernel only: the commit-time self-test keeps the real table reads */
        const uint32_t addr = (lut_addr & 1u) ? (lut_addr - 1u) + 4u * (__lane_id() + 64u * (uint32_t)i)
                                              : __float_as_uint(m) + (lut_addr - 0x4A000000u);
#else
        const uint32_t addr = __float_as_uint(m) + (lut_addr - 0x4A000000u);
#endif
        typedef const __attribute__((address_space(3))) float *lds_fp;
        lds_fp p = (lds_fp)(uintptr_t)addr;
        t0[i] = p[0];
        dt[i] = p[256];
#endif
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float prod = dt[i] * fr[i];
        const float interp = t0[i] + prod;             /* :130-131, unfused */
        const float base = (z[i] < MFM_TAN_MAP_RES_F) ? z[i] : interp;
        /* :134-163: sign(y) * (K + u), K in {0, pi, pi/2}, u = +-base */
        const bool x_nonneg = s_re[i] >= 0, wide = __builtin_fabsf(x[i]) > __builtin_fabsf(y[i]);
        const float k = wide ? (x_nonneg ? 0.0f : MFM_PI_F) : MFM_HALF_PI_F;
        const float u = (x_nonneg == wide) ? base : -base;
        const float mag = k + u;
        const float lo = mag * MFM_Q14_OVER_PI_LO;
        const float sc = __builtin_fmaf(mag, MFM_Q14_OVER_PI_HI, lo);
        int signed_sc;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(signed_sc) : "s"(0x7fffffff), "v"(sc), "v"(s_im[i]));
        pcm[i] = (int)__int_as_float(signed_sc);       /* NaN (from (0, 0)) converts to 0 = fast_atan2f.c:111-112 */
    }
    }
#else
    (void)s_re, (void)s_im, (void)lut_addr, (void)pcm;
#endif
}

/* low word of a clock stamp, pinned to ONE scalar register (left alone the compiler keeps the 64-bit pair alive to the end) */
static __device__ __forceinline__ uint32_t mfm3_stamp_lo(uint64_t t)
{
    uint32_t lo = (uint32_t)t;
    asm volatile("" : "+s"(lo));
    return lo;
}

/* the workgroup's duration in shader-clock ticks and in 100 MHz reference ticks, folded into the launch's maximum
 * (mfm_launch_v3::cyc); t0 / r0: the low words of the stamps taken at the workgroup's start (a launch is shorter than 2^32
 * ticks of either clock; two SGPRs across the kernel instead of four - the 128-register instances parked four in a VGPR lane) */
static __device__ __forceinline__ void mfm3_stamp_end(const mfm_launch_v3 &L, uint32_t t0, uint32_t r0)
{
    if (L.cyc != nullptr && threadIdx.x == 0) {
        const uint64_t tag = (uint64_t)(L.cyc_tag & 0xffffffu) << 40;
        const uint64_t dt = (uint32_t)((uint32_t)__builtin_amdgcn_s_memtime() - t0), dr = (uint32_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - r0);
        atomicMax(L.cyc, (unsigned long long)(tag | dt));
        atomicMax(L.cyc + 1, (unsigned long long)(tag | dr));
        if (blockIdx.x == 0) {
            L.cyc_clear[0] = 0ull; /* the slot of the launch half a ring from now */
            L.cyc_clear[1] = 0ull;
        }
    }
}

/* one work item = (chunk of consecutive tiles, 64-channel slice); XCD-aware: the slices of a chunk run back to back
 * on one XCD, so the chunk's input is fetched from HBM once and re-read from that XCD's L2 */
static __device__ __forceinline__ bool mfm3_decode_item(const mfm_launch_v3 &L, uint32_t item, uint32_t *chunk, uint32_t *slice)
{
    const uint32_t xcd = item & 7u, seq = item >> 3;
    *chunk = (seq / L.nslices) * 8u + xcd;
    *slice = seq % L.nslices;
    return item < L.nitems && *chunk < L.nchunks;
}

