/*
 * mfm_kernel.hip - the fused multifm channel kernel for gfx950 (MI355X).
 *
 * One launch turns a block of wideband int16 IQ into int16 PCM for every channel:
 *
 *   complex-tap decimating FIR   filter/direct_fir.c:328-417 (+ filter/complex.h:40-46)
 *   Q14 round + derotation       filter/direct_fir.c:406-413, :151-172
 *   FM discriminator             multifm/fm_demod.c:53-79 (+ multifm/fast_atan2f.c:101-174)
 *
 * Mapping (see DESIGN.md "Kernel"):
 *   - lanes  = consecutive OUTPUT samples (64*OPL outputs per tile, lane j / slot o -> output
 *     tile_first + 64*o + j).  The first output of a tile is the previous tile's last one,
 *     recomputed so the discriminator's one-sample history never crosses a workgroup.
 *   - the tile's input window is staged ONCE into LDS, transposed to [sample mod D][sample / D]
 *     so that the 64 lanes of a wave read 64 consecutive dwords for any tap (no bank conflicts
 *     at any decimation) and every wave of the workgroup reuses it for its own channels.
 *   - taps are wave-uniform: they arrive through the scalar cache (s_load_dwordx16) and feed
 *     v_dot2c_i32_i16 as the SGPR operand; int16 IQ pairs are the packed VGPR operand.  Two
 *     dot2 per complex tap, exact wrap-around int32 accumulation.
 *   - the rotator recurrence is sequential and non-associative, but input independent: the
 *     engine tabulates it (pre-period + one period) and the kernel indexes the table.
 */
#include <hip/hip_runtime.h>

#include "mfm_kernel.h"
#include "mfm_numerics.h"

typedef short mfm_short2 __attribute__((ext_vector_type(2)));

/* Read-only tables (taps, tap offsets, channel descriptors) are read through the constant address
 * space so that wave-uniform reads become scalar-cache loads (s_load_*) and reach the VALU as SGPR
 * operands.  Nothing in a launch writes these tables. */
#define MFM_CONST_AS __attribute__((address_space(4)))
typedef const MFM_CONST_AS uint32_t *mfm_cptr_u32;
template <typename T> static __device__ __forceinline__ const MFM_CONST_AS T *mfm_as_const(const T *p)
{
    return (const MFM_CONST_AS T *)(p);
}

static __device__ __forceinline__ int mfm_dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(mfm_short2, a), __builtin_bit_cast(mfm_short2, b), c,
                                  false);
}

/* prev-lane value; lane 0 receives `lane0` (DPP wave_shr:1, bound_ctrl off) */
static __device__ __forceinline__ uint32_t mfm_shift_up1(uint32_t v, uint32_t lane0)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf,
                                                 false);
}

template <int OPL, bool DBG_IQ>
__global__ __launch_bounds__(MFM_NT) void mfm_channel_kernel(const mfm_launch L)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    constexpr int OT = MFM_WAVE * OPL; /* outputs per tile, including the recomputed first one */
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (MFM_WAVE - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    /* XCD-aware decode: blocks b, b+8, b+16.. share an XCD (and its L2); give them the slices of
     * the same tile back to back so the tile's input is fetched from HBM once. */
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, seq = bid >> 3;
    const uint32_t tile = (seq / L.nslices) * 8u + xcd;
    const uint32_t slice = seq % L.nslices;
    if (tile >= L.ntiles) {
        return;
    }

    const uint32_t D = L.decim;
    const uint32_t rs2 = L.rs2;

    /* ---- stage the tile: samples [tile*(OT-1)*D - D, +nstage) -> lds[(s % D)*rs2 + s / D] ---- */
    {
        const int64_t s0 = (int64_t)tile * (OT - 1) * D - (int64_t)D;
        uint32_t row = tid / D, r = tid % D;
        const uint32_t drow = MFM_NT / D, dr = MFM_NT % D;
        /* a batch of global loads is issued before any is consumed; one pass per load would cost one
         * HBM round trip each */
        constexpr int SB = 8;
        for (uint32_t idx0 = tid; idx0 < L.nstage; idx0 += SB * MFM_NT) {
            uint32_t v[SB];
#pragma unroll
            for (int b = 0; b < SB; b++) {
                const uint32_t idx = idx0 + b * MFM_NT;
                const int64_t g = s0 + idx;
                v[b] = 0;
                if (idx < L.nstage && g >= 0 && g < (int64_t)L.n_avail) {
                    v[b] = L.x[g];
                }
            }
#pragma unroll
            for (int b = 0; b < SB; b++) {
                if (idx0 + b * MFM_NT < L.nstage) {
                    lds[r * rs2 + row] = v[b];
                }
                r += dr;
                row += drow;
                if (r >= D) {
                    r -= D;
                    row += 1;
                }
            }
        }
        /* atan LUT: 256 x {T[i], T[i+1]-T[i]} = 512 dwords */
        const uint32_t *lut_g = reinterpret_cast<const uint32_t *>(L.lut);
        for (uint32_t i = tid; i < 512; i += MFM_NT) {
            lds[L.lut_off + i] = lut_g[i];
        }
    }
    __syncthreads();

    const float2 *lut = reinterpret_cast<const float2 *>(lds + L.lut_off);
    const uint32_t g_end = min(L.ngroups, (slice + 1) * L.gpw);
    const int rel0 = (int)(tile * (OT - 1)) - 1; /* output index (relative to this pass) of lane 0 / slot 0 */

    for (uint32_t g = slice * L.gpw + wave; g < g_end; g += MFM_NW) {
        int acc_re[OPL][MFM_CG], acc_im[OPL][MFM_CG];
#pragma unroll
        for (int o = 0; o < OPL; o++) {
#pragma unroll
            for (int c = 0; c < MFM_CG; c++) {
                acc_re[o][c] = 0;
                acc_im[o][c] = 0;
            }
        }

        /* ---- FIR: T taps x CG channels x OPL outputs per lane ---- */
        mfm_cptr_u32 cp = mfm_as_const(L.coef) + (size_t)g * L.nchunks * (MFM_TG * MFM_CG * 2);
        mfm_cptr_u32 tp = mfm_as_const(L.tapoff);
        for (uint32_t ch = 0; ch < L.nchunks; ch++) {
#pragma unroll
            for (int k = 0; k < MFM_TG; k++) {
                const uint32_t a = (tp[k] >> 2) + lane;
                uint32_t xv[OPL];
#pragma unroll
                for (int o = 0; o < OPL; o++) {
                    xv[o] = lds[a + MFM_WAVE * o];
                }
#pragma unroll
                for (int c = 0; c < MFM_CG; c++) {
                    const uint32_t w_re = cp[(k * MFM_CG + c) * 2 + 0]; /* (cr, -ci) */
                    const uint32_t w_im = cp[(k * MFM_CG + c) * 2 + 1]; /* (ci,  cr) */
#pragma unroll
                    for (int o = 0; o < OPL; o++) {
                        acc_re[o][c] = mfm_dot2(w_re, xv[o], acc_re[o][c]);
                        acc_im[o][c] = mfm_dot2(w_im, xv[o], acc_im[o][c]);
                    }
                }
            }
            cp += MFM_TG * MFM_CG * 2;
            tp += MFM_TG;
        }

        /* ---- epilogue: round, derotate, discriminate, store ---- */
#pragma unroll
        for (int c = 0; c < MFM_CG; c++) {
            const uint32_t chn = g * MFM_CG + c;
            if (chn >= L.nchan) {
                break;
            }
            /* scalar loads of the channel descriptor and carried state */
            mfm_cptr_u32 ip = mfm_as_const(reinterpret_cast<const uint32_t *>(L.info)) + (size_t)chn * 8;
            mfm_cptr_u32 sp = mfm_as_const(reinterpret_cast<const uint32_t *>(L.st_in)) + (size_t)chn * 2;
            mfm_chan_info ci;
            ci.rot_base = (uint64_t)ip[0] | ((uint64_t)ip[1] << 32);
            ci.mu = ip[2];
            ci.lam = ip[3];
            ci.lam_magic = ip[4];
            mfm_chan_state st;
            st.carry_q = sp[0];
            st.kb = sp[1];

            /* rotator table index of lane 0 / slot 0, folded into [0, mu + lam) */
            int ks = (int)st.kb + rel0;
            if (ks >= (int)ci.mu) {
                const uint32_t x = (uint32_t)ks - ci.mu;
                uint32_t r = x - __umulhi(x, ci.lam_magic) * ci.lam;
                r = (r >= ci.lam) ? r - ci.lam : r;
                ks = (int)(ci.mu + r);
            }
            const uint2 *rot = L.rot + ci.rot_base + (int64_t)ks + lane;

            uint32_t q_prev_slot = 0;
#pragma unroll
            for (int o = 0; o < OPL; o++) {
                /* filter/direct_fir.c:406-409: f = r14(acc); o = f * rot */
                const uint32_t f = mfm_pack16(mfm_r14_wide(acc_re[o][c]), mfm_r14_wide(acc_im[o][c]));
                const uint2 rv = rot[MFM_WAVE * o];
                const int o_re = mfm_dot2(f, rv.x, 0); /* fr*rr - fi*ri */
                const int o_im = mfm_dot2(f, rv.y, 0); /* fr*ri + fi*rr */
                /* :412-413 */
                uint32_t q = mfm_pack16(mfm_r14_wide(o_re), mfm_r14_wide(o_im));

                if (o == 0 && tile == 0) {
                    /* output "-1" of this pass is the last one of the previous pass */
                    q = (lane == 0) ? st.carry_q : q;
                }
                uint32_t lane0_prev = 0;
                if (o > 0) {
                    lane0_prev = (uint32_t)__builtin_amdgcn_readlane((int)q_prev_slot, MFM_WAVE - 1);
                }
                const uint32_t p = mfm_shift_up1(q, lane0_prev);
                q_prev_slot = q;

                /* multifm/fm_demod.c:55-64: s = q * conj(p), int32 wrap-around */
                const int q_re = mfm_lo16(q), q_im = mfm_hi16(q), p_re = mfm_lo16(p), p_im = mfm_hi16(p);
                const int s_re = mfm_dot2(q, p, 0);
                const int s_im = (int)((uint32_t)(q_im * p_re) - (uint32_t)(q_re * p_im));
                const int pcm = mfm_discriminate(s_re, s_im, lut);

                const int rel = rel0 + MFM_WAVE * o + (int)lane;
                const bool first = (o == 0) && (lane == 0);
                if (!first && rel < (int)L.n_new) {
                    L.pcm[(size_t)chn * L.out_stride + rel] = (int16_t)pcm;
                    if (DBG_IQ) {
                        L.iq_dbg[(size_t)chn * L.out_stride + rel] = q;
                    }
                    if (rel == (int)L.n_new - 1) {
                        L.st_out[chn].carry_q = q;
                    }
                }
            }

            if (tile == 0 && lane == 0) {
                /* rotator index of the next pass's first output */
                uint32_t kn = st.kb + L.n_new;
                if (kn >= ci.mu) {
                    const uint32_t x = kn - ci.mu;
                    uint32_t r = x - __umulhi(x, ci.lam_magic) * ci.lam;
                    r = (r >= ci.lam) ? r - ci.lam : r;
                    kn = ci.mu + r;
                }
                L.st_out[chn].kb = kn;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------- */

/* which instance runs (outputs per lane, filtered-IQ output): asked once at commit, where its LDS limit is raised */
/* The scalar discriminator (mfm_numerics.h) on caller-supplied products (tests, the engine's division self-test) */
__global__ __launch_bounds__(256) void mfm_disc_test_kernel(const int *s_re, const int *s_im, int *pcm, uint32_t n, const float2 *lut)
{
    __shared__ float2 tbl[256];
    tbl[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n) {
        pcm[t] = mfm_discriminate(s_re[t], s_im[t], tbl);
    }
}

extern "C" hipError_t mfm_disc_test_dot2(const int *s_re, const int *s_im, int *pcm, uint32_t n, const float2 *lut, hipStream_t stream)
{
    hipLaunchKernelGGL(mfm_disc_test_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, s_re, s_im, pcm, n, lut);
    return hipGetLastError();
}

extern "C" hipError_t mfm_select_channel_kernel(int opl, int dbg_iq, const void **kfn_out)
{
    *kfn_out = nullptr;
    if (opl == 2) {
        *kfn_out = dbg_iq ? reinterpret_cast<const void *>(&mfm_channel_kernel<2, true>)
                          : reinterpret_cast<const void *>(&mfm_channel_kernel<2, false>);
    } else if (opl == 1) {
        *kfn_out = dbg_iq ? reinterpret_cast<const void *>(&mfm_channel_kernel<1, true>)
                          : reinterpret_cast<const void *>(&mfm_channel_kernel<1, false>);
    } else {
        return hipErrorInvalidValue;
    }
    return hipSuccess;
}

extern "C" hipError_t mfm_launch_channel_kernel(const void *kfn, const mfm_launch *L, uint32_t lds_bytes, hipStream_t stream)
{
    const uint32_t tiles8 = (L->ntiles + 7u) / 8u;
    const dim3 grid(tiles8 * 8u * L->nslices), block(MFM_NT);
    if (L->ntiles == 0) {
        return hipSuccess;
    }
    void *args[] = { const_cast<mfm_launch *>(L) };
    return hipLaunchKernel(kfn, grid, block, args, lds_bytes, stream);
}
