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
 * i.e. four v_mfma_i32_32x32x32_i8 per 32 rows x 32 outputs x 32 elements, int32 accumulators that
 * wrap exactly like the reference's int32 sums (tools/ubench_mfma_i8.hip checks the wrap, the k-slot
 * pairing and the C/D layout on hardware).  The last term is a per-row constant added once.
 *
 * Per workgroup: 4 waves x 32 rows (= 64 channels) share one LDS image of the input tile, stored as
 * two byte planes (Eh, El) in rows of 2*D bytes with an odd 16-byte row stride, so the B operand of
 * every lane is one conflict-free ds_read_b128.  The A operand (taps) is loaded once per wave and
 * stays in 64 VGPRs.  Everything after the accumulators - Q14 round, derotation by the tabulated
 * rotator, the fast_atan2f discriminator, PCM store - is the same exact arithmetic as the v_dot2
 * kernel (mfm_numerics.h), applied to the MFMA C/D layout (each lane holds re/im of 8 channels for
 * one output).
 */
#include <hip/hip_runtime.h>

#include "mfm_kernel.h"
#include "mfm_numerics.h"

typedef int mfm_v4i __attribute__((ext_vector_type(4)));
typedef int mfm_v16i __attribute__((ext_vector_type(16)));
typedef short mfm_s2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ int mfm_dot2m(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(mfm_s2, a), __builtin_bit_cast(mfm_s2, b), c, false);
}

template <int KS, bool DBG_IQ>
__global__ __launch_bounds__(MFM_MFMA_NW * 64, 2) void mfm_channel_kernel_mfma(const mfm_launch_mfma L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t g = lane >> 5, n = lane & 31u;
    const uint32_t D = L.decim, row_bytes = 2u * D, rs = L.rs;

    uint8_t *plane_h = smem, *plane_l = smem + L.plane_bytes;
    const float2 *lut = reinterpret_cast<const float2 *>(smem + L.lut_off);
    const int32_t *krow_s = reinterpret_cast<const int32_t *>(smem + L.krow_off);

    /* per-lane LDS byte offset of the B fragment of k-step ks for output column n of N-tile 0 */
    uint32_t boff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const uint32_t e = 32u * ks + 16u * g;
        boff[ks] = (n + e / row_bytes) * rs + e % row_bytes;
    }

    mfm_v4i a_h[KS], a_l[KS];
    uint32_t slice_loaded = 0xffffffffu;

    for (uint32_t item = blockIdx.x; item < L.nitems; item += gridDim.x) {
        /* XCD-aware decode (see mfm_kernel.hip): the slices of one tile run back to back on one XCD */
        const uint32_t xcd = item & 7u, seq = item >> 3;
        const uint32_t tile = (seq / L.nslices) * 8u + xcd;
        const uint32_t slice = seq % L.nslices;
        if (tile >= L.ntiles) {
            continue;
        }
        const uint32_t rb = slice * MFM_MFMA_NW + wave; /* this wave's block of 32 rows */
        const bool rb_valid = rb < L.nrb;

        if (slice != slice_loaded) {
            /* A operand: the wave's 32 rows x (32*KS) elements, both byte planes, in fragment order */
            if (rb_valid) {
                const mfm_v4i *ap = reinterpret_cast<const mfm_v4i *>(L.afrag) + (size_t)rb * KS * 2 * 64 + lane;
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    a_h[ks] = ap[(ks * 2 + 0) * 64];
                    a_l[ks] = ap[(ks * 2 + 1) * 64];
                }
            }
            slice_loaded = slice;
        }

        __syncthreads(); /* everyone is done reading the previous tile's LDS image */

        /* ---- stage: 4 samples (16 B) per thread per pass -> 8 bytes into each plane ---- */
        {
            const int64_t s0 = ((int64_t)tile * (L.ot - 1) - 1) * (int64_t)D;
            const uint32_t nchunk = L.nstage >> 2;
            uint32_t p8 = tid * 8u;
            uint32_t row = p8 / row_bytes, off = p8 % row_bytes;
            const uint32_t drow = 2048u / row_bytes, doff = 2048u % row_bytes; /* 256 threads x 8 bytes */
            for (uint32_t q = tid; q < nchunk; q += MFM_MFMA_NW * 64) {
                const int64_t gs = s0 + (int64_t)q * 4;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (gs >= 0 && gs + 3 < (int64_t)L.n_avail) {
                    v = *reinterpret_cast<const uint4 *>(L.x + gs);
                } else if (gs + 3 >= 0 && gs < (int64_t)L.n_avail) {
                    uint32_t t[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int64_t gk = gs + k;
                        t[k] = (gk >= 0 && gk < (int64_t)L.n_avail) ? L.x[gk] : 0u;
                    }
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                }
                /* dword = [lo0 hi0 lo1 hi1]: gather high bytes / low bytes of four int16 into one dword */
                uint2 hi, lo;
                hi.x = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
                hi.y = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
                lo.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u) ^ 0x80808080u;
                lo.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u) ^ 0x80808080u;
                const uint32_t at = row * rs + off;
                *reinterpret_cast<uint2 *>(plane_h + at) = hi;
                *reinterpret_cast<uint2 *>(plane_l + at) = lo;
                off += doff;
                row += drow;
                if (off >= row_bytes) {
                    off -= row_bytes;
                    row += 1;
                }
            }
            uint32_t *lut_s = reinterpret_cast<uint32_t *>(smem + L.lut_off);
            const uint32_t *lut_g = reinterpret_cast<const uint32_t *>(L.lut);
            for (uint32_t i = tid; i < 512; i += MFM_MFMA_NW * 64) {
                lut_s[i] = lut_g[i];
            }
            if (tid < MFM_MFMA_NW * 32) {
                const uint32_t r = slice * MFM_MFMA_NW * 32 + tid;
                reinterpret_cast<int32_t *>(smem + L.krow_off)[tid] = (r < L.nrb * 32u) ? L.krow[r] : 0;
            }
        }
        __syncthreads();

        if (!rb_valid) {
            continue;
        }

        /* ---- per tile, per channel pair: where this lane's first column sits in the rotator table ---- */
        const int rel_first = (int)(tile * (L.ot - 1)) - 1; /* output index (this pass) of column 0 */
        uint32_t q_prev[8]; /* previous N-tile's filtered samples, per channel pair */
        uint32_t k_abs[8], k_wrap[8], k_lam[8];
#pragma unroll
        for (int rp = 0; rp < 8; rp++) {
            q_prev[rp] = 0;
            const uint32_t chn = rb * 16u + (uint32_t)(rp & 1) + 4u * (uint32_t)(rp >> 1) + 2u * g;
            const uint32_t chs = chn < L.nchan ? chn : 0u;
            const uint32_t *ip = reinterpret_cast<const uint32_t *>(L.info) + (size_t)chs * 8;
            const uint4 inf = *reinterpret_cast<const uint4 *>(ip);
            const uint32_t lam_magic = ip[4];
            const uint32_t kb = L.st_in[chs].kb;
            const uint32_t mu = inf.z, lam = inf.w;
            int k = (int)kb + rel_first; /* >= -1; entry -1 of every table is a readable dummy */
            if (k >= (int)mu) {
                const uint32_t x = (uint32_t)k - mu;
                uint32_t m = x - __umulhi(x, lam_magic) * lam;
                m = (m >= lam) ? m - lam : m;
                k = (int)(mu + m);
            }
            k_abs[rp] = inf.x + (uint32_t)k; /* table index fits 32 bits (engine checks) */
            k_wrap[rp] = inf.x + mu + lam;
            k_lam[rp] = lam;                 /* >= 128: one conditional subtraction folds a whole tile */
        }

        const uint32_t ntiles_n = L.ot / MFM_MFMA_NT;
        for (uint32_t tau = 0; tau < ntiles_n; tau++) {
            /* rotator entries for this N-tile: addresses do not depend on data, so issue the loads
             * before the matrix work and let them land underneath it */
            uint2 rv[8];
#pragma unroll
            for (int rp = 0; rp < 8; rp++) {
                uint32_t idx = k_abs[rp] + tau * MFM_MFMA_NT + n;
                idx = (idx >= k_wrap[rp]) ? idx - k_lam[rp] : idx;
                rv[rp] = L.rot[idx];
            }

            /* ---- GEMM: 32 rows x 32 outputs x (32*KS) elements, four byte-plane products ---- */
            mfm_v16i acc_hh, acc_md, acc_ll;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc_hh[r] = 0;
                acc_md[r] = 0;
                /* + 128 * sum_k W[row][k]: the constant of the El offset */
                acc_ll[r] = krow_s[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g];
            }
            const uint32_t tbase = tau * MFM_MFMA_NT * rs;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const mfm_v4i b_h = *reinterpret_cast<const mfm_v4i *>(plane_h + tbase + boff[ks]);
                const mfm_v4i b_l = *reinterpret_cast<const mfm_v4i *>(plane_l + tbase + boff[ks]);
                acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_h[ks], b_h, acc_hh, 0, 0, 0);
                acc_md = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_h[ks], b_l, acc_md, 0, 0, 0);
                acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_l[ks], b_l, acc_ll, 0, 0, 0);
                acc_md = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_l[ks], b_h, acc_md, 0, 0, 0);
            }

            /* ---- epilogue: this lane holds (re, im) of 8 channels for output column n ---- */
            const int rel = rel_first + (int)(tau * MFM_MFMA_NT + n);
            const bool first_col = (tau == 0) && (n == 0); /* the recomputed previous output */
#pragma unroll
            for (int rp = 0; rp < 8; rp++) {
                const int r = 2 * rp;
                const uint32_t chn = rb * 16u + (uint32_t)(rp & 1) + 4u * (uint32_t)(rp >> 1) + 2u * g;
                const bool ch_ok = chn < L.nchan;

                const uint32_t a_re = ((uint32_t)acc_hh[r] << 16) + ((uint32_t)acc_md[r] << 8) + (uint32_t)acc_ll[r];
                const uint32_t a_im =
                    ((uint32_t)acc_hh[r + 1] << 16) + ((uint32_t)acc_md[r + 1] << 8) + (uint32_t)acc_ll[r + 1];

                /* filter/direct_fir.c:406-413 */
                const uint32_t f = mfm_pack16(mfm_r14_wide((int)a_re), mfm_r14_wide((int)a_im));
                const int o_re = mfm_dot2m(f, rv[rp].x, 0);
                const int o_im = mfm_dot2m(f, rv[rp].y, 0);
                uint32_t q = mfm_pack16(mfm_r14_wide(o_re), mfm_r14_wide(o_im));
                if (tile == 0 && tau == 0) {
                    /* column 0 of the pass is the last filtered sample of the previous pass */
                    const uint32_t carry = L.st_in[ch_ok ? chn : 0u].carry_q;
                    q = (n == 0) ? carry : q;
                }

                /* previous output of the same channel: lane n-1, or the last column of the previous N-tile */
                uint32_t p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)q_prev[rp], 31);
                const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)q_prev[rp], 63);
                p = (n == 0) ? (g ? e1 : e0) : p;
                q_prev[rp] = q;

                /* multifm/fm_demod.c:55-72 */
                const int q_re = mfm_lo16(q), q_im = mfm_hi16(q), p_re = mfm_lo16(p), p_im = mfm_hi16(p);
                const int s_re = mfm_dot2m(q, p, 0);
                const int s_im = (int)((uint32_t)(q_im * p_re) - (uint32_t)(q_re * p_im));
                const int pcm = mfm_discriminate(s_re, s_im, lut);

                if (ch_ok && !first_col && rel < (int)L.n_new) {
                    L.pcm[(size_t)chn * L.out_stride + rel] = (int16_t)pcm;
                    if (DBG_IQ) {
                        L.iq_dbg[(size_t)chn * L.out_stride + rel] = q;
                    }
                    if (rel == (int)L.n_new - 1) {
                        L.st_out[chn].carry_q = q;
                    }
                }
            }
        }

        if (tile == 0 && n == 0) {
            /* rotator index of the next pass's first output, one lane per channel */
#pragma unroll
            for (int rp = 0; rp < 8; rp++) {
                const uint32_t chn = rb * 16u + (uint32_t)(rp & 1) + 4u * (uint32_t)(rp >> 1) + 2u * g;
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
    }
}

extern "C" hipError_t mfm_launch_channel_kernel_mfma(const mfm_launch_mfma *L, int dbg_iq, uint32_t lds_bytes,
                                                     uint32_t grid, hipStream_t stream)
{
    if (L->ntiles == 0) {
        return hipSuccess;
    }
#define MFM_LAUNCH_M(KS_, DBG_)                                                                              \
    do {                                                                                                     \
        auto kfn = mfm_channel_kernel_mfma<KS_, DBG_>;                                                       \
        static uint32_t lds_set_ = 0;                                                                        \
        if (lds_bytes > lds_set_) {                                                                          \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                         \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e_ != hipSuccess) {                                                                          \
                return e_;                                                                                   \
            }                                                                                                \
            lds_set_ = lds_bytes;                                                                            \
        }                                                                                                    \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(MFM_MFMA_NW * 64), lds_bytes, stream, *L);                  \
    } while (0)
#define MFM_LAUNCH_KS(KS_)                                                                                   \
    do {                                                                                                     \
        if (dbg_iq) {                                                                                        \
            MFM_LAUNCH_M(KS_, true);                                                                         \
        } else {                                                                                             \
            MFM_LAUNCH_M(KS_, false);                                                                        \
        }                                                                                                    \
    } while (0)

    switch (L->ks) {
    case 1: MFM_LAUNCH_KS(1); break;
    case 2: MFM_LAUNCH_KS(2); break;
    case 4: MFM_LAUNCH_KS(4); break;
    case 8: MFM_LAUNCH_KS(8); break;
    default: return hipErrorInvalidValue;
    }
#undef MFM_LAUNCH_KS
#undef MFM_LAUNCH_M
    return hipGetLastError();
}
