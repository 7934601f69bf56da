/*
 * mfm_resampler.hip - the PCM stage behind the channel FIFOs, batched over all channels on the GPU:
 * real-valued rational resampler (filter/polyphase_fir.c, filter/utils.c) and DC blocker
 * (filter/dc_blocker.h).  See include/multifm_hip.h for the boundary and the reference lines.
 *
 * Kernel shape: grid = (blocks of 1024 outputs, channels).  All channels share the phase walk (they receive the
 * same number of samples), so output j of a call starts at sample (p0 + j*D) / I with phase (p0 + j*D) % I.
 * A workgroup copies the input window of its 1024 outputs (about 1024 * D / I + phase length samples) into LDS -
 * straight from where the samples lie: the unconsumed tail of the previous call (a few dozen samples per channel,
 * ping-pong buffers) followed by the caller's PCM, no staging copy - and every thread computes 4 outputs as
 * v_dot2_i32_i16 over sample pairs (int32 wrap-around like filter/utils.c:94-103; an odd window start costs one
 * funnel shift per pair).  When 256 * D is a multiple of I - 4/5 for POCSAG, 16/25 for FLEX - a thread's four
 * outputs share one phase and its coefficient pairs stay in registers; otherwise they are read from LDS.
 * HBM traffic: 2 bytes in + 2 * I / D bytes out per PCM sample.
 *
 * Matrix-core form (round 2; used when 16 D / I is an integer - 4/5 and 16/25 both - and the window fits 256 bytes).
 * The v_dot2 form is bound by the vector ALU, not by memory: 16/25 with the reference's 821 taps is 26 coefficient
 * pairs = 78 instructions per output, 0.11 ms per block for 64 channels where the channel kernel in front of it takes
 * 0.125.  Sixteen consecutive outputs consume exactly R = 16 D / I samples and their phases repeat from block to
 * block, so  y[16 m + q] = sum_s G[q][s] x[R m + s]  with ONE 16 x K matrix G (row q = the phase of output q shifted
 * to where its window starts, zero elsewhere; one G per value of the carried phase) - a GEMM with M = 16 = the matrix
 * instruction's M, columns = blocks of 16 outputs.  int16 x int16 -> wrapping int32 is done exactly as in the channel
 * kernel (mfm_kernel_mfma.hip): W = 256 Wh + Wl, x = 256 Xh + Xl + 128, four v_mfma_i32_16x16x64_i8 per 64 elements.
 * The input lies in LDS as two byte planes in rows of R samples padded to 16-byte multiples (zero coefficients over
 * the padding), so that a lane's B operand is one aligned ds_read_b128: the window of block m is rows m, m+1, ...
 * A lane ends up with four consecutive outputs of one block: one 8-byte store.
 * The DC blocker is a sequential IIR with a truncating shift in the loop (not associative): one thread
 * per channel.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/multifm_hip.h"

extern "C" __attribute__((visibility("hidden"))) void mfm_internal_set_error(const char *msg);
#include "mfm_numerics.h"

namespace {

struct RsLaunch {
    const int16_t *tail;  /* [C][tail_cap]: what the previous call left unconsumed */
    const int16_t *pcm;   /* [C][in_stride]: this call's samples, where the caller has them */
    int16_t *y;           /* [C][out_cap] */
    const int16_t *phase; /* [I][plen] */
    int16_t *tail_out;    /* [C][tail_cap]: what this call leaves unconsumed (mfm_rs_tail_kernel) */
    size_t in_stride;
    uint32_t tail_len, nr_in, tail_cap, out_cap, n_out, plen, interp, decim, p0, nchan, invert;
    uint32_t pos_end, new_tail;
};

constexpr uint32_t RS_NT = 256, RS_OPT = 4, RS_OPB = RS_NT * RS_OPT; /* threads, outputs per thread / per block */
constexpr uint32_t RS_PAIRS_MAX = 32;                                /* register-resident phase: up to 64 taps */

/* sample v of channel c in the virtual stream "tail, then the new samples"; decoder -i negates on int16 storage */
__device__ __forceinline__ int16_t rs_sample(const RsLaunch &L, uint32_t c, uint32_t v)
{
    int16_t s = 0;
    if (v < L.tail_len) {
        s = L.tail[(size_t)c * L.tail_cap + v];
    } else if (v - L.tail_len < L.nr_in) {
        s = L.pcm[(size_t)c * L.in_stride + (v - L.tail_len)];
    }
    return L.invert ? (int16_t)(-s) : s; /* decoder.c:624: samp[i] *= -1 */
}

/* NP > 0: coefficient pairs of the thread's phase in registers, NP = pairs per phase rounded up to a multiple of 4
 * (the padding pairs are zero); NP = 0: pairs read from LDS, any phase length */
template <int NP>
__global__ __launch_bounds__(RS_NT) void mfm_resample_kernel(const RsLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t rs_smem[];
    uint32_t *ph_s = reinterpret_cast<uint32_t *>(rs_smem);                 /* [I][plen / 2] coefficient pairs */
    const uint32_t npairs = L.plen / 2u;                                     /* plen is a multiple of 4 */
    int16_t *x_s = reinterpret_cast<int16_t *>(rs_smem + (size_t)L.interp * L.plen * 2u);
    const uint32_t c = blockIdx.y, tid = threadIdx.x;
    const uint32_t j0 = blockIdx.x * RS_OPB;
    const uint32_t j1 = (j0 + RS_OPB < L.n_out ? j0 + RS_OPB : L.n_out) - 1u; /* last output of the block */
    /* filter/polyphase_fir.c:206-211 unrolled to output j: position (p0 + j D) / I, phase (p0 + j D) % I */
    /* p0 + j D fits 32 bits for every output of a call (create() checks out_cap * D < 2^32) */
    const uint32_t in_lo = ((L.p0 + j0 * L.decim) / L.interp) & ~1u;
    const uint32_t in_hi = (L.p0 + j1 * L.decim) / L.interp + L.plen + 2u + 8u; /* + the zero-padded pairs */
    for (uint32_t i = tid; i < L.interp * npairs; i += RS_NT) {
        ph_s[i] = reinterpret_cast<const uint32_t *>(L.phase)[i];
    }
    for (uint32_t v = in_lo + tid; v < in_hi; v += RS_NT) {
        x_s[v - in_lo] = rs_sample(L, c, v);
    }
    __syncthreads();
    const uint32_t *x32 = reinterpret_cast<const uint32_t *>(x_s);

    /* one division per thread: its outputs are RS_NT apart, so position and phase advance by constants */
    const uint32_t t_first = L.p0 + (j0 + tid) * L.decim;
    uint32_t pos_abs = t_first / L.interp, ph = t_first - pos_abs * L.interp;
    const uint32_t step_pos = (RS_NT * L.decim) / L.interp, step_ph = (RS_NT * L.decim) % L.interp;
    constexpr bool REGCOEF = NP > 0;
    uint32_t cw[REGCOEF ? NP : 1];
    if (REGCOEF) {
        /* 256 * D is a multiple of I: outputs tid, tid + 256, ... of this block have the same phase */
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)NP; i++) {
            cw[i] = i < npairs ? ph_s[ph * npairs + i] : 0u;
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < RS_OPT; u++) {
        const uint32_t j = j0 + tid + u * RS_NT;
        if (j >= L.n_out) {
            break;
        }
        const uint32_t pos = pos_abs - in_lo;
        const uint32_t *xw = x32 + (pos >> 1);
        const uint32_t sh = (pos & 1u) * 16u;
        int32_t acc = 0; /* filter/utils.c:94-103, int32 wrap-around */
        uint32_t lo = xw[0];
        if (REGCOEF) {
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)NP; i++) {
                const uint32_t hi = xw[i + 1];
                const uint32_t pr = __builtin_amdgcn_alignbit(hi, lo, sh); /* samples pos + 2i, pos + 2i + 1 */
                asm("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(acc) : "v"(pr), "v"(cw[i]));
                lo = hi;
            }
        } else {
            const uint32_t *cp = ph_s + ph * npairs;
            for (uint32_t i = 0; i < npairs; i++) {
                const uint32_t hi = xw[i + 1];
                const uint32_t pr = __builtin_amdgcn_alignbit(hi, lo, sh);
                asm("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(acc) : "v"(pr), "v"(cp[i]));
                lo = hi;
            }
        }
        asm volatile("s_nop 2" : "+v"(acc)); /* a DOT result needs 3 wait states before other VALU code reads it */
        L.y[(size_t)c * L.out_cap + j] = (int16_t)mfm_r14_wide(acc); /* utils.c:112 */
        pos_abs += step_pos;
        ph += step_ph;
        if (ph >= L.interp) {
            ph -= L.interp;
            pos_abs += 1u;
        }
    }
}

/* what this call leaves unconsumed: samples pos_end .. pos_end + new_tail of the virtual stream (new_tail <= plen) */
__global__ void mfm_rs_tail_kernel(const RsLaunch L)
{
    const uint32_t c = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < L.new_tail; i += blockDim.x) {
        int16_t s = 0;
        const uint32_t v = L.pos_end + i;
        if (v < L.tail_len) {
            s = L.tail[(size_t)c * L.tail_cap + v];
        } else if (v - L.tail_len < L.nr_in) {
            s = L.pcm[(size_t)c * L.in_stride + (v - L.tail_len)];
        }
        L.tail_out[(size_t)c * L.tail_cap + i] = s; /* stored as received: inversion is applied on use */
    }
}

/* ---- matrix-core form ------------------------------------------------------------------------------------- */

typedef int rs_v4i __attribute__((ext_vector_type(4)));

struct RsMLaunch {
    RsLaunch b;
    const rs_v4i *gfrag; /* [2 planes: Wh, Wl][KS][64 lanes] A fragments of G for this call's carried phase */
    const int32_t *krow; /* [16]: 128 * sum of row q of G (the x = ... + 128 term) */
    uint32_t R;          /* samples per block of 16 outputs = 16 D / I */
    uint32_t rp;         /* bytes per row in LDS: R rounded up to a multiple of 16 */
    uint32_t plane;      /* bytes per byte plane in LDS */
};

constexpr uint32_t RSM_NT = 256;  /* threads per workgroup (4 waves) */
constexpr uint32_t RSM_NB = 256;  /* blocks of 16 outputs per workgroup: each wave does four column groups of 16 blocks */

struct __attribute__((packed, aligned(2))) RsPcm8 {
    int16_t v[8];
};

template <int KS>
__global__ __launch_bounds__(RSM_NT) void mfm_resample_mfma_kernel(const RsMLaunch M)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t rs_smem[];
    const RsLaunch &L = M.b;
    uint8_t *const pl_h = rs_smem, *const pl_l = rs_smem + M.plane;
    const uint32_t c = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t b0 = blockIdx.x * RSM_NB;                  /* first block of this workgroup */
    const uint32_t nrows = RSM_NB + (64u * KS + M.rp - 1u) / M.rp; /* image rows: the last block's window is 64 KS bytes long */
    const uint32_t v0 = b0 * M.R;

    /* ---- stage: a work item = eight samples of one row, as two 8-byte LDS writes (one per byte plane).  The last
     * group of a row runs into the padding (8 ceil(R / 8) <= rp) and carries the next row's first samples there -
     * G is zero over the padding ---- */
    const uint32_t g8 = (M.R + 7u) / 8u;                       /* groups per row */
    for (uint32_t it = tid; it < nrows * g8; it += RSM_NT) {
        const uint32_t row = it / g8, grp = it - row * g8;
        const uint32_t v = v0 + row * M.R + 8u * grp;
        uint32_t w[4];
        if (v >= L.tail_len && v + 8u <= L.tail_len + L.nr_in) {
            const RsPcm8 p = *reinterpret_cast<const RsPcm8 *>(L.pcm + (size_t)c * L.in_stride + (v - L.tail_len));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                w[i] = mfm_pack16(p.v[2 * i], p.v[2 * i + 1]);
            }
            if (L.invert) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    w[i] = mfm_pack16(-(int32_t)p.v[2 * i], -(int32_t)p.v[2 * i + 1]); /* decoder.c:624 on int16 storage */
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) { /* the tail of the previous call, the seam, zeros past the end */
                w[i] = mfm_pack16(rs_sample(L, c, v + 2u * (uint32_t)i), rs_sample(L, c, v + 2u * (uint32_t)i + 1u));
            }
        }
        uint2 lo, hi;
        lo.x = __builtin_amdgcn_perm(w[1], w[0], 0x06040200u) ^ 0x80808080u; /* Xl = (x & 255) - 128 */
        lo.y = __builtin_amdgcn_perm(w[3], w[2], 0x06040200u) ^ 0x80808080u;
        hi.x = __builtin_amdgcn_perm(w[1], w[0], 0x07050301u);                /* Xh = x >> 8 */
        hi.y = __builtin_amdgcn_perm(w[3], w[2], 0x07050301u);
        const uint32_t at = row * M.rp + 8u * grp;
        *reinterpret_cast<uint2 *>(pl_l + at) = lo;
        *reinterpret_cast<uint2 *>(pl_h + at) = hi;
    }
    __syncthreads();

    /* ---- multiply: lane (kg, n) reads 16 bytes of column (block) n at element 64 ks + 16 kg ---- */
    const uint32_t kg = lane >> 4, n = lane & 15u;
    rs_v4i a_h[KS], a_l[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        a_h[ks] = M.gfrag[(0 * KS + ks) * 64 + lane];
        a_l[ks] = M.gfrag[(1 * KS + ks) * 64 + lane];
    }
    rs_v4i kr;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        kr[i] = M.krow[4u * kg + (uint32_t)i];
    }
#pragma unroll
    for (uint32_t g = 0; g < 4; g++) {
        const uint32_t blk = wave * 64u + g * 16u + n;          /* block within the workgroup */
        const uint32_t off = blk * M.rp + 16u * kg;
        rs_v4i hh = { 0, 0, 0, 0 }, md = hh, ll = hh;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const rs_v4i b_h = *reinterpret_cast<const rs_v4i *>(pl_h + off + 64u * ks);
            const rs_v4i b_l = *reinterpret_cast<const rs_v4i *>(pl_l + off + 64u * ks);
            hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[ks], b_h, hh, 0, 0, 0);
            md = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_h[ks], b_l, md, 0, 0, 0);
            ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[ks], b_l, ll, 0, 0, 0);
            md = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_l[ks], b_h, md, 0, 0, 0);
        }
        /* rows 4 kg .. 4 kg + 3 of column n = outputs 16 (b0 + blk) + 4 kg + i */
        const uint32_t j = 16u * (b0 + blk) + 4u * kg;
        int32_t y[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t acc = (uint32_t)ll[i] + ((uint32_t)md[i] << 8) + ((uint32_t)hh[i] << 16) + (uint32_t)kr[i];
            y[i] = mfm_r14_wide((int32_t)acc); /* utils.c:112 */
        }
        int16_t *dst = L.y + (size_t)c * L.out_cap + j;
        if (j + 4u <= L.n_out) {
            uint2 w;
            w.x = mfm_pack16(y[0], y[1]);
            w.y = mfm_pack16(y[2], y[3]);
            *reinterpret_cast<uint2 *>(dst) = w; /* out_cap and j are multiples of 4: 8-byte aligned */
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (j + (uint32_t)i < L.n_out) {
                    dst[i] = (int16_t)y[i];
                }
            }
        }
    }
}

struct DcState {
    int32_t x_n_1, y_n_1, acc;
};

/*
 * filter/dc_blocker.h:80-90, one lane per channel.  The loop is a feedback loop with a truncating shift inside (not
 * associative, and two trajectories that start apart take ~2^14 / p samples to meet, so there is nothing to split in
 * time): what can be done is to keep the per-sample work down to the recurrence itself.  With acc_n = (x_n << 14) + r_n
 * the reference's four statements are  r_n = r_(n-1) - p y_(n-1),  y_n = (int32)((x_n << 14) + r_n) >> 14  (all mod
 * 2^32, as the reference's int32 arithmetic wraps); samples are read and written eight at a time (16-byte accesses, the
 * next eight requested before the current eight are worked on), so a sample costs a few instructions instead of a global
 * load - store round trip (94 ms -> 6.2 ms per 447 392-sample block in round 2; 5.8 ms with the three-instruction chain of
 * round 5).  What bounds it now is the chain itself: y -> multiply-add -> multiply-add -> shift -> y is three dependent
 * instructions, 26 cycles per sample as measured, and one wave of 64 channels has nothing to put in between; a second
 * formulation with a chain of two (u_n = y_(n-1) * -p + (u_(n-1) + dX_n), y_n = u_n >> 14) needs five instructions per
 * sample and would land at ~4.3 ms.  The (Y, r) form (acc = 2^14 Y + r) has the same chain length as this one.  Off by
 * default in the reference (decoder -b), not on multifm's path.
 */
__global__ __launch_bounds__(64) void mfm_dc_block_kernel(int16_t *y, uint32_t out_cap, uint32_t n_out, uint32_t nchan, int32_t p,
                                                          DcState *st)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchan) {
        return;
    }
    const DcState s0 = st[c];
    /* r = acc - x_(n-1) << 14 in the reference's variables */
    uint32_t r = (uint32_t)s0.acc - (uint32_t)s0.x_n_1;
    int32_t yp = s0.y_n_1, xl = 0;
    int16_t *yy = y + (size_t)c * out_cap;
    uint32_t i = 0;
    /* 64 samples per trip: the next 64 are requested (eight 16-byte loads in flight) before the current 64 are worked
     * on - a lane's row is its own cache lines, so a load that is waited for costs a full memory round trip */
    constexpr uint32_t V = 8;
    if (n_out >= 8u * V) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); /* (HIP's uint4 is a struct of unions: arrays of it end up in scratch) */
        const u32x4 *src = reinterpret_cast<const u32x4 *>(yy); /* out_cap is a multiple of 8: rows are 16-byte aligned */
        u32x4 *dst = reinterpret_cast<u32x4 *>(yy);
        u32x4 cur[V], nxt[V];
#pragma unroll
        for (uint32_t k = 0; k < V; k++) {
            cur[k] = src[k];
        }
        const int32_t np = -p; /* |p| < 2^15 and |y| <= 2^17: the products fit the 24-bit multiplier (one v_mad_i32_i24) */
        for (; i + 8u * V <= n_out; i += 8u * V) {
            /* the next 64 samples; behind the last whole group the same ones again (a valid address, nobody uses them) */
            const u32x4 *pn = src + (i + 16u * V <= n_out ? i / 8u + V : i / 8u);
#pragma unroll
            for (uint32_t k = 0; k < V; k++) {
                nxt[k] = pn[k];
            }
#pragma unroll
            for (uint32_t k = 0; k < V; k++) {
                const u32x4 w = cur[k];
                u32x4 o;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    /* three instructions per sample on the chain y -> r -> acc -> y (round 5; five before): the multiply-add of
                     * the leak, (x << 14) + r as ONE v_mad_i32_i16 that picks the sample's half of the packed word itself
                     * (x * 16384 + r, wrapping like the reference's int32), the arithmetic shift; half a v_perm to pack */
                    uint32_t acc;
                    r += (uint32_t)__mul24(yp, np);
                    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(acc) : "v"(w[d]), "s"(16384), "v"(r));
                    yp = (int32_t)acc >> 14;
                    const uint32_t lo = (uint32_t)yp;
                    r += (uint32_t)__mul24(yp, np);
                    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(acc) : "v"(w[d]), "s"(16384), "v"(r));
                    yp = (int32_t)acc >> 14;
                    o[d] = __builtin_amdgcn_perm((uint32_t)yp, lo, 0x05040100u); /* low halves of (lo, yp) */
                }
                xl = (int32_t)w[3] >> 16; /* the group's last sample: x_(n-1) of whatever follows */
                dst[i / 8u + k] = o;
            }
#pragma unroll
            for (uint32_t k = 0; k < V; k++) {
                cur[k] = nxt[k];
            }
        }
    }
    for (; i < n_out; i++) {
        r -= (uint32_t)p * (uint32_t)yp;
        xl = (int32_t)yy[i];
        yp = (int32_t)(((uint32_t)xl << 14) + r) >> 14;
        yy[i] = (int16_t)yp;
    }
    if (n_out) {
        DcState s;
        s.x_n_1 = (int32_t)((uint32_t)xl << 14);
        s.acc = (int32_t)(((uint32_t)xl << 14) + r);
        s.y_n_1 = yp;
        st[c] = s;
    }
}

thread_local char g_rs_error[256] = "";

} /* namespace */

struct mfm_resampler {
    mfm_resampler_config cfg{};
    uint32_t plen = 0;
    uint32_t in_cap = 0, out_cap = 0;
    int16_t *d_phase = nullptr;
    int16_t *d_x[2] = { nullptr, nullptr }; /* [C][tail_cap] ping-pong: the samples the last call left unconsumed */
    uint32_t tail_cap = 0;
    uint32_t lds_bytes = 0;
    bool reg_coef = false;
    int16_t *d_y = nullptr;
    DcState *d_dc = nullptr;
    int32_t dc_p = 0;
    int cur = 0;
    uint32_t tail = 0; /* unconsumed samples in d_x[cur] */
    uint32_t phase_id = 0;
    int16_t *d_stage = nullptr; /* process_host_to_device: [C][max_in_samples] */
    /* matrix-core form */
    bool use_mfma = false;
    uint32_t m_ks = 0, m_R = 0, m_rp = 0, m_plane = 0, m_lds = 0;
    rs_v4i *d_gfrag = nullptr; /* [I][2][KS][64] */
    int32_t *d_krow = nullptr; /* [I][16] */
};

#define RS_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t err_ = (expr);                                                                            \
        if (err_ != hipSuccess) {                                                                            \
            snprintf(g_rs_error, sizeof(g_rs_error), "%s failed: %s", #expr, hipGetErrorString(err_));       \
            mfm_internal_set_error(g_rs_error);                                                              \
            return MFM_E_DEVICE;                                                                             \
        }                                                                                                    \
    } while (0)

extern "C" {

int mfm_resampler_create(struct mfm_resampler **pr, const struct mfm_resampler_config *cfg, const int16_t *coeffs,
                         size_t nr_coeffs)
{
    if (!pr || !cfg || !coeffs || 0 == nr_coeffs) {
        return MFM_E_INVAL;
    }
    *pr = nullptr;
    if (cfg->abi_version != MFM_ABI_VERSION || 0 == cfg->interpolate || 0 == cfg->decimate || 0 == cfg->nr_channels ||
        0 == cfg->max_in_samples) {
        return MFM_E_INVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) {
        return MFM_E_DEVICE; /* no CPU path */
    }
    mfm_resampler *r = new (std::nothrow) mfm_resampler();
    if (!r) {
        return MFM_E_NOMEM;
    }
    r->cfg = *cfg;
    /* filter/polyphase_fir.c:70-83 */
    uint32_t plen = (uint32_t)((nr_coeffs + cfg->interpolate - 1) / cfg->interpolate);
    plen = (plen + 3u) & ~3u;
    r->plen = plen;
    /* the walk must not step past the samples it has (one output consumes at most ceil(D/I) samples) */
    if ((cfg->decimate + cfg->interpolate - 1) / cfg->interpolate > plen) {
        delete r;
        return MFM_E_INVAL;
    }
    std::vector<int16_t> ph((size_t)cfg->interpolate * plen, 0);
    for (size_t i = 0; i < nr_coeffs; i++) {
        ph[(i % cfg->interpolate) * plen + (i / cfg->interpolate)] = coeffs[i];
    }
    r->in_cap = cfg->max_in_samples + plen + 64;
    r->out_cap = (uint32_t)(((uint64_t)r->in_cap * cfg->interpolate) / cfg->decimate + 8);
    r->out_cap = (r->out_cap + 7u) & ~7u; /* rows of the output start 16-byte aligned (the matrix-core form stores four outputs at
                                           * once, the DC blocker reads and writes eight) */
    if ((uint64_t)r->out_cap * cfg->decimate >= (1ull << 32)) {
        delete r;
        return MFM_E_INVAL;
    }
    if (cfg->dc_block) {
        r->dc_p = (int16_t)((1.0 - cfg->dc_pole) * 16384.0); /* filter/dc_blocker.h:56 */
    }
    *pr = r;
    RS_TRY(hipSetDevice(cfg->device));
    RS_TRY(hipMalloc(&r->d_phase, ph.size() * 2));
    RS_TRY(hipMemcpy(r->d_phase, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
    r->tail_cap = plen + 8; /* never more than plen samples are left over (see process_device) */
    for (int i = 0; i < 2; i++) {
        RS_TRY(hipMalloc(&r->d_x[i], (size_t)cfg->nr_channels * r->tail_cap * 2));
        RS_TRY(hipMemset(r->d_x[i], 0, (size_t)cfg->nr_channels * r->tail_cap * 2));
    }
    /* LDS of a workgroup: coefficient pairs + the input window of RS_OPB outputs */
    r->lds_bytes = (uint32_t)((size_t)cfg->interpolate * plen * 2 +
                              (((uint64_t)RS_OPB * cfg->decimate) / cfg->interpolate + plen + 32) * 2);
    r->lds_bytes = (r->lds_bytes + 15u) & ~15u;
    if (r->lds_bytes > 150u * 1024u) {
        snprintf(g_rs_error, sizeof(g_rs_error), "resampling ratio %u/%u with %u taps per phase needs %u bytes of LDS per workgroup",
                 cfg->interpolate, cfg->decimate, plen, r->lds_bytes);
        mfm_internal_set_error(g_rs_error);
        return MFM_E_INVAL;
    }
    r->reg_coef = ((uint64_t)RS_NT * cfg->decimate) % cfg->interpolate == 0 && plen / 2 <= RS_PAIRS_MAX;
    if (r->lds_bytes > 48u * 1024u) {
        /* only extreme decimation ratios get here: they use the LDS-coefficient variant */
        r->reg_coef = false;
        RS_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mfm_resample_kernel<0>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
    }
    /* ---- matrix-core form: y[16 m + q] = sum_s G[q][s] x[R m + s], one G per carried phase (see the head of the file) ---- */
    {
        const uint32_t I = cfg->interpolate, D = cfg->decimate;
        bool ok = (16u * D) % I == 0u && !(cfg->flags & MFM_RS_FORCE_DOT2);
        const uint32_t R = ok ? 16u * D / I : 0u;
        const uint32_t rp = (R + 15u) & ~15u;
        /* output q of a block starts its window floor((phi + q D) / I) samples into the block, phi < I */
        const uint32_t max_off = (I - 1u + 15u * D) / I;
        uint32_t K = 0;
        if (ok) {
            const uint32_t last = max_off + plen - 1u;                  /* last sample index (from the block's first) with a coefficient */
            K = (last / R) * rp + last % R + 1u;                        /* its byte position in the padded rows, + 1 */
            K = (K + 63u) & ~63u;
            ok = K <= 256u && R <= 240u;
        }
        for (size_t i = 0; ok && i < ph.size(); i++) {
            ok = ph[i] >= -32639 && ph[i] <= 32639;                     /* W = 256 Wh + Wl with both in [-128, 127] */
        }
        if (ok) {
            const uint32_t KS = K / 64u;
            std::vector<int8_t> frag((size_t)I * 2u * KS * 64u * 16u, 0);
            std::vector<int32_t> krow((size_t)I * 16u, 0);
            for (uint32_t phi = 0; phi < I; phi++) {
                for (uint32_t q = 0; q < 16; q++) {
                    const uint32_t t = phi + q * D, off = t / I, phq = t % I;
                    uint32_t sum = 0;
                    for (uint32_t kk = 0; kk < K; kk++) {
                        const uint32_t row = kk / rp, col = kk % rp;
                        int32_t w = 0;
                        if (col < R) {
                            const int64_t tap = (int64_t)(row * R + col) - (int64_t)off;
                            if (tap >= 0 && tap < (int64_t)plen) {
                                w = ph[(size_t)phq * plen + (size_t)tap];
                            }
                        }
                        sum += (uint32_t)w;
                        const int32_t wl = (int8_t)(w & 0xff), wh = (w - wl) >> 8;
                        /* v_mfma_i32_16x16x64_i8 A operand: lane (kg = lane >> 4, i = lane & 15) holds row i, elements 64 ks + 16 kg + j */
                        const uint32_t ks = kk / 64u, kgq = (kk % 64u) / 16u, j = kk % 16u, ln = kgq * 16u + q;
                        frag[((((size_t)phi * 2u + 0u) * KS + ks) * 64u + ln) * 16u + j] = (int8_t)wh;
                        frag[((((size_t)phi * 2u + 1u) * KS + ks) * 64u + ln) * 16u + j] = (int8_t)wl;
                    }
                    krow[(size_t)phi * 16u + q] = (int32_t)(128u * sum);
                }
            }
            r->use_mfma = true;
            r->m_ks = KS;
            r->m_R = R;
            r->m_rp = rp;
            const uint32_t nrows = RSM_NB + (K + rp - 1u) / rp;
            r->m_plane = (nrows * rp + 63u) & ~63u;
            r->m_lds = 2u * r->m_plane;
            RS_TRY(hipMalloc(&r->d_gfrag, frag.size()));
            RS_TRY(hipMemcpy(r->d_gfrag, frag.data(), frag.size(), hipMemcpyHostToDevice));
            RS_TRY(hipMalloc(&r->d_krow, krow.size() * 4));
            RS_TRY(hipMemcpy(r->d_krow, krow.data(), krow.size() * 4, hipMemcpyHostToDevice));
        }
    }
    RS_TRY(hipMalloc(&r->d_y, (size_t)cfg->nr_channels * r->out_cap * 2));
    RS_TRY(hipMalloc(&r->d_dc, (size_t)cfg->nr_channels * sizeof(DcState)));
    RS_TRY(hipMemset(r->d_dc, 0, (size_t)cfg->nr_channels * sizeof(DcState)));
    RS_TRY(hipDeviceSynchronize());
    return MFM_OK;
}

void mfm_resampler_destroy(struct mfm_resampler **pr)
{
    if (!pr || !*pr) {
        return;
    }
    mfm_resampler *r = *pr;
    (void)hipSetDevice(r->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(r->d_phase);
    (void)hipFree(r->d_x[0]);
    (void)hipFree(r->d_x[1]);
    (void)hipFree(r->d_y);
    (void)hipFree(r->d_dc);
    (void)hipFree(r->d_stage);
    (void)hipFree(r->d_gfrag);
    (void)hipFree(r->d_krow);
    delete r;
    *pr = nullptr;
}

size_t mfm_resampler_max_out(const struct mfm_resampler *r)
{
    return r ? r->out_cap : 0;
}

int mfm_resampler_process_device(struct mfm_resampler *r, const int16_t *d_pcm, size_t in_stride, size_t nr_in,
                                 void *stream, int16_t **d_out, size_t *out_stride, size_t *nr_out)
{
    if (!r || !d_pcm || !d_out || !out_stride || !nr_out) {
        return MFM_E_INVAL;
    }
    if (nr_in > r->cfg.max_in_samples) {
        return MFM_E_INVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t C = r->cfg.nr_channels, I = r->cfg.interpolate, D = r->cfg.decimate;
    RS_TRY(hipSetDevice(r->cfg.device));
    const uint32_t total = r->tail + (uint32_t)nr_in;
    /* outputs m with total - pos_m > plen  <=>  p0 + m*D < (total - plen) * I   (polyphase_fir.c:184) */
    uint32_t n_out = 0;
    if (total > r->plen) {
        const uint64_t lim = (uint64_t)(total - r->plen) * I;
        if (lim > r->phase_id) {
            n_out = (uint32_t)((lim - r->phase_id + D - 1) / D);
        }
    }
    const uint64_t t_end = (uint64_t)r->phase_id + (uint64_t)n_out * D;
    const uint32_t pos_end = (uint32_t)(t_end / I);
    /* pos_end <= total (create() checked ceil(D/I) <= plen) and total - pos_end <= plen: with outputs, t_end >= (total -
     * plen) * I; without, total <= plen */
    const uint32_t new_tail = total - pos_end;
    RsLaunch L{ r->d_x[r->cur], d_pcm, r->d_y, r->d_phase, r->d_x[r->cur ^ 1], in_stride, r->tail, (uint32_t)nr_in, r->tail_cap,
                r->out_cap, n_out, r->plen, I, D, r->phase_id, C, r->cfg.invert, pos_end, new_tail };
    if (n_out && r->use_mfma) {
        RsMLaunch M{ L, r->d_gfrag + (size_t)r->phase_id * 2u * r->m_ks * 64u, r->d_krow + (size_t)r->phase_id * 16u, r->m_R, r->m_rp,
                     r->m_plane };
        const dim3 grid((n_out + 16u * RSM_NB - 1u) / (16u * RSM_NB), C);
        switch (r->m_ks) {
        case 1: hipLaunchKernelGGL(mfm_resample_mfma_kernel<1>, grid, dim3(RSM_NT), r->m_lds, s, M); break;
        case 2: hipLaunchKernelGGL(mfm_resample_mfma_kernel<2>, grid, dim3(RSM_NT), r->m_lds, s, M); break;
        case 3: hipLaunchKernelGGL(mfm_resample_mfma_kernel<3>, grid, dim3(RSM_NT), r->m_lds, s, M); break;
        default: hipLaunchKernelGGL(mfm_resample_mfma_kernel<4>, grid, dim3(RSM_NT), r->m_lds, s, M); break;
        }
        RS_TRY(hipGetLastError());
    } else if (n_out) {
        const dim3 grid((n_out + RS_OPB - 1) / RS_OPB, C);
        const uint32_t np4 = r->reg_coef ? (r->plen / 2u + 3u) / 4u : 0u; /* register variant: pairs rounded up to 4 */
        switch (np4) {
        case 1: hipLaunchKernelGGL(mfm_resample_kernel<4>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        case 2: hipLaunchKernelGGL(mfm_resample_kernel<8>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        case 3: hipLaunchKernelGGL(mfm_resample_kernel<12>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        case 4: hipLaunchKernelGGL(mfm_resample_kernel<16>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        case 5: hipLaunchKernelGGL(mfm_resample_kernel<20>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        case 6: hipLaunchKernelGGL(mfm_resample_kernel<24>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        case 7: hipLaunchKernelGGL(mfm_resample_kernel<28>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        case 8: hipLaunchKernelGGL(mfm_resample_kernel<32>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        default: hipLaunchKernelGGL(mfm_resample_kernel<0>, grid, dim3(RS_NT), r->lds_bytes, s, L); break;
        }
        RS_TRY(hipGetLastError());
    }
    if (n_out) {
        if (r->cfg.dc_block) {
            hipLaunchKernelGGL(mfm_dc_block_kernel, dim3((C + 63) / 64), dim3(64), 0, s, r->d_y, r->out_cap, n_out, C,
                               r->dc_p, r->d_dc);
            RS_TRY(hipGetLastError());
        }
    }
    if (new_tail) {
        hipLaunchKernelGGL(mfm_rs_tail_kernel, dim3(C), dim3(64), 0, s, L);
        RS_TRY(hipGetLastError());
    }
    r->phase_id = (uint32_t)(t_end % I);
    r->tail = new_tail;
    r->cur ^= 1;
    *d_out = r->d_y;
    *out_stride = r->out_cap;
    *nr_out = n_out;
    return MFM_OK;
}

int mfm_resampler_process_host_to_device(struct mfm_resampler *r, const int16_t *pcm, size_t in_stride, size_t nr_in,
                                         void *stream, int16_t **d_out, size_t *out_stride, size_t *nr_out)
{
    if (!r || (!pcm && nr_in) || nr_in > r->cfg.max_in_samples) {
        return MFM_E_INVAL;
    }
    RS_TRY(hipSetDevice(r->cfg.device));
    const uint32_t C = r->cfg.nr_channels;
    if (!r->d_stage) {
        RS_TRY(hipMalloc(&r->d_stage, (size_t)C * r->cfg.max_in_samples * 2));
    }
    if (nr_in) {
        RS_TRY(hipMemcpy2DAsync(r->d_stage, (size_t)r->cfg.max_in_samples * 2, pcm, in_stride * 2, nr_in * 2, C,
                                hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    }
    return mfm_resampler_process_device(r, r->d_stage, r->cfg.max_in_samples, nr_in, stream, d_out, out_stride, nr_out);
}

int mfm_resampler_process_host(struct mfm_resampler *r, const int16_t *pcm, size_t in_stride, size_t nr_in,
                               int16_t *out, size_t out_stride, size_t *nr_out)
{
    if (!r || !pcm || !out || !nr_out) {
        return MFM_E_INVAL;
    }
    RS_TRY(hipSetDevice(r->cfg.device));
    int16_t *d_in = nullptr;
    const uint32_t C = r->cfg.nr_channels;
    RS_TRY(hipMalloc(&d_in, (size_t)C * (nr_in ? nr_in : 1) * 2));
    if (nr_in) {
        RS_TRY(hipMemcpy2D(d_in, nr_in * 2, pcm, in_stride * 2, nr_in * 2, C, hipMemcpyHostToDevice));
    }
    int16_t *d_out = nullptr;
    size_t ostr = 0, n = 0;
    int rc = mfm_resampler_process_device(r, d_in, nr_in, nr_in, nullptr, &d_out, &ostr, &n);
    if (rc == MFM_OK && n) {
        if (n > out_stride) {
            rc = MFM_E_INVAL;
        } else {
            hipError_t e = hipMemcpy2D(out, out_stride * 2, d_out, ostr * 2, n * 2, C, hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                rc = MFM_E_DEVICE;
            }
        }
    }
    (void)hipDeviceSynchronize();
    (void)hipFree(d_in);
    *nr_out = n;
    return rc;
}

} /* extern "C" */
