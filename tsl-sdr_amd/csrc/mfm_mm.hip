/*
 * mfm_mm.hip - Mueller-Muller clock recovery for all channels of a PCM block (pager/mueller_muller.c:10-115;
 * BASELINE.json configs[3] names it as the slicer in front of the pager stage).
 *
 * The loop is a feedback loop - where the next decision is taken depends on the last one - so a channel is
 * sequential and its time is the length of that dependent chain (about 15 float operations and one sample read per
 * decision); channels are independent.  One lane per channel, MM_CH channels per one-wave workgroup: few channels per
 * wave, because the chain costs a wave the same time whether 16 or 64 of its lanes walk, while the staging it has to
 * do grows with the rows.  The samples go through an LDS ring of two chunks per row: the whole wave requests the next
 * chunk of every row into registers (coalesced 16-byte loads, all in flight at once), the lanes walk the part of the
 * ring that is already there while those loads are under way, then the registers are written behind it.  Same float
 * operations in the same order as the reference (no contraction: mul, mul, sub; mul, add; compare/clamp; mul, add,
 * add; floorf), so the decisions are bit-identical to oracle/pocsag_oracle.c's restatement.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <new>

#include "../../include/multifm_hip.h"

extern "C" void mfm_internal_set_error(const char *msg);

namespace {

constexpr uint32_t MM_CH = 16;                    /* channels (walking lanes) per workgroup */
constexpr uint32_t MM_W = 1024;                   /* samples per row and chunk */
constexpr uint32_t MM_R = 2 * MM_W;               /* ring: the chunk being walked and the one being fetched */
constexpr uint32_t MM_PITCH = MM_R * 2 + 16;      /* bytes per ring row: 16-byte aligned, rows four banks apart */
constexpr uint32_t MM_NV = MM_CH * MM_W / 8 / 64; /* 16-byte vectors per lane and chunk */
constexpr uint32_t MM_LDS = MM_CH * MM_PITCH;

typedef uint32_t mm_u32x4 __attribute__((ext_vector_type(4)));

struct MmState {
    float w, m, next_offset, last_sample;
};

struct MmLaunch {
    const int16_t *pcm;
    int16_t *dec;
    uint32_t *counts;
    MmState *st;
    uint32_t in_stride, nr_in, readable, dec_stride, dec_cap, nchan, smin;
    float kw, km, error_min, error_max;
};

/* VEC: rows of at least 8 readable samples (anything shorter goes through the partial-vector path alone) */
template <bool VEC>
__global__ __launch_bounds__(64) void mfm_mm_kernel(const MmLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_ring[];
    const uint32_t lane = threadIdx.x, c = blockIdx.x * MM_CH + lane;
    const bool live = lane < MM_CH && c < L.nchan;
    MmState s = live ? L.st[c] : MmState{ 0.f, 0.f, 0.f, 0.f };
    const float nr_f = (float)L.nr_in;                      /* mueller_muller.c:58 */
    uint32_t idx = (uint32_t)(s.next_offset + 0.5f);        /* :66 */
    uint32_t n_dec = 0;
    int16_t *dec = L.dec + (size_t)(live ? c : 0u) * L.dec_stride;

    /* the ring starts at the slowest lane's next sample */
    uint32_t s0 = live && idx < L.nr_in ? idx : 0xffffffffu;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)s0, off);
        s0 = o < s0 ? o : s0;
    }
    if (s0 != 0xffffffffu) {
        s0 &= ~(MM_W - 1u); /* chunks are MM_W-aligned: a chunk is one half of the ring */
        /* vector j of this lane: row j / 2, samples (j % 2) * 512 + lane * 8 .. + 7 of the chunk */
        const uint32_t col8 = lane * 8u;
        const uint32_t last_full = VEC ? L.readable - 8u : 0u;
        const uint32_t tail0 = L.readable & ~7u; /* samples [tail0, readable) are not covered by whole vectors */
        mm_u32x4 v[MM_NV];

        auto fetch = [&](uint32_t base) {
#pragma unroll
            for (uint32_t j = 0; j < MM_NV; j++) {
                const uint32_t r = j >> 1, ch = blockIdx.x * MM_CH + r;
                const int16_t *row = L.pcm + (size_t)(ch < L.nchan ? ch : L.nchan - 1u) * L.in_stride;
                uint32_t pos = base + (j & 1u) * 512u + col8;
                pos = pos < last_full ? pos : last_full; /* stays inside the row; what lies at or behind tail0 is redone below */
                mm_u32x4 t = { 0u, 0u, 0u, 0u };
                if (VEC) {
                    __builtin_memcpy(&t, row + pos, 16);
                }
                v[j] = t;
            }
        };
        /* the first eight samples of the ring are repeated behind its end, so that a run of four samples that
         * starts near the end reads on without wrapping */
        auto commit = [&](uint32_t base) {
            const bool first_half = 0u == (base & MM_W);
#pragma unroll
            for (uint32_t j = 0; j < MM_NV; j++) {
                const uint32_t r = j >> 1, pos = base + (j & 1u) * 512u + col8;
                *reinterpret_cast<mm_u32x4 *>(mm_ring + r * MM_PITCH + (pos & (MM_R - 1u)) * 2u) = v[j];
                if (0u == (j & 1u) && first_half && 0u == lane) {
                    *reinterpret_cast<mm_u32x4 *>(mm_ring + r * MM_PITCH + MM_R * 2u) = v[j];
                }
            }
            if (tail0 >= base && tail0 < base + MM_W) { /* the row ends in this chunk: its last, partial vector */
                for (uint32_t f = lane; f < MM_CH * 8u; f += 64u) {
                    const uint32_t r = f >> 3, pos = tail0 + (f & 7u), ch = blockIdx.x * MM_CH + r;
                    if (pos < L.readable && ch < L.nchan) {
                        const int16_t x = L.pcm[(size_t)ch * L.in_stride + pos];
                        const uint32_t slot = pos & (MM_R - 1u);
                        *reinterpret_cast<int16_t *>(mm_ring + r * MM_PITCH + slot * 2u) = x;
                        if (slot < 8u) {
                            *reinterpret_cast<int16_t *>(mm_ring + r * MM_PITCH + (MM_R + slot) * 2u) = x;
                        }
                    }
                }
            }
        };

        fetch(s0);
        commit(s0);
        uint32_t hi = s0 + MM_W;
        const unsigned char *my = mm_ring + lane * MM_PITCH;
        /* The index of a decision is kept as an integer: the reference's float position (:58-66, :96) only ever
         * holds whole numbers below 2^23 here (mfm_mm_create bounds the block length and the step), so
         * (uint32)(cur + 0.5f) is cur and cur < nr_f is idx < nr_in. */
        float sl = (float)(s.last_sample > 0.f) - (float)(s.last_sample < 0.f);
        for (;;) {
            const bool more = hi < L.nr_in; /* every index walked is below nr_in */
            fetch(hi);                      /* always requested (the addresses are clamped): one wait pattern */
            __syncthreads();                /* one wave: the ring writes before the walk's reads */
            if (live) {
                /* The walk.  A lone wave issues one instruction every four cycles, so the time of a decision is
                 * the number of instructions it takes; and of those only the float arithmetic is on the dependent
                 * chain: the four samples the next decision can fall on - steps of L.smin .. L.smin + 3, which is
                 * every step the loop can take once it has settled - are read from the ring as one 8-byte word as
                 * soon as the index of this one is known, and the one it did fall on is shifted out when the step
                 * is. */
                const uint32_t lim = more ? hi : L.nr_in;
                bool go = idx < lim;
                int32_t raw = go ? (int32_t)*reinterpret_cast<const int16_t *>(my + (idx & (MM_R - 1u)) * 2u) : 0; /* :67 */
                while (go) {
                    const uint32_t pb = idx + L.smin;
                    uint64_t cand;
                    __builtin_memcpy(&cand, my + (pb & (MM_R - 1u)) * 2u, 8);
                    const float sample = (float)raw;
                    dec[n_dec < L.dec_cap ? n_dec : L.dec_cap] = (int16_t)raw; /* :71; the row has a spare slot */
                    n_dec++;
                    const int32_t sgn = max(-1, min(1, raw));
                    const float sc = (float)sgn;                            /* (sample > 0) - (sample < 0), :77 */
                    const float w_error = sl * sample - sc * s.last_sample; /* :77 */
                    const float w1 = s.w + w_error * L.kw;                  /* :80 */
                    s.w = __builtin_amdgcn_fmed3f(w1, L.error_min, L.error_max); /* :87-91: finite, error_min <= error_max */
                    s.m += s.w + L.km * sample; /* :93 */
                    const float fl = floorf(s.m);
                    idx += (uint32_t)fl;        /* :96 */
                    s.m -= fl;                  /* :98 */
                    s.last_sample = sample;     /* :101 */
                    sl = sc;
                    go = idx < lim;
                    const uint32_t d = idx - pb;
                    raw = (int32_t)(int16_t)(cand >> ((d & 3u) * 16u));
                    if (go && d > 3u) { /* a step outside the four: the first decisions of a stream, an unsettled loop */
                        raw = (int32_t)*reinterpret_cast<const int16_t *>(my + (idx & (MM_R - 1u)) * 2u);
                    }
                }
            }
            if (!more) {
                break;
            }
            __syncthreads();
            commit(hi);
            hi += MM_W;
        }
    }
    if (live) {
        s.next_offset = (float)idx - nr_f; /* :109 */
        L.st[c] = s;
        L.counts[c] = n_dec;
    }
}

void mm_launch(const MmLaunch &L, hipStream_t s)
{
    const dim3 grid((L.nchan + MM_CH - 1u) / MM_CH);
    if (L.readable >= 8u) {
        hipLaunchKernelGGL(mfm_mm_kernel<true>, grid, dim3(64), MM_LDS, s, L);
    } else {
        hipLaunchKernelGGL(mfm_mm_kernel<false>, grid, dim3(64), MM_LDS, s, L);
    }
}

thread_local char g_mm_error[256] = "";

} /* namespace */

struct mfm_mm {
    mfm_mm_config cfg{};
    uint32_t dec_cap = 0;
    int16_t *d_dec = nullptr;
    uint32_t *d_counts = nullptr;
    MmState *d_st = nullptr;
    int16_t *d_stage = nullptr;
    uint32_t dec_pitch = 0; /* dec_cap + 8: the slot behind a row takes the decisions that do not fit it */
    uint32_t smin = 0;
};

#define MM_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t err_ = (expr);                                                                            \
        if (err_ != hipSuccess) {                                                                            \
            snprintf(g_mm_error, sizeof(g_mm_error), "%s failed: %s", #expr, hipGetErrorString(err_));       \
            mfm_internal_set_error(g_mm_error);                                                              \
            return MFM_E_DEVICE;                                                                             \
        }                                                                                                    \
    } while (0)

extern "C" {

/* inside mfm_mm_create(): a device failure releases what exists and leaves *pm NULL */
#define MM_TRY_C(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t err_ = (expr);                                                                            \
        if (err_ != hipSuccess) {                                                                            \
            snprintf(g_mm_error, sizeof(g_mm_error), "%s failed: %s", #expr, hipGetErrorString(err_));       \
            mfm_internal_set_error(g_mm_error);                                                              \
            mfm_mm_destroy(pm);                                                                              \
            return MFM_E_DEVICE;                                                                             \
        }                                                                                                    \
    } while (0)

int mfm_mm_create(struct mfm_mm **pm, const struct mfm_mm_config *cfg)
{
    if (!pm || !cfg) {
        return MFM_E_INVAL;
    }
    *pm = nullptr;
    if (cfg->abi_version != MFM_ABI_VERSION || 0 == cfg->nr_channels || 0 == cfg->max_in_samples ||
        !(cfg->error_max >= cfg->error_min) || !(cfg->samples_per_bit >= 1.0f) ||
        !(cfg->error_min - fabsf(cfg->km) * 32768.0f >= 1.0f) ||
        !(cfg->error_max + fabsf(cfg->km) * 32768.0f < 1048576.0f) || !std::isfinite(cfg->kw) ||
        cfg->max_in_samples >= (1u << 22)) {
        /* a step that can fall below one sample would never leave the loop (:93-96); the upper bounds keep the
         * reference's float sample position a whole number the float holds exactly (block + one step < 2^23) */
        return MFM_E_INVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) {
        return MFM_E_DEVICE; /* no CPU path */
    }
    mfm_mm *m = new (std::nothrow) mfm_mm();
    if (!m) {
        return MFM_E_NOMEM;
    }
    m->cfg = *cfg;
    /* no step is shorter than floor(error_min - |km| * 32768) >= 1 samples (:87-96; the check above): a row is sized
     * for a block walked in steps of that length - the worst any accepted configuration can do - and the kernel never
     * writes past it */
    {
        const double min_step = floor((double)cfg->error_min - fabs((double)cfg->km) * 32768.0);
        m->dec_cap = (uint32_t)((double)cfg->max_in_samples / (min_step < 1.0 ? 1.0 : min_step) + 16.0);
    }
    *pm = m;
    if (hipSetDevice(cfg->device) != hipSuccess) {
        mfm_mm_destroy(pm); /* an error means *pm == NULL, and nothing of a half-built object survives */
        return MFM_E_DEVICE;
    }
    MM_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void *>(mfm_mm_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)MM_LDS));
    MM_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void *>(mfm_mm_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)MM_LDS));
    m->dec_pitch = m->dec_cap + 8u;
    MM_TRY_C(hipMalloc(&m->d_dec, (size_t)cfg->nr_channels * m->dec_pitch * 2));
    MM_TRY_C(hipMalloc(&m->d_counts, (size_t)cfg->nr_channels * 4));
    /* no step is shorter than floor(error_min - |km| * 32768) (:87-96); one below it for the rounding of the sums */
    m->smin = (uint32_t)floorf(cfg->error_min - fabsf(cfg->km) * 32768.0f) - 1u;
    MM_TRY_C(hipMalloc(&m->d_st, (size_t)cfg->nr_channels * sizeof(MmState)));
    /* mm_init, mueller_muller.c:17-26 */
    MmState init{ cfg->samples_per_bit, cfg->samples_per_bit, 0.0f, 0.0f };
    for (uint32_t c = 0; c < cfg->nr_channels; c++) {
        MM_TRY_C(hipMemcpy(m->d_st + c, &init, sizeof(init), hipMemcpyHostToDevice));
    }
    return MFM_OK;
}

void mfm_mm_destroy(struct mfm_mm **pm)
{
    if (!pm || !*pm) {
        return;
    }
    mfm_mm *m = *pm;
    (void)hipSetDevice(m->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(m->d_dec);
    (void)hipFree(m->d_counts);
    (void)hipFree(m->d_st);
    (void)hipFree(m->d_stage);
    delete m;
    *pm = nullptr;
}

size_t mfm_mm_max_decisions(const struct mfm_mm *m)
{
    return m ? m->dec_cap : 0;
}

int mfm_mm_process_device(struct mfm_mm *m, const int16_t *d_pcm, size_t in_stride, size_t nr_in, void *stream,
                          int16_t **d_decisions, size_t *dec_stride, uint32_t **d_counts)
{
    if (!m || !d_pcm || !d_decisions || !dec_stride || !d_counts || 0 == nr_in || nr_in > m->cfg.max_in_samples ||
        in_stride < nr_in) {
        return MFM_E_INVAL;
    }
    MM_TRY(hipSetDevice(m->cfg.device));
    MmLaunch L{};
    L.pcm = d_pcm;
    L.dec = m->d_dec;
    L.counts = m->d_counts;
    L.smin = m->smin;
    L.st = m->d_st;
    L.in_stride = (uint32_t)in_stride;
    L.nr_in = (uint32_t)nr_in;
    L.readable = (uint32_t)(in_stride > nr_in ? nr_in + 1 : nr_in);
    L.dec_stride = m->dec_pitch;
    L.dec_cap = m->dec_cap;
    L.nchan = m->cfg.nr_channels;
    L.kw = m->cfg.kw;
    L.km = m->cfg.km;
    L.error_min = m->cfg.error_min;
    L.error_max = m->cfg.error_max;
    mm_launch(L, static_cast<hipStream_t>(stream));
    MM_TRY(hipGetLastError());
    *d_decisions = m->d_dec;
    *dec_stride = m->dec_pitch;
    *d_counts = m->d_counts;
    return MFM_OK;
}

int mfm_mm_process_host(struct mfm_mm *m, const int16_t *pcm, size_t in_stride, size_t nr_in, int16_t *decisions,
                        size_t dec_stride, uint32_t *counts)
{
    if (!m || !pcm || !decisions || !counts || 0 == nr_in || nr_in > m->cfg.max_in_samples || in_stride < nr_in) {
        return MFM_E_INVAL;
    }
    MM_TRY(hipSetDevice(m->cfg.device));
    const uint32_t C = m->cfg.nr_channels;
    const size_t cols = in_stride > nr_in ? nr_in + 1 : nr_in; /* with the look-ahead sample when the caller has one */
    if (!m->d_stage) {
        MM_TRY(hipMalloc(&m->d_stage, (size_t)C * (m->cfg.max_in_samples + 1) * 2));
    }
    MM_TRY(hipMemcpy2D(m->d_stage, (size_t)(m->cfg.max_in_samples + 1) * 2, pcm, in_stride * 2, cols * 2, C,
                       hipMemcpyHostToDevice));
    /* the staged rows are max_in_samples + 1 apart; the look-ahead column is readable only if the caller had it */
    MmLaunch L{};
    L.pcm = m->d_stage;
    L.dec = m->d_dec;
    L.counts = m->d_counts;
    L.smin = m->smin;
    L.st = m->d_st;
    L.in_stride = m->cfg.max_in_samples + 1;
    L.nr_in = (uint32_t)nr_in;
    L.readable = (uint32_t)cols;
    L.dec_stride = m->dec_pitch;
    L.dec_cap = m->dec_cap;
    L.nchan = C;
    L.kw = m->cfg.kw;
    L.km = m->cfg.km;
    L.error_min = m->cfg.error_min;
    L.error_max = m->cfg.error_max;
    mm_launch(L, nullptr);
    MM_TRY(hipGetLastError());
    int16_t *d_dec = m->d_dec;
    const size_t dstr = m->dec_pitch;
    uint32_t *d_cnt = m->d_counts;
    MM_TRY(hipDeviceSynchronize());
    MM_TRY(hipMemcpy(counts, d_cnt, (size_t)C * 4, hipMemcpyDeviceToHost));
    uint32_t mx = 0;
    for (uint32_t c = 0; c < C; c++) {
        mx = counts[c] > mx ? counts[c] : mx;
    }
    if (mx > dec_stride || mx > m->dec_cap) {
        return MFM_E_NOMEM;
    }
    if (mx) {
        MM_TRY(hipMemcpy2D(decisions, dec_stride * 2, d_dec, dstr * 2, (size_t)mx * 2, C, hipMemcpyDeviceToHost));
    }
    return MFM_OK;
}

} /* extern "C" */
