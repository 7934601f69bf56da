/*
 * mfm_mm.hip - Mueller-Muller clock recovery for all channels of a PCM block (pager/mueller_muller.c:10-115;
 * BASELINE.json configs[3] names it as the slicer in front of the pager stage).
 *
 * The loop is a feedback loop - where the next decision is taken depends on the last one - so a channel is
 * sequential; channels are independent.  One lane per channel; a workgroup (one wave) stages a window of every
 * channel's samples into LDS with coalesced loads and the lanes walk their own rows, the window following the
 * slowest lane.  Same float operations in the same order as the reference (no contraction: mul, mul, sub; mul, add;
 * compare/clamp; mul, add, add; floorf), so the decisions are bit-identical to oracle/pocsag_oracle.c's restatement.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <new>

#include "../../include/multifm_hip.h"

extern "C" void mfm_internal_set_error(const char *msg);

namespace {

constexpr uint32_t MM_W = 256; /* samples per channel and window */

struct MmState {
    float w, m, next_offset, last_sample;
};

struct MmLaunch {
    const int16_t *pcm;
    int16_t *dec;
    uint32_t *counts;
    MmState *st;
    uint32_t in_stride, nr_in, readable, dec_stride, nchan;
    float kw, km, error_min, error_max;
};

__global__ __launch_bounds__(64) void mfm_mm_kernel(const MmLaunch L)
{
    __shared__ int16_t win[64][MM_W + 8];
    const uint32_t lane = threadIdx.x, c = blockIdx.x * 64u + lane;
    const bool live = c < L.nchan;
    MmState s = live ? L.st[c] : MmState{ 0.f, 0.f, 0.f, 0.f };
    float cur = s.next_offset;
    const float nr_f = (float)L.nr_in; /* mueller_muller.c:58 */
    uint32_t n_dec = 0;
    int16_t *dec = L.dec + (size_t)(live ? c : 0u) * L.dec_stride;

    for (;;) {
        const bool active = live && cur < nr_f; /* :66 */
        const uint32_t idx = active ? (uint32_t)(cur + 0.5f) : 0xffffffffu;
        /* the window starts at the slowest active lane's next sample */
        uint32_t ws = idx;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)ws, off);
            ws = o < ws ? o : ws;
        }
        if (ws == 0xffffffffu) {
            break; /* nobody has anything left */
        }
        ws &= ~3u;
        __syncthreads(); /* one wave: orders the LDS accesses of the previous window */
        for (uint32_t r = 0; r < 64u; r++) {
            const uint32_t ch = blockIdx.x * 64u + r;
            if (ch < L.nchan) {
                const int16_t *row = L.pcm + (size_t)ch * L.in_stride;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    uint32_t v = ws + lane * 4u + j;
                    v = v < L.readable ? v : L.readable - 1u;
                    win[r][lane * 4u + j] = row[v];
                }
            }
        }
        __syncthreads();
        if (active) {
            uint32_t i = idx;
            while (cur < nr_f && i < ws + MM_W) {
                const uint32_t ic = i < L.readable ? i : L.readable - 1u;
                const float sample = (float)win[lane][ic - ws]; /* :67 */
                if (n_dec < L.dec_stride) {
                    dec[n_dec] = (int16_t)sample; /* :71 */
                }
                n_dec++;
                const float sl = (float)(s.last_sample > 0.f) - (float)(s.last_sample < 0.f);
                const float sc = (float)(sample > 0.f) - (float)(sample < 0.f);
                const float w_error = sl * sample - sc * s.last_sample; /* :77 */
                s.w += w_error * L.kw;                                   /* :80 */
                if (L.error_min > s.w) {                                 /* :87-91 */
                    s.w = L.error_min;
                } else if (L.error_max < s.w) {
                    s.w = L.error_max;
                }
                s.m += s.w + L.km * sample; /* :93 */
                const float fl = floorf(s.m);
                cur += fl;                  /* :96 */
                s.m -= fl;                  /* :98 */
                s.last_sample = sample;     /* :101 */
                i = (uint32_t)(cur + 0.5f);
            }
        }
    }
    if (live) {
        s.next_offset = cur - nr_f; /* :109 */
        L.st[c] = s;
        L.counts[c] = n_dec;
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

int mfm_mm_create(struct mfm_mm **pm, const struct mfm_mm_config *cfg)
{
    if (!pm || !cfg) {
        return MFM_E_INVAL;
    }
    *pm = nullptr;
    if (cfg->abi_version != MFM_ABI_VERSION || 0 == cfg->nr_channels || 0 == cfg->max_in_samples ||
        !(cfg->error_max >= cfg->error_min) || !(cfg->samples_per_bit >= 1.0f) ||
        !(cfg->error_min - fabsf(cfg->km) * 32768.0f >= 1.0f)) {
        return MFM_E_INVAL; /* a step that can fall below one sample would never leave the loop (:93-96) */
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
    /* every step advances by at least floor(error_min + km * sample) >= ... samples; sized for steps of
     * error_min / 2, and the kernel never writes past it */
    m->dec_cap = (uint32_t)((double)cfg->max_in_samples / (cfg->error_min * 0.5) + 16.0);
    *pm = m;
    MM_TRY(hipSetDevice(cfg->device));
    MM_TRY(hipMalloc(&m->d_dec, (size_t)cfg->nr_channels * m->dec_cap * 2));
    MM_TRY(hipMalloc(&m->d_counts, (size_t)cfg->nr_channels * 4));
    MM_TRY(hipMalloc(&m->d_st, (size_t)cfg->nr_channels * sizeof(MmState)));
    /* mm_init, mueller_muller.c:17-26 */
    MmState init{ cfg->samples_per_bit, cfg->samples_per_bit, 0.0f, 0.0f };
    for (uint32_t c = 0; c < cfg->nr_channels; c++) {
        MM_TRY(hipMemcpy(m->d_st + c, &init, sizeof(init), hipMemcpyHostToDevice));
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
    L.st = m->d_st;
    L.in_stride = (uint32_t)in_stride;
    L.nr_in = (uint32_t)nr_in;
    L.readable = (uint32_t)(in_stride > nr_in ? nr_in + 1 : nr_in);
    L.dec_stride = m->dec_cap;
    L.nchan = m->cfg.nr_channels;
    L.kw = m->cfg.kw;
    L.km = m->cfg.km;
    L.error_min = m->cfg.error_min;
    L.error_max = m->cfg.error_max;
    hipLaunchKernelGGL(mfm_mm_kernel, dim3((L.nchan + 63u) / 64u), dim3(64), 0, static_cast<hipStream_t>(stream), L);
    MM_TRY(hipGetLastError());
    *d_decisions = m->d_dec;
    *dec_stride = m->dec_cap;
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
    L.st = m->d_st;
    L.in_stride = m->cfg.max_in_samples + 1;
    L.nr_in = (uint32_t)nr_in;
    L.readable = (uint32_t)cols;
    L.dec_stride = m->dec_cap;
    L.nchan = C;
    L.kw = m->cfg.kw;
    L.km = m->cfg.km;
    L.error_min = m->cfg.error_min;
    L.error_max = m->cfg.error_max;
    hipLaunchKernelGGL(mfm_mm_kernel, dim3((C + 63u) / 64u), dim3(64), 0, nullptr, L);
    MM_TRY(hipGetLastError());
    int16_t *d_dec = m->d_dec;
    const size_t dstr = m->dec_cap;
    uint32_t *d_cnt = m->d_counts;
    MM_TRY(hipDeviceSynchronize());
    MM_TRY(hipMemcpy(counts, d_cnt, (size_t)C * 4, hipMemcpyDeviceToHost));
    uint32_t mx = 0;
    for (uint32_t c = 0; c < C; c++) {
        mx = counts[c] > mx ? counts[c] : mx;
    }
    if (mx > dec_stride || mx > dstr) {
        return MFM_E_NOMEM;
    }
    if (mx) {
        MM_TRY(hipMemcpy2D(decisions, dec_stride * 2, d_dec, dstr * 2, (size_t)mx * 2, C, hipMemcpyDeviceToHost));
    }
    return MFM_OK;
}

} /* extern "C" */
