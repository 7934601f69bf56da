/*
 * mfm_resampler.hip - the PCM stage behind the channel FIFOs, batched over all channels on the GPU:
 * real-valued rational resampler (filter/polyphase_fir.c, filter/utils.c) and DC blocker
 * (filter/dc_blocker.h).  See include/multifm_hip.h for the boundary and the reference lines.
 *
 * Kernel shape: grid = (output blocks, channels); one thread per output.  All channels share the phase
 * walk (they receive the same number of samples), so output j of a call starts at sample
 * (p0 + j*D) / I with phase (p0 + j*D) % I.  The phase filters (I x phase_len int16) sit in LDS; the PCM
 * windows overlap heavily between neighbouring threads and come through L1/L2.  This stage moves two
 * bytes per output and is nowhere near any roof; it exists so the whole chain stays on the device.
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
    const int16_t *x; /* [C][in_cap]: carried tail followed by the new samples */
    int16_t *y;       /* [C][out_cap] */
    const int16_t *phase; /* [I][plen] */
    uint32_t in_cap, out_cap, n_out, plen, interp, decim, p0, nchan, invert;
};

__global__ __launch_bounds__(256) void mfm_resample_kernel(const RsLaunch L)
{
    extern __shared__ int16_t ph_s[];
    for (uint32_t i = threadIdx.x; i < L.interp * L.plen; i += blockDim.x) {
        ph_s[i] = L.phase[i];
    }
    __syncthreads();
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (j >= L.n_out) {
        return;
    }
    /* filter/polyphase_fir.c:206-211 unrolled to output j */
    const uint64_t t = (uint64_t)L.p0 + (uint64_t)j * L.decim;
    const uint32_t pos = (uint32_t)(t / L.interp), ph = (uint32_t)(t % L.interp);
    const int16_t *xw = L.x + (size_t)c * L.in_cap + pos;
    const int16_t *cf = ph_s + ph * L.plen;
    uint32_t acc = 0; /* filter/utils.c:94-103, int32 wrap-around */
    for (uint32_t k = 0; k < L.plen; k++) {
        int32_t s = xw[k];
        if (L.invert) {
            s = (int16_t)(-s); /* decoder.c:624: samp[i] *= -1 on int16 */
        }
        acc += (uint32_t)(s * (int32_t)cf[k]);
    }
    L.y[(size_t)c * L.out_cap + j] = (int16_t)mfm_r14_wide((int32_t)acc); /* utils.c:112 */
}

struct DcState {
    int32_t x_n_1, y_n_1, acc;
};

__global__ void mfm_dc_block_kernel(int16_t *y, uint32_t out_cap, uint32_t n_out, uint32_t nchan, int32_t p, DcState *st)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchan) {
        return;
    }
    DcState s = st[c];
    int16_t *yy = y + (size_t)c * out_cap;
    for (uint32_t i = 0; i < n_out; i++) { /* filter/dc_blocker.h:80-90 */
        s.acc = (int32_t)((uint32_t)s.acc - (uint32_t)s.x_n_1);
        s.x_n_1 = (int32_t)((uint32_t)(int32_t)yy[i] << 14);
        s.acc = (int32_t)((uint32_t)s.acc + (uint32_t)s.x_n_1 - (uint32_t)(p * s.y_n_1));
        s.y_n_1 = s.acc >> 14;
        yy[i] = (int16_t)s.y_n_1;
    }
    st[c] = s;
}

thread_local char g_rs_error[256] = "";

} /* namespace */

struct mfm_resampler {
    mfm_resampler_config cfg{};
    uint32_t plen = 0;
    uint32_t in_cap = 0, out_cap = 0;
    int16_t *d_phase = nullptr;
    int16_t *d_x[2] = { nullptr, nullptr };
    int16_t *d_y = nullptr;
    DcState *d_dc = nullptr;
    int32_t dc_p = 0;
    int cur = 0;
    uint32_t tail = 0; /* unconsumed samples at the front of d_x[cur] */
    uint32_t phase_id = 0;
    int16_t *d_stage = nullptr; /* process_host_to_device: [C][max_in_samples] */
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
    for (int i = 0; i < 2; i++) {
        RS_TRY(hipMalloc(&r->d_x[i], (size_t)cfg->nr_channels * r->in_cap * 2));
        RS_TRY(hipMemset(r->d_x[i], 0, (size_t)cfg->nr_channels * r->in_cap * 2));
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
    if (nr_in > r->cfg.max_in_samples || r->tail + nr_in > r->in_cap) {
        return MFM_E_INVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t C = r->cfg.nr_channels, I = r->cfg.interpolate, D = r->cfg.decimate;
    RS_TRY(hipSetDevice(r->cfg.device));
    int16_t *x = r->d_x[r->cur];
    if (nr_in) {
        RS_TRY(hipMemcpy2DAsync(x + r->tail, (size_t)r->in_cap * 2, d_pcm, in_stride * 2, nr_in * 2, C,
                                hipMemcpyDeviceToDevice, s));
    }
    const uint32_t total = r->tail + (uint32_t)nr_in;
    /* outputs m with total - pos_m > plen  <=>  p0 + m*D < (total - plen) * I   (polyphase_fir.c:184) */
    uint32_t n_out = 0;
    if (total > r->plen) {
        const uint64_t lim = (uint64_t)(total - r->plen) * I;
        if (lim > r->phase_id) {
            n_out = (uint32_t)((lim - r->phase_id + D - 1) / D);
        }
    }
    if (n_out) {
        RsLaunch L{ x, r->d_y, r->d_phase, r->in_cap, r->out_cap, n_out, r->plen, I, D, r->phase_id, C, r->cfg.invert };
        const dim3 grid((n_out + 255) / 256, C);
        hipLaunchKernelGGL(mfm_resample_kernel, grid, dim3(256), (size_t)I * r->plen * 2, s, L);
        RS_TRY(hipGetLastError());
        if (r->cfg.dc_block) {
            hipLaunchKernelGGL(mfm_dc_block_kernel, dim3((C + 63) / 64), dim3(64), 0, s, r->d_y, r->out_cap, n_out, C,
                               r->dc_p, r->d_dc);
            RS_TRY(hipGetLastError());
        }
    }
    const uint64_t t_end = (uint64_t)r->phase_id + (uint64_t)n_out * D;
    const uint32_t pos_end = (uint32_t)(t_end / I);
    r->phase_id = (uint32_t)(t_end % I);
    const uint32_t new_tail = total - pos_end; /* pos_end <= total: create() checked ceil(D/I) <= plen */
    if (new_tail) {
        RS_TRY(hipMemcpy2DAsync(r->d_x[r->cur ^ 1], (size_t)r->in_cap * 2, x + pos_end, (size_t)r->in_cap * 2,
                                (size_t)new_tail * 2, C, hipMemcpyDeviceToDevice, s));
    }
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
